#!/usr/bin/env python3
"""Generate tests/golden/mulaw_pcm.npz from the REFERENCE's own Python (authoring container only):
    python tests/golden/make_mulaw_golden.py
pcm_A[y] = (MAX_WAV_VALUE * mu_law_decode_numpy(y, A)).astype('int16') for every sample index y,
i.e. pytorch/utils.py:62-70 followed by pytorch/inference.py:58-60."""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, "/root/reference/pytorch")
import utils  # noqa: E402  (the reference's utils.py)

out = {}
with warnings.catch_warnings():
    warnings.simplefilter("ignore")      # the top bin overflows int16 in the reference, too
    for A in (256, 512, 1024):
        audio = utils.mu_law_decode_numpy(np.arange(A), A)
        out["pcm_%d" % A] = (utils.MAX_WAV_VALUE * audio).astype("int16")
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mulaw_pcm.npz"), **out)
print({k: (v[:3], v[-3:]) for k, v in out.items()})
