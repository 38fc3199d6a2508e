#!/usr/bin/env python3
"""Three tiles per workgroup (organisation 8) against two: identical samples, then kHz per utterance (C3 fp16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
w = bench.make_weights()


def samples(B, N, org):
    e = bench.build_engine(w, B, N, organisation=org)
    Lh, sel = bench.device_inputs(B, N, 5)
    e.setInputs(Lh, sel)
    y = np.full((B, N), -1, dtype=np.int32)
    assert e.run(N, B, y)
    e.synchronize()
    info = e.kernelInfo(B, False)
    e.close()
    return y, info


for B in (40, 100):
    y3, i3 = samples(B, 300, 8)
    y2, i2 = samples(B, 300, 3)
    print("B=%d identical=%s  %s | %s" % (B, np.array_equal(y3, y2), i3, i2), flush=True)
cases = ((48, 512, 8), (32, 512, 3), (12288, 128, 8), (8192, 128, 3), (12288, 128, 3))
for B, N, org in cases:
    khz, info = bench.measure_khz(w, B, N, organisation=org)
    print("B=%5d org=%d  %.2f kHz  %.2f us/sample  %s" % (B, org, khz, 1e3 / khz, info), flush=True)
