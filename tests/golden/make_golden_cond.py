#!/usr/bin/env python3
"""Generate tests/golden/cond_*.npz from the REFERENCE's own WaveNet.get_cond_input (pytorch/wavenet.py:190-202).

Runs only in the authoring container (imports /root/reference/pytorch/wavenet.py):
    python tests/golden/make_golden_cond.py

For every tests/condgen.py case the reference's WaveNet module is built with the case's dimensions, its `upsample` and
`cond_layers` parameters are overwritten with the case's seeded tensors (condgen.make_cond_model), and get_cond_input(features) is
evaluated on the CPU in fp32 by the reference's own code.  Stored: the tensor's shape, float64 sums and 4096 sampled values -- the
inputs are rebuilt from the seed by the tests, so the fixture stays a few KB.
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, "/root/reference/pytorch")

import cases  # noqa: E402
import condgen  # noqa: E402


def reference_cond_input(cc, shape, m):
    import wavenet  # the reference's module
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = wavenet.WaveNet(n_in_channels=shape.A, n_layers=shape.L, max_dilation=shape.maxD, n_residual_channels=shape.R,
                              n_skip_channels=shape.S, n_out_channels=shape.A, n_cond_channels=cc.n_cond, upsamp_window=cc.window,
                              upsamp_stride=cc.stride)
    with torch.no_grad():
        net.upsample.weight.copy_(torch.from_numpy(m["up_w"]))
        net.upsample.bias.copy_(torch.from_numpy(m["up_b"]))
        net.cond_layers.conv.weight.copy_(torch.from_numpy(m["cond_w"]))
        net.cond_layers.conv.bias.copy_(torch.from_numpy(m["cond_b"]))
        out = net.get_cond_input(torch.from_numpy(m["features"]))
    return out.contiguous().numpy()


def main():
    torch.set_num_threads(1)
    for cc in condgen.COND_CASES:
        shape = cases.BY_NAME[cc.case_name].shape
        m = condgen.make_cond_model(cc, shape)
        ci = reference_cond_input(cc, shape, m)
        assert ci.shape == (2 * shape.R, shape.B, shape.L, shape.N), ci.shape
        rec = condgen.record_of(ci)
        np.savez(os.path.join(HERE, cc.name + ".npz"), **rec)
        print("%-20s cond_input %s std %.3f  (fixture %d bytes)" % (cc.name, ci.shape, ci.std(), os.path.getsize(os.path.join(HERE, cc.name + ".npz"))))


if __name__ == "__main__":
    main()
