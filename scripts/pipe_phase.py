#!/usr/bin/env python3
"""Phase timing of wavenet_pipe's layer stages (experiment build -DWN_CHAIN_TIMING, NVW_LIB): group 0 of chain 0."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nv_wavenet_amd import WavenetEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N = 32
w = bench.make_weights()
e = bench.build_engine(w, B, N, organisation=7)
info = e.kernelInfo(B, False)
print(info)
K = int(info.split("stages=")[1].split()[0])
Lh, sel = bench.device_inputs(B, N, 3)
e.setInputs(Lh, sel)
assert e.run(N, B, None, 1, False)
e.synchronize()
assert e.chainStatus() == 0
raw = e.getP().view(np.uint64).reshape(-1)[:(K - 1) * 8 * 16].reshape(K - 1, 8, 16).astype(np.int64)
us = lambda a: a * 0.01
names = ["drain+barrier", "recv x", "layers", "send x", "recv skip", "skip gemm", "send skip"]
fine = ["acc init+cond (li0)", "prev gemm", "ring st+prefetch", "cur gemm+gate+put", "barrier h", "get h+res gemm", "put x+barrier"]
for s in range(K - 1):
    r = raw[s]
    print("stage %d: step %.2f us | " % (s, us(np.diff(r[:, 0]).mean())) + "  ".join("%s %.2f" % (n, v) for n, v in zip(names, us(np.diff(r[:, :8], axis=1).mean(0)))))
    ev = np.stack([r[:, 2], r[:, 8], r[:, 9], r[:, 10], r[:, 11], r[:, 12], r[:, 13], r[:, 14]], 1)
    print("         layer 0: " + "  ".join("%s %.3f" % (n, v) for n, v in zip(fine, us(np.diff(ev, axis=1).mean(0)))))
e.close()
