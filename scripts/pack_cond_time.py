#!/usr/bin/env python3
"""The conditioning pack alone (setInputs of 256 samples x 8192 utterances, C3 shape, device source), 3 times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
w = bench.make_weights()
B, N = 8192, 256
e = bench.build_engine(w, B, N)
Lh, sel = bench.device_inputs(B, N, 5)
for i in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e.setConditioning(Lh)
    torch.cuda.synchronize()
    print("setConditioning(%d x %d): %.2f ms  (source %.1f GB fp32, packed %.1f GB fp16)" %
          (N, B, 1e3 * (time.perf_counter() - t0), Lh.numel() * 4 / 1e9, Lh.numel() * 2 / 1e9))
e.close()
