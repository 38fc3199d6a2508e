#!/usr/bin/env python3
"""Packed vs in-place conditioning, C3 fp16: kHz per utterance of one launch (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
w = bench.make_weights()
for B, N in ((16, 1024), (4096, 256), (8192, 128)):
    e = bench.build_engine(w, B, N)
    Lh, sel = bench.device_inputs(B, N, 11)
    e.setInputs(Lh, sel)
    torch.cuda.synchronize()
    e.time_runs(1, min(N, 64), B)
    ms = e.time_runs(1, N, B)
    e.setConditioningDirect(Lh)
    e.time_runs(1, min(N, 64), B)
    ms2 = e.time_runs(1, N, B)
    print("B=%5d  packed %.2f kHz   in place %.2f kHz   %s" % (B, N / ms, N / ms2, e.kernelInfo(B, False).split(" ")[0]), flush=True)
    e.close(); del Lh; torch.cuda.empty_cache()
