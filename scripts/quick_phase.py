#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of wavenet_wg (needs the WN_TIMING experiment build via NVW_LIB).
usage: quick_phase.py [batch] [samples] [organisation: 2 = wg one tile (default), 3 = two tiles, 4 = three tiles]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.argv = [sys.argv[0]] + (sys.argv[1:] or ["16", "512"])
B, N = int(sys.argv[1]), int(sys.argv[2])
ORG = int(sys.argv[3]) if len(sys.argv) > 3 else 2
import bench
w = bench.make_weights()
e = bench.build_engine(w, B, N, organisation=ORG)
Lh, sel = bench.device_inputs(B, N, 1)
e.setInputs(Lh, sel)
ms = e.time_runs(1, N, B)
P = e.getP().reshape(-1)[:12]
names = ["embed+barrier", "xb read + tap gemm (prev layer's tail)", "cur gemm+ring st+prefetch", "gate || skip gemm + publish tap", "barrier h",
         "hb/xp read+res gemm+put x", "cond mfma(+dump)", "barrier x", "head gemms", "pad takes+barrier", "softmax+ybarrier", "sel load"]
tot = P.sum()
print("B=%d N=%d: %.2f us/sample; wave0 clock total %.0f per sample (=%.2f us @2.4GHz... clock is 100MHz-based if small)" % (B, N, 1e3*ms/N, tot/N, tot/N/2400))
L = bench.L
for n, v in zip(names, P):
    per = v / N
    print("  %-32s %10.0f clk/sample  %6.1f%%   (%.0f per layer)" % (n, per, 100*v/tot, per / L))
