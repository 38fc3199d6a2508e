#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of wavenet_split, wave 0 (role A) and wave 4 (role B) of workgroup 0 (needs the
-DWN_SPLIT_TIMING experiment build via NVW_LIB).   usage: split_phase.py [batch] [samples] [organisation 8/9/10]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
a = sys.argv[1:] + ["48", "256", "10"][len(sys.argv) - 1:]
B, N, ORG = int(a[0]), int(a[1]), int(a[2])
import bench
w = bench.make_weights()
e = bench.build_engine(w, B, N, organisation=ORG)
Lh, sel = bench.device_inputs(B, N, 1)
e.setInputs(Lh, sel)
ms = e.time_runs(1, N, B)
P = e.getP().reshape(-1)[:32]
L = bench.L
print("B=%d N=%d org=%d: %.2f us/sample  %s" % (B, N, ORG, 1e3 * ms / N, e.kernelInfo(B, False).split(" ")[0]))
namesA = ["acc/xb reads", "cur gemm", "gate + h store", "barrier H wait", "hb read + bres", "res gemm + x store", "barrier X wait", "-", "embedding",
          "barrier E wait", "barrier S wait", "head", "-", "-", "-", "-"]
namesB = ["xp/xs reads", "preact + prev gemm", "ring st + skip gemm", "prefetch issue", "barrier H wait", "Q2: acc/hb/publish", "barrier X wait", "tail skip + sk store",
          "sel + zero skip", "barrier E wait", "barrier S wait", "head", "-", "-", "-", "-"]
for role, names, off in (("A (wave 0)", namesA, 0), ("B (wave 4)", namesB, 16)):
    v = P[off:off + 16]
    tot = v.sum()
    print("role %s: %.0f clk per sample" % (role, tot / N))
    for i, (n, x) in enumerate(zip(names, v)):
        if n != "-":
            per = x / N
            print("   %-24s %8.0f clk/sample %5.1f%%  %s" % (n, per, 100 * x / tot, ("(%.0f per layer)" % (per / L)) if i < 7 else ""))
