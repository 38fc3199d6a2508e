#!/usr/bin/env python3
"""us per sample of wavenet_wg (C3 fp16) at 1 / 2 / 3 tiles per workgroup, one workgroup and a full GPU (for ablation builds via NVW_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
w = bench.make_weights()
out = []
for B, N, org in ((16, 512, 2), (32, 512, 3), (48, 512, 8), (8192, 128, 3), (12288, 128, 8)):
    khz, info = bench.measure_khz(w, B, N, organisation=org)
    out.append("B=%d/org%d %.2f" % (B, org, 1e3 / khz))
print("  ".join(out), flush=True)
