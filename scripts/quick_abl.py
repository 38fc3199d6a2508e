#!/usr/bin/env python3
"""us per sample of wavenet_wg (C3 fp16) for A/B builds (library chosen with NVW_LIB): one workgroup at 1 / 2 / 3 tiles
(latency), and full-GPU launches at two and three tiles per CU AT STEADY STATE (samples 640..1151, all taps live).
usage: quick_abl.py [points]   points: comma list of w1,w2,w3,g2,g3,g3raw16,g3raw32 (default all but the raw ones)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
pts = (sys.argv[1] if len(sys.argv) > 1 else "w1,w2,w3,g2,g3").split(",")
w = bench.make_weights()
ncu = torch.cuda.get_device_properties(0).multi_processor_count
out = []
for p in pts:
    if p in ("w1", "w2", "w3"):
        bt = int(p[1])
        khz, info = bench.measure_steady_khz(w, 16 * bt, 512, organisation=1 + bt)
    else:
        bt = int(p[1])
        ip = {"raw16": torch.float16, "raw32": torch.float32}.get(p[2:], None)
        khz, info = bench.measure_steady_khz(w, 16 * bt * ncu, 256, in_place=ip)
    out.append("%s %.2f" % (p, 1e3 / khz))
print("  ".join(out), flush=True)
