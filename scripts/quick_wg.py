#!/usr/bin/env python3
"""kHz per utterance of the C3 fp16 engine at a few batch sizes / organisations (HIP events, pre-packed inputs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
w = bench.make_weights()
for B, N, org in ((16, 1024, 2), (4096, 256, 0), (8192, 128, 0), (16384, 128, 4)):
    khz, info = bench.measure_khz(w, B, N, organisation=org)
    print("B=%5d  %.2f kHz  %.2f us/sample  %s" % (B, khz, 1e3 / khz, info.split(" ")[0]), flush=True)
