#!/usr/bin/env python3
"""kHz per utterance and samples/s of wavenet_pipe at a few batch sizes (C3 fp16, HIP events, pre-packed inputs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
w = bench.make_weights()
for B in [int(x) for x in (sys.argv[1:] or ["4096", "8192", "16384", "24576"])]:
    khz, info = bench.measure_khz(w, B, 128, organisation=7)
    print("B=%5d  %.2f kHz  %.2f us/sample  %.1f M samples/s  %s" % (B, khz, 1e3 / khz, B * khz / 1e3, info), flush=True)
