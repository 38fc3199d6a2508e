#!/usr/bin/env python3
"""A/B of two engine libraries (C3 fp16, wavenet_wg): us per sample, interleaved repeats.  usage: quick_ab.py libA libB"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
code = r'''
import sys, os
sys.path.insert(0, os.path.dirname(%r))
import bench
w = bench.make_weights()
out = []
for B, N, org in ((16, 512, 2), (32, 512, 3), (4096, 192, 2), (8192, 128, 3)):
    try:
        khz, info = bench.measure_khz(w, B, N, organisation=org)
        out.append("%%d:%%.2f" %% (B, 1e3 / khz))
    except Exception as e:
        out.append("%%d:err" %% B)
print(" ".join(out))
''' % here
for rep in range(3):
    for lib in sys.argv[1:]:
        env = dict(os.environ, NVW_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(os.path.basename(os.path.dirname(lib)), r.stdout.strip().split("\n")[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
