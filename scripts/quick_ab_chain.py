#!/usr/bin/env python3
"""A/B of engine libraries on C3 fp16: reference-definition kHz of the chain and the one-tile workgroup. usage: quick_ab_chain.py libA libB ..."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
code = r'''
import sys, os
sys.path.insert(0, os.path.dirname(%r))
import bench
out = []
for impl in (3, 1):
    r = bench.reference_definition_khz(bench.C3, impl)
    out.append("impl%%d:%%.2f" %% (impl, r["khz_per_utterance"]))
print(" ".join(out))
''' % here
for rep in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ, NVW_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(os.path.basename(os.path.dirname(lib)), r.stdout.strip().split("\n")[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
