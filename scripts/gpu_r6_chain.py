#!/usr/bin/env python3
"""Round 6: the C4 chain with several tiles per chain, steady-state kHz per utterance at 4 / 5 / 6 tiles per chain (16 chains x 16 CUs),
for the library NVW_LIB selects -- the HOIST instantiation (a unit's packed conditioning requested up front) against the plain one
(-DWN_CHAIN_HOIST_FROM=99).  usage: [NVW_LIB=...] gpu_r6_chain.py <tag> [tiles per chain ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402

tag = sys.argv[1]
tpcs = [int(x) for x in sys.argv[2:]] or [4, 5, 6]
sh = bench.C4
ncu = torch.cuda.get_device_properties(0).multi_processor_count
chains = ncu // 16
for tpc in tpcs:
    shb = bench.Shape(sh.name, sh.R, sh.S, sh.A, sh.L, sh.maxD, 16 * chains * tpc)
    ks = []
    for _ in range(2):
        k, info = bench.measure_steady_khz(bench.make_weights(shb, seed=1), shb.B, 1024, sh=shb, impl=3)
        ks.append(round(k, 3))
    rec = dict(tag=tag, lib=os.environ.get("NVW_LIB", "shipped"), tiles_per_chain=tpc, batch=shb.B, steady_khz=ks, kernel=info)
    print(json.dumps(rec), flush=True)
    open(os.path.join(ROOT, "gpurun_out", "r6_chain.jsonl"), "a").write(json.dumps(rec) + "\n")
