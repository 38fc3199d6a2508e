#!/usr/bin/env python3
"""GPU-box side of the profiles: reduce one rocprofv3 output directory (ROCm 7.2 rocpd SQLite) to a small JSON beside it and
DELETE the directory (a round's databases together exceed what gpurun copies back).

  prof_extract.py <dir> <out.json>

JSON: {"kernels": {name: {"calls", "durations_ns": [...], "grid", "wg", "lds", "vgpr", "agpr", "sgpr", "scratch"}},
       "pmc": {name: {counter: {"sum": s, "calls": n}}}}   (durations in dispatch order; counters summed over dispatches)"""
import glob
import json
import os
import shutil
import sqlite3
import sys
from collections import defaultdict

d, out = sys.argv[1], sys.argv[2]
dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
res = {"kernels": {}, "pmc": {}, "dbs": len(dbs)}
for db in dbs:
    c = sqlite3.connect(db)
    rows = c.execute("select k.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x, d.group_segment_size, "
                     "k.arch_vgpr_count, k.accum_vgpr_count, k.sgpr_count, d.private_segment_size "
                     "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol k on d.kernel_id = k.id order by d.start").fetchall()
    for name, s, e, gx, wx, lds, vg, ag, sg, scr in rows:
        k = res["kernels"].setdefault(name, {"calls": 0, "durations_ns": [], "grid": gx, "wg": wx, "lds": lds, "vgpr": vg, "agpr": ag,
                                             "sgpr": sg, "scratch": scr})
        k["calls"] += 1
        if len(k["durations_ns"]) < 64:
            k["durations_ns"].append(e - s)
        k["total_ns"] = k.get("total_ns", 0) + (e - s)
    try:
        rows = c.execute("select k.kernel_name, p.name, e.value from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
                         "join rocpd_kernel_dispatch d on d.event_id = e.event_id join rocpd_info_kernel_symbol k on d.kernel_id = k.id").fetchall()
    except sqlite3.Error:
        rows = []
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for name, pmc, val in rows:
        agg[name][pmc][0] += val
        agg[name][pmc][1] += 1
    for name, cs in agg.items():
        res["pmc"].setdefault(name, {}).update({p: {"sum": v[0], "calls": v[1]} for p, v in cs.items()})
    c.close()
json.dump(res, open(out, "w"))
shutil.rmtree(d, ignore_errors=True)
print(out, "kernels", len(res["kernels"]), "pmc kernels", len(res["pmc"]))
