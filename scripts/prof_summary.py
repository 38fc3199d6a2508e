#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2 rocpd SQLite) results into text for profiles/.

  python scripts/prof_summary.py kernel gpurun_out/prof_kt/kt_results.db
  python scripts/prof_summary.py pmc    gpurun_out/prof_fetch/f_results.db [more.db ...]
"""
import sqlite3
import sys
from collections import defaultdict


def kernel_stats(db):
    c = sqlite3.connect(db)
    rows = c.execute("select k.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x, d.group_segment_size, "
                     "k.arch_vgpr_count, k.accum_vgpr_count, k.sgpr_count, d.private_segment_size "
                     "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol k on d.kernel_id = k.id").fetchall()
    agg = defaultdict(list)
    meta = {}
    for name, s, e, gx, wx, lds, vg, ag, sg, scr in rows:
        agg[name].append(e - s)
        meta[name] = (gx, wx, lds, vg, ag, sg, scr)
    tot = sum(sum(v) for v in agg.values())
    print("%-90s %6s %12s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-90s %6d %12d %12d %12d %12d %6.2f%%" % (name[:90], len(v), sum(v), sum(v) // len(v), min(v), max(v),
                                                         100.0 * sum(v) / tot))
        gx, wx, lds, vg, ag, sg, scr = meta[name]
        print("    grid=%d wg=%d lds=%d B vgpr=%d agpr=%d sgpr=%d scratch=%d B/lane" % (gx, wx, lds, vg, ag, sg, scr))


def pmc_stats(dbs):
    for db in dbs:
        c = sqlite3.connect(db)
        rows = c.execute(
            "select k.kernel_name, p.name, e.value from rocpd_pmc_event e "
            "join rocpd_info_pmc p on e.pmc_id = p.id "
            "join rocpd_kernel_dispatch d on d.event_id = e.event_id "
            "join rocpd_info_kernel_symbol k on d.kernel_id = k.id").fetchall()
        agg = defaultdict(list)
        for name, pmc, val in rows:
            agg[(name, pmc)].append(val)
        print("# %s" % db)
        print("%-90s %-14s %6s %16s %16s" % ("kernel", "counter", "calls", "avg", "max"))
        for (name, pmc), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            print("%-90s %-14s %6d %16.3f %16.3f" % (name[:90], pmc, len(v), sum(v) / len(v), max(v)))


if __name__ == "__main__":
    if sys.argv[1] == "kernel":
        kernel_stats(sys.argv[2])
    else:
        pmc_stats(sys.argv[2:])
