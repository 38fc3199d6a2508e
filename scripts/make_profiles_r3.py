#!/usr/bin/env python3
"""Authoring side of the round-3 profiles: summarises the rocprofv3 databases collected by scripts/prof_collect_r3.sh
(gpurun_out/prof3_*) into the tracked files under profiles/ (r03_*), plus the static instruction mix of the shipped
code object (scripts/isa_stats.py)."""
import json
import os
import sqlite3
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
G, P = "gpurun_out", "profiles"
KERN = "wavenet_wg"
CMD = "python bench.py --batch 12288 --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
B, NSTEP, STEADY = 12288, 256, 640


def dispatches(db, sub):
    c = sqlite3.connect(db)
    rows = c.execute("select k.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol k "
                     "on d.kernel_id = k.id order by d.start").fetchall()
    return [(e - s) for n, s, e in rows if sub in n]


def pmc(db, sub):
    c = sqlite3.connect(db)
    rows = c.execute("select k.kernel_name, p.name, e.value from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
                     "join rocpd_kernel_dispatch d on d.event_id = e.event_id join rocpd_info_kernel_symbol k on d.kernel_id = k.id").fetchall()
    tot = defaultdict(float)
    for n, name, v in rows:
        if sub in n:
            tot[name] += v
    return dict(tot)


def sh(*a):
    return subprocess.run(list(a), capture_output=True, text=True).stdout


line = json.load(open(f"{G}/prof3_bench_line.json"))
dur = dispatches(f"{G}/prof3_kt/p_results.db", KERN)
# the run holds one launch of STEADY samples (the untimed run-in) and warmup + steps launches of NSTEP samples
timed = sorted(dur)[:-1]
samples_total = STEADY + NSTEP * len(timed)
with open(f"{P}/r03_kernel_trace_stats_wg_b12288.txt", "w") as f:
    f.write(f"# round 3: {CMD}  under  rocprofv3 --kernel-trace --stats\n")
    f.write("# wn::wavenet_wg<fp16,64,256,256,BT=3,EMBLDS=1,DUMP=0,RAW=0>: three tiles of 16 utterances per workgroup, 256 workgroups;\n")
    f.write(f"# every timed launch generates samples {STEADY}..{STEADY + NSTEP - 1} of 12 288 utterances (steady state: all dilated taps live);\n")
    f.write(f"# the run also holds ONE launch of {STEADY} samples (the untimed run-in from sample 0), listed separately below.\n")
    f.write("# bench.py's own line of this run: value %.1f M samples/s, kernel_ms %.3f (HIP events), khz_per_utterance %.2f, roofline.frac %.4f\n" %
            (line["value"] / 1e6, line["roofline"]["kernel_ms"], line["khz_per_utterance"], line["roofline"]["frac"]))
    f.write("# wavenet_wg launches of %d samples: n=%d avg %.3f ms min %.3f ms max %.3f ms  (the first one after the run-in is the warm-up step)\n" %
            (NSTEP, len(timed), sum(timed) / len(timed) / 1e6, min(timed) / 1e6, max(timed) / 1e6))
    f.write("# run-in launch of %d samples: %.3f ms = %.2f us per sample\n" % (STEADY, max(dur) / 1e6, max(dur) / 1e3 / STEADY))
    avg = sum(timed) / len(timed) * 1e-9
    f.write("# MFMA roofline from the profiler's average: %.1f TFLOP/s = %.4f of 2500 dense fp16\n" %
            (B * NSTEP * 1736704 / avg / 1e12, B * NSTEP * 1736704 / avg / 2.5e15))
    f.write(sh(sys.executable, "scripts/prof_summary.py", "kernel", f"{G}/prof3_kt/p_results.db"))

c = {}
for d in ("fetch", "write", "l2", "sq", "lds", "ldsbw"):
    c.update(pmc(f"{G}/prof3_{d}/p_results.db", KERN))
us = samples_total * B                       # utterance-samples of all profiled launches together
wgs = samples_total * (B // 48)              # workgroup-samples
hbm_r, hbm_w = 2 * c["FETCH_SIZE"] * 1024 / us, c["WRITE_SIZE"] * 1024 / us
alg_r, alg_w = 20 * 2 * 64 * 2 + 20 * 64 * 2 + 4, 20 * 64 * 2 + 4
lds_b = (c["SQ_INSTS_LDS_LOAD_BANDWIDTH"] + c["SQ_INSTS_LDS_STORE_BANDWIDTH"]) * 64 / wgs
launch_hbm = (hbm_r + hbm_w) * B * NSTEP
launch_lds = lds_b * (B // 48) * NSTEP
kms = line["roofline"]["kernel_ms"] * 1e-3
with open(f"{P}/r03_pmc_wg_b12288.txt", "w") as f:
    f.write(f"# round 3, wn::wavenet_wg<fp16,64,256,256,BT=3,EMBLDS=1,DUMP=0,RAW=0> at 12 288 utterances, steady state ({CMD})\n")
    f.write("# separate runs, --kernel-trace only: --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum | SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE "
            "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES | SQ_INSTS_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_ACTIVE_INST_LDS | "
            "SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_INSTS_SALU SQ_INSTS_SMEM\n")
    f.write("# counters are summed over every wavenet_wg launch of the run (%d samples of %d utterances) and divided by the work\n" % (samples_total, B))
    f.write("# HBM (FETCH_SIZE x2: gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md; units KB):\n")
    f.write("#   read  %.0f B per utterance-sample (algorithmic %d: conditioning 5120 + dilated taps 2560 + selector)  %.2fx\n" % (hbm_r, alg_r, hbm_r / alg_r))
    f.write("#   write %.0f B per utterance-sample (algorithmic %d: ring 2560 + sample)  %.2fx\n" % (hbm_w, alg_w, hbm_w / alg_w))
    f.write("#   per timed launch (%d samples): %.2f GB; at kernel_ms %.3f: %.2f TB/s = %.1f %% of 8 TB/s\n" %
            (NSTEP, launch_hbm / 1e9, kms * 1e3, launch_hbm / kms / 1e12, 100 * launch_hbm / kms / 8e12))
    f.write("# L2 hit rate %.1f %%\n" % (100 * c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])))
    f.write("# LDS (rocprof-reported): SQ_INSTS_LDS_{LOAD,STORE}_BANDWIDTH count 64-byte units (a wave-wide ds_read_b128 = 1024 B = 16 units:\n")
    f.write("#   LOAD_BANDWIDTH / INSTS_LDS_LOAD = %.1f, STORE_BANDWIDTH / INSTS_LDS_STORE = %.1f):\n" %
            (c["SQ_INSTS_LDS_LOAD_BANDWIDTH"] / c["SQ_INSTS_LDS_LOAD"], c["SQ_INSTS_LDS_STORE_BANDWIDTH"] / c["SQ_INSTS_LDS_STORE"]))
    f.write("#   %.0f LDS instructions and %.2f MB of LDS traffic per workgroup-sample (bench.py's algorithmic figure: 2.46 MB);\n" % (c["SQ_INSTS_LDS"] / wgs, lds_b / 1e6))
    f.write("#   per timed launch %.1f GB; at kernel_ms: %.1f TB/s = %.1f %% of the 157 TB/s LDS peak (256 CUs x 256 B/clk x 2.4 GHz)\n" %
            (launch_lds / 1e9, launch_lds / kms / 1e12, 100 * launch_lds / kms / 157.3e12))
    f.write("#   bank-conflict cycles / LDS-active cycles = %.1f %%\n" % (100 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]))
    f.write("# instruction issue (SQ counters are per shader engine): VALU : MFMA = %.2f, SALU : MFMA = %.2f, MFMA per wave and tile-sample = %.0f\n" %
            (c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"], c["SQ_INSTS_SALU"] / c["SQ_INSTS_MFMA"], c["SQ_INSTS_MFMA"] / (wgs * 4 * 3)))
    for k in sorted(c):
        f.write("%-32s %20.0f\n" % (k, c[k]))
json.dump({"batch": B, "samples": NSTEP, "hbm_bytes_per_launch": launch_hbm, "lds_bytes_per_launch": launch_lds,
           "hbm_read_bytes_per_utterance_sample": hbm_r, "hbm_write_bytes_per_utterance_sample": hbm_w,
           "lds_bytes_per_workgroup_sample": lds_b, "valu_per_mfma": c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"],
           "note": "rocprofv3 PMC, separate --pmc passes (scripts/prof_collect_r3.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md; "
                   "LDS bytes = (SQ_INSTS_LDS_LOAD_BANDWIDTH + SQ_INSTS_LDS_STORE_BANDWIDTH) x 64 B; wn::wavenet_wg<BT=3>, steady state"},
          open(f"{P}/traffic_r03.json", "w"), indent=1)
json.dump(line, open(f"{P}/r03_bench_line_under_rocprof_b12288.json", "w"))

# two tiles per workgroup
l2 = json.load(open(f"{G}/prof3_bench_line_b8192.json"))
d2 = sorted(dispatches(f"{G}/prof3_kt_b8192/p_results.db", KERN))[:-1]
with open(f"{P}/r03_kernel_trace_stats_wg_b8192.txt", "w") as f:
    f.write("# round 3: python bench.py --batch 8192 --steps 5 --warmup 1 --no-cpu-baseline --no-extras  under  rocprofv3 --kernel-trace --stats\n")
    f.write("# wn::wavenet_wg<fp16,64,256,256,BT=2,...>, steady state, %d samples per launch; bench.py: kernel_ms %.3f, khz_per_utterance %.2f, roofline.frac %.4f\n" %
            (l2["config"]["samples_per_step"], l2["roofline"]["kernel_ms"], l2["khz_per_utterance"], l2["roofline"]["frac"]))
    f.write("# wavenet_wg launches of that size: n=%d avg %.3f ms min %.3f ms\n" % (len(d2), sum(d2) / len(d2) / 1e6, min(d2) / 1e6))
    f.write(sh(sys.executable, "scripts/prof_summary.py", "kernel", f"{G}/prof3_kt_b8192/p_results.db"))
with open(f"{P}/r03_kernel_trace_stats_chain_c4.txt", "w") as f:
    f.write("# round 3: python scripts/nv_wavenet_perf.py -r 128 -s 256 -a 256 -l 30 -b 8 -m 3 -n 4096 -t 2048  under  rocprofv3 --kernel-trace --stats\n")
    f.write("# BASELINE config C4 on the multi-CU chain (16 workgroups, weights resident); every chain launch is bracketed by the state snapshot,\n")
    f.write("# chain_restore_kernel, the gated wavenet_wg fallback and chain_settle_kernel (empty launches when the chain completes), which the\n")
    f.write("# profiler serialises: 21.5 kHz under rocprofv3, 26.6 kHz without (bench.py reference_definition.C4)\n")
    for ln in open(f"{G}/prof3_kt_c4.log"):
        if "kernel:" in ln or "Sample rate" in ln:
            f.write("# " + ln)
    f.write("\n".join(sh(sys.executable, "scripts/prof_summary.py", "kernel", f"{G}/prof3_kt_c4/p_results.db").split("\n")[:14]) + "\n")
with open(f"{P}/r03_isa_mix.txt", "w") as f:
    f.write("# round 3: static instruction mix of the shipped gfx950 code object (scripts/isa_stats.py mix / regs; no GPU needed)\n")
    for k in ("wavenet_wg<true, 64, 256, 256, 3, true, false, 0>", "wavenet_wg<true, 64, 256, 256, 2, true, false, 0>",
              "wavenet_wg<true, 64, 256, 256, 3, true, false, 2>"):
        f.write(sh(sys.executable, "scripts/isa_stats.py", "mix", k))
    f.write("\n# registers / scratch of every kernel of the C3 fp16 instantiation\n")
    f.write(sh(sys.executable, "scripts/isa_stats.py", "regs"))
print(open(f"{P}/r03_pmc_wg_b12288.txt").read()[:3000])
print(open(f"{P}/r03_kernel_trace_stats_wg_b12288.txt").read()[:1500])
