#!/usr/bin/env python3
"""Authoring side of the round-4 profiles: turns the per-pass JSONs that scripts/prof_collect_r4.sh leaves in gpurun_out/
(prof4_*.json: per-kernel durations and counter sums, reduced on the GPU box by scripts/prof_extract.py) into the tracked
files under profiles/ (r04_*), plus the static instruction mix of the shipped code object (scripts/isa_stats.py)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
G, P = "gpurun_out", "profiles"
NSTEP, STEADY = 256, 640
FLOP = 1736704                       # per utterance-sample, C3
PEAK = 2.5e15


def sh(*a):
    return subprocess.run(list(a), capture_output=True, text=True).stdout


def load(name):
    return json.load(open(f"{G}/{name}.json"))


def kernel_of(doc, sub):
    ks = [(k, v) for k, v in doc["kernels"].items() if sub in k]
    assert len(ks) == 1, [k for k, _ in ks]
    return ks[0]


def pmc_of(doc, sub):
    out = {}
    for k, v in doc["pmc"].items():
        if sub in k:
            for c, x in v.items():
                out[c] = out.get(c, 0.0) + x["sum"]
    return out


def demangle(name):
    return subprocess.run(["c++filt", name[:-3] if name.endswith(".kd") else name], capture_output=True, text=True).stdout.strip() or name


def stats_table(doc):
    rows = sorted(((demangle(k), v) for k, v in doc["kernels"].items()), key=lambda kv: -kv[1]["total_ns"])
    tot = sum(v["total_ns"] for _, v in rows)
    out = ["%-100s %6s %14s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "pct")]
    for k, v in rows[:12]:
        out.append("%-100s %6d %14d %12d %6.2f%%" % (k[:100], v["calls"], v["total_ns"], v["total_ns"] // v["calls"], 100.0 * v["total_ns"] / tot))
        out.append("    grid=%d wg=%d lds=%d B vgpr=%d agpr=%d sgpr=%d scratch=%d B/lane" %
                   (v["grid"], v["wg"], v["lds"], v["vgpr"], v["agpr"], v["sgpr"], v["scratch"]))
    return "\n".join(out) + "\n"


def trace_file(tag, sub, B, cmd, what, out):
    line = json.load(open(f"{G}/{tag}_bench_line.json"))
    NSTEP = line["config"]["samples_per_step"]          # (16 384 utterances time 64-sample steps: bench.py keeps the packed conditioning under 60 GB)
    doc = load(f"{tag}_kt")
    name, k = kernel_of(doc, sub)
    dur = k["durations_ns"]
    timed = sorted(dur)[:-1]                 # the run also holds ONE launch of STEADY samples (the untimed run-in)
    avg = sum(timed) / len(timed) * 1e-9
    with open(out, "w") as f:
        f.write(f"# round 4: {cmd}  under  rocprofv3 --kernel-trace --stats\n")
        f.write(f"# {what}\n")
        f.write(f"# every timed launch generates samples {STEADY}..{STEADY + NSTEP - 1} of {B} utterances (steady state: all dilated taps live);\n")
        f.write(f"# the run also holds ONE launch of {STEADY} samples (the untimed run-in from sample 0), listed separately below.\n")
        f.write("# bench.py's own line of this run: value %.1f M samples/s, kernel_ms %.3f (HIP events), khz_per_utterance %.2f, roofline.frac %.4f\n" %
                (line["value"] / 1e6, line["roofline"]["kernel_ms"], line["khz_per_utterance"], line["roofline"]["frac"]))
        f.write("# launches of %d samples: n=%d avg %.3f ms min %.3f ms max %.3f ms  (the first one after the run-in is the warm-up step)\n" %
                (NSTEP, len(timed), avg * 1e3, min(timed) / 1e6, max(timed) / 1e6))
        f.write("# run-in launch of %d samples: %.3f ms = %.2f us per sample\n" % (STEADY, max(dur) / 1e6, max(dur) / 1e3 / STEADY))
        f.write("# MFMA roofline from the profiler's average: %.1f TFLOP/s = %.4f of 2500 dense fp16 (minimum launch: %.4f)\n" %
                (B * NSTEP * FLOP / avg / 1e12, B * NSTEP * FLOP / avg / PEAK, B * NSTEP * FLOP / (min(timed) * 1e-9) / PEAK))
        f.write(stats_table(doc))
    return line, len(timed), avg, NSTEP


def issue_lines(c, f, waves_per_cu=4):
    """what the SQ busy / wait counters say (MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES)"""
    wc = c.get("SQ_WAVE_CYCLES")
    if not wc:
        return
    f.write("# issue picture of a wave (fractions of SQ_WAVE_CYCLES; the kernel runs one wave per SIMD):\n")
    for k, what in (("SQ_WAIT_ANY", "parked in s_waitcnt / s_barrier"), ("SQ_WAIT_INST_ANY", "waiting to issue (dependency / pipe busy)"),
                    ("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_ACTIVE_INST_VALU", "  of which VALU + MFMA issue"),
                    ("SQ_ACTIVE_INST_LDS", "  LDS issue"), ("SQ_ACTIVE_INST_VMEM", "  vector-memory issue"),
                    ("SQ_ACTIVE_INST_SCA", "  scalar issue"), ("SQ_ACTIVE_INST_MISC", "  other issue")):
        if k in c:
            f.write("#   %-22s %5.1f %%   %s\n" % (k, 100.0 * c[k] / wc, what))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        # units, from the counters themselves: SQ_VALU_MFMA_BUSY_CYCLES = 16 x SQ_INSTS_MFMA exactly (clocks: a 16x16x32 f16 MFMA holds the
        # pipe for 16), SQ_WAVE_CYCLES x 4 = the clocks the waves were resident (quad-cycles, MI355X_MICROARCH.md), GRBM_GUI_ACTIVE =
        # clocks per XCD
        f.write("#   matrix pipe busy: SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES) = %.1f %% of the clocks a wave (= a SIMD) was resident\n" %
                (100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * wc)))
        if "GRBM_GUI_ACTIVE" in c and "SQ_INSTS_MFMA" in c:
            f.write("#     (check: SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA = %.2f clk per MFMA)\n" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_INSTS_MFMA"]))
        if "GRBM_GUI_ACTIVE" in c:
            gpu_clk = c["GRBM_GUI_ACTIVE"] / 8.0            # summed over the 8 XCDs
            f.write("#   ... / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) = %.1f %% of every SIMD clock of the launches (the MFMA roofline at the clock granted)\n" %
                    (100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gpu_clk * 1024)))
    if "SQ_ACTIVE_INST_ANY" in c and "SQ_INSTS_VALU" in c:
        pass


# ---- the headline kernel at 12 288 utterances ------------------------------------------------------------------------------
B = 12288
CMD = f"python bench.py --batch {B} --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
KERN = "wavenet_wgI"          # (this rocprofv3 stores mangled names)
line, nl, avg, NSTEP = trace_file("prof4", KERN, B, CMD,
                           "wn::wavenet_wg<fp16,64,256,256,BT=3,EMBLDS=1,DUMP=0,RAW=0>: three tiles of 16 utterances per workgroup, 256 workgroups",
                           f"{P}/r04_kernel_trace_stats_wg_b12288.txt")
c = {}
for d in ("fetch", "write", "sq", "ldsbw", "issue", "busy"):
    c.update(pmc_of(load(f"prof4_{d}"), KERN))
samples_total = STEADY + NSTEP * nl
us = samples_total * B
wgs = samples_total * (B // 48)
hbm_r, hbm_w = 2 * c["FETCH_SIZE"] * 1024 / us, c["WRITE_SIZE"] * 1024 / us
alg_r, alg_w = 20 * 2 * 64 * 2 + 20 * 64 * 2 + 4, 20 * 64 * 2 + 4
lds_b = (c["SQ_INSTS_LDS_LOAD_BANDWIDTH"] + c["SQ_INSTS_LDS_STORE_BANDWIDTH"]) * 64 / wgs
launch_hbm = (hbm_r + hbm_w) * B * NSTEP
launch_lds = lds_b * (B // 48) * NSTEP
kms = line["roofline"]["kernel_ms"] * 1e-3
with open(f"{P}/r04_pmc_wg_b12288.txt", "w") as f:
    f.write(f"# round 4, wn::wavenet_wg<fp16,64,256,256,BT=3,EMBLDS=1,DUMP=0,RAW=0> at 12 288 utterances, steady state ({CMD})\n")
    f.write("# separate runs, --kernel-trace only (scripts/prof_collect_r4.sh): --pmc FETCH_SIZE | WRITE_SIZE | SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE "
            "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES | SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_INSTS_SALU SQ_INSTS_SMEM "
            "SQ_INSTS_LDS | SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE | "
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE\n")
    f.write("# counters are summed over every wavenet_wg launch of the run (%d samples of %d utterances) and divided by the work\n" % (samples_total, B))
    f.write("# HBM (FETCH_SIZE x2: gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md; units KB):\n")
    f.write("#   read  %.0f B per utterance-sample (algorithmic %d: conditioning 5120 + dilated taps 2560 + selector)  %.2fx\n" % (hbm_r, alg_r, hbm_r / alg_r))
    f.write("#   write %.0f B per utterance-sample (algorithmic %d: ring 2560 + sample)  %.2fx\n" % (hbm_w, alg_w, hbm_w / alg_w))
    f.write("#   per timed launch (%d samples): %.2f GB; at kernel_ms %.3f: %.2f TB/s = %.1f %% of 8 TB/s\n" %
            (NSTEP, launch_hbm / 1e9, kms * 1e3, launch_hbm / kms / 1e12, 100 * launch_hbm / kms / 8e12))
    f.write("# LDS (rocprof-reported, SQ_INSTS_LDS_{LOAD,STORE}_BANDWIDTH in 64-byte units): %.0f LDS instructions and %.2f MB per workgroup-sample;\n" %
            (c["SQ_INSTS_LDS"] / wgs, lds_b / 1e6))
    f.write("#   per timed launch %.1f GB; at kernel_ms: %.1f TB/s = %.1f %% of the 157 TB/s LDS peak (256 CUs x 256 B/clk x 2.4 GHz)\n" %
            (launch_lds / 1e9, launch_lds / kms / 1e12, 100 * launch_lds / kms / 157.3e12))
    f.write("#   bank-conflict cycles / LDS-active cycles = %.1f %%\n" % (100 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]))
    f.write("# instruction counts: VALU : MFMA = %.2f, SALU : MFMA = %.2f, MFMA per wave and tile-sample = %.0f\n" %
            (c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"], c["SQ_INSTS_SALU"] / c["SQ_INSTS_MFMA"], c["SQ_INSTS_MFMA"] / (wgs * 4 * 3)))
    issue_lines(c, f)
    f.write("# shader clock during the timed launches (workgroup 0's s_memtime / wall clock, bench.py roofline.shader_clock_ghz): %s GHz\n" %
            line["roofline"].get("shader_clock_ghz"))
    for k in sorted(c):
        f.write("%-32s %20.0f\n" % (k, c[k]))
json.dump({"batch": B, "samples": NSTEP, "hbm_bytes_per_launch": launch_hbm, "lds_bytes_per_launch": launch_lds,
           "hbm_read_bytes_per_utterance_sample": hbm_r, "hbm_write_bytes_per_utterance_sample": hbm_w,
           "lds_bytes_per_workgroup_sample": lds_b, "valu_per_mfma": c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"],
           "wave_parked_frac": c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], "wave_issue_stall_frac": c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"],
           "wave_issuing_frac": c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"],
           "note": "rocprofv3 PMC, separate --pmc passes (scripts/prof_collect_r4.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md; "
                   "LDS bytes = (SQ_INSTS_LDS_LOAD_BANDWIDTH + SQ_INSTS_LDS_STORE_BANDWIDTH) x 64 B; wn::wavenet_wg<BT=3>, steady state"},
          open(f"{P}/traffic_r04.json", "w"), indent=1)
json.dump(line, open(f"{P}/r04_bench_line_under_rocprof_b12288.json", "w"))

# ---- wavenet_bcast at 16 384 utterances ------------------------------------------------------------------------------------
B2 = 16384
CMD2 = f"python bench.py --batch {B2} --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
K2 = "wavenet_bcastI"
line2, nl2, avg2, NS2 = trace_file("prof4_bc", K2, B2, CMD2,
                              "wn::wavenet_bcast<fp16,64,256,256,BTW=1,EMBLDS=1,DUMP=0>: every wave one tile (four per workgroup), 256 workgroups; "
                              "not real time (the engine's choice between three and four tiles per CU)",
                              f"{P}/r04_kernel_trace_stats_bcast_b16384.txt")
c2 = {}
for d in ("ldsbw", "issue"):
    c2.update(pmc_of(load(f"prof4_bc_{d}"), K2))
st2 = STEADY + NS2 * nl2
wg2 = st2 * (B2 // 64)
with open(f"{P}/r04_pmc_bcast_b16384.txt", "w") as f:
    f.write(f"# round 4, wn::wavenet_bcast<fp16,64,256,256,BTW=1,EMBLDS=1,DUMP=0> at 16 384 utterances, steady state ({CMD2})\n")
    lds2 = (c2["SQ_INSTS_LDS_LOAD_BANDWIDTH"] + c2["SQ_INSTS_LDS_STORE_BANDWIDTH"]) * 64 / wg2
    k2 = line2["roofline"]["kernel_ms"] * 1e-3
    f.write("# LDS (rocprof-reported): %.2f MB per workgroup-sample (every wave reads every weight fragment: 4 x 1.7 MB + the copies' 1.7 MB arrive by DMA);\n" % (lds2 / 1e6))
    f.write("#   per timed launch (%d samples) %.1f GB; at kernel_ms %.3f: %.1f TB/s = %.1f %% of the 157 TB/s LDS peak\n" %
            (NS2, lds2 * (B2 // 64) * NS2 / 1e9, k2 * 1e3, lds2 * (B2 // 64) * NS2 / k2 / 1e12, 100 * lds2 * (B2 // 64) * NS2 / k2 / 157.3e12))
    f.write("#   bank-conflict cycles / LDS-active cycles = %.1f %%\n" % (100 * c2["SQ_LDS_BANK_CONFLICT"] / c2["SQ_LDS_IDX_ACTIVE"]))
    f.write("# VALU : MFMA = %.2f, MFMA per wave and tile-sample = %.0f\n" % (c2["SQ_INSTS_VALU"] / c2["SQ_INSTS_MFMA"], c2["SQ_INSTS_MFMA"] / (wg2 * 4)))
    issue_lines(c2, f)
    f.write("# shader clock during the timed launches: %s GHz\n" % line2["roofline"].get("shader_clock_ghz"))
    for k in sorted(c2):
        f.write("%-32s %20.0f\n" % (k, c2[k]))

with open(f"{P}/r04_isa_mix.txt", "w") as f:
    f.write("# round 4: static instruction mix of the shipped gfx950 code object (scripts/isa_stats.py mix / regs; no GPU needed)\n")
    for k in ("wavenet_wg<true, 64, 256, 256, 3, true, false, 0>", "wavenet_wg<true, 64, 256, 256, 2, true, false, 0>",
              "wavenet_bcast<true, 64, 256, 256, 1, true, false>"):
        f.write(sh(sys.executable, "scripts/isa_stats.py", "mix", "inst_64_256_256_p16.o", k))
    f.write("\n# registers / scratch of every kernel of the C3 fp16 instantiation\n")
    f.write(sh(sys.executable, "scripts/isa_stats.py", "regs"))
if os.path.exists(f"{G}/r4_bench_default.log"):
    for ln in open(f"{G}/r4_bench_default.log"):
        if ln.startswith("{"):
            json.dump(json.loads(ln), open(f"{P}/r04_bench_default_run.json", "w"))
print(open(f"{P}/r04_pmc_wg_b12288.txt").read()[:3500])
print(open(f"{P}/r04_kernel_trace_stats_wg_b12288.txt").read()[:1200])
print(open(f"{P}/r04_pmc_bcast_b16384.txt").read()[:2500])
