#!/usr/bin/env python3
"""Authoring side of the round-6 profiles: turns the per-pass JSONs that scripts/prof_collect_r6.sh leaves in gpurun_out/
(prof6*.json: per-kernel durations and counter sums, reduced on the GPU box by scripts/prof_extract.py) into the tracked files under
profiles/ (r06_*), plus traffic_r06.json -- which names the kernel and the sha256 of its instruction stream (scripts/isa_stats.py sha), so
that bench.py drops the counters once the device code changes."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import isa_stats  # noqa: E402

G, P = "gpurun_out", "profiles"
STEADY = 640
FLOP = 1736704                       # per utterance-sample, C3
FLOP_FEAT = FLOP + 2 * 20 * 128 * 80  # + the conditioning GEMM (2R x n_cond per layer) when it is computed in the kernel
PEAK = 2.5e15


def load(name):
    return json.load(open(f"{G}/{name}.json"))


def have(name):
    return os.path.exists(f"{G}/{name}.json")


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


def stats_table(doc, top=12):
    rows = sorted(((demangle(k), v) for k, v in doc["kernels"].items()), key=lambda kv: -kv[1]["total_ns"])
    tot = sum(v["total_ns"] for _, v in rows)
    out = ["%-100s %6s %14s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "pct")]
    for k, v in rows[:top]:
        out.append("%-100s %6d %14d %12d %6.2f%%" % (k[:100], v["calls"], v["total_ns"], v["total_ns"] // v["calls"], 100.0 * v["total_ns"] / tot))
        out.append("    grid=%d wg=%d lds=%d B vgpr=%d agpr=%d sgpr=%d scratch=%d B/lane" %
                   (v["grid"], v["wg"], v["lds"], v["vgpr"], v["agpr"], v["sgpr"], v["scratch"]))
    return "\n".join(out) + "\n"


def issue_lines(c, f):
    wc = c.get("SQ_WAVE_CYCLES")
    if not wc:
        return
    f.write("# issue picture of a wave (fractions of SQ_WAVE_CYCLES; one wave per SIMD):\n")
    for k, what in (("SQ_WAIT_ANY", "parked in s_waitcnt / s_barrier"), ("SQ_WAIT_INST_ANY", "waiting to issue (dependency / pipe busy)"),
                    ("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_ACTIVE_INST_VALU", "  of which VALU + MFMA issue"),
                    ("SQ_ACTIVE_INST_LDS", "  LDS issue"), ("SQ_ACTIVE_INST_VMEM", "  vector-memory issue"),
                    ("SQ_ACTIVE_INST_SCA", "  scalar issue"), ("SQ_ACTIVE_INST_MISC", "  other issue")):
        if k in c:
            f.write("#   %-22s %5.1f %%   %s\n" % (k, 100.0 * c[k] / wc, what))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        f.write("#   matrix pipe busy: SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES) = %.1f %% of the clocks a wave (= a SIMD) was resident\n" %
                (100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * wc)))
        if "GRBM_GUI_ACTIVE" in c:
            f.write("#   ... / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) = %.1f %% of every SIMD clock of the launches\n" %
                    (100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024)))


def wg_launch(tag, B, bt, label, alg_r, alg_w, alg_what, out_trace, out_pmc, out_traffic=None):
    """a wavenet_wg launch shape of bench.py --batch B under the profiler (packed conditioning; bt tiles per workgroup)"""
    KERN = "wavenet_wgI"
    line = json.load(open(f"{G}/{tag}_bench_line.json"))
    NSTEP = line["config"]["samples_per_step"]
    kname = line["roofline"]["kernel"]
    assert ("BT=%d" % bt) in kname, kname
    doc = load(f"{tag}_kt")
    name, k = kernel_of(doc, KERN)
    dur = k["durations_ns"]
    timed = sorted(dur)[:-1]                 # the run also holds ONE launch of STEADY samples (the untimed run-in)
    avg = sum(timed) / len(timed) * 1e-9
    cmd = "python bench.py --batch %d --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-power" % B
    wg_per_sample = (B // 16 + bt - 1) // bt
    with open(out_trace, "w") as f:
        f.write(f"# round 6: {cmd}  under  rocprofv3 --kernel-trace --stats\n# {kname}: {label}\n")
        f.write(f"# every timed launch generates samples {STEADY}..{STEADY + NSTEP - 1} of {B} utterances (steady state: all dilated taps live) on {wg_per_sample} workgroups;\n")
        f.write(f"# the run also holds ONE launch of {STEADY} samples (the untimed run-in from sample 0).\n")
        f.write("# bench.py's own line of this run: value %.1f M samples/s, kernel_ms %.3f (HIP events), khz_per_utterance %.2f, roofline.frac %.4f, shader clock %s GHz\n" %
                (line["value"] / 1e6, line["roofline"]["kernel_ms"], line["khz_per_utterance"], line["roofline"]["frac"], line["roofline"].get("shader_clock_ghz")))
        f.write("# launches of %d samples: n=%d avg %.3f ms min %.3f ms max %.3f ms\n" % (NSTEP, len(timed), avg * 1e3, min(timed) / 1e6, max(timed) / 1e6))
        f.write("# MFMA accounting from the profiler's average: %.1f TFLOP/s = %.4f of 2500 dense fp16 (minimum launch: %.4f); flops per utterance-sample: %d\n" %
                (B * NSTEP * FLOP / avg / 1e12, B * NSTEP * FLOP / avg / PEAK, B * NSTEP * FLOP / (min(timed) * 1e-9) / PEAK, FLOP))
        f.write(stats_table(doc))
    c = {}
    for d in ("fetch", "write", "sq", "ldsbw", "issue", "busy"):
        if have(f"{tag}_{d}"):
            c.update(pmc_of(load(f"{tag}_{d}"), KERN))
    samples_total = STEADY + NSTEP * len(timed)
    us = samples_total * B
    wgs = samples_total * wg_per_sample
    hbm_r, hbm_w = 2 * c["FETCH_SIZE"] * 1024 / us, c["WRITE_SIZE"] * 1024 / us
    lds_b = (c["SQ_INSTS_LDS_LOAD_BANDWIDTH"] + c["SQ_INSTS_LDS_STORE_BANDWIDTH"]) * 64 / wgs
    launch_hbm = (hbm_r + hbm_w) * B * NSTEP
    launch_lds = lds_b * wg_per_sample * NSTEP
    kms = line["roofline"]["kernel_ms"] * 1e-3
    with open(out_pmc, "w") as f:
        f.write(f"# round 6, {kname} at {B} utterances ({wg_per_sample} workgroups of {bt} tiles), steady state ({cmd})\n")
        f.write("# separate runs, --kernel-trace only (scripts/prof_collect_r6.sh): --pmc FETCH_SIZE | WRITE_SIZE | SQ_LDS_BANK_CONFLICT ... | "
                "SQ_INSTS_LDS_*_BANDWIDTH ... | SQ_WAIT_* ... | SQ_VALU_MFMA_BUSY_CYCLES ...\n")
        f.write("# counters are summed over every wavenet_wg launch of the run (%d samples of %d utterances) and divided by the work\n" % (samples_total, B))
        f.write("# HBM (FETCH_SIZE x2: gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md; units KB):\n")
        f.write("#   read  %.0f B per utterance-sample (algorithmic %d: %s)  %.2fx\n" % (hbm_r, alg_r, alg_what, hbm_r / alg_r))
        f.write("#   write %.0f B per utterance-sample (algorithmic %d: ring + sample)  %.2fx\n" % (hbm_w, alg_w, hbm_w / alg_w))
        f.write("#   per timed launch (%d samples): %.2f GB; at kernel_ms %.3f: %.2f TB/s = %.1f %% of 8 TB/s\n" %
                (NSTEP, launch_hbm / 1e9, kms * 1e3, launch_hbm / kms / 1e12, 100 * launch_hbm / kms / 8e12))
        f.write("# LDS (SQ_INSTS_LDS_{LOAD,STORE}_BANDWIDTH in 64-byte units): %.0f LDS instructions and %.2f MB per workgroup-sample; per timed launch %.1f GB = %.1f TB/s = %.1f %% of the 157 TB/s LDS peak\n" %
                (c["SQ_INSTS_LDS"] / wgs, lds_b / 1e6, launch_lds / 1e9, launch_lds / kms / 1e12, 100 * launch_lds / kms / 157.3e12))
        f.write("#   bank-conflict cycles / LDS-active cycles = %.1f %%; LDS-active cycles / (4 x SQ_WAVE_CYCLES) = %.1f %% of a wave's resident clocks\n" %
                (100 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 100 * c["SQ_LDS_IDX_ACTIVE"] / (4.0 * c["SQ_WAVE_CYCLES"])))
        f.write("# instruction counts: VALU : MFMA = %.2f, SALU : MFMA = %.2f, MFMA per wave and tile-sample = %.0f\n" %
                (c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"], c["SQ_INSTS_SALU"] / c["SQ_INSTS_MFMA"], c["SQ_INSTS_MFMA"] / (samples_total * (B // 16) * 4)))
        issue_lines(c, f)
        for kk in sorted(c):
            f.write("%-32s %20.0f\n" % (kk, c[kk]))
    if out_traffic:
        sub = isa_stats.kernel_sub_of(kname)
        sha = isa_stats.kernel_sha("inst_64_256_256_p16.o", sub)
        assert sha and sha == line["roofline"].get("kernel_sha256"), ("the library here is not the one that was profiled", sha, line["roofline"].get("kernel_sha256"))
        json.dump({"batch": B, "samples": NSTEP, "kernel": kname, "kernel_sha256": sha,
                   "hbm_bytes_per_launch": launch_hbm, "lds_bytes_per_launch": launch_lds,
                   "hbm_read_bytes_per_utterance_sample": hbm_r, "hbm_write_bytes_per_utterance_sample": hbm_w,
                   "lds_bytes_per_workgroup_sample": lds_b, "valu_per_mfma": c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"],
                   "lds_bank_conflict_frac": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"],
                   "wave_parked_frac": c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], "wave_issue_stall_frac": c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"],
                   "wave_issuing_frac": c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"],
                   "note": "rocprofv3 PMC, separate --pmc passes (scripts/prof_collect_r6.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md; LDS bytes = "
                           "(SQ_INSTS_LDS_LOAD_BANDWIDTH + SQ_INSTS_LDS_STORE_BANDWIDTH) x 64 B; steady state; valid for the kernel whose "
                           "instruction stream hashes to kernel_sha256 (scripts/isa_stats.py sha)"},
                  open(out_traffic, "w"), indent=1)
    json.dump(line, open(out_trace.replace("kernel_trace_stats", "bench_line_under_rocprof").replace(".txt", ".json"), "w"))
    return hbm_r, hbm_w


# ring bytes that still go to HBM: the two d = 1 layers of C3 (of 20) keep their slots in LDS at three / four tiles per workgroup
RING_L = 18
if have("prof6a_kt"):
    r = wg_launch("prof6a", 12288, 3, "conditioning pre-packed in fragment order, three tiles per workgroup on every CU (the timed launch of bench.py)",
                  20 * 2 * 64 * 2 + RING_L * 64 * 2 + 4, RING_L * 64 * 2 + 4, "conditioning 5120 + dilated taps of the 18 layers with d > 1: 2304 + selector",
                  f"{P}/r06_kernel_trace_stats_wg_b12288.txt", f"{P}/r06_pmc_wg_b12288.txt", f"{P}/traffic_r06.json")
    print("12288 (BT=3): HBM read %.0f write %.0f B per utterance-sample" % r)
if have("prof6b_kt"):
    r = wg_launch("prof6b", 13824, 4, "conditioning pre-packed, FOUR tiles per workgroup on 216 CUs (the largest real-time batch's launch shape)",
                  20 * 2 * 64 * 2 + RING_L * 64 * 2 + 4, RING_L * 64 * 2 + 4, "conditioning 5120 + dilated taps of the 18 layers with d > 1: 2304 + selector",
                  f"{P}/r06_kernel_trace_stats_wg_b13824.txt", f"{P}/r06_pmc_wg_b13824.txt", f"{P}/traffic_r06_b13824.json")
    print("13824 (BT=4): HBM read %.0f write %.0f B per utterance-sample" % r)
# ---- C: the chain at C4, five tiles per chain (HOIST) ---------------------------------------------------------------------------------------
if have("prof6c_kt"):
    doc = load("prof6c_kt")
    line = json.load(open(f"{G}/prof6c_line.json"))
    ks = [(k, v) for k, v in doc["kernels"].items() if "wavenet_chain" in k]
    FLOP4 = 7143424
    B4 = line["batch"]
    with open(f"{P}/r06_chain_c4_five_tiles_per_chain.txt", "w") as f:
        f.write("# round 6: python scripts/gpu_r6_chain.py prof 5  under  rocprofv3 --kernel-trace --stats (+ FETCH_SIZE / WRITE_SIZE passes in their own runs)\n")
        f.write("# %s\n" % line["kernel"])
        f.write("# 16 chains x 16 CUs (15 stages of 2 layers + head), FIVE tiles per chain = %d utterances per GPU; the stages request a unit's packed\n"
                "# conditioning of both layers up front (HOIST).  The script's own line: steady-state kHz per utterance %s (samples 640..1663)\n" % (B4, line["steady_khz"]))
        for name, k in ks:
            d = sorted(k["durations_ns"])
            f.write("# %s: %d launches; the longest (1024 steady-state samples x %d utterances): %.2f ms = %.2f us per sample = %.2f kHz per utterance; "
                    "%.1f TFLOP/s = %.4f of 2500 dense fp16\n" % (demangle(name)[:70], k["calls"], B4, d[-1] / 1e6, d[-1] / 1e3 / 1024, 1024 / (d[-1] / 1e6),
                                                                 B4 * 1024 * FLOP4 / (d[-1] * 1e-9) / 1e12, B4 * 1024 * FLOP4 / (d[-1] * 1e-9) / PEAK))
        f.write(stats_table(doc, 8))
        c = {}
        for dd in ("fetch", "write"):
            if have(f"prof6c_{dd}"):
                c.update(pmc_of(load(f"prof6c_{dd}"), "wavenet_chain"))
        if "FETCH_SIZE" in c:
            tot_us = B4 * 2 * (640 + 1024)          # two probes, each a 640-sample run-in and 1024 timed samples
            f.write("# HBM over all chain launches of the run (%d utterance-samples): read %.0f B, written %.0f B per utterance-sample\n" %
                    (tot_us, 2 * c["FETCH_SIZE"] * 1024 / tot_us, c["WRITE_SIZE"] * 1024 / tot_us))
            f.write("#   (algorithmic: conditioning 15 360 + dilated taps 7 680 read, ring 7 680 written; the rest is the hand-off -- 8-byte {value, tag} granules,\n"
                    "#    1 KiB of x per utterance and stage + 2 KiB of skip sums over 15 + 16 hops -- as in round 5)\n")
        for kk in sorted(c):
            f.write("%-32s %20.0f\n" % (kk, c[kk]))
print("done")
