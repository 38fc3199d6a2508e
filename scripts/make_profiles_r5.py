#!/usr/bin/env python3
"""Authoring side of the round-5 profiles: turns the per-pass JSONs that scripts/prof_collect_r5.sh leaves in gpurun_out/
(prof5*.json: per-kernel durations and counter sums, reduced on the GPU box by scripts/prof_extract.py) into the tracked files under
profiles/ (r05_*), plus traffic_r05.json -- which names the kernel and the sha256 of its instruction stream (scripts/isa_stats.py sha), so
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


def wg_launch(tag, label, raw, flop, alg_r, alg_w, alg_what, out_trace, out_pmc, out_traffic=None):
    """the wavenet_wg launch shape of bench.py --batch 12288 under the profiler (packed conditioning, or computed from the features)"""
    B = 12288
    KERN = "wavenet_wgI"
    line = json.load(open(f"{G}/{tag}_bench_line.json"))
    NSTEP = line["config"]["samples_per_step"]
    kname = line["roofline"]["kernel"]
    assert ("RAW=%d" % raw) in kname, kname
    doc = load(f"{tag}_kt")
    name, k = kernel_of(doc, KERN)
    dur = k["durations_ns"]
    timed = sorted(dur)[:-1]                 # the run also holds ONE launch of STEADY samples (the untimed run-in)
    avg = sum(timed) / len(timed) * 1e-9
    cmd = "python bench.py --batch 12288 --steps 5 --warmup 1 --no-cpu-baseline --no-extras" + (" --conditioning features" if raw == 3 else "")
    with open(out_trace, "w") as f:
        f.write(f"# round 5: {cmd}  under  rocprofv3 --kernel-trace --stats\n# {kname}: {label}\n")
        f.write(f"# every timed launch generates samples {STEADY}..{STEADY + NSTEP - 1} of {B} utterances (steady state: all dilated taps live);\n")
        f.write(f"# the run also holds ONE launch of {STEADY} samples (the untimed run-in from sample 0).\n")
        f.write("# bench.py's own line of this run: value %.1f M samples/s, kernel_ms %.3f (HIP events), khz_per_utterance %.2f, roofline.frac %.4f, shader clock %s GHz\n" %
                (line["value"] / 1e6, line["roofline"]["kernel_ms"], line["khz_per_utterance"], line["roofline"]["frac"], line["roofline"].get("shader_clock_ghz")))
        f.write("# launches of %d samples: n=%d avg %.3f ms min %.3f ms max %.3f ms\n" % (NSTEP, len(timed), avg * 1e3, min(timed) / 1e6, max(timed) / 1e6))
        f.write("# MFMA roofline from the profiler's average: %.1f TFLOP/s = %.4f of 2500 dense fp16 (minimum launch: %.4f); flops per utterance-sample: %d\n" %
                (B * NSTEP * flop / avg / 1e12, B * NSTEP * flop / avg / PEAK, B * NSTEP * flop / (min(timed) * 1e-9) / PEAK, flop))
        f.write(stats_table(doc))
    c = {}
    for d in ("fetch", "write", "sq", "ldsbw", "issue", "busy"):
        if have(f"{tag}_{d}"):
            c.update(pmc_of(load(f"{tag}_{d}"), KERN))
    samples_total = STEADY + NSTEP * len(timed)
    us = samples_total * B
    wgs = samples_total * (B // 48)
    hbm_r, hbm_w = 2 * c["FETCH_SIZE"] * 1024 / us, c["WRITE_SIZE"] * 1024 / us
    lds_b = (c["SQ_INSTS_LDS_LOAD_BANDWIDTH"] + c["SQ_INSTS_LDS_STORE_BANDWIDTH"]) * 64 / wgs
    launch_hbm = (hbm_r + hbm_w) * B * NSTEP
    launch_lds = lds_b * (B // 48) * NSTEP
    kms = line["roofline"]["kernel_ms"] * 1e-3
    with open(out_pmc, "w") as f:
        f.write(f"# round 5, {kname} at 12 288 utterances, steady state ({cmd})\n")
        f.write("# separate runs, --kernel-trace only (scripts/prof_collect_r5.sh): --pmc FETCH_SIZE | WRITE_SIZE | SQ_LDS_BANK_CONFLICT ... | "
                "SQ_INSTS_LDS_*_BANDWIDTH ... | SQ_WAIT_* ... | SQ_VALU_MFMA_BUSY_CYCLES ...\n")
        f.write("# counters are summed over every wavenet_wg launch of the run (%d samples of %d utterances) and divided by the work\n" % (samples_total, B))
        f.write("# HBM (FETCH_SIZE x2: gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md; units KB):\n")
        f.write("#   read  %.0f B per utterance-sample (algorithmic %d: %s)  %.2fx\n" % (hbm_r, alg_r, alg_what, hbm_r / alg_r))
        f.write("#   write %.0f B per utterance-sample (algorithmic %d: ring 2560 + sample)  %.2fx\n" % (hbm_w, alg_w, hbm_w / alg_w))
        f.write("#   per timed launch (%d samples): %.2f GB; at kernel_ms %.3f: %.2f TB/s = %.1f %% of 8 TB/s\n" %
                (NSTEP, launch_hbm / 1e9, kms * 1e3, launch_hbm / kms / 1e12, 100 * launch_hbm / kms / 8e12))
        f.write("# LDS (SQ_INSTS_LDS_{LOAD,STORE}_BANDWIDTH in 64-byte units): %.0f LDS instructions and %.2f MB per workgroup-sample; per timed launch %.1f GB = %.1f TB/s = %.1f %% of the 157 TB/s LDS peak\n" %
                (c["SQ_INSTS_LDS"] / wgs, lds_b / 1e6, launch_lds / 1e9, launch_lds / kms / 1e12, 100 * launch_lds / kms / 157.3e12))
        f.write("#   bank-conflict cycles / LDS-active cycles = %.1f %%; LDS-active cycles / (4 x SQ_WAVE_CYCLES) = %.1f %% of a wave's resident clocks\n" %
                (100 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 100 * c["SQ_LDS_IDX_ACTIVE"] / (4.0 * c["SQ_WAVE_CYCLES"])))
        f.write("# instruction counts: VALU : MFMA = %.2f, SALU : MFMA = %.2f, MFMA per wave and tile-sample = %.0f\n" %
                (c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"], c["SQ_INSTS_SALU"] / c["SQ_INSTS_MFMA"], c["SQ_INSTS_MFMA"] / (wgs * 4 * 3)))
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
                   "note": "rocprofv3 PMC, separate --pmc passes (scripts/prof_collect_r5.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md; LDS bytes = "
                           "(SQ_INSTS_LDS_LOAD_BANDWIDTH + SQ_INSTS_LDS_STORE_BANDWIDTH) x 64 B; steady state; valid for the kernel whose "
                           "instruction stream hashes to kernel_sha256 (scripts/isa_stats.py sha)"},
                  open(out_traffic, "w"), indent=1)
    json.dump(line, open(out_trace.replace("kernel_trace_stats", "bench_line_under_rocprof").replace(".txt", ".json"), "w"))
    return hbm_r, hbm_w


if have("prof5a_kt"):
    r = wg_launch("prof5a", "conditioning pre-packed in fragment order (the headline launch)", 0, FLOP, 20 * 2 * 64 * 2 + 20 * 64 * 2 + 4, 20 * 64 * 2 + 4,
                  "conditioning 5120 + dilated taps 2560 + selector", f"{P}/r05_kernel_trace_stats_wg_b12288.txt", f"{P}/r05_pmc_wg_b12288.txt",
                  f"{P}/traffic_r05.json")
    print("packed: HBM read %.0f write %.0f B per utterance-sample" % r)
if have("prof5b_kt"):
    r = wg_launch("prof5b", "conditioning computed in the kernel from the upsampled features", 3, FLOP_FEAT, 192 + 20 * 64 * 2 + 4, 20 * 64 * 2 + 4,
                  "features 192 + dilated taps 2560 + selector", f"{P}/r05_kernel_trace_stats_wg_features_b12288.txt", f"{P}/r05_pmc_wg_features_b12288.txt")
    print("features: HBM read %.0f write %.0f B per utterance-sample" % r)

# ---- C: the features-in loop: upsample_features_kernel + wavenet_wg<RAW=3> per chunk ----------------------------------------------------
if have("prof5c_kt"):
    doc = load("prof5c_kt")
    B, CH = 12288, 256
    with open(f"{P}/r05_features_in_loop.txt", "w") as f:
        f.write("# round 5: python scripts/gpu_r5_stream.py 12288  under  rocprofv3 --kernel-trace --stats (+ --pmc FETCH_SIZE | WRITE_SIZE in their own runs)\n")
        f.write("# the features-in loop (nvw_generate_stream): per chunk of 256 samples x 12 288 utterances, wn::upsample_features_kernel (the model's\n"
                "# ConvTranspose1d, window 1024 / stride 256, on MFMAs, into feature fragments) then wn::wavenet_wg<..,RAW=3> (conditioning computed in the kernel)\n")
        f.write("# the script's own output (wall clock around each variant, min of 3):\n")
        for ln in open(f"{G}/prof5c_stdout.txt"):
            f.write("#   " + ln)
        f.write(stats_table(doc, 8))
        for sub, what, units in (("upsample_features_kernel", "upsampling", None), ("pack_features_kernel", "mel frames -> fragments", None)):
            ks = [(k, v) for k, v in doc["kernels"].items() if sub in k]
            if not ks:
                continue
            name, k = ks[0]
            d = sorted(k["durations_ns"])
            f.write("# %s (%s): %d launches, median %.3f ms\n" % (sub, what, k["calls"], d[len(d) // 2] / 1e6))
        if have("prof5c_fetch") and have("prof5c_write"):
            fe, wr = pmc_of(load("prof5c_fetch"), "upsample_features_kernel"), pmc_of(load("prof5c_write"), "upsample_features_kernel")
            n = [v for k, v in load("prof5c_fetch")["kernels"].items() if "upsample_features_kernel" in k][0]
            # launches: chunk-sized ones and four-chunk ones (the script times both); bytes per utterance-sample over all of them
            f.write("# upsample_features_kernel counters over its %d launches: FETCH_SIZE x2 = %.2f GB read, WRITE_SIZE = %.2f GB written\n" %
                    (n["calls"], 2 * fe["FETCH_SIZE"] * 1024 / 1e9, wr["WRITE_SIZE"] * 1024 / 1e9))
            f.write("#   (algorithmic: 192 B of feature fragments written per utterance-sample = 0.604 GB per chunk of 256 x 12 288; the mel frames and\n"
                    "#    the 15.7 MB operand table are read from L2)\n")

# ---- D: the chain at C4, four tiles per chain ----------------------------------------------------------------------------------------------
if have("prof5d_kt"):
    doc = load("prof5d_kt")
    line = json.load(open(f"{G}/prof5d_line.json"))
    name, k = kernel_of(doc, "ELb0EEEvNS_6ParamsENS_11ChainParamsE")      # (the dump-free variant; the harness warms up with the dumping one)
    FLOP4 = 7143424
    with open(f"{P}/r05_chain_c4_tiles_per_chain.txt", "w") as f:
        f.write("# round 5: python scripts/gpu_r5_chain.py C4 4  under  rocprofv3 --kernel-trace --stats (+ PMC passes in their own runs)\n")
        f.write("# wn::wavenet_chain<fp16,128,256,256,DUMP=0>: 16 chains x 16 CUs (15 stages of 2 layers + head), FOUR tiles per chain = 1024 utterances per GPU\n")
        f.write("# the script's own line: %s\n" % json.dumps({kk: line[kk] for kk in ("khz_per_utterance", "steady_khz", "batch", "samples", "chunk", "shader_clock_ghz", "kernel")}))
        d = sorted(k["durations_ns"])
        f.write("# chain launches: %d; the longest (steady-state launch of 2048 samples x 1024 utterances): %.2f ms = %.2f us per sample = %.2f kHz per utterance;\n" %
                (k["calls"], d[-1] / 1e6, d[-1] / 1e3 / 2048, 2048 / (d[-1] / 1e6)))
        f.write("#   MFMA roofline: %.1f TFLOP/s = %.4f of 2500 dense fp16 (7 143 424 flop per utterance-sample)\n" %
                (1024 * 2048 * FLOP4 / (d[-1] * 1e-9) / 1e12, 1024 * 2048 * FLOP4 / (d[-1] * 1e-9) / PEAK))
        f.write(stats_table(doc, 8))
        c = {}
        for dd in ("fetch", "write", "issue"):
            if have(f"prof5d_{dd}"):
                c.update(pmc_of(load(f"prof5d_{dd}"), "ELb0EEEvNS_6ParamsENS_11ChainParamsE"))
        if "FETCH_SIZE" in c:
            tot_us = sum(1024 * n for n in (64, line["samples"] // 2, line["samples"] // 2, 640, 2048))      # utterance-samples of the script's chain launches
            f.write("# HBM over all chain launches of the run (%d utterance-samples): read %.0f B, written %.0f B per utterance-sample\n" %
                    (tot_us, 2 * c["FETCH_SIZE"] * 1024 / tot_us, c["WRITE_SIZE"] * 1024 / tot_us))
            f.write("#   (algorithmic: conditioning 15 360 + dilated taps 7 680 read, ring 7 680 written.  The rest is the hand-off: 8-byte granules,\n"
                    "#    1 KiB of x per utterance and stage + 2 KiB of skip sums = 46 KB per utterance-sample over 15 + 16 hops, stored with\n"
                    "#    relaxed atomics that are written through to memory, and the consumers' sweeps that miss L2 -- 0.1 TB/s at this rate)\n")
        issue_lines(c, f)
        for kk in sorted(c):
            f.write("%-32s %20.0f\n" % (kk, c[kk]))
print("done")
