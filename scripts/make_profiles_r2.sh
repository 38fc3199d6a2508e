#!/bin/bash
# Authoring side of the round-2 profiles: summarises the rocprofv3 databases collected by
# scripts/prof_collect_r2.sh (gpurun_out/prof2_*) into the tracked files under profiles/.
set -e
cd "$(dirname "$0")/.."
P=profiles
BENCH="python bench.py --batch 8192 --samples 256 --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
{
echo "# round 2: $BENCH  under  rocprofv3 --kernel-trace --stats"
echo "# (two tiles of 16 utterances per workgroup, 256 workgroups; bench.py's own line of this run:"
echo "#  $(python - <<'PY'
import json
j = json.load(open("gpurun_out/prof2_bench_line.json"))
print("value %.1f M samples/s, kernel_ms %.3f (HIP events), khz_per_utterance %.2f, roofline.frac %.4f)" % (j["value"] / 1e6, j["roofline"]["kernel_ms"], j["khz_per_utterance"], j["roofline"]["frac"]))
PY
)"
python scripts/prof_summary.py kernel gpurun_out/prof2_kt/p_results.db; } > $P/r02_kernel_trace_stats_wg_b8192.txt
{
echo "# round 2: python scripts/nv_wavenet_perf.py -r 128 -s 256 -a 256 -l 30 -b 8 -m 3 -n 4096 -t 2048  under  rocprofv3 --kernel-trace --stats"
echo "# BASELINE config C4 (R128/S256/A256, 30 layers, batch 8, fp16) on the multi-CU chain: 16 workgroups on 16 CUs, weights resident;"
echo "# kHz per utterance = samples of a launch / its duration (the profiler's serialisation costs ~8 %: 24.1 kHz under rocprofv3, 26.5 kHz without)"
grep -h "kernel:\|Sample rate" gpurun_out/prof2_kt_c4.log | sed 's/^/# /'
python scripts/prof_summary.py kernel gpurun_out/prof2_kt_c4/p_results.db | head -8; } > $P/r02_kernel_trace_stats_chain_c4.txt
{
echo "# round 2: python scripts/nv_wavenet_perf.py -r 64 -s 256 -a 256 -l 20 -b 16 -m 3 -n 8192 -t 2048  under  rocprofv3 --kernel-trace --stats"
echo "# BASELINE config C3 (R64/S256/A256, 20 layers, batch 16, fp16) on the multi-CU chain: 5 workgroups"
grep -h "kernel:\|Sample rate" gpurun_out/prof2_kt_c3.log | sed 's/^/# /'
python scripts/prof_summary.py kernel gpurun_out/prof2_kt_c3/p_results.db | head -8; } > $P/r02_kernel_trace_stats_chain_c3.txt
{
echo "# round 2: python scripts/pack_cond_time.py (setConditioning of 256 samples x 8192 utterances, C3 shape, fp32 source on the device)"
echo "# under rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE and --pmc WRITE_SIZE in their own runs"
echo "# round 1's scalar-gather pack_cond_kernel: 19.0 ms, FETCH_SIZE 48.0 GB raw; this LDS-tiled kernel: source and destination bytes once"
python scripts/prof_summary.py kernel gpurun_out/prof2_kt_pack/p_results.db | head -4
python scripts/prof_summary.py pmc gpurun_out/prof2_fetch_pack/p_results.db gpurun_out/prof2_write_pack/p_results.db | grep "pack_cond\|^#\|^kernel"
echo "# FETCH_SIZE x2 (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md) = 21.5 GB = the fp32 source; WRITE_SIZE 10.7 GB = the fp16 fragments;"
echo "# 32.2 GB in 6.1 ms = 5.3 TB/s (84 % of the 6.3 TB/s a streaming copy reaches on this chip)"; } > $P/r02_pack_cond.txt
python - <<'PY'
import json, subprocess, sys, re
out = subprocess.run([sys.executable, "scripts/prof_summary.py", "pmc", "gpurun_out/prof2_fetch/p_results.db",
                      "gpurun_out/prof2_write/p_results.db", "gpurun_out/prof2_l2/p_results.db", "gpurun_out/prof2_sq/p_results.db"],
                     capture_output=True, text=True).stdout
v = {}
for line in out.splitlines():
    m = re.match(r"\S*wavenet_wg\S*\s+(\w+)\s+\d+\s+([\d.]+)", line)
    if m:
        v[m.group(1)] = float(m.group(2))
f, w, hit, miss = v["FETCH_SIZE"], v["WRITE_SIZE"], v["TCC_HIT_sum"], v["TCC_MISS_sum"]
hbm = (2 * f + w) * 1024
json.dump({"batch": 8192, "samples": 256, "fetch_size_kb": f, "write_size_kb": w, "hbm_bytes_per_launch": hbm,
           "l2_hit_rate": hit / (hit + miss),
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); "
                   "separate --pmc passes; wn::wavenet_wg, two tiles per workgroup"},
          open("profiles/traffic_r02.json", "w"), indent=1)
hdr = ["# round 2, wn::wavenet_wg<fp16,64,256,256,BT=2,EMBLDS=1,DUMP=0> at batch 8192 x 256 samples (python bench.py --batch 8192 --samples 256 --steps 5 --warmup 1 --no-cpu-baseline --no-extras)",
       "# separate runs: rocprofv3 --kernel-trace --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc TCC_HIT_sum TCC_MISS_sum | --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES",
       "# FETCH_SIZE / WRITE_SIZE in KB per dispatch; gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of wide coalesced reads -> x2",
       "# HBM bytes per launch = (2*%.1f + %.1f) KB = %.2fe9 B  vs algorithmic 21.8e9 B (cond 10.7 + ring r/w 10.9 + sel/yOut 0.02): %.2fx" % (f, w, hbm / 1e9, hbm / 21.76e9),
       "# L2 hit rate = %.0f / (%.0f + %.0f) = %.1f %%" % (hit, hit, miss, 100 * hit / (hit + miss)),
       "# SQ counters are per shader engine (32 of them): VALU : MFMA = %.2f, LDS bank-conflict cycles / LDS active cycles = %.1f %%" %
       (v["SQ_INSTS_VALU"] / v["SQ_INSTS_MFMA"], 100 * v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"])]
keep = [l for l in out.splitlines() if re.search(r"wavenet_wg|^#|^kernel", l)]
open("profiles/r02_pmc_wg_b8192.txt", "w").write("\n".join(hdr + keep) + "\n")
print("\n".join(hdr))
PY
# ---- three tiles per workgroup at 12288 utterances ----
B3="python bench.py --batch 12288 --samples 128 --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
{
echo "# round 2: $B3  under  rocprofv3 --kernel-trace --stats"
echo "# (three tiles of 16 utterances per workgroup, 256 workgroups: the launch shape of 8193 .. 12288 utterances per GPU; bench.py's own line of this run:"
echo "#  $(python - <<'PY'
import json
j = json.load(open("gpurun_out/prof2_bench_line_b12288.json"))
print("value %.1f M samples/s, kernel_ms %.3f (HIP events), khz_per_utterance %.2f, roofline.frac %.4f)" % (j["value"] / 1e6, j["roofline"]["kernel_ms"], j["khz_per_utterance"], j["roofline"]["frac"]))
PY
)"
python scripts/prof_summary.py kernel gpurun_out/prof2_kt_b12288/p_results.db; } > $P/r02_kernel_trace_stats_wg_b12288.txt
python - <<'PY'
import json, subprocess, sys, re
out = subprocess.run([sys.executable, "scripts/prof_summary.py", "pmc", "gpurun_out/prof2_fetch_b12288/p_results.db",
                      "gpurun_out/prof2_write_b12288/p_results.db", "gpurun_out/prof2_sq_b12288/p_results.db"],
                     capture_output=True, text=True).stdout
v = {}
for line in out.splitlines():
    m = re.match(r"\S*wavenet_wg\S*\s+(\w+)\s+\d+\s+([\d.]+)", line)
    if m:
        v[m.group(1)] = float(m.group(2))
f, w = v["FETCH_SIZE"], v["WRITE_SIZE"]
hbm = (2 * f + w) * 1024
alg = 12288 * 128 * (20 * 2 * 64 * 2 * 2 + 8)      # conditioning (2R) + ring read + ring write (R each), fp16, + selector + sample
json.dump({"batch": 12288, "samples": 128, "fetch_size_kb": f, "write_size_kb": w, "hbm_bytes_per_launch": hbm,
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); "
                   "separate --pmc passes; wn::wavenet_wg, three tiles per workgroup"},
          open("profiles/traffic_r02_b12288.json", "w"), indent=1)
hdr = ["# round 2, wn::wavenet_wg<fp16,64,256,256,BT=3,EMBLDS=1,DUMP=0> at batch 12288 x 128 samples (python bench.py --batch 12288 --samples 128 --steps 5 --warmup 1 --no-cpu-baseline --no-extras)",
       "# separate runs: rocprofv3 --kernel-trace --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES",
       "# HBM bytes per launch = (2*%.1f + %.1f) KB = %.2fe9 B  vs algorithmic %.2fe9 B: %.2fx" % (f, w, hbm / 1e9, alg / 1e9, hbm / alg),
       "# SQ counters are per shader engine (32 of them): VALU : MFMA = %.2f, LDS bank-conflict cycles / LDS active cycles = %.1f %%" %
       (v["SQ_INSTS_VALU"] / v["SQ_INSTS_MFMA"], 100 * v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"])]
keep = [l for l in out.splitlines() if re.search(r"wavenet_wg|^#|^kernel", l)]
open("profiles/r02_pmc_wg_b12288.txt", "w").write("\n".join(hdr + keep) + "\n")
print("\n".join(hdr))
PY
cp gpurun_out/prof2_bench_line_b12288.json $P/r02_bench_line_under_rocprof_b12288.json
cp gpurun_out/prof2_bench_line.json $P/r02_bench_line_under_rocprof.json
ls -la $P
