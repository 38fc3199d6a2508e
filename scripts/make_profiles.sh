#!/bin/bash
# Turns the rocprofv3 databases of the last profiling gpurun (gpurun_out/prof_*) into the tracked
# summaries under profiles/.  usage: scripts/make_profiles.sh <round> <kt_ms> <kt4096_ms> <stream_ms>
set -e
R=$1; KT=$2; KT4=$3; KTS=$4
cd "$(dirname "$0")/.."
{
echo "# round $R, final engine: python bench.py --batch 8192 --samples 256 --steps 5 --warmup 1 --no-cpu-baseline under rocprofv3 --kernel-trace --stats"
echo "# (the default bench.py run picks this batch itself: the largest that stays real time)"
echo "# bench.py's own HIP-event kernel_ms for the same run: $KT ms (rocprof's average below includes the first, cold launch)"
python scripts/prof_summary.py kernel gpurun_out/prof_kt/kt_results.db; } > profiles/r${R}_kernel_trace_stats_wg_b8192.txt
{
echo "# round $R, final engine: python bench.py --batch 4096 --samples 512 --steps 3 --warmup 1 --no-cpu-baseline under rocprofv3 --kernel-trace --stats"
echo "# bench.py's own HIP-event kernel_ms for the same run: $KT4 ms"
python scripts/prof_summary.py kernel gpurun_out/prof_kt4096/kt_results.db; } > profiles/r${R}_kernel_trace_stats_wg_b4096.txt
{
echo "# round $R, final engine: python bench.py --batch 16384 --samples 128 --steps 3 --warmup 1 --no-cpu-baseline under rocprofv3 --kernel-trace --stats"
echo "# bench.py's own HIP-event kernel_ms for the same run: $KTS ms"
python scripts/prof_summary.py kernel gpurun_out/prof_kt_stream/kt_results.db; } > profiles/r${R}_kernel_trace_stats_stream_b16384.txt
python - "$R" <<'PY'
import json, subprocess, sys, re
R = sys.argv[1]
out = subprocess.run([sys.executable, "scripts/prof_summary.py", "pmc", "gpurun_out/prof_fetch/f_results.db",
                      "gpurun_out/prof_write/w_results.db", "gpurun_out/prof_l2/l2_results.db"],
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
          open("profiles/traffic_r%s.json" % R, "w"), indent=1)
hdr = ["# round %s, final engine, wn::wavenet_wg<BT=2> at batch 8192 x 256 samples (python bench.py --batch 8192 --samples 256 --steps 3 --warmup 1 --no-cpu-baseline)" % R,
       "# separate passes: rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc TCC_HIT_sum TCC_MISS_sum | --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES",
       "# FETCH_SIZE / WRITE_SIZE in KB per dispatch; gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of wide coalesced reads -> x2",
       "# HBM bytes per launch = (2*%.1f + %.1f) KB = %.1fe9 B  vs algorithmic 21.8e9 B (cond 10.7 + ring r/w 10.9 + sel/yOut 0.02): %.2fx" % (f, w, hbm / 1e9, hbm / 21.76e9),
       "# L2 hit rate = %.0f / (%.0f + %.0f) = %.1f %%" % (hit, hit, miss, 100 * hit / (hit + miss))]
full = subprocess.run([sys.executable, "scripts/prof_summary.py", "pmc", "gpurun_out/prof_fetch/f_results.db",
                       "gpurun_out/prof_write/w_results.db", "gpurun_out/prof_l2/l2_results.db",
                       "gpurun_out/prof_sq/sq_results.db"], capture_output=True, text=True).stdout
keep = [l for l in full.splitlines() if not re.search(r"rocprim|at6native|distribution", l)]
open("profiles/r%s_pmc_wg_b8192.txt" % R, "w").write("\n".join(hdr + keep) + "\n")
print("HBM GB/launch %.2f, L2 hit %.3f" % (hbm / 1e9, hit / (hit + miss)))
PY
