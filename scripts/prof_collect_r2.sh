#!/bin/bash
# GPU side of the round-2 profiles: rocprofv3 kernel traces and PMC passes (each counter set in its own run,
# --kernel-trace only, as MI355X_MICROARCH.md prescribes); the databases land in gpurun_out/prof2_* and are
# summarised into profiles/ by scripts/make_profiles_r2.sh on the authoring side.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
mkdir -p gpurun_out
BENCH="python bench.py --batch 8192 --samples 256 --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
run() { d=$1; shift; rm -rf gpurun_out/$d; timeout 600 rocprofv3 "$@" -d gpurun_out/$d -o p -- ${CMD} > gpurun_out/$d.log 2>&1; echo "$d rc=$?"; }
CMD="$BENCH"
run prof2_kt --kernel-trace --stats
run prof2_fetch --kernel-trace --pmc FETCH_SIZE
run prof2_write --kernel-trace --pmc WRITE_SIZE
run prof2_l2 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum
run prof2_sq --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES
# three tiles per workgroup (8193 .. 12288 utterances per GPU)
CMD="python bench.py --batch 12288 --samples 128 --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
run prof2_kt_b12288 --kernel-trace --stats
run prof2_fetch_b12288 --kernel-trace --pmc FETCH_SIZE
run prof2_write_b12288 --kernel-trace --pmc WRITE_SIZE
run prof2_sq_b12288 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES
grep -h "^{" gpurun_out/prof2_kt_b12288.log | tail -1 > gpurun_out/prof2_bench_line_b12288.json
CMD="python scripts/nv_wavenet_perf.py -r 128 -s 256 -a 256 -l 30 -b 8 -m 3 -n 4096 -t 2048"
run prof2_kt_c4 --kernel-trace --stats
CMD="python scripts/nv_wavenet_perf.py -r 64 -s 256 -a 256 -l 20 -b 16 -m 3 -n 8192 -t 2048"
run prof2_kt_c3 --kernel-trace --stats
CMD="python scripts/pack_cond_time.py"
run prof2_kt_pack --kernel-trace --stats
run prof2_fetch_pack --kernel-trace --pmc FETCH_SIZE
run prof2_write_pack --kernel-trace --pmc WRITE_SIZE
grep -h "^{" gpurun_out/prof2_kt.log | tail -1 > gpurun_out/prof2_bench_line.json
grep -h "Sample rate\|kernel:" gpurun_out/prof2_kt_c4.log gpurun_out/prof2_kt_c3.log
find gpurun_out -name "*.db" -path "*prof2*" | head -20
du -sh gpurun_out
