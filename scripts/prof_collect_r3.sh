#!/bin/bash
# GPU side of the round-3 profiles: rocprofv3 kernel traces and PMC passes (each counter set in its own run, --kernel-trace
# only, as MI355X_MICROARCH.md prescribes) of the DEFAULT headline launch shape at steady state (bench.py --batch 12288:
# every step = samples 640..895 of 12 288 utterances, wavenet_wg<BT=3>); databases land in gpurun_out/prof3_* and are
# summarised into profiles/ by scripts/make_profiles_r3.sh on the authoring side.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
mkdir -p gpurun_out
run() { d=$1; shift; rm -rf gpurun_out/$d; timeout 900 rocprofv3 "$@" -d gpurun_out/$d -o p -- ${CMD} > gpurun_out/$d.log 2>&1; echo "$d rc=$?"; }
CMD="python bench.py --batch 12288 --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
run prof3_kt --kernel-trace --stats
run prof3_fetch --kernel-trace --pmc FETCH_SIZE
run prof3_write --kernel-trace --pmc WRITE_SIZE
run prof3_l2 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum
run prof3_sq --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES
run prof3_lds --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_ACTIVE_INST_LDS
run prof3_ldsbw --kernel-trace --pmc SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_INSTS_SALU SQ_INSTS_SMEM
grep -h "^{" gpurun_out/prof3_kt.log | tail -1 > gpurun_out/prof3_bench_line.json
# two tiles per workgroup (8192 utterances)
CMD="python bench.py --batch 8192 --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
run prof3_kt_b8192 --kernel-trace --stats
grep -h "^{" gpurun_out/prof3_kt_b8192.log | tail -1 > gpurun_out/prof3_bench_line_b8192.json
# the multi-CU chain on C4 / C3 (reference definition: 16384 samples in chunks of 2048)
CMD="python scripts/nv_wavenet_perf.py -r 128 -s 256 -a 256 -l 30 -b 8 -m 3 -n 4096 -t 2048"
run prof3_kt_c4 --kernel-trace --stats
grep -h "Sample rate\|kernel:" gpurun_out/prof3_kt_c4.log
for d in gpurun_out/prof3_*; do [ -d $d ] && find $d -name "*.db" | head -1; done
du -sh gpurun_out
