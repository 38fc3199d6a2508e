#!/bin/bash
# GPU side of the round-4 profiles: rocprofv3 kernel traces and PMC passes (each counter set in its own run, --kernel-trace
# only, as MI355X_MICROARCH.md prescribes) of
#   * the DEFAULT headline launch shape at steady state (bench.py --batch 12288: wavenet_wg<BT=3>, samples 640..895), now with
#     the issue / busy counter sets the round-3 review asked for (matrix-pipe busy cycles, active / waiting / stalled cycles);
#   * the organisation between three and four tiles per CU (bench.py --batch 16384: wavenet_bcast, four tiles per workgroup).
# Every database is reduced on the box to gpurun_out/prof4_*.json (scripts/prof_extract.py: per-kernel durations and counter sums) and
# deleted -- together they exceed what gpurun copies back; scripts/make_profiles_r4.py turns the JSONs into profiles/r04_* on the authoring side.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
mkdir -p gpurun_out
run() { d=$1; shift; rm -rf gpurun_out/$d; timeout 900 rocprofv3 "$@" -d gpurun_out/$d -o p -- ${CMD} > gpurun_out/$d.log 2>&1; echo "$d rc=$?";
        python scripts/prof_extract.py gpurun_out/$d gpurun_out/$d.json; }
CMD="python bench.py --batch 12288 --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
run prof4_kt --kernel-trace --stats
run prof4_fetch --kernel-trace --pmc FETCH_SIZE
run prof4_write --kernel-trace --pmc WRITE_SIZE
run prof4_sq --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES
run prof4_ldsbw --kernel-trace --pmc SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS
run prof4_issue --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run prof4_busy --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE
grep -h "^{" gpurun_out/prof4_kt.log | tail -1 > gpurun_out/prof4_bench_line.json
CMD="python bench.py --batch 16384 --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
run prof4_bc_kt --kernel-trace --stats
run prof4_bc_ldsbw --kernel-trace --pmc SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run prof4_bc_issue --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
grep -h "^{" gpurun_out/prof4_bc_kt.log | tail -1 > gpurun_out/prof4_bc_bench_line.json
du -sh gpurun_out
