#!/bin/bash
# GPU side of the round-5 profiles: rocprofv3 kernel traces and PMC passes (each counter set in its own run, --kernel-trace only, as
# MI355X_MICROARCH.md prescribes) of
#   A  the headline launch shape at steady state (bench.py --batch 12288: wavenet_wg<BT=3, RAW=0>, conditioning pre-packed);
#   B  the same launch with the conditioning computed in the kernel from the features (--conditioning features: RAW=3);
#   C  the features-in loop (scripts/gpu_r5_stream.py: nvw_generate_stream = upsample_features_kernel + wavenet_wg<RAW=3> per chunk);
#   D  the multi-CU chain at C4 with four tiles per chain, 1024 utterances (scripts/gpu_r5_chain.py C4 4).
# Every database is reduced on the box to gpurun_out/prof5_*.json (scripts/prof_extract.py) and deleted; scripts/make_profiles_r5.py turns
# the JSONs into profiles/r05_* on the authoring side.   usage: prof_collect_r5.sh [A] [B] [C] [D]   (default: all)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
mkdir -p gpurun_out
WHAT="${*:-A B C D}"
run() { d=$1; shift; rm -rf gpurun_out/$d; timeout 900 rocprofv3 "$@" -d gpurun_out/$d -o p -- ${CMD} > gpurun_out/$d.log 2>&1; echo "$d rc=$?";
        python scripts/prof_extract.py gpurun_out/$d gpurun_out/$d.json; }
for part in $WHAT; do
case $part in
A|B)
  if [ $part = A ]; then T=prof5a; CMD="python bench.py --batch 12288 --steps 5 --warmup 1 --no-cpu-baseline --no-extras";
  else T=prof5b; CMD="python bench.py --batch 12288 --steps 5 --warmup 1 --no-cpu-baseline --no-extras --conditioning features"; fi
  run ${T}_kt --kernel-trace --stats
  run ${T}_fetch --kernel-trace --pmc FETCH_SIZE
  run ${T}_write --kernel-trace --pmc WRITE_SIZE
  run ${T}_sq --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES
  run ${T}_ldsbw --kernel-trace --pmc SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS
  run ${T}_issue --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
  run ${T}_busy --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE
  grep -h "^{" gpurun_out/${T}_kt.log | tail -1 > gpurun_out/${T}_bench_line.json ;;
C)
  CMD="python scripts/gpu_r5_stream.py 12288"
  run prof5c_kt --kernel-trace --stats
  run prof5c_fetch --kernel-trace --pmc FETCH_SIZE
  run prof5c_write --kernel-trace --pmc WRITE_SIZE
  grep -E "^(upsampling|generation|nvw_)" gpurun_out/prof5c_kt.log > gpurun_out/prof5c_stdout.txt ;;
D)
  CMD="python scripts/gpu_r5_chain.py C4 4"
  run prof5d_kt --kernel-trace --stats
  run prof5d_fetch --kernel-trace --pmc FETCH_SIZE
  run prof5d_write --kernel-trace --pmc WRITE_SIZE
  run prof5d_issue --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
  grep -h "^{" gpurun_out/prof5d_kt.log | tail -1 > gpurun_out/prof5d_line.json ;;
esac
done
du -sh gpurun_out
