#!/bin/bash
# GPU side of the round-6 profiles: rocprofv3 kernel traces and PMC passes (each counter set in its own run, --kernel-trace only, as
# MI355X_MICROARCH.md prescribes) of
#   A  the timed launch shape of bench.py (--batch 12288: wavenet_wg<BT=3, RAW=0, LR=1>, conditioning pre-packed, d = 1 ring slots in LDS);
#   B  the largest real-time batch's launch shape (--batch 13824: wavenet_wg<BT=4, RAW=0, LR=1> on 216 CUs);
#   C  the C4 chain with five tiles per chain (scripts/gpu_r6_chain.py prof 5: wavenet_chain<.., HOIST=1>, 1 280 utterances).
# Every database is reduced on the box to gpurun_out/prof6_*.json (scripts/prof_extract.py) and deleted; scripts/make_profiles_r6.py turns
# the JSONs into profiles/r06_* on the authoring side.   usage: prof_collect_r6.sh [A] [B] [C]   (default: A B)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
mkdir -p gpurun_out
WHAT="${*:-A B}"
run() { d=$1; shift; rm -rf gpurun_out/$d; timeout 900 rocprofv3 "$@" -d gpurun_out/$d -o p -- ${CMD} > gpurun_out/$d.log 2>&1; echo "$d rc=$?";
        python scripts/prof_extract.py gpurun_out/$d gpurun_out/$d.json; }
for part in $WHAT; do
  if [ $part = C ]; then
    # C  the multi-CU chain at C4 with FIVE tiles per chain (1 280 utterances: the HOIST instantiation of round 6)
    CMD="python scripts/gpu_r6_chain.py prof 5"
    run prof6c_kt --kernel-trace --stats
    run prof6c_fetch --kernel-trace --pmc FETCH_SIZE
    run prof6c_write --kernel-trace --pmc WRITE_SIZE
    grep -h "^{" gpurun_out/prof6c_kt.log | tail -1 > gpurun_out/prof6c_line.json
    continue
  fi
  if [ $part = A ]; then T=prof6a; BATCH=12288; else T=prof6b; BATCH=13824; fi
  CMD="python bench.py --batch $BATCH --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-power"
  run ${T}_kt --kernel-trace --stats
  run ${T}_fetch --kernel-trace --pmc FETCH_SIZE
  run ${T}_write --kernel-trace --pmc WRITE_SIZE
  run ${T}_sq --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES
  run ${T}_ldsbw --kernel-trace --pmc SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS
  run ${T}_issue --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
  run ${T}_busy --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE
  grep -h "^{" gpurun_out/${T}_kt.log | tail -1 > gpurun_out/${T}_bench_line.json
done
du -sh gpurun_out
