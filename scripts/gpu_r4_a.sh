#!/bin/bash
# round 4, GPU call A: suite still green with the clock probe in Params; what clock / power the headline launch runs at
# (scripts/clock_probe.py: workgroup-count sweep, then the ablation builds at the full chip); issue / busy counters.
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider ) > gpurun_out/r4a_tests.log 2>&1
tail -3 gpurun_out/r4a_tests.log
timeout 600 python scripts/clock_probe.py --wgs 1,64,128,192,256 --out gpurun_out/r4a_clock_base.json 2> gpurun_out/r4a_clock_base.log
cat gpurun_out/r4a_clock_base.log
for v in nowl noact hot nohbm; do
  NVW_LIB=scripts/ubench/bld_$v/libwavenet_infer.so timeout 300 python scripts/clock_probe.py --wgs 256 --out gpurun_out/r4a_clock_$v.json 2> gpurun_out/r4a_clock_$v.log
  echo "== $v"; cat gpurun_out/r4a_clock_$v.log | tail -2
done
timeout 300 python scripts/clock_probe.py --wgs 256 --bt 2 --out gpurun_out/r4a_clock_bt2.json 2> gpurun_out/r4a_clock_bt2.log; tail -1 gpurun_out/r4a_clock_bt2.log
timeout 300 python scripts/clock_probe.py --wgs 256 --bt 1 --out gpurun_out/r4a_clock_bt1.json 2> gpurun_out/r4a_clock_bt1.log; tail -1 gpurun_out/r4a_clock_bt1.log
cd /tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo; cd $R
rocprofv3 -L > gpurun_out/r4a_counters.txt 2>&1
CMD="python bench.py --batch 12288 --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
run() { d=$1; shift; rm -rf gpurun_out/$d; timeout 600 rocprofv3 "$@" -d gpurun_out/$d -o p -- ${CMD} > gpurun_out/$d.log 2>&1; echo "$d rc=$?"; }
run prof4_issue --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MFMA GRBM_GUI_ACTIVE
run prof4_busy --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
ls gpurun_out/prof4_issue gpurun_out/prof4_busy 2>&1 | head
du -sh gpurun_out
