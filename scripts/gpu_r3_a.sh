#!/bin/bash
# round 3, GPU call A: whole GPU suite + A/B timing of the experiment builds
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "=== PERF base"; timeout 600 python scripts/quick_abl.py w1,w2,w3,g2,g3,g3raw16,g3raw32 2>&1 | tail -3
for v in pk noslp hot; do
  echo "=== PERF $v"; NVW_LIB=scripts/ubench/bld_$v/libwavenet_infer.so timeout 600 python scripts/quick_abl.py w1,w3,g2,g3 2>&1 | tail -3
done
} > gpurun_out/r3a_perf.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/r3a_tests.log 2>&1
tail -60 gpurun_out/r3a_tests.log | cut -c1-400
cat gpurun_out/r3a_perf.log
