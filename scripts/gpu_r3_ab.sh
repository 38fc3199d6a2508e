#!/bin/bash
# round 3: A/B timing of experiment builds of wavenet_wg (scripts/build_variant.sh), us per sample at steady state
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "=== base"; timeout 600 python scripts/quick_abl.py w1,w3,g2,g3 2>&1 | tail -1
for v in "$@"; do
  echo "=== $v"; NVW_LIB=scripts/ubench/bld_$v/libwavenet_infer.so timeout 600 python scripts/quick_abl.py w1,w3,g2,g3 2>&1 | tail -1
done
echo "=== base again"; timeout 600 python scripts/quick_abl.py g2,g3 2>&1 | tail -1
} > gpurun_out/r3_ab.log 2>&1
cat gpurun_out/r3_ab.log
