#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for o in 10 9 8; do
NVW_LIB=scripts/ubench/bld_splt/libwavenet_infer.so timeout 300 python scripts/split_phase.py $((16*(o-7))) 256 $o 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r3g.log 2>&1
cat gpurun_out/r3g.log
