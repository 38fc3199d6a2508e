#!/bin/bash
# round 3, wavenet_split: bit-identity against wavenet_wg, then steady-state timing
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 300 python scripts/split_check.py check 2>&1 | grep -v amdgpu.ids
echo "rc=$?"
timeout 900 python scripts/split_check.py time 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r3f.log 2>&1
cat gpurun_out/r3f.log
