#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 300 python scripts/split_check.py check 2>&1 | grep -v amdgpu.ids | cut -c1-200
timeout 600 python scripts/split_check.py time s1 s2 s3 S1g S2g S3g 2>&1 | grep -v amdgpu.ids
NVW_LIB=scripts/ubench/bld_splt/libwavenet_infer.so timeout 300 python scripts/split_phase.py 48 256 10 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r3i.log 2>&1
cat gpurun_out/r3i.log
