#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for v in bmfma; do
echo "=== $v"
NVW_LIB=scripts/ubench/bld_$v/libwavenet_infer.so timeout 600 python scripts/split_check.py time s1 s2 s3 S2g S3g 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r3h.log 2>&1
cat gpurun_out/r3h.log
