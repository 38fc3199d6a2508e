#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/r3c_tests.log 2>&1
tail -25 gpurun_out/r3c_tests.log | cut -c1-300
{
echo "=== PF1 variant on C3 (after the fix)"
NVW_LIB=scripts/ubench/bld_pf1/libwavenet_infer.so timeout 600 python scripts/debug_r3.py C3:wg:16:64:64 2>&1 | grep -v amdgpu.ids
echo "=== bench default"
timeout 1500 python bench.py 2>&1 | tail -3
} > gpurun_out/r3c_misc.log 2>&1
cat gpurun_out/r3c_misc.log | cut -c1-6000
