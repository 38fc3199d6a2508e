#!/bin/bash
mkdir -p gpurun_out
{
echo "=== default (PFMAX 18)"; timeout 300 python scripts/quick_wg.py
echo "=== PFMAX 12"; NVW_LIB=scripts/ubench/bld_pf12/libwavenet_infer.so timeout 300 python scripts/quick_wg.py
} > gpurun_out/r2_f.log 2>&1
cat gpurun_out/r2_f.log | grep -v amdgpu.ids
