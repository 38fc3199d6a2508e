#!/bin/bash
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "harness and R64S256A256 or benchmarked or teacher" 2>&1 | tail -4
echo "=== default (PFMAX 12 -> 9)"; timeout 300 python scripts/quick_wg.py
echo "=== PFMAX 18"; NVW_LIB=scripts/ubench/bld_pf18/libwavenet_infer.so timeout 300 python scripts/quick_wg.py
} > gpurun_out/r2_f.log 2>&1
cat gpurun_out/r2_f.log | grep -v amdgpu.ids
