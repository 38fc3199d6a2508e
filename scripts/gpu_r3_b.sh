#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "=== DEBUG"
timeout 900 python scripts/debug_r3.py S8R:wg:16 S8R:chain:16 R32:wg:16 oddL_ragged:wg:32 oddL_ragged:wg:32:40 oddL_ragged:chain:32:40 oddL_ragged:wg:32:13:39 2>&1 | grep -v amdgpu.ids
echo "=== PF1 variant on C3"
NVW_LIB=scripts/ubench/bld_pf1/libwavenet_infer.so timeout 600 python scripts/debug_r3.py C3:wg:16:64:64 2>&1 | grep -v amdgpu.ids
echo "=== PERF raw"
timeout 600 python scripts/quick_abl.py g2raw16,g2raw32 2>&1 | tail -1
NVW_LIB=scripts/ubench/bld_rawaux0/libwavenet_infer.so timeout 600 python scripts/quick_abl.py g3raw16,g3raw32,g2raw16,g2raw32 2>&1 | tail -1
echo "=== COUNTERS"
rocprofv3 -L 2>/dev/null | grep -i "lds\|SQ_INSTS\|SQ_ACTIVE_INST\|SQ_INST_CYCLES" | cut -c1-200 | sort -u | head -60
} > gpurun_out/r3b.log 2>&1
cat gpurun_out/r3b.log
