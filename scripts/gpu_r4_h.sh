#!/bin/bash
# round 4: an experiment build of wavenet_bcast: full-chip parity against wavenet_wg on the benchmarked sequence, then timings
#   gpu_r4_h.sh <variant> [points]
mkdir -p gpurun_out; export TMPDIR=/tmp
v=$1; lib=scripts/ubench/bld_$v/libwavenet_infer.so
{
echo "=== $v parity"
NVW_LIB=$lib timeout 600 python scripts/gpu_r4_f.py 2>&1 | grep -v amdgpu.ids
echo "=== $v timing"
NVW_LIB=$lib R4_POINTS="${2:-bcast1:8:16384,bcast1:8:64,wg3:4:12288}" timeout 600 python scripts/gpu_r4_b.py time 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r4h_$v.log 2>&1
cat gpurun_out/r4h_$v.log
