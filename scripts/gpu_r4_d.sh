#!/bin/bash
# round 4: timing of wavenet_bcast experiment builds (one workgroup: what a CU costs without contention)
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for v in "" "$@"; do
  echo "=== ${v:-base}"
  lib=""; [ -n "$v" ] && lib=scripts/ubench/bld_$v/libwavenet_infer.so
  NVW_LIB=$lib R4_POINTS="${PTS:-bcast1:8:64,bcast2:9:128}" timeout 300 python scripts/gpu_r4_b.py time 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r4d.log 2>&1
cat gpurun_out/r4d.log
