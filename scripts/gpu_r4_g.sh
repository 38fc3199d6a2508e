#!/bin/bash
# round 4: steady-state timings of the organisations (and of experiment builds named on the command line) with the clock probe
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for v in "" "$@"; do
  echo "=== ${v:-base}"
  lib=""; [ -n "$v" ] && lib=scripts/ubench/bld_$v/libwavenet_infer.so
  NVW_LIB=$lib R4_POINTS="${PTS:-wg3:4:12288,bcast1:8:16384,bcast1:8:14336,auto:0:24576,bcast1:8:64}" timeout 600 python scripts/gpu_r4_b.py time 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r4g.log 2>&1
cat gpurun_out/r4g.log
