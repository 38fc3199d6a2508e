#!/bin/bash
# round 4: A/B of wavenet_wg experiment builds: sample checksums (must equal the shipped library's), then steady-state timings
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for v in "" "$@"; do
  echo "=== ${v:-base}"
  lib=""; [ -n "$v" ] && lib=scripts/ubench/bld_$v/libwavenet_infer.so
  NVW_LIB=$lib timeout 300 python scripts/ab_check.py wg3 wg 2>&1 | grep crc
  NVW_LIB=$lib timeout 600 python scripts/quick_abl.py ${PTS:-w3,g3} 2>&1 | tail -1
done
echo "=== base again"; timeout 600 python scripts/quick_abl.py ${PTS:-w3,g3} 2>&1 | tail -1
} > gpurun_out/r4e.log 2>&1
cat gpurun_out/r4e.log
