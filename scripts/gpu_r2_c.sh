#!/bin/bash
mkdir -p gpurun_out
{
for sl in 0 4 16; do
  echo "--- sleep $sl"
  NVW_LIB=scripts/ubench/bld_s$sl/libwavenet_infer.so timeout 300 python scripts/nv_wavenet_perf.py -r 128 -s 256 -a 256 -l 30 -b 8 -m 3 -n 4096 -t 2048 2>&1 | grep -E "Sample rate|timed out|rror"
done
} > gpurun_out/r2_c.log 2>&1
cat gpurun_out/r2_c.log
