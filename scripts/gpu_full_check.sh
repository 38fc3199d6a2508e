#!/bin/bash
# whole GPU suite + the perf-harness points of the BASELINE configs
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40
echo "=== PERF"
for args in "-r 64 -s 256 -a 256 -l 20 -b 16 -m 1" "-r 64 -s 256 -a 256 -l 20 -b 16 -m 3" "-r 128 -s 256 -a 256 -l 30 -b 8 -m 3" "-r 64 -s 128 -a 256 -l 20 -b 4 -m 3"; do
  echo "--- $args"
  timeout 300 python scripts/nv_wavenet_perf.py $args -n 16384 -t 2048 2>&1 | grep -E "kernel:|Sample rate|timed out|rror"
done
} > gpurun_out/r2_full.log 2>&1
tail -40 gpurun_out/r2_full.log
