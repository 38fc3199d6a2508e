#!/bin/bash
# round 2, first GPU call: chain kernel bring-up (parity subset) + latency of the BASELINE configs
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "harness and (R64S256A256_impl3 or R32S128A256_impl1 or R128S256A256_impl3)" 2>&1 | tail -15
echo "=== PERF"
for args in "-r 64 -s 256 -a 256 -l 20 -b 16 -m 1" "-r 64 -s 256 -a 256 -l 20 -b 16 -m 3" "-r 64 -s 256 -a 256 -l 20 -b 16 -m 4" \
            "-r 128 -s 256 -a 256 -l 30 -b 8 -m 1" "-r 128 -s 256 -a 256 -l 30 -b 8 -m 3" "-r 128 -s 256 -a 256 -l 30 -b 8 -m 4" \
            "-r 64 -s 128 -a 256 -l 20 -b 4 -m 1" "-r 64 -s 128 -a 256 -l 20 -b 4 -m 3"; do
  echo "--- $args"
  timeout 300 python scripts/nv_wavenet_perf.py $args -n 8192 -t 2048 2>&1 | grep -E "kernel:|Sample rate|timed out|rror"
done
} > gpurun_out/r2_a.log 2>&1
tail -60 gpurun_out/r2_a.log
