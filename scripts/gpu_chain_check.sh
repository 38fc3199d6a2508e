#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "chain" 2>&1 | tail -8
echo "=== PHASES C4"
NVW_LIB=scripts/ubench/bld_ct/libwavenet_infer.so timeout 120 python scripts/chain_phase.py 128 256 256 30 16 5
echo "=== PHASES C3"
NVW_LIB=scripts/ubench/bld_ct/libwavenet_infer.so timeout 120 python scripts/chain_phase.py 64 256 256 20 16 5
echo "=== PERF"
for args in "-r 64 -s 256 -a 256 -l 20 -b 16 -m 3" "-r 128 -s 256 -a 256 -l 30 -b 8 -m 3" "-r 64 -s 128 -a 256 -l 20 -b 4 -m 3"; do
  echo "--- $args"
  timeout 300 python scripts/nv_wavenet_perf.py $args -n 8192 -t 2048 2>&1 | grep -E "kernel:|Sample rate|timed out|rror"
done
} > gpurun_out/r2_b.log 2>&1
tail -70 gpurun_out/r2_b.log
