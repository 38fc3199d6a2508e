#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "chain" 2>&1 | tail -15
echo "=== PHASES C3"
NVW_LIB=scripts/ubench/bld_ct/libwavenet_infer.so timeout 120 python scripts/chain_phase.py 64 256 256 20 16 5
echo "=== PHASES C4"
NVW_LIB=scripts/ubench/bld_ct/libwavenet_infer.so timeout 120 python scripts/chain_phase.py 128 256 256 30 8 5
} > gpurun_out/r2_b.log 2>&1
tail -80 gpurun_out/r2_b.log
