#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 900 python -m pytest tests/test_primitives_gpu.py -m gpu -q -s 2>&1 | tail -25
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "benchmarked or binding or wrapper" 2>&1 | tail -25
} > gpurun_out/r2_e.log 2>&1
tail -60 gpurun_out/r2_e.log
