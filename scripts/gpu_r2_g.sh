#!/bin/bash
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "pipe" 2>&1 | tail -15
timeout 600 python scripts/quick_pipe.py 4096 8192 16384 24576
} > gpurun_out/r2_g.log 2>&1
cat gpurun_out/r2_g.log | grep -v amdgpu.ids | tail -40
