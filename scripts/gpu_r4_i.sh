#!/bin/bash
# round 4: GPU tests that touch wn::wavenet_bcast / the organisation choice, then timings of the shipped library
mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x -k "${1:-bcast or benchmarked or full_chip or organisation or kernel_info or capi}" ) > gpurun_out/r4i_tests.log 2>&1
tail -n 8 gpurun_out/r4i_tests.log | cut -c1-300
R4_POINTS="bcast1:8:16384,bcast1:8:64,wg3:4:12288,auto:0:24576" timeout 600 python scripts/gpu_r4_b.py time 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4i_time.log | cut -c1-220
