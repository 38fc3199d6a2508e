#!/bin/bash
# round 4: the GPU suite + smoke(), the default bench line, then the profile collection (one box, one call)
mkdir -p gpurun_out; export TMPDIR=/tmp
bash scripts/gpu_r4_tests.sh
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r4_bench_default.log 2> gpurun_out/r4_bench_default.err
tail -c 3000 gpurun_out/r4_bench_default.log
bash scripts/prof_collect_r4.sh 2>&1 | tail -40
