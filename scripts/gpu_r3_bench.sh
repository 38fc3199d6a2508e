#!/bin/bash
# round 3: the default bench run (what the driver runs), kept for the record
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python bench.py ) > gpurun_out/r3_bench_default.log 2>&1
tail -4 gpurun_out/r3_bench_default.log | cut -c1-12000
