#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "=== base"; timeout 600 python scripts/quick_abl.py w1,w2,w3,g2,g3 2>&1 | tail -1
timeout 600 python scripts/quick_abl.py g2,g3 2>&1 | tail -1
} > gpurun_out/r3e.log 2>&1
cat gpurun_out/r3e.log
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > gpurun_out/r3e_tests.log 2>&1
tail -8 gpurun_out/r3e_tests.log | cut -c1-300
