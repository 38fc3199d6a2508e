#!/bin/bash
# round 3: the whole GPU suite + smoke(), what the driver runs at round end
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2700 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rA --durations=15 ) > gpurun_out/r3_tests.log 2>&1
grep -E "^fp16 |passed|failed|error" gpurun_out/r3_tests.log | tail -60 | cut -c1-400
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r3_smoke.log 2>&1
tail -8 gpurun_out/r3_smoke.log | cut -c1-300
