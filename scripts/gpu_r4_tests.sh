#!/bin/bash
# round 4: the whole GPU suite + smoke(), what the driver runs at round end
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2700 python -m pytest tests -m gpu -q --timeout 1200 -p no:cacheprovider -rA --durations=15 ) > gpurun_out/r4_tests.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|error" gpurun_out/r4_tests.log | tail -40 | cut -c1-300
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r4_smoke.log 2>&1
tail -6 gpurun_out/r4_smoke.log | cut -c1-300
