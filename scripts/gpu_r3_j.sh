#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 900 python -m pytest tests/test_parity_gpu.py -q -x --timeout 600 -p no:cacheprovider -k "fragment_order or get_cond_input or consumed_in_place" -rA 2>&1 | grep -E "fragment-order|passed|failed|Error|error|assert" | cut -c1-300 | tail -20
timeout 600 python - <<'PY'
import bench, torch
w = bench.make_weights()
for ip in ("fragments", None, torch.float16):
    k, info = bench.measure_steady_khz(w, 12288, 256, in_place=ip)
    print(ip, "%.2f kHz" % k, info.split(" ")[0], flush=True)
PY
} > gpurun_out/r3j.log 2>&1
cat gpurun_out/r3j.log
