set -x
cd $GRAFT_REPO_ROOT
NVW_LIB=$PWD/scripts/ubench/bld_ct32/libwavenet_infer.so python scripts/chain_phase.py 64 256 256 20 8 5 32 > gpurun_out/ct32b.log 2>&1
grep -E "sample period|last layer stage|head residence|stage  6:|stage  3:" gpurun_out/ct32b.log
python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "fp32 or harness or baseline_config or wavenet_infer or run_equals or native or pybind or no_tanh or replication" 2>&1 | tail -5
python - <<'P'
import json, bench
r = bench.dropin_fp32(bench.C3)
print(json.dumps({k: r[k] for k in ("persistent", "auto", "generation_only")}))
P
