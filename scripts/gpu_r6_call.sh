# the round's standard GPU call: smoke, the GPU suite, the driver-format bench line (kept as the last form of the scratch script
# that every measurement of LABNOTES round 6 went through: gpurun -- 'bash scripts/gpu_r6_call.sh')
set -x
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/gpu_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default_run.json 2> gpurun_out/bench_default_run.err
tail -2 gpurun_out/smoke.log; tail -3 gpurun_out/gpu_tests.log; tail -2 gpurun_out/bench_default_run.err
