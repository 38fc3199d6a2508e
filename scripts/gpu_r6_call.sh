set -x
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke17.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke17.log
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/t17.log
python bench.py --steps 20 --warmup 5 > gpurun_out/b17.json 2> gpurun_out/b17.err
tail -3 gpurun_out/smoke17.log; tail -3 gpurun_out/t17.log; tail -2 gpurun_out/b17.err
