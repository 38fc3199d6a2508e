set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/t19.log
python bench.py --steps 20 --warmup 5 > gpurun_out/b19.json 2> gpurun_out/b19.err
tail -3 gpurun_out/t19.log; tail -2 gpurun_out/b19.err
