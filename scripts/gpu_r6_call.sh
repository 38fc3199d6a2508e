set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/t8.log
python bench.py --steps 20 --warmup 5 > gpurun_out/b8.json 2> gpurun_out/b8.err
tail -4 gpurun_out/t8.log; tail -3 gpurun_out/b8.err
