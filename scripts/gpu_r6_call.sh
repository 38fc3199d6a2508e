set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/t14.log
python bench.py --steps 20 --warmup 5 > gpurun_out/b14.json 2> gpurun_out/b14.err
tail -3 gpurun_out/t14.log; tail -2 gpurun_out/b14.err
