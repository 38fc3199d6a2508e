set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/t10.log
tail -4 gpurun_out/t10.log
