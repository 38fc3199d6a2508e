set -x
cd $GRAFT_REPO_ROOT
bash scripts/prof_collect_r6.sh C > gpurun_out/prof6c.log 2>&1
tail -4 gpurun_out/prof6c.log; cat gpurun_out/prof6c_line.json | cut -c1-300
