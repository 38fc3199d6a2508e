set -x
cd $GRAFT_REPO_ROOT
python scripts/energy_ubench.py --wgs 256,128 --seconds 4 > gpurun_out/en4.log 2>&1
cp gpurun_out/r06_energy_ubench.json gpurun_out/r06_energy_ubench.txt profiles/
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/t4.log
python bench.py --steps 20 --warmup 5 > gpurun_out/b4.json 2> gpurun_out/b4.err
tail -4 gpurun_out/t4.log; tail -3 gpurun_out/b4.err; tail -22 gpurun_out/en4.log
