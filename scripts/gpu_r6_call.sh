set -x
cd $GRAFT_REPO_ROOT
bash scripts/prof_collect_r6.sh A B > gpurun_out/prof6.log 2>&1
for r in -1 0 -1 0; do
  python scripts/gpu_r6_ab.py ringpow$r --ring $r --batches 12288,13824 --crc-modes "" >> gpurun_out/ab12.log 2>&1
done
tail -5 gpurun_out/prof6.log
grep -h "^{" gpurun_out/ab12.log | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['tag'],r['B'],r['us_per_sample'],r['clock_ghz'],r['cycles_per_sample'],r.get('socket_w'),r.get('uj_per_utterance_sample'))"
