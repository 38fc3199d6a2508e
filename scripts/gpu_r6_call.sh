set -x
cd $GRAFT_REPO_ROOT
for v in base hr3 base hr3 base hr3; do
  NVW_LIB=$PWD/scripts/ubench/bld_$v/libwavenet_infer.so python scripts/gpu_r6_ab.py $v --batches 12288 --crc-modes wg3 >> gpurun_out/ab20.log 2>&1
done
grep -h "^{" gpurun_out/ab20.log | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['tag'],r['B'],r['us_per_sample'],r['clock_ghz'],r['cycles_per_sample'],r.get('socket_w'),r.get('uj_per_utterance_sample'),r['crc'])"
