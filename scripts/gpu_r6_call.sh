set -x
cd $GRAFT_REPO_ROOT
for v in shipped p18 p3 p19 p16 shipped p18 p3 p19 p16; do
  if [ $v = shipped ]; then unset NVW_LIB; else export NVW_LIB=$PWD/scripts/ubench/bld_$v/libwavenet_infer.so; fi
  python scripts/gpu_r6_ab.py $v --batches 12288 --crc-modes wg3 >> gpurun_out/ab18.log 2>&1
done
grep -h "^{" gpurun_out/ab18.log | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['tag'],r['B'],r['us_per_sample'],r['clock_ghz'],r['cycles_per_sample'],r.get('socket_w'),r.get('uj_per_utterance_sample'),r['crc'])"
