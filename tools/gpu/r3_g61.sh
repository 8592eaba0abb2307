cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
for mode in dsatur firstfit; do
if [ $mode = firstfit ]; then export QLDPC_FIRST_FIT_LAYERS=1; else unset QLDPC_FIRST_FIT_LAYERS; fi
timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-early --no-layered --no-fp16 --no-int8 --no-config3 --no-cpu --config5-frames 64,256,1024 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['config5_layered_1e6']
print('$mode', 'config5 64:', round(c['early_exit']['value']), c['early_exit']['avg_sweeps'], c['early_exit']['sweeps_launched'], ' 256:', c['at_256_frames']['early_exit'], ' 1024:', c['at_1024_frames']['early_exit'])
for p in d['fer_deep']['layered_schedule']: print('   layered fer', p['qber'], p['frames'], p['frame_errors'], round(p['avg_iterations'],4), p['max_iterations'], round(p['decode_Mbit_s']))
"
done
