cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/g48_tests.log 2>&1 || { tail -40 gpurun_out/g48_tests.log; exit 1; }
tail -2 gpurun_out/g48_tests.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-early --no-fp16 --no-int8 --no-config3 --no-cpu --no-fer-deep --config5-frames 64,128,256,1024 2>gpurun_out/g48_bench.err > gpurun_out/g48_bench.json
python3 -c "
import json
d=json.loads(open('gpurun_out/g48_bench.json').read().strip().splitlines()[-1]); c=d['config5_layered_1e6']
for k in ('fixed','early_exit'): print(k, round(c[k]['value']), {kk: round(vv,3) if isinstance(vv,float) else vv for kk,vv in c[k]['roofline'].items() if kk!='kernel'})
for f in (128,256,1024): print(f, c['at_%d_frames'%f])
"
