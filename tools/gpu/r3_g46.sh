cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "layered or chain or config5" > gpurun_out/g46_tests.log 2>&1 || { tail -30 gpurun_out/g46_tests.log; exit 1; }
tail -2 gpurun_out/g46_tests.log
for w in auto 0 2 3 4; do
if [ $w = auto ]; then unset QLDPC_LAYER_CHAIN QLDPC_CHAIN_WAVES; elif [ $w = 0 ]; then export QLDPC_LAYER_CHAIN=0; else export QLDPC_LAYER_CHAIN=1 QLDPC_CHAIN_WAVES=$w; fi
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-early --no-fp16 --no-int8 --no-config3 --no-cpu --no-fer-deep --config5-frames 64,128,256 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config5_layered_1e6']
print('%-6s  64: %s  128: %s  256: %s' % ('$w', {k:(round(c[k]['value']), round(c[k]['roofline']['frac'],3)) for k in ('fixed','early_exit')}, {k:round(c['at_128_frames'][k]['value']) for k in ('fixed','early_exit')}, {k:round(c['at_256_frames'][k]['value']) for k in ('fixed','early_exit')}))
"
done
