cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
for chain in 0 1; do
QLDPC_LAYER_CHAIN=$chain timeout -k 10 500 python bench.py --steps 2 --warmup 1 --no-early --no-fp16 --no-int8 --no-config3 --no-cpu --no-fer-deep --config5-frames 64,128,256,512,1024 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config5_layered_1e6']
print('chain=$chain   64:', {k:(round(c[k]['value'],1), round(c[k]['roofline']['frac'],3)) for k in ('fixed','early_exit')})
for f in (128,256,512,1024):
    a=c['at_%d_frames'%f]; print('chain=$chain %4d:'%f, {k:(round(a[k]['value'],1), round(a[k]['roofline_frac'],3)) for k in ('fixed','early_exit')})
"
done
# config-2-shape layered (30 layers x 437 checks, 64 groups)
for chain in 0 1; do
QLDPC_LAYER_CHAIN=$chain timeout -k 10 300 python bench.py --steps 3 --warmup 1 --schedule hlayered --no-config3 --no-config5 --no-fer-deep 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config-2 batch, layered, chain=$chain:', round(d['value'],1), round(d['ms_per_step'],2), 'early', round(d['early_exit']['value'],1))
"
done
