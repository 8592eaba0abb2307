cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
for rep in 1 2; do
for lib in layers w4 w3; do
if [ $lib = layers ]; then export QLDPC_LAYER_CHAIN=0; unset QLDPC_LIB; elif [ $lib = w4 ]; then export QLDPC_LAYER_CHAIN=1; unset QLDPC_LIB; else export QLDPC_LAYER_CHAIN=1; export QLDPC_LIB=$GRAFT_REPO_ROOT/qcrypto-ldpc_amd/variants/libqldpc_chain_$lib.so; fi
timeout -k 10 500 python bench.py --steps 3 --warmup 1 --no-early --no-fp16 --no-int8 --no-config3 --no-cpu --no-fer-deep --config5-frames 64,128,256 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config5_layered_1e6']
print('%-6s  64: %s  128: %s  256: %s' % ('$lib', {k:(round(c[k]['value']), round(c[k]['roofline']['frac'],3)) for k in ('fixed','early_exit')}, {k:round(c['at_128_frames'][k]['value']) for k in ('fixed','early_exit')}, {k:round(c['at_256_frames'][k]['value']) for k in ('fixed','early_exit')}))
"
done; done
