cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
export QLDPC_LAYER_CHAIN=1
for lib in base w3 w2; do
if [ $lib = base ]; then unset QLDPC_LIB; else export QLDPC_LIB=$GRAFT_REPO_ROOT/qcrypto-ldpc_amd/variants/libqldpc_chain_$lib.so; fi
timeout -k 10 500 python bench.py --steps 2 --warmup 1 --no-early --no-fp16 --no-int8 --no-config3 --no-cpu --no-fer-deep --config5-frames 64,128 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config5_layered_1e6']
print('$lib   64:', {k:(round(c[k]['value'],1), round(c[k]['roofline']['frac'],3)) for k in ('fixed','early_exit')}, ' 128:', {k:(round(c['at_128_frames'][k]['value'],1), round(c['at_128_frames'][k]['roofline_frac'],3)) for k in ('fixed','early_exit')})
"
done
