cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
# compressed check state: parity first
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "layered or chain or config5 or fuzz" > gpurun_out/g47_tests.log 2>&1 || { tail -30 gpurun_out/g47_tests.log; exit 1; }
tail -2 gpurun_out/g47_tests.log
run() {
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-early --no-fp16 --no-int8 --no-config3 --no-cpu --no-fer-deep --config5-frames 64,128,256 2>gpurun_out/g47_$1.err | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config5_layered_1e6']
print('%-14s  64: %s  128: %s  256: %s' % ('$1', {k:(round(c[k]['value']), round(c[k]['roofline']['frac'],3)) for k in ('fixed','early_exit')}, {k:round(c['at_128_frames'][k]['value']) for k in ('fixed','early_exit')}, {k:round(c['at_256_frames'][k]['value']) for k in ('fixed','early_exit')}))
"
}
export QLDPC_LAYER_CHAIN=0
QLDPC_LAYER_CST=1 run cst_layers
QLDPC_LAYER_CST=0 run msg_layers
export QLDPC_LAYER_CST=0 QLDPC_LAYER_CHAIN=1
QLDPC_CHAIN_WAVES=3 QLDPC_CHAIN_LDS=0 run chain3_grid
QLDPC_CHAIN_WAVES=3 QLDPC_CHAIN_LDS=40960 run chain3_lds40k
QLDPC_CHAIN_WAVES=4 QLDPC_CHAIN_LDS=0 run chain4_grid
QLDPC_CHAIN_WAVES=5 QLDPC_CHAIN_LDS=0 run chain5_grid
QLDPC_DEBUG=1 QLDPC_CHAIN_WAVES=3 timeout 120 python bench.py --steps 1 --warmup 0 --no-early --no-fp16 --no-int8 --no-config3 --no-cpu --no-fer-deep --config5-frames 64 2>&1 >/dev/null | grep "resident" | sort | uniq -c
