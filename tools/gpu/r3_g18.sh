cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
for cpw in 1 2 3 4 6 8; do
echo "== cpw=$cpw" >> gpurun_out/g18.log
QLDPC_LAYER_CPW=$cpw timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-early --no-fp16 --no-int8 --no-config3 --no-cpu --no-fer-deep --config5-frames 64 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config5_layered_1e6']
print({k:(round(c[k]['value'],1), round(c[k]['roofline']['frac'],3), round(c[k]['roofline']['avg_sweep_ms'],3), c[k]['fer']) for k in ('fixed','early_exit')})
" >> gpurun_out/g18.log 2>&1
done
cat gpurun_out/g18.log
