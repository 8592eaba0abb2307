cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
# small layered parity tests first, everything forced through the one-launch sweep; bounded so that a hang cannot hold the box
QLDPC_LAYER_CHAIN=1 timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "layer or hlayered" > gpurun_out/g30_tests.log 2>&1; echo "pytest(chain forced) rc=$?" >> gpurun_out/g30_tests.log; tail -4 gpurun_out/g30_tests.log
grep -q "rc=0" gpurun_out/g30_tests.log || exit 1
timeout -k 10 300 python -m pytest tests/test_baseline_configs_gpu.py -m gpu -q -x -k "config5 or million" >> gpurun_out/g30_tests.log 2>&1; echo "pytest(config5) rc=$?" >> gpurun_out/g30_tests.log; tail -3 gpurun_out/g30_tests.log
for chain in 0 1; do
QLDPC_LAYER_CHAIN=$chain timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-early --no-fp16 --no-int8 --no-config3 --no-cpu --no-fer-deep 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config5_layered_1e6']
print('chain=$chain', {k:(round(c[k]['value'],1), round(c[k]['roofline']['frac'],3), round(c[k]['roofline']['avg_sweep_ms'],3), round(c[k]['ms_per_step'],2), c[k]['fer']) for k in ('fixed','early_exit')}, {k:round(v['value'],1) for k,v in c['at_256_frames'].items()})
"
done
