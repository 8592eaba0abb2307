cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_baseline_configs_gpu.py tests/test_recon_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x > gpurun_out/g34_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/g34_tests.log; tail -4 gpurun_out/g34_tests.log
timeout -k 10 500 python bench.py --steps 2 --warmup 1 --no-early --no-fp16 --no-int8 --no-config3 --no-cpu --no-fer-deep 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config5_layered_1e6']
print('auto   64:', {k:(round(c[k]['value'],1), round(c[k]['roofline']['frac'],3)) for k in ('fixed','early_exit')}, ' 256:', c['at_256_frames'])
"
