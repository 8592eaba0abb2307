cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
QLDPC_LAYER_CHAIN=1 timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "layer or hlayered or one_launch" > gpurun_out/g38_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/g38_tests.log; tail -3 gpurun_out/g38_tests.log
grep -q "rc=0" gpurun_out/g38_tests.log || exit 1
QLDPC_DEBUG=1 QLDPC_LAYER_CHAIN=1 timeout -k 10 500 python bench.py --steps 2 --warmup 1 --no-early --no-fp16 --no-int8 --no-config3 --no-cpu --no-fer-deep --config5-frames 64,128,256 2> gpurun_out/g38.err | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config5_layered_1e6']
print('chain   64:', {k:(round(c[k]['value'],1), round(c[k]['roofline']['frac'],3)) for k in ('fixed','early_exit')}, ' 128:', round(c['at_128_frames']['fixed']['roofline_frac'],3), round(c['at_128_frames']['early_exit']['value'],1), ' 256:', round(c['at_256_frames']['fixed']['roofline_frac'],3), round(c['at_256_frames']['early_exit']['value'],1))
"
grep "one-launch layered sweeps:" gpurun_out/g38.err | sort | uniq -c | sort -rn | head -4
