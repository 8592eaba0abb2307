cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_bench_gpu.py > gpurun_out/g60_tests.log 2>&1 || { tail -40 gpurun_out/g60_tests.log; exit 1; }
tail -2 gpurun_out/g60_tests.log
timeout -k 10 200 python tests/fuzz_parity.py 90 11 2>&1 | tail -1
for rep in 1 2; do
for mode in dsatur firstfit; do
if [ $mode = firstfit ]; then export QLDPC_FIRST_FIT_LAYERS=1; else unset QLDPC_FIRST_FIT_LAYERS; fi
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-fp16 --no-int8 --no-config3 --no-cpu --no-fer-deep 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
l=d['layered_schedule']; c=d['config5_layered_1e6']
print('%-8s headline %d  layered config 2: %d %d   config5: %d (%.3f) %d (%.2f sweeps)  256: %d %d' % ('$mode', d['value'], l['fixed']['value'], l['early_exit']['value'], c['fixed']['value'], c['fixed']['roofline']['frac'], c['early_exit']['value'], c['early_exit']['avg_sweeps'], c['at_256_frames']['fixed']['value'], c['at_256_frames']['early_exit']['value']))
"
done; done
