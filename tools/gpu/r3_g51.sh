cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_bench_gpu.py > gpurun_out/g51_tests.log 2>&1 || { tail -40 gpurun_out/g51_tests.log; exit 1; }
tail -2 gpurun_out/g51_tests.log
timeout -k 10 300 python tools/config5_breakdown.py 64 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-fp16 --no-int8 --no-config3 --no-cpu --no-fer-deep 2>gpurun_out/g51_bench.err > gpurun_out/g51_bench.json
python3 -c "
import json
d=json.loads(open('gpurun_out/g51_bench.json').read().strip().splitlines()[-1])
print('headline', round(d['value']), 'early', round(d['early_exit']['value']))
l=d['layered_schedule']; print('layered', round(l['fixed']['value']), round(l['early_exit']['value']))
c=d['config5_layered_1e6']; print('config5', round(c['fixed']['value']), round(c['early_exit']['value']), '256:', round(c['at_256_frames']['fixed']['value']), round(c['at_256_frames']['early_exit']['value']))
"
