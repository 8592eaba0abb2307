cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
timeout -k 10 300 python tools/config5_breakdown.py 64 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-fp16 --no-int8 --no-config3 --no-config5 --no-cpu --no-fer-deep 2>gpurun_out/g50_bench.err > gpurun_out/g50_bench.json
python3 -c "
import json
d=json.loads(open('gpurun_out/g50_bench.json').read().strip().splitlines()[-1])
print('headline', round(d['value']), 'early', round(d['early_exit']['value']))
print(json.dumps(d['layered_schedule'], indent=1))
"
