cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --deselect tests/test_bench_gpu.py > gpurun_out/g65_tests.log 2>&1 || { tail -40 gpurun_out/g65_tests.log; exit 1; }
tail -2 gpurun_out/g65_tests.log
timeout -k 10 500 python bench.py --steps 3 --warmup 1 --no-early --no-layered --no-fp16 --no-int8 --no-config5 --no-cpu --no-fer-deep 2>gpurun_out/g65_bench.err > gpurun_out/g65_bench.json
python3 -c "
import json
d=json.loads(open('gpurun_out/g65_bench.json').read().strip().splitlines()[-1])
c=d['config3_multirate_stream']
print('config3', round(c['value']), c['fer'], c['leaked_fraction'], round(c['ms_total'],2), round(c['wall_frac'],3), c['avg_iterations'], c['undetected_errors'], c['roofline']['kernel'][:40])
for k in ('peg_mothers_round2_gaps','seeded_shuffle_mothers','flooding_schedule'): print(k, round(c[k]['value']), c[k]['fer'], c[k]['leaked_fraction'], round(c[k]['ms_total'],2), round(c[k]['wall_frac'],3), c[k]['avg_iterations'])
"
