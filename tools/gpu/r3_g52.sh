cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
timeout -k 10 500 python bench.py --steps 3 --warmup 1 --no-early --no-layered --no-fp16 --no-int8 --no-config5 --no-cpu --no-fer-deep 2>gpurun_out/g52_bench.err > gpurun_out/g52_bench.json
python3 -c "
import json
d=json.loads(open('gpurun_out/g52_bench.json').read().strip().splitlines()[-1])
c=d['config3_multirate_stream']
print('config3', round(c['value']), c['leaked_fraction'], c['ms_total'], c['wall_frac'], c['avg_iterations'])
for k in ('peg_mothers_round2_gaps','seeded_shuffle_mothers'): print(k, round(c[k]['value']), c[k]['leaked_fraction'], c[k]['ms_total'])
"
qcrypto-ldpc_amd/host/qldpc_stream -b 256 -r 5 2>&1 | tail -5
