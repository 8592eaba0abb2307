cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "layered or recon or ragged or config5 or fuzz or compressed or harness" > gpurun_out/g67_tests.log 2>&1 || { tail -40 gpurun_out/g67_tests.log; exit 1; }
tail -2 gpurun_out/g67_tests.log
for b in 512 256; do for m in -f -l; do qcrypto-ldpc_amd/host/qldpc_stream -b $b -r 5 $m 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('b=$b $m', d['reconciled'], round(d['ms_best'],2), round(d['ms_mean'],2), round(d['Mbit_s_best']), d['leaked_fraction'], d['avg_iterations'])
"; done; done
timeout -k 10 500 python bench.py --steps 3 --warmup 1 --no-early --no-layered --no-fp16 --no-int8 --no-config5 --no-cpu --no-fer-deep 2>gpurun_out/g67_bench.err > gpurun_out/g67_bench.json
python3 -c "
import json
d=json.loads(open('gpurun_out/g67_bench.json').read().strip().splitlines()[-1])
c=d['config3_multirate_stream']
print('config3', round(c['value']), c['fer'], c['leaked_fraction'], round(c['ms_total'],2), round(c['wall_frac'],3), c['avg_iterations'], c['undetected_errors'])
for k in ('peg_mothers_round2_gaps','seeded_shuffle_mothers','flooding_schedule'): print(k, round(c[k]['value']), c[k]['fer'], c[k]['leaked_fraction'], round(c[k]['ms_total'],2), round(c[k]['wall_frac'],3), c[k]['avg_iterations'])
"
