# experiment: records {check, first edge, degree, slots} for the fp32 flooding check pass (QLDPC_CN_REC=1): parity, then same-box A/B of the headline leg
set -o pipefail
QLDPC_CN_REC=1 timeout -k 10 500 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_compaction_gpu.py -x -q -m gpu > gpurun_out/s9_parity.log 2>&1; rc=$?; echo "parity (QLDPC_CN_REC=1) rc=$rc"; tail -2 gpurun_out/s9_parity.log
[ $rc -eq 0 ] || exit 1
B="--steps 10 --warmup 1 --no-fp16 --no-int8 --no-config3 --no-config5 --no-cpu --no-fer-deep --no-layered"
for pass in 1 2 3; do
  for rec in 0 1; do
    QLDPC_CN_REC=$rec timeout -k 10 200 python bench.py $B > gpurun_out/s9_bench.json 2> gpurun_out/s9_bench.err || exit 1
    python - <<P
import json
d=json.loads(open('gpurun_out/s9_bench.json').read().strip().splitlines()[-1])
r=d['roofline']; s=d['spa_rule']
print('cn_rec=${rec} pass ${pass}: headline %.1f Mbit/s  cn %.4f ms frac %.4f  vn %.4f ms | early %.0f | spa fixed %.0f cn %.0f GB/s early %.0f' % (d['value'], r['avg_launch_ms'], r['frac'], r['vn_update']['avg_pass_ms'], d['early_exit']['value'], s['fixed']['value'], s['fixed']['cn_update_GBs'], s['early_exit']['value']))
P
  done
done
