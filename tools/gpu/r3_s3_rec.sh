# third session of round 3: layer records (bucket::d_rec) -- parity subset, then same-box A/B of the layered legs (QLDPC_LAYER_REC=0 = the old index walk)
set -o pipefail
timeout -k 10 700 python -m pytest tests/test_parity_gpu.py tests/test_baseline_configs_gpu.py tests/test_fuzz_gpu.py tests/test_recon_gpu.py tests/test_compaction_gpu.py -x -q -m gpu > gpurun_out/s5_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/s5_tests.log
[ $rc -eq 0 ] || exit 1
B="--steps 3 --warmup 1 --no-early --no-fp16 --no-int8 --no-config3 --no-cpu --no-fer-deep --no-spa"
for pass in 1 2; do
  for rec in 1 0; do
    QLDPC_LAYER_REC=$rec timeout -k 10 200 python bench.py $B > gpurun_out/s5_bench_rec${rec}_p${pass}.json 2> gpurun_out/s5_bench_rec${rec}_p${pass}.err || exit 1
    python - <<P
import json
d=json.loads(open('gpurun_out/s5_bench_rec${rec}_p${pass}.json').read().strip().splitlines()[-1])
l=d['layered_schedule']; c=d['config5_layered_1e6']
print('rec=${rec} pass ${pass}: layered fixed %.0f early %.0f sweep_ms %.3f | config5 fixed %.0f (frac %.3f moved %.3f) early %.0f | 256: fixed %.0f early %.0f' % (l['fixed']['value'], l['early_exit']['value'], l['fixed']['avg_sweep_ms'], c['fixed']['value'], c['fixed']['roofline']['frac'], c['fixed']['roofline']['moved_frac'], c['early_exit']['value'], c['at_256_frames']['fixed']['value'], c['at_256_frames']['early_exit']['value']))
P
    QLDPC_LAYER_REC=$rec qcrypto-ldpc_amd/host/qldpc_stream -b 512 -r 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   stream rec=$rec: ms_mean %.3f best %.3f Mbit/s %.0f' % (d['ms_mean'], d['ms_best'], d['Mbit_s_mean']))"
  done
done
