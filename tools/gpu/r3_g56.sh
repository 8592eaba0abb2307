cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "compressed_check_state or layered or syndrome or fuzz or config5 or baseline" > gpurun_out/g56_tests.log 2>&1 || { tail -40 gpurun_out/g56_tests.log; exit 1; }
tail -2 gpurun_out/g56_tests.log
timeout -k 10 200 python tests/fuzz_parity.py 120 5 2>&1 | tail -1
for bp in 0 1; do
if [ $bp = 1 ]; then export QLDPC_LAYER_BALLOT_PASS=1; else unset QLDPC_LAYER_BALLOT_PASS; fi
echo "separate ballot pass: $bp"
timeout -k 10 300 python tools/config5_breakdown.py 64 2>&1 | grep -v amdgpu.ids | grep -v "load\|fetch\|status"
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-fp16 --no-int8 --no-config3 --no-cpu --no-fer-deep 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
l=d['layered_schedule']; print('  layered config 2', round(l['fixed']['value']), round(l['early_exit']['value']))
c=d['config5_layered_1e6']; print('  config5', round(c['fixed']['value']), round(c['early_exit']['value']), '256:', round(c['at_256_frames']['fixed']['value']), round(c['at_256_frames']['early_exit']['value']))
"
done
