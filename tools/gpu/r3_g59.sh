cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "compressed_check_state or layered or config5 or fuzz" > gpurun_out/g59_tests.log 2>&1 || { tail -40 gpurun_out/g59_tests.log; exit 1; }
tail -2 gpurun_out/g59_tests.log
for rep in 1 2; do
for lib in exact runtime prefuse; do
unset QLDPC_LIB QLDPC_CST_EXACT_DEG
if [ $lib = prefuse ]; then export QLDPC_LIB=$GRAFT_REPO_ROOT/qcrypto-ldpc_amd/variants/libqldpc_prefuse.so; fi
if [ $lib = runtime ]; then export QLDPC_CST_EXACT_DEG=0; fi
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-fp16 --no-int8 --no-config3 --no-cpu --no-fer-deep 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
l=d['layered_schedule']; c=d['config5_layered_1e6']
print('%-8s headline %d  layered config 2: %d %d   config5: %d (%.3f) %d   256: %d %d' % ('$lib', d['value'], l['fixed']['value'], l['early_exit']['value'], c['fixed']['value'], c['fixed']['roofline']['frac'], c['early_exit']['value'], c['at_256_frames']['fixed']['value'], c['at_256_frames']['early_exit']['value']))
"
done; done
