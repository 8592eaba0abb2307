cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
for rep in 1 2; do
for lib in base NT LDNT BOTH; do
if [ $lib = base ]; then unset QLDPC_LIB; else export QLDPC_LIB=$GRAFT_REPO_ROOT/qcrypto-ldpc_amd/variants/libqldpc_post_$lib.so; fi
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-early --no-fp16 --no-int8 --no-config3 --no-cpu --no-fer-deep 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
l=d['layered_schedule']; c=d['config5_layered_1e6']
print('%-6s layered config 2: %d %d   config5: %d (%.3f) %d   256: %d %d' % ('$lib', l['fixed']['value'], l['early_exit']['value'], c['fixed']['value'], c['fixed']['roofline']['frac'], c['early_exit']['value'], c['at_256_frames']['fixed']['value'], c['at_256_frames']['early_exit']['value']))
"
done; done
