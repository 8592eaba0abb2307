cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
for rep in 1 2; do for lib in old new; do
if [ $lib = old ]; then export QLDPC_LIB=$GRAFT_REPO_ROOT/qcrypto-ldpc_amd/variants/libqldpc_r3pre_edge.so; else unset QLDPC_LIB; fi
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-early --no-fp16 --no-int8 --no-config3 --no-cpu --no-fer-deep 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config5_layered_1e6']
print('$lib', {k:(round(c[k]['value'],1), round(c[k]['roofline']['frac'],3), round(c[k]['ms_per_step'],2)) for k in ('fixed','early_exit')}, {k:round(v['value'],1) for k,v in c['at_256_frames'].items()})
"
done; done
