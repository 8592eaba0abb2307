cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
S=./qcrypto-ldpc_amd/host/qldpc_stream
$S -r 1 > /dev/null 2>&1   # fill the code cache
run() { echo "== $*" >> gpurun_out/g23.log; env "$@" 2>&1 | grep -v "^W2" | tail -6 >> gpurun_out/g23.log; }
run QLDPC_DEBUG=1 timeout -k 10 120 $S -r 3
run timeout -k 10 120 $S -r 5 -G 1
run timeout -k 10 120 $S -r 5 -b 256
run timeout -k 10 120 $S -r 5 -S 7
timeout -k 10 600 python bench.py --steps 5 --warmup 1 --no-early --no-fp16 --no-int8 --no-config5 --no-cpu --no-fer-deep 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config3_multirate_stream']
keep=('value','fer','leaked_fraction','ms_total','avg_iterations','wall_frac','epochs_per_rate')
print({k:c[k] for k in keep}); print({k:c['peg_mothers_round2_gaps'][k] for k in keep}); print({k:c['seeded_shuffle_mothers'][k] for k in keep})
" >> gpurun_out/g23.log 2>&1
cat gpurun_out/g23.log
