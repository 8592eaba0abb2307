cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
S=./qcrypto-ldpc_amd/host/qldpc_stream
for kb in 7000 15000 30000 44000; do for sc in 0.1,0.85,1.0 0.2,0.85,1.0 0.3,0.9,1.0 0.6,0.9,1.0; do
echo "== key_bits=$kb scale=$sc" >> gpurun_out/g16.log
QLDPC_RECON_GAP_SCALE=$sc timeout -k 10 200 $S -b 256 -r 1 -e 2048 -k $kb -S 11 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l.strip()); continue
    print({k:d[k] for k in ('reconciled','epochs','leaked_fraction','avg_iterations','ms_best','epochs_per_rate','failed_per_rate')})
" >> gpurun_out/g16.log
done; done
cat gpurun_out/g16.log
