cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
S=./qcrypto-ldpc_amd/host/qldpc_stream
for P in 2 0; do for sc in 0.6,0.9,1.0 0.4,0.9,1.0 0.3,0.9,1.0 0.2,0.9,1.0 0.1,0.9,1.0 0.3,0.85,1.0 0.3,0.9,0.95; do for seed in 42 7; do
echo "== P=$P scale=$sc seed=$seed" >> gpurun_out/g15.log
QLDPC_RECON_GAP_SCALE=$sc timeout -k 10 200 $S -b 256 -r 1 -e 2048 -P $P -S $seed 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l.strip()); continue
    print({k:d[k] for k in ('reconciled','epochs','leaked_fraction','avg_iterations','ms_best','epochs_per_rate','failed_per_rate')})
" >> gpurun_out/g15.log
done; done; done
cat gpurun_out/g15.log
