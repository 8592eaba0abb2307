cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
S=./qcrypto-ldpc_amd/host/qldpc_stream
for P in 2 0; do for g in 0.035 0.030 0.025 0.020 0.015; do for seed in 42 7; do
echo "== P=$P gap=$g seed=$seed" >> gpurun_out/g14.log
timeout -k 10 200 $S -b 256 -r 1 -e 2048 -P $P -g $g -S $seed 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l.strip()); continue
    print({k:d[k] for k in ('reconciled','epochs','leaked_fraction','avg_iterations','ms_best','epochs_per_rate','failed_per_rate')})
" >> gpurun_out/g14.log
done; done; done
cat gpurun_out/g14.log
