cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
S=./qcrypto-ldpc_amd/host/qldpc_stream
$S -r 1 > /dev/null 2>&1
run() { echo "== $*" >> gpurun_out/g31.log; env "$@" 2>&1 | grep -v "^W2" | tail -3 | cut -c1-600 >> gpurun_out/g31.log; }
run timeout -k 10 120 $S -r 5
run timeout -k 10 120 $S -r 5 -l
run QLDPC_LAYER_CHAIN=0 timeout -k 10 120 $S -r 5 -l
run timeout -k 10 120 $S -r 5 -l -G 1
run timeout -k 10 120 $S -r 5 -l -p
cat gpurun_out/g31.log
