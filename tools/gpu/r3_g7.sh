cd $GRAFT_REPO_ROOT
S=./qcrypto-ldpc_amd/host/qldpc_stream
run() { echo "== $*" >> gpurun_out/g7.log; env "$@" >> gpurun_out/g7.log 2>&1; echo "rc=$?" >> gpurun_out/g7.log; }
run timeout -k 10 120 $S -b 256 -r 4
run timeout -k 10 120 $S -b 256 -r 4 -T 2
run timeout -k 10 120 $S -b 256 -r 4 -T 3
run GPU_MAX_HW_QUEUES=8 timeout -k 10 120 $S -b 256 -r 4 -T 2
run GPU_MAX_HW_QUEUES=8 timeout -k 10 120 $S -b 256 -r 4
cat gpurun_out/g7.log
