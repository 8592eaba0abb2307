cd $GRAFT_REPO_ROOT
S=./qcrypto-ldpc_amd/host/qldpc_stream
run() { echo "== $*" >> gpurun_out/g6.log; env "$@" >> gpurun_out/g6.log 2>&1; echo "rc=$?" >> gpurun_out/g6.log; }
run QLDPC_DEBUG=1 timeout -k 10 120 $S -b 256 -r 3 -l -p
run QLDPC_MSG_HALF=1 timeout -k 10 120 $S -b 256 -r 3 -p
run QLDPC_POLL_EVERY=4 timeout -k 10 120 $S -b 256 -r 3
run QLDPC_POLL_EVERY=1 timeout -k 10 120 $S -b 256 -r 3
run QLDPC_FRAMES_PER_LANE=2 timeout -k 10 120 $S -b 256 -r 3
cat gpurun_out/g6.log
