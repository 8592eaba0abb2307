cd $GRAFT_REPO_ROOT
S=./qcrypto-ldpc_amd/host/qldpc_stream
run() { echo "== $*" >> gpurun_out/g5.log; env "$@" >> gpurun_out/g5.log 2>&1; echo "rc=$?" >> gpurun_out/g5.log; }
run QLDPC_RECON_NOSORT=1 QLDPC_DEBUG=1 timeout -k 10 120 $S -b 256 -r 2 -p
run QLDPC_DEBUG=1 timeout -k 10 120 $S -b 256 -r 2 -p
run timeout -k 10 120 $S -b 256 -r 5 -S 7
run timeout -k 10 120 $S -b 128 -r 5
run QLDPC_COMPACT=1 timeout -k 10 120 $S -b 256 -r 5
cat gpurun_out/g5.log
