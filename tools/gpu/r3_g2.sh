cd $GRAFT_REPO_ROOT
S=./qcrypto-ldpc_amd/host/qldpc_stream
run() { echo "== $*" >> gpurun_out/g2.log; env "$@" >> gpurun_out/g2.log 2>&1; echo "rc=$?" >> gpurun_out/g2.log; }
run QLDPC_RECON_LANES=1 QLDPC_POLL_EVERY=0 timeout -k 10 120 $S -b 256 -r 2
run QLDPC_RECON_LANES=1 timeout -k 10 120 $S -b 256 -r 2
run QLDPC_RECON_LANES=4 QLDPC_POLL_EVERY=0 timeout -k 10 120 $S -b 256 -r 2
run QLDPC_RECON_LANES=4 QLDPC_COMPACT=2 timeout -k 10 120 $S -b 256 -r 2
run QLDPC_RECON_LANES=4 timeout -k 10 120 $S -b 256 -r 2
run QLDPC_RECON_LANES=4 timeout -k 10 120 $S -b 256 -r 2 -S 7
run QLDPC_RECON_LANES=4 HIP_LAUNCH_BLOCKING=1 timeout -k 10 120 $S -b 256 -r 2
cat gpurun_out/g2.log
