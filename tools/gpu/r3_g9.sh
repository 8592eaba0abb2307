cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
S=$GRAFT_REPO_ROOT/qcrypto-ldpc_amd/host/qldpc_stream
run() { echo "== $*" >> gpurun_out/g9.log; env "$@" >> gpurun_out/g9.log 2>&1; echo "rc=$?" >> gpurun_out/g9.log; }
run timeout -k 10 120 $S -b 256 -r 5
run QLDPC_RECON_STREAMS_INTERLEAVED=1 timeout -k 10 120 $S -b 256 -r 5
run QLDPC_RECON_LANES=2 timeout -k 10 120 $S -b 256 -r 5
run QLDPC_RECON_LANES=3 timeout -k 10 120 $S -b 256 -r 5
mkdir -p gpurun_out/prof_c3b
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c3b -o c3 --output-format csv -- $S -b 256 -r 5 >> gpurun_out/g9.log 2>&1
cat gpurun_out/g9.log | grep -v "^W2\|^E2\|rocprof" | cut -c1-400
