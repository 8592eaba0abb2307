cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
timeout -k 10 600 python -m pytest tests/test_recon_gpu.py -m gpu -q -x > gpurun_out/g36_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/g36_tests.log; tail -4 gpurun_out/g36_tests.log
S=./qcrypto-ldpc_amd/host/qldpc_stream
run() { echo "== $*" >> gpurun_out/g36.log; env "$@" 2>&1 | grep -v "^W2" | tail -7 | cut -c1-420 >> gpurun_out/g36.log; }
run QLDPC_RECON_REM_MAX=0 timeout -k 10 120 $S -r 5
run QLDPC_DEBUG=1 timeout -k 10 120 $S -r 3
run QLDPC_RECON_REM_MAX=24 timeout -k 10 120 $S -r 5
run QLDPC_RECON_REM_MAX=63 timeout -k 10 120 $S -r 5
run timeout -k 10 120 $S -r 5 -S 7
run QLDPC_RECON_REM_MAX=0 timeout -k 10 120 $S -r 5 -S 7
cat gpurun_out/g36.log
