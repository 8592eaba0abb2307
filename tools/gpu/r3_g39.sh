cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
timeout -k 10 420 python3 tests/fuzz_parity.py 300 31 > gpurun_out/g39_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -4 gpurun_out/g39_fuzz.log; grep -c "chain=on" gpurun_out/g39_fuzz.log
