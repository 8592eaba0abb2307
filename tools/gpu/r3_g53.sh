cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "compressed_check_state or layered or syndrome or fuzz" > gpurun_out/g53_tests.log 2>&1 || { tail -40 gpurun_out/g53_tests.log; exit 1; }
tail -2 gpurun_out/g53_tests.log
S=$(date +%s)
timeout -k 10 900 python bench.py 2>gpurun_out/g53_bench.err > gpurun_out/g53_bench.json; echo "bench rc=$? seconds=$(( $(date +%s) - S ))"
python3 tools/benchsum.py gpurun_out/g53_bench.json 2>/dev/null | tail -30 || true
