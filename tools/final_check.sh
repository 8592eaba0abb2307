set -o pipefail
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/final_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/final_gpu_tests.log
timeout -k 10 400 python bench.py > gpurun_out/bench_r02_c.json 2> gpurun_out/bench_r02_c.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu > /tmp/prof_c.log 2>&1; echo "rocprof rc=$?"
f=$(find /tmp/prof_c -name "*kernel_stats.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r02c_bench_kernel_stats.csv; head -5 $GRAFT_REPO_ROOT/gpurun_out/r02c_bench_kernel_stats.csv | cut -c1-160
