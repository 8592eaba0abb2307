# round-end rehearsal on the GPU box: the driver's sequence (smoke, GPU suite, default bench) + the rocprofv3 kernel stats of the bench command
set -o pipefail
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/final_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/final_gpu_tests.log
timeout -k 10 400 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu > /tmp/prof_c.log 2>&1; echo "rocprof rc=$?"
f=$(find /tmp/prof_c -name "*kernel_stats.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/final_bench_kernel_stats.csv; head -3 $GRAFT_REPO_ROOT/gpurun_out/final_bench_kernel_stats.csv | cut -c1-120
