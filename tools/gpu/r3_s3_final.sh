# round-end rehearsal of the third session: the driver's sequence, a 2-rank gloo rehearsal of the N > 1 path on the one GPU, and the rocprofv3 kernel stats of the default bench command
set -o pipefail
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 || exit 1
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/final_gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 gpurun_out/final_gpu_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; rc=$?; echo "bench rc=$rc"
[ $rc -eq 0 ] || exit 1
QLDPC_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --frames 2048 > gpurun_out/rehearsal2.json 2> gpurun_out/rehearsal2.err; rc=$?; echo "rehearsal rc=$rc"; tail -c 600 gpurun_out/rehearsal2.json | head -c 300; echo
[ $rc -eq 0 ] || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu > /tmp/prof_c.log 2>&1; echo "rocprof rc=$?"
f=$(find /tmp/prof_c -name "*kernel_stats.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/final_bench_kernel_stats.csv; head -4 $GRAFT_REPO_ROOT/gpurun_out/final_bench_kernel_stats.csv | cut -c1-160
