cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/g57_tests.log 2>&1 || { tail -40 gpurun_out/g57_tests.log; exit 1; }
tail -2 gpurun_out/g57_tests.log
python __graft_entry__.py smoke 2>&1 | tail -1
S=$(date +%s)
timeout -k 10 900 python bench.py 2>gpurun_out/g57_bench.err > gpurun_out/g57_bench.json; echo "bench rc=$? seconds=$(( $(date +%s) - S ))"
QLDPC_DIST_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --frames 2048 > gpurun_out/rehearsal2_r3b.json 2> gpurun_out/rehearsal2_r3b.err; echo "rehearsal rc=$?"
tail -1 gpurun_out/rehearsal2_r3b.json | cut -c1-400
