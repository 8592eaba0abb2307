cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
QLDPC_DEBUG=1 QLDPC_LAYER_CHAIN=1 timeout -k 10 500 python bench.py --steps 1 --warmup 1 --no-early --no-fp16 --no-int8 --no-config3 --no-cpu --no-fer-deep --config5-frames 64,256 2>&1 >/dev/null | grep "one-launch layered sweeps:" | sort | uniq -c | sort -rn | head -8
