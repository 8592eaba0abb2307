cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
timeout -k 10 600 python3 tools/edge_remainder_probe.py 2>&1 | grep -v amdgpu.ids
