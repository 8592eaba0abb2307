cd $GRAFT_REPO_ROOT
for rep in 1 2; do
echo "== old (round-3 tree before the edge-kernel change)"; QLDPC_LIB=$GRAFT_REPO_ROOT/qcrypto-ldpc_amd/variants/libqldpc_r3pre_edge.so python3 tools/edge_latency.py 2>&1 | grep "F="
echo "== new"; python3 tools/edge_latency.py 2>&1 | grep "F="
done
