cd $GRAFT_REPO_ROOT
for rep in 1 2; do
echo "== new, launches"; python3 tools/edge_latency.py 2>&1 | grep "F="
echo "== new, QLDPC_GRAPH=1"; QLDPC_GRAPH=1 python3 tools/edge_latency.py 2>&1 | grep "F="
echo "== new, QLDPC_GRAPH=1 QLDPC_POLL_EVERY=4"; QLDPC_GRAPH=1 QLDPC_POLL_EVERY=4 python3 tools/edge_latency.py 2>&1 | grep "F="
done
