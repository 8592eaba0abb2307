cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
mkdir -p gpurun_out
python3 -c "
import sys; sys.path.insert(0,'.'); import _qldpc_loader; q=_qldpc_loader.load(); q.Recon(preload=True); print('cache warm')"
timeout -k 10 500 python3 tests/ecd2_loop.py ecd2_ldpc_urandom single=96 b4,w50 2>&1 | tail -3
timeout -k 10 300 python3 tests/ecd2_loop.py ecd2_ldpc_urandom single=48 m20 2>&1 | tail -3
timeout -k 10 300 python3 tests/ecd2_loop.py ecd2_ldpc_urandom 6 2>&1 | tail -3
