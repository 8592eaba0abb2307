cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
python3 -c "
import sys; sys.path.insert(0,'.'); import _qldpc_loader; q=_qldpc_loader.load(); q.Recon(preload=True); print('cache warm')"
timeout -k 10 1000 python3 tools/daemon_yield.py 128 m0 m10 m20 m30 G1,m0 G1,m10 G1,m20 2>&1 | tee gpurun_out/g29_daemon_yield.txt
