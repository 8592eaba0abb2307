cd $GRAFT_REPO_ROOT
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
# daemon soak on the layered session decoders: one daemon pair, 64 blocks, batched ingest of 16 (layered) and of 4 (flooding / edge engine)
timeout -k 10 400 python tests/ecd2_loop.py ecd2_ldpc_urandom single=64 b16,w50 2>&1 | grep -v amdgpu.ids | tail -2
timeout -k 10 400 python tests/ecd2_loop.py ecd2_ldpc_urandom single=64 b4,w50 2>&1 | grep -v amdgpu.ids | tail -2
timeout -k 10 400 python tests/fuzz_parity.py 240 2026 2>&1 | tail -1
