# experiment: a layered sweep as ONE launch with a device-wide barrier between the colour layers (qk_cn_layer_grid, QLDPC_LAYER_GRID = workgroups)
set -o pipefail
QLDPC_LAYER_GRID=128 QLDPC_LAYER_CST=0 timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "hlayered or natural_layer or layer_records" > gpurun_out/s8_parity.log 2>&1; rc=$?; echo "parity (grid sweep) rc=$rc"; tail -3 gpurun_out/s8_parity.log
[ $rc -eq 0 ] || exit 1
QLDPC_LAYER_GRID=128 timeout -k 10 300 python -m pytest tests/test_recon_gpu.py -x -q -m gpu > gpurun_out/s8_recon.log 2>&1; rc=$?; echo "recon (grid sweep) rc=$rc"; tail -3 gpurun_out/s8_recon.log
[ $rc -eq 0 ] || exit 1
for pass in 1 2; do
  for wg in 0 64 128 256; do
    if [ $wg = 0 ]; then unset QLDPC_LAYER_GRID; else export QLDPC_LAYER_GRID=$wg; fi
    timeout -k 10 120 qcrypto-ldpc_amd/host/qldpc_stream -b 512 -r 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stream grid=$wg pass $pass: ms_mean %.3f best %.3f Mbit/s %.0f reconciled %d' % (d['ms_mean'], d['ms_best'], d['Mbit_s_mean'], d['reconciled']))" || exit 1
  done
done
unset QLDPC_LAYER_GRID
