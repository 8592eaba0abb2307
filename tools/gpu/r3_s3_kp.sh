# kernarg preload (-mllvm -amdgpu-kernarg-preload-count=16 on qldpc_launch.hip) for the small layer launches: same-box A/B through QLDPC_LIB
set -o pipefail
B="--steps 3 --warmup 1 --no-early --no-fp16 --no-int8 --no-config3 --no-cpu --no-fer-deep --no-spa"
for pass in 1 2; do
  for lib in qcrypto-ldpc_amd/libqldpc.so qcrypto-ldpc_amd/variants/libqldpc_kp.so; do
    QLDPC_LIB=$PWD/$lib timeout -k 10 200 python bench.py $B > gpurun_out/s6_bench.json 2> gpurun_out/s6_bench.err || exit 1
    python - <<P
import json
d=json.loads(open('gpurun_out/s6_bench.json').read().strip().splitlines()[-1])
l=d['layered_schedule']; c=d['config5_layered_1e6']
print('$lib pass ${pass}: headline %.0f | layered fixed %.0f early %.0f | config5 fixed %.0f (moved %.3f) early %.0f | 256: fixed %.0f early %.0f' % (d['value'], l['fixed']['value'], l['early_exit']['value'], c['fixed']['value'], c['fixed']['roofline']['moved_frac'], c['early_exit']['value'], c['at_256_frames']['fixed']['value'], c['at_256_frames']['early_exit']['value']))
P
  done
done
