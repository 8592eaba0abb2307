cd $GRAFT_REPO_ROOT
timeout -k 10 800 python bench.py --frames 32768 --steps 3 --warmup 1 --no-config3 --no-config5 --no-fer-deep --no-fp16 --no-int8 --no-cpu 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config 4 shard on one GPU (32768 frames):', round(d['value'],1), 'Mbit/s fixed-50,', round(d['ms_per_step'],1), 'ms/step, FER', d['fer'], ', check pass frac', round(d['roofline']['frac'],3), ', early exit', round(d['early_exit']['value'],1), 'useful', round(d['early_exit']['useful_work'],3))
" | tee gpurun_out/g41_config4_shard.txt
