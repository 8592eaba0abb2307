cd $GRAFT_REPO_ROOT
export QLDPC_DIST_BACKEND=gloo
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --frames 2048 > gpurun_out/rehearsal2_r3.json 2> gpurun_out/rehearsal2_r3.err; echo "rc=$?"
tail -3 gpurun_out/rehearsal2_r3.err
python3 -c "
import json
d=json.loads(open('gpurun_out/rehearsal2_r3.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('metric','value','n_gpus','steps','ms_per_step','scaling','fer','frames_per_step')}, d['config']['parallelism'], d['early_exit']['value'], d['config3_multirate_stream'], d['fer_deep'])
"
