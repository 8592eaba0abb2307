for f in 64 128 256 512; do
  QLDPC_BENCH_C5_FRAMES=$f python bench.py --steps 1 --warmup 1 --no-cpu --no-config3 --no-early --no-fp16 --no-int8 > gpurun_out/c5_$f.json 2>gpurun_out/c5_$f.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/c5_$f.json").read().strip().splitlines()[-1])["config5_layered_1e6"]
print($f, "fixed", d["fixed"]["value"], d["fixed"]["roofline"]["frac"], "early", d["early_exit"]["value"], d["early_exit"]["roofline"]["frac"], d["early_exit"]["avg_sweeps"], flush=True)
PY
done
