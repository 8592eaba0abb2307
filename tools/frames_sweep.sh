# developer sweep: the headline leg (+ early exit) against the batch size; usage: bash tools/frames_sweep.sh 64 128 ...
for f in "$@"; do
  python bench.py --frames $f --steps 2 --warmup 1 --no-cpu --no-config3 --no-config5 --no-fp16 --no-int8 > gpurun_out/sweep_$f.json 2>gpurun_out/sweep_$f.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/sweep_$f.json").read().strip().splitlines()[-1])
r=d["roofline"]
print($f, "value", round(d["value"]), "ms", round(d["ms_per_step"],2), "cn", round(r["achieved"]), "vn", round(r["vn_update"].get("achieved")), "whole", round(r["whole_step"].get("moved_frac"),3), "early", round(d["early_exit"]["value"]), flush=True)
PY
done
