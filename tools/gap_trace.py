#!/usr/bin/env python3
"""Developer tool: where the time of one early-exit step goes, kernel by kernel, from a rocprofv3 kernel trace.
  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/gap_trace.py run [f32|i8|f16] [fixed]
  python3 tools/gap_trace.py read $OUT
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "run":
    import torch

    import _qldpc_loader
    import bench
    q = _qldpc_loader.load()
    dt = sys.argv[2] if len(sys.argv) > 2 else "f32"
    F = 4096
    dev = torch.device("cuda", 0)
    code = q.Code.ira(65536, 52429, 0.125, 11, 3, 7)
    enc = q.Encoder(code, "IRA")
    cw, rx = bench.make_frames(q, torch, code, enc, F, 0.02, 1000, dev)
    mag = torch.full((F,), q.bsc_llr(0.02), dtype=torch.float32, device=dev)
    cls = torch.zeros(code.N, dtype=torch.uint8, device=dev)
    cls[enc.K:] = 1
    dec = q.Decoder(code, enc.K, 50, rule="NMS", rule_param=0.75, enable_syndrome=(len(sys.argv) < 4 or sys.argv[3] != "fixed"), n_frames=F, msg_dtype=dt)
    dec.set_stream(torch.cuda.current_stream())
    out = torch.empty((F, 2048), dtype=torch.int32, device=dev)
    for rep in range(3):
        torch.cuda.synchronize()
        mark = torch.zeros(1 + rep, device=dev).fill_(1.0)      # a recognisable kernel between the steps (fill of 1, 2, 3 elements)
        dec.load_bits(rx, mag, cls)
        dec.run()
        dec.fetch_packed(out)
        torch.cuda.synchronize()
    sys.exit(0)

rows = []
for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last step: everything after the last qk_load_bits-like kernel ("load" family starts the step)
starts = [i for i, n in enumerate(names) if "load_bits" in n or "qk_load" in n or "qi_load" in n]
i0 = starts[-1]
step = rows[i0:]
t0, t1 = int(step[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in step)
busy, gaps, fam = 0, [], {}
prev_end = None
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    k = r["Kernel_Name"].split("<")[0].split("(")[0].replace("void ", "")
    a = fam.setdefault(k, [0, 0, 0])
    a[0] += 1
    a[1] += e - s
    if prev_end is not None:
        a[2] += max(0, s - prev_end)
        gaps.append((max(0, s - prev_end), k))
    prev_end = max(prev_end or 0, e)
print("step: %.3f ms from first kernel start to last kernel end, %.3f ms inside kernels, %.3f ms idle between them, %d launches" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(step)))
print("%-28s %7s %10s %12s" % ("kernel", "calls", "busy ms", "idle before ms"))
for k, a in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print("%-28s %7d %10.3f %12.3f" % (k, a[0], a[1] / 1e6, a[2] / 1e6))
gaps.sort(reverse=True)
print("largest gaps (us, before kernel):", [(round(g / 1e3, 1), k) for g, k in gaps[:12]])
