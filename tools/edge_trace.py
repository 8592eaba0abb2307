#!/usr/bin/env python3
"""Where one edge-engine decode spends its time: run under `rocprofv3 --kernel-trace -- python3 tools/edge_trace.py run`,
then `python3 tools/edge_trace.py report <kernel_trace.csv>` prints, for the one-block decode (N = 65 536, rate 0.8, NMS 0.75, QBER 2 %):
kernels per decode, time inside kernels by kind, idle time between consecutive kernels, and the decode's span on the device."""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ROOT))


def run():
    import numpy as np
    import torch
    import _qldpc_loader
    q = _qldpc_loader.load()
    torch.cuda.init()
    code = q.Code.ira(65536, 52429, 0.125, 11, 3, 7)
    enc = q.Encoder(code, "IRA")
    rng = np.random.default_rng(1)
    cw = enc.encode(rng.integers(0, 2, (1, enc.K)))
    noisy = cw.copy()
    noisy[:, :enc.K] ^= rng.random((1, enc.K)) < 0.02
    bits = torch.from_numpy(q.pack_bits(noisy).view(np.int32)).cuda()
    mag = torch.full((1,), q.bsc_llr(0.02), dtype=torch.float32, device="cuda")
    cls = torch.zeros(code.N, dtype=torch.uint8, device="cuda")
    cls[enc.K:] = 1
    dec = q.Decoder(code, enc.K, 50, rule="NMS", rule_param=0.75, n_frames=1, engine="edges")
    for _ in range(60):
        dec.load_bits(bits, mag, cls)
        dec.run()
        out = dec.fetch_packed()
    torch.cuda.synchronize()
    it, ok = dec.fetch_status()
    print("iterations", it.cpu().numpy().tolist(), "ok", bool(ok.all()), "correct", bool((out.cpu().numpy().view(np.uint32) == q.pack_bits(cw)).all()))


def report(path):
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(path))), key=lambda x: x[0])
    starts = [i for i, r in enumerate(rows) if "qe_load_bits" in r[2]]
    decodes = [rows[a:b] for a, b in zip(starts, starts[1:])][10:]      # skip the warm-up decodes
    n = len(decodes)
    kinds, gaps, span, count = {}, 0.0, 0.0, 0
    for d in decodes:
        d = [k for k in d if k[2].startswith(("void qe_", "qe_"))]
        count += len(d)
        span += (d[-1][1] - d[0][0]) / 1e3
        for k in d:
            name = k[2].replace("void ", "")
            kinds[name] = kinds.get(name, [0, 0.0])
            kinds[name][0] += 1
            kinds[name][1] += (k[1] - k[0]) / 1e3
        gaps += sum(max(0, b[0] - a[1]) for a, b in zip(d, d[1:])) / 1e3
    print("%d decodes: %.1f kernels each, device span %.1f us per decode (first kernel start -> last kernel end), idle between kernels %.1f us (%.2f us per boundary)" % (
        n, count / n, span / n, gaps / n, gaps / max(1, count - n)))
    for name, (c, t) in sorted(kinds.items(), key=lambda kv: -kv[1][1]):
        print("  %-28s %5.1f launches per decode, %6.2f us each, %6.1f us per decode" % (name, c / n, t / c, t / n))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
