#!/usr/bin/env python3
"""Per-kernel-family time split of one early-exit decode of the bench workload (developer tool)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import _qldpc_loader  # noqa: E402
import bench  # noqa: E402

q = _qldpc_loader.load()
F = int(os.environ.get("FRAMES", "4096"))
DTYPE = os.environ.get("DTYPE", "f32")
COMPACT = os.environ.get("COMPACT", "auto")
dev = torch.device("cuda", 0)
code = q.Code.ira(65536, 52429, 0.125, 11, 3, 7)
enc = q.Encoder(code, "IRA")
cw, rx = bench.make_frames(q, torch, code, enc, F, 0.02, 1000, dev)
mag = torch.full((F,), q.bsc_llr(0.02), dtype=torch.float32, device=dev)
cls = torch.zeros(code.N, dtype=torch.uint8, device=dev)
cls[enc.K:] = 1
for synd in (True, False):
    dec = q.Decoder(code, enc.K, 50, rule="NMS", rule_param=0.75, enable_syndrome=synd, n_frames=F, msg_dtype=DTYPE, compact=COMPACT)
    dec.set_stream(torch.cuda.current_stream())
    import time
    for rep in range(4):
        dec.profile(rep == 3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dec.load_bits(rx, mag, cls)
        dec.run()
        out = dec.fetch_packed()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
    print("wall of the step before the profiled one: %.3f ms (%s)" % (wall, dec.last_run_stats()))
    it, ok = dec.fetch_status()
    st = dec.profile_read()
    tot = sum(s["total_ms"] for s in st)
    print("enable_syndrome=%s: launched %d iterations, avg %.2f, max %d; kernel time %.2f ms" % (synd, dec.last_run_iterations, it.float().mean().item(), it.max().item(), tot))
    hist = np.bincount(it.cpu().numpy(), minlength=20)
    if synd:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.save(os.path.join(ROOT, "gpurun_out", "iters_%d.npy" % F), it.cpu().numpy())
    print("  iteration histogram:", {i: int(c) for i, c in enumerate(hist) if c})
    g = it.cpu().numpy().reshape(-1, 64).max(axis=1)
    print("  per-64-frame-group max: mean %.2f  -> ideal group-granular work %.1f%% of launched" % (g.mean(), 100 * g.mean() / dec.last_run_iterations))
    for s in st:
        print("  %-14s launches %4d  total %8.3f ms  avg %7.3f ms  alg %7.1f GB/s" % (s["name"], s["launches"], s["total_ms"], s["total_ms"] / s["launches"],
                                                                                 s["alg_bytes"] / max(s["total_ms"], 1e-9) / 1e6))
