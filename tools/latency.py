#!/usr/bin/env python3
"""Single-block latency of the two engines (developer tool): one 65 536-VN block at QBER 2 %, the daemon's case."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _qldpc_loader  # noqa: E402

q = _qldpc_loader.load()
import torch  # noqa: E402

torch.cuda.init()

code = q.Code.ira(65536, 52429, 0.125, 11, 3, 7)
enc = q.Encoder(code, "IRA")
rng = np.random.default_rng(1)
for F in (1, 4, 16):
    cw = enc.encode(rng.integers(0, 2, (F, enc.K)))
    noisy = cw.copy()
    noisy[:, :enc.K] ^= rng.random((F, enc.K)) < 0.02
    bits = torch.from_numpy(q.pack_bits(noisy).astype(np.int64).astype(np.uint32).view(np.int32)).cuda()
    mag = torch.full((F,), q.bsc_llr(0.02), dtype=torch.float32, device="cuda")
    cls = torch.zeros(code.N, dtype=torch.uint8, device="cuda")
    cls[enc.K:] = 1
    for engine in ("edges", "frames"):
        dec = q.Decoder(code, enc.K, 50, rule="NMS", rule_param=0.75, n_frames=F, engine=engine)
        dec.set_stream(torch.cuda.current_stream())
        out = None
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 20
            for _ in range(n):
                dec.load_bits(bits, mag, cls)
                dec.run()
                out = dec.fetch_packed()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        it, ok = dec.fetch_status()
        good = bool((out.cpu().numpy().view(np.uint32) == q.pack_bits(cw)).all())
        print("F=%2d engine=%-6s %8.1f us/decode  %7.1f Mbit/s key  iters %s launched %d ok %s correct %s" % (
            F, engine, dt * 1e6, F * enc.K / dt / 1e6, it.cpu().numpy().tolist()[:4], dec.last_run_iterations, bool(ok.all()), good), flush=True)
# host-buffer reconciliation session (what the ecd2 handler calls), one 60 000-bit block
r = q.Recon(max_blocks=1)
a = rng.integers(0, 2, 60000).astype(np.uint8)
b = a ^ (rng.random(60000) < 0.02)
aw, bw = q.pack_bits(a), q.pack_bits(b)
msg, par = r.encode(aw, 60000, 0.02)
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(10):
        ok, fixed, c, l, it = r.decode(bw, 60000, 0.02, msg, par)
    dt = (time.perf_counter() - t0) / 10
print("recon_decode (host buffers, 60000-bit block): %.1f us, ok %s, %d iterations, %d corrected, leak %d" % (dt * 1e6, ok, it, c, l))
t0 = time.perf_counter()
for _ in range(10):
    r.encode(aw, 60000, 0.02)
print("recon_encode: %.1f us" % ((time.perf_counter() - t0) / 10 * 1e6))

# ---- first-block and cache-miss latency of the sessions: a code per block size (round 1) vs preloaded mother codes ----------
sizes = [60000, 41935, 33000, 52429, 20011, 64999, 60000]
for label, kw in (("per-size codes, built on first use (mother_step=0)", dict(mother_step=0)),
                  ("mother codes, built on first use", dict()),
                  ("mother codes preloaded at create (ldpc_init)", dict(preload=True))):
    t0 = time.perf_counter()
    ra, rb = q.Recon(max_blocks=1, **kw), q.Recon(max_blocks=1, **kw)
    t_create = time.perf_counter() - t0
    lat = []
    for kb in sizes:
        x = rng.integers(0, 2, kb).astype(np.uint8)
        y = x ^ (rng.random(kb) < 0.02)
        m, p = ra.encode(q.pack_bits(x), kb, 0.02)
        before = rb.entries_created
        t0 = time.perf_counter()
        okb, *_ = rb.decode(q.pack_bits(y), kb, 0.02, m, p)
        lat.append(((time.perf_counter() - t0) * 1e3, rb.entries_created - before, bool(okb)))
    print("%-52s create %7.1f ms | decode latency per block [ms (codes built)]: %s" % (
        label, t_create * 1e3, "  ".join("%.2f (%d)%s" % (l, n, "" if o else "!") for l, n, o in lat)), flush=True)
