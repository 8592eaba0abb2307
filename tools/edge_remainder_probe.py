#!/usr/bin/env python3
"""Developer probe: a ragged remainder of a rate group (14 / 19 / 35 frames) on the edge engine against the same frames as one padded 64-lane
group of the frames engine (SPA, mother code K = 57 344, early exit): is it worth sending remainders to the edge engine in the sessions?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _qldpc_loader
q = _qldpc_loader.load()
import torch
torch.cuda.init()
rng = np.random.default_rng(1)
K = 57344
for R, qber in ((0.7, 0.045), (0.8, 0.028), (0.9, 0.011)):
    M = int(round(K * (1 - R) / R)); N = K + M
    code = q.Code.ira_peg(N, K, 0.125, 11, 3, 2, 7)
    enc = q.Encoder(code, "IRA")
    for F in (14, 19, 35):
        cw = enc.encode(rng.integers(0, 2, (F, K)))
        noisy = cw.copy(); noisy[:, :K] ^= rng.random((F, K)) < qber
        bits = torch.from_numpy(q.pack_bits(noisy).view(np.int32)).cuda()
        mag = torch.full((F,), q.bsc_llr(qber), dtype=torch.float32, device="cuda")
        cls = torch.zeros(N, dtype=torch.uint8, device="cuda"); cls[K:] = 1
        out = {}
        for eng in ("edges", "frames"):
            dec = q.Decoder(code, K, 60, rule="SPA", n_frames=F, engine=eng)
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(5):
                    dec.load_bits(bits, mag, cls); dec.run(); o = dec.fetch_packed()
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            it, ok = dec.fetch_status()
            out[eng] = (dt * 1e3, float(it.float().mean()), int(it.max()), bool(ok.all()))
        print("rate %.1f F=%2d: edges %.2f ms (it %.1f max %d ok %s) | frames %.2f ms (it %.1f max %d ok %s)" % ((R, F) + out["edges"] + out["frames"]), flush=True)
