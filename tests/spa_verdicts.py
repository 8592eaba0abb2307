#!/usr/bin/env python3
"""SURVEY.md section 8c, the tolerance stated for the float-LLR SPA variant: >= 99.9 % identical frame verdicts against the oracle
and FER within the binomial 95 % interval, over >= 10^4 frames (test infrastructure; run on the GPU box).
usage: spa_verdicts.py [frames] [qber]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import _qldpc_loader  # noqa: E402
import oracle as O  # noqa: E402

q = _qldpc_loader.load()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
qber = float(sys.argv[2]) if len(sys.argv) > 2 else 0.03
code = q.Code.ira(65536, 52429)
enc = q.Encoder(code, "IRA")
K, N = enc.K, code.N
var, chk = code.edges()
og = O.Graph.from_edges(N, code.M, var, chk)
rng = np.random.default_rng(2)
mag = np.float32(q.bsc_llr(qber))
dec = q.Decoder(code, K, 50, rule="SPA", n_frames=2048)
agree = n = gpu_fail = orc_fail = 0
same_word = 0
for lo in range(0, F, 2048):
    nb = min(2048, F - lo)
    info = rng.integers(0, 2, (nb, K)).astype(np.uint8)
    cw = enc.encode(info)
    noisy = cw.copy()
    noisy[:, :K] ^= rng.random((nb, K)) < qber
    llr = np.where(noisy == 1, -mag, mag).astype(np.float32)
    llr[:, K:] = np.where(cw[:, K:] == 1, -np.float32(q.CONFIRMED_BIT_LLR), np.float32(q.CONFIRMED_BIT_LLR))
    ref = O.decode(og, llr, "SPA", 0.0, 50, n_threads=os.cpu_count() or 8)
    dec.load_llr(torch.from_numpy(llr).cuda())
    dec.run()
    hard = q.unpack_bits(dec.fetch_packed().cpu().numpy().view(np.uint32), N)[:nb]
    it, ok = dec.fetch_status()
    ok = ok.cpu().numpy()[:nb]
    g_good = (hard == cw).all(1) & (ok == 1)
    o_good = (ref["hard"] == cw).all(1) & (ref["synd_ok"] == 1)
    agree += int((g_good == o_good).sum())
    same_word += int(((hard == ref["hard"]).all(1))[o_good].sum())
    gpu_fail += int((~g_good).sum())
    orc_fail += int((~o_good).sum())
    n += nb
    print("%d frames: verdicts agree %.4f, FER gpu %.4f oracle %.4f" % (n, agree / n, gpu_fail / n, orc_fail / n), flush=True)
p = orc_fail / n
ci = 1.96 * np.sqrt(max(p * (1 - p), 1e-9) / n)
print("SPA, QBER %.3f, %d frames: identical verdicts %.5f; FER gpu %.5f, oracle %.5f +- %.5f (95 %%); identical words among oracle-converged %.5f"
      % (qber, n, agree / n, gpu_fail / n, p, ci, same_word / max(1, n - orc_fail)))
