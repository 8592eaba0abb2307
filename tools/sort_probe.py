#!/usr/bin/env python3
"""How well does a load-time difficulty predictor group frames for early exit? (developer probe)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import _qldpc_loader  # noqa: E402
import bench  # noqa: E402

q = _qldpc_loader.load()
F = 4096
dev = torch.device("cuda", 0)
code = q.Code.ira(65536, 52429, 0.125, 11, 3, 7)
enc = q.Encoder(code, "IRA")
cw, rx = bench.make_frames(q, torch, code, enc, F, 0.02, 1000, dev)
mag = torch.full((F,), q.bsc_llr(0.02), dtype=torch.float32, device=dev)
cls = torch.zeros(code.N, dtype=torch.uint8, device=dev)
cls[enc.K:] = 1
dec = q.Decoder(code, enc.K, 50, rule="NMS", rule_param=0.75, n_frames=F)
dec.load_bits(rx, mag, cls)
dec.run()
it = dec.fetch_status()[0].cpu().numpy()
x = (rx ^ cw).cpu().numpy().view(np.uint32)
nerr = np.array([int(np.unpackbits(r.view(np.uint8)).sum()) for r in x])
s = dec.syndrome_of(rx).cpu().numpy().view(np.uint32)
sw = np.array([int(np.unpackbits(r.view(np.uint8)).sum()) for r in s])


def eff(order):
    g = it[order].reshape(-1, 64).max(axis=1)
    return it.sum() / (g.sum() * 64.0), g.mean()


print("mean iters %.2f" % it.mean())
print("corr(iters, #errors) = %.3f   corr(iters, syndrome weight) = %.3f" % (np.corrcoef(it, nerr)[0, 1], np.corrcoef(it, sw)[0, 1]))
for name, order in (("arrival order", np.arange(F)), ("sorted by #errors (oracle knowledge)", np.argsort(nerr)), ("sorted by syndrome weight", np.argsort(sw)),
                    ("sorted by true iterations (upper bound)", np.argsort(it))):
    e, gm = eff(order)
    print("  %-42s useful work %.1f %%, group max mean %.2f" % (name, 100 * e, gm))
