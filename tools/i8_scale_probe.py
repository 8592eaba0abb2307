#!/usr/bin/env python3
"""Developer probe: frame-error rate of the 8-bit fixed-point variant against its quantiser scale (steps per LLR unit),
next to the fp32 decoder, on the config-2 code near the waterfall.  Run on the GPU box."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _qldpc_loader  # noqa: E402

q = _qldpc_loader.load()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
code = q.Code.ira(65536, 52429)
enc = q.Encoder(code, "IRA")
K, N = enc.K, code.N
cls = torch.zeros(N, dtype=torch.uint8, device="cuda")
cls[K:] = q.VN_PINNED
rng = np.random.default_rng(1)
info = torch.from_numpy(rng.integers(0, 2 ** 31, (F, (K + 31) // 32), dtype=np.int64).astype(np.int32)).cuda()
cw = enc.encode_packed(info) if hasattr(enc, "encode_packed") else None
for qber in (0.028, 0.030, 0.031, 0.032):
    flips = torch.from_numpy(q.pack_bits((rng.random((F, N)) < qber).astype(np.uint8)).view(np.int32)).cuda()
    mask = torch.from_numpy(q.pack_bits(np.concatenate([np.ones(K, np.uint8), np.zeros(N - K, np.uint8)])[None, :]).view(np.int32)).cuda()
    rx = cw ^ (flips & mask)
    mag = torch.full((F,), q.bsc_llr(qber), dtype=torch.float32, device="cuda")
    out = torch.empty_like(cw)
    row = []
    for name, kw in [("f32", {})] + [("i8 s=%g" % s, dict(msg_dtype="i8", quant_scale=s)) for s in (2.0, 3.0, 4.0, 6.0, 8.0)] + \
                    [("i8 OMS.5 s=4", dict(msg_dtype="i8", quant_scale=4.0, rule="OMS", rule_param=0.5))]:
        a = dict(rule="NMS", rule_param=0.75)
        a.update(kw)
        dec = q.Decoder(code, K, 50, n_frames=F, **a)
        dec.load_bits(rx, mag, cls)
        dec.run()
        dec.fetch_packed(out)
        it, ok = dec.fetch_status()
        good = ((out == cw).all(dim=1)) & (ok == 1)
        row.append("%s: FER %.3f it %.1f" % (name, 1 - good.float().mean().item(), it.float().mean().item()))
        del dec
    print("QBER %.3f | " % qber + " | ".join(row), flush=True)
