#!/usr/bin/env python3
"""Developer probe: frame error rate of a shortened + punctured IRA mother code at the configured efficiency.

For a block of KEY bits on a mother code (K_m, rate R): the key VNs past KEY are shortened (known zeros), p = M - ceil(f h(q) KEY)
parity VNs are punctured (evenly spaced along the accumulator chain, or at random), Bob decodes with NMS.  Prints FER and
mean iterations per (mother, QBER, pattern)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _qldpc_loader  # noqa: E402

q = _qldpc_loader.load()
KEY = int(os.environ.get("KEY", "52429"))
F = int(os.environ.get("FRAMES", "128"))
EFF = float(os.environ.get("EFF", "1.4"))
RULE, ALPHA = os.environ.get("RULE", "NMS"), float(os.environ.get("ALPHA", "0.75"))
dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)


def h(x):
    return -x * np.log2(x) - (1 - x) * np.log2(1 - x)


def i32(a):
    return np.ascontiguousarray(a).view(np.int32)


for Km in [int(x) for x in os.environ.get("MOTHERS", "65536,53248").split(",")]:
    for R in (0.5, 0.7, 0.8, 0.9):
        M = int(round(Km * (1 - R) / R))
        N = Km + M
        code = enc = dec = None
        GAP = float(os.environ.get("GAP", "0"))      # minimum distance of the effective rate from the BSC capacity 1 - h(q)
        for qber in [float(x) for x in os.environ.get("QBERS", "0.005,0.01,0.02,0.03,0.04,0.06").split(",")]:
            need = min(1.0 / (1.0 + EFF * h(qber)), 1.0 - h(qber) - GAP)
            rates = [r for r in (0.5, 0.7, 0.8, 0.9) if r <= need]
            if not rates or max(rates) != R:
                continue
            d = int(np.ceil((1.0 / need - 1.0) * KEY))
            p = M - d
            if p < 0:
                print("K_m %d R %.1f q %.3f: needs %d parity bits, mother has %d" % (Km, R, qber, d, M))
                continue
            if code is None:
                code = q.Code.ira_peg(N, Km, depth=int(os.environ["PEG"]), seed=7) if os.environ.get("PEG") else q.Code.ira(N, Km, 0.125, 11, 3, 7)
                enc = q.Encoder(code, "IRA")
                dec = q.Decoder(code, Km, int(os.environ.get("N_ITE", "50")), rule=RULE, rule_param=ALPHA, n_frames=F)
            info = np.zeros((F, Km), np.uint8)
            info[:, :KEY] = rng.integers(0, 2, (F, KEY))
            cw = q.unpack_bits(enc.encode_packed(torch.from_numpy(i32(q.pack_bits(info))).to(dev)).cpu().numpy().view(np.uint32), N)
            rx = cw.copy()
            rx[:, :KEY] ^= (rng.random((F, KEY)) < qber).astype(np.uint8)
            for pattern in ("even",):
                cls = np.zeros(N, np.uint8)
                cls[Km:] = q.VN_PINNED
                j = np.arange(M, dtype=np.int64)
                pun = ((j + 1) * p // M > j * p // M) if pattern == "even" else np.isin(j, rng.permutation(M)[:p])
                cls[Km:][pun] = q.VN_PUNCTURED
                assert pun.sum() == p
                nch = torch.full((F,), KEY, dtype=torch.int32, device=dev)
                mag = torch.full((F,), q.bsc_llr(qber), dtype=torch.float32, device=dev)
                dec.load_bits(torch.from_numpy(i32(q.pack_bits(rx))).to(dev), mag, torch.from_numpy(cls).to(dev), n_channel=nch)
                dec.run()
                out = q.unpack_bits(dec.fetch_packed().cpu().numpy().view(np.uint32), N)
                it, ok = dec.fetch_status()
                good = (out[:, :KEY] == cw[:, :KEY]).all(axis=1) & (ok.cpu().numpy() == 1)
                print("K_m %6d R %.1f q %.3f key %d: M %6d disclosed %6d punctured %6d (%.0f %%) %-6s FER %.3f  mean it %.1f  leak/key %.3f f_eff %.2f" % (
                    Km, R, qber, KEY, M, d, p, 100.0 * p / M, pattern, 1 - good.mean(), it.float().mean().item(), (d + 32) / KEY, d / KEY / h(qber)), flush=True)
