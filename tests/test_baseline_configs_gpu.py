"""GPU suite: the BASELINE.json `configs` that are parity-test cases rather than bench lines
(SURVEY.md section 8d defines each one's code, frames and seeds)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    return torch


def decode(q, torch, dec, llr):
    dec.load_llr(torch.from_numpy(llr).cuda())
    dec.run()
    hard = q.unpack_bits(dec.fetch_packed().cpu().numpy().view(np.uint32), dec.N)
    it, ok = dec.fetch_status()
    return hard, it.cpu().numpy(), ok.cpu().numpy()


def test_config1_plumbing_case(q, O, torch, gold):
    """configs[0]: BP-flooding SPA, rate-1/2 PEGReg504x1008 (the tree's closest object to "N = 1024"), BSC p = 0.05
    -> |LLR| = 2.944, 1 frame, n_ite = 10, seed 0 (VAR/main.cpp (alist-v1.0.1):21,41)."""
    p = os.path.join(gold, "PEGReg504x1008.alist")
    code, og = q.Code.from_alist(p), O.Graph.from_alist(p)
    enc = q.Encoder(code, "IDENTITY")
    rng = np.random.default_rng(0)
    cw = enc.encode(rng.integers(0, 2, (1, enc.K)))
    mag = np.float32(q.bsc_llr(0.05))
    assert abs(mag - 2.944) < 1e-3
    y = cw ^ (rng.random((1, 1008)) < 0.05)
    llr = np.where(y == 1, -mag, mag).astype(np.float32)
    ref = O.decode(og, llr, "SPA", 0.0, 10)
    for engine in ("edges", "frames"):
        dec = q.Decoder(code, enc.K, 10, info_bits_pos=enc.info_bits_pos, rule="SPA", n_frames=1, engine=engine)
        V = dec.decode_siho(llr)
        assert (V[0] == ref["hard"][0][enc.info_bits_pos]).all(), engine
    assert ref["synd_ok"][0] == 1 and (ref["hard"][0] == cw[0]).all()


def test_config5_million_bit_layered(q, O, torch):
    """configs[4]: irregular IRA N = 10^6, rate 0.8, horizontal-layered NMS, syndrome check every iteration, batch 64.
    The dual-diagonal chain makes the natural check order sequential, so the layered sweep runs over the exported
    colour-class order; the oracle gets H with its rows in that same order.  4 of the 64 frames are checked bit-exactly
    (oracle cost), all 64 through the code's own parity checks."""
    N, K = 1_000_000, 800_000
    code = q.Code.ira(N, K, 0.125, 11, 3, 7)
    assert code.n_layers < 200 and not code.layer_order()[2]
    enc = q.Encoder(code, "IRA")
    rng = np.random.default_rng(50)
    F = 64
    cw = enc.encode(rng.integers(0, 2, (F, K)))
    mag = np.float32(q.bsc_llr(0.02))
    y = cw.copy()
    y[:, :K] ^= rng.random((F, K)) < 0.02
    llr = np.where(y == 1, -mag, mag).astype(np.float32)
    llr[:, K:] = np.where(cw[:, K:] == 1, -np.float32(q.CONFIRMED_BIT_LLR), np.float32(q.CONFIRMED_BIT_LLR))
    dec = q.Decoder(code, K, 50, rule="NMS", rule_param=0.75, n_frames=F, schedule="hlayered")
    hard, it, ok = decode(q, torch, dec, llr)
    assert (ok == 1).all() and (hard == cw).all()
    assert it.max() <= 12                                   # layered converges in about half the flooding iterations
    order, _, _ = code.layer_order()
    var, chk = code.edges()
    inv = np.empty(code.M, np.int32)
    inv[order] = np.arange(code.M, dtype=np.int32)
    newc = inv[chk]
    idx = np.argsort(newc, kind="stable")
    og = O.Graph.from_edges(N, code.M, var[idx], newc[idx])
    ref = O.decode(og, llr[:4], "NMS", 0.75, 50, "hlayered", True, 1, n_threads=4)
    assert (hard[:4] == ref["hard"]).all() and (it[:4] == ref["iters"]).all()


def test_config4_frame_sharding_is_replica_exact(q, O, torch):
    """configs[3] on one GPU: a rank decoding frames [lo, hi) alone gives the same words as the same frames inside a
    bigger batch (no cross-frame coupling anywhere), which is what makes contiguous frame sharding exact."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qcrypto-ldpc_amd"))
    import shard
    code = q.Code.ira(8192, 6554, 0.125, 11, 3, 7)
    rng = np.random.default_rng(4)
    F = 700
    mag = np.float32(q.bsc_llr(0.02))
    llr = np.where(rng.random((F, code.N)) < 0.02, -mag, mag).astype(np.float32)
    llr[:, 6554:] = np.float32(q.CONFIRMED_BIT_LLR)
    whole = decode(q, torch, q.Decoder(code, 6554, 30, rule="NMS", rule_param=0.75, n_frames=F), llr)
    parts = []
    for r in range(3):
        lo, hi = shard.frame_range(F, 3, r)
        parts.append(decode(q, torch, q.Decoder(code, 6554, 30, rule="NMS", rule_param=0.75, n_frames=hi - lo), llr[lo:hi]))
    assert (np.concatenate([p[0] for p in parts]) == whole[0]).all()
    assert (np.concatenate([p[1] for p in parts]) == whole[1]).all()


@pytest.mark.parametrize("F", [4096, 32768])
def test_config2_and_config4_shard_at_full_size_properties(q, torch, F):
    """BASELINE configs 2 (4 096 frames) and 4 (its per-GPU shard of 32 768 frames, 63 GB of decoder state) at full size, through
    size-independent properties: encode -> BSC -> decode round trip (every word equals Alice's), H x = 0 for every decoded
    word computed on the device, linearity (the XOR of two decoded words is a codeword), success flags, and iteration counts in
    the band the small-sample parity tests see.  No oracle at this size: those tests cover the arithmetic."""
    import bench
    dev = torch.device("cuda", 0)
    code = q.Code.ira(65536, 52429, 0.125, 11, 3, 7)
    enc = q.Encoder(code, "IRA")
    K, N = enc.K, code.N
    cw, rx = bench.make_frames(q, torch, code, enc, F, 0.02, 4242, dev)
    assert (cw != rx).any(dim=1).all()                                    # every frame really carries channel errors
    mag = torch.full((F,), q.bsc_llr(0.02), dtype=torch.float32, device=dev)
    cls = torch.zeros(N, dtype=torch.uint8, device=dev)
    cls[K:] = q.VN_PINNED
    dec = q.Decoder(code, K, 50, rule="NMS", rule_param=0.75, n_frames=F)
    dec.load_bits(rx, mag, cls)
    dec.run()
    out = dec.fetch_packed()
    it, ok = dec.fetch_status()
    assert bool((ok == 1).all()) and bool((out == cw).all())
    assert 8 <= int(it.min()) and int(it.max()) <= 20 and 10.5 < float(it.float().mean()) < 12.5
    assert not bool(dec.syndrome_of(out).any())                           # H x = 0 on the device, all frames
    mixed = out[: F // 2] ^ out[F // 2:]
    assert not bool(dec.syndrome_of(mixed.contiguous()).any())            # linearity
    assert bool(dec.syndrome_of(rx).any(dim=1).all())                     # while Bob's raw words are not codewords
    del dec
    torch.cuda.empty_cache()
