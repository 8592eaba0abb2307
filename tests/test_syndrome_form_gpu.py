"""GPU suite: syndrome-form (coset) decoding -- every check must come out with Alice's parity s = H x_A instead of 0
("support both in the ABI").  Works on any H without an encoder; checked bit-for-bit against the
oracle's coset mode (min-sum family)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    return torch


def i32(words):
    return words.astype(np.int64).astype(np.uint32).view(np.int32)


def setup(q, O, gold, rng, F, qber):
    p = os.path.join(gold, "PEGReg504x1008.alist")
    code, og = q.Code.from_alist(p), O.Graph.from_alist(p)
    x = rng.integers(0, 2, (F, 1008)).astype(np.uint8)
    s = np.stack([og.syndrome(xx)[1] for xx in x])
    y = x ^ (rng.random((F, 1008)) < qber)
    mag = np.float32(q.bsc_llr(qber))
    return code, og, x, s, y, mag


@pytest.mark.parametrize("engine,F,sched,V", [("frames", 150, "flooding", 1), ("frames", 150, "flooding", 2), ("frames", 150, "hlayered", 1),
                                             ("edges", 5, "flooding", 1), ("edges", 1, "flooding", 1)])
def test_syndrome_form_bit_exact(q, O, torch, gold, engine, F, sched, V):
    rng = np.random.default_rng(F + V)
    code, og, x, s, y, mag = setup(q, O, gold, rng, F, 0.045)
    llr = np.where(y == 1, -mag, mag).astype(np.float32)
    og2 = og
    s2 = s
    if sched == "hlayered":
        order, _, _ = code.layer_order()
        var, chk = og.edges()
        inv = np.empty(code.M, np.int32); inv[order] = np.arange(code.M, dtype=np.int32)
        newc = inv[chk]; idx = np.argsort(newc, kind="stable")
        og2 = O.Graph.from_edges(code.N, code.M, var[idx], newc[idx])
        s2 = s[:, order]                                             # row-permuted H -> row-permuted syndrome
    ref = O.decode(og2, llr, "NMS", 0.75, 30, sched, True, 1, n_threads=8, target=s2)
    dec = q.Decoder(code, 1008, 30, rule="NMS", rule_param=0.75, n_frames=F, schedule=sched, engine=engine, frames_per_lane=V, freeze_messages=True)
    dec.load_bits(torch.from_numpy(i32(q.pack_bits(y))).cuda(), torch.full((F,), float(mag), device="cuda"))
    dec.load_syndrome(torch.from_numpy(i32(q.pack_bits(s))).cuda())
    dec.run()
    hard = q.unpack_bits(dec.fetch_packed().cpu().numpy().view(np.uint32), 1008)
    it, ok = dec.fetch_status()
    it, ok = it.cpu().numpy(), ok.cpu().numpy()
    assert (hard == ref["hard"]).all() and (it == ref["iters"]).all() and (ok == ref["synd_ok"]).all()
    post = dec.fetch_post().cpu().numpy()
    assert (post.view(np.uint32) == ref["post"].view(np.uint32)).all()
    good = ok == 1
    assert good.mean() > 0.8 and (hard[good] == x[good]).all()      # Bob ends with Alice's key


def test_syndrome_of_and_reload_clears_the_target(q, O, torch, gold):
    rng = np.random.default_rng(3)
    code, og, x, s, y, mag = setup(q, O, gold, rng, 70, 0.03)
    dec = q.Decoder(code, 1008, 20, rule="NMS", rule_param=0.75, n_frames=70)
    got = dec.syndrome_of(torch.from_numpy(i32(q.pack_bits(x))).cuda()).cpu().numpy().view(np.uint32)
    assert (q.unpack_bits(got, 504) == s).all()
    bits = torch.from_numpy(i32(q.pack_bits(y))).cuda()
    m = torch.full((70,), float(mag), device="cuda")
    with pytest.raises(q.QldpcError) as e:
        dec.load_syndrome(torch.from_numpy(i32(q.pack_bits(s))).cuda())      # nothing loaded yet
    assert e.value.status == -8
    dec.load_bits(bits, m)
    dec.load_syndrome(torch.from_numpy(i32(q.pack_bits(s))).cuda())
    dec.run()
    ok1 = dec.fetch_status()[1].cpu().numpy()
    dec.load_bits(bits, m)                                           # a fresh load means H x = 0 again
    dec.run()
    ok2 = dec.fetch_status()[1].cpu().numpy()
    assert ok1.mean() > 0.9 and ok2.mean() < 0.1


@pytest.mark.parametrize("V", [1, 2])
def test_syndrome_form_with_binary16_messages(q, O, torch, gold, V):
    """V = 2 takes the packed check-node kernel (qldpc_kernels_h16.h), V = 1 the generic one: both must equal the rounding oracle."""
    rng = np.random.default_rng(40 + V)
    F = 200
    code, og, x, s, y, mag = setup(q, O, gold, rng, F, 0.045)
    llr = np.where(y == 1, -mag, mag).astype(np.float32)
    for rule, param in (("NMS", 0.75), ("OMS", 0.3), ("NMS", 0.7)):       # 0.7 is not a binary16 number: the rule must stay in fp32
        ref = O.decode(og, llr, rule, param, 30, "flooding", True, 1, n_threads=8, target=s, msg_fp16=True)
        dec = q.Decoder(code, 1008, 30, rule=rule, rule_param=param, n_frames=F, frames_per_lane=V, msg_dtype="f16")
        dec.load_bits(torch.from_numpy(i32(q.pack_bits(y))).cuda(), torch.full((F,), float(mag), device="cuda"))
        dec.load_syndrome(torch.from_numpy(i32(q.pack_bits(s))).cuda())
        dec.run()
        hard = q.unpack_bits(dec.fetch_packed().cpu().numpy().view(np.uint32), 1008)
        it, ok = dec.fetch_status()
        assert (hard == ref["hard"]).all() and (it.cpu().numpy() == ref["iters"]).all() and (ok.cpu().numpy() == ref["synd_ok"]).all()
