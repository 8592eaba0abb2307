"""Active-frame compaction of the early-exit mode (qldpc_decoder_cfg.compact, csrc/qldpc_kernels_compact.h) -- GPU parity (-m gpu).

AFF3CT stops every frame on its own (Decoder_LDPC_BP_flooding with enable_syndrome, VAR/main.cpp (alist-v1.0.1):203-218); the
batched decoder keeps a converged frame in its lane and, with compaction, deals the frames still running into fewer groups.
That must not change anything a caller can see: hard decisions, iteration counts and success flags stay those of the oracle
(and of the uncompacted decoder), in the caller's frame order, for every message width, LLR form and the syndrome form.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


@pytest.fixture(scope="module")
def peg(q, O, gold):
    import os
    p = os.path.join(gold, "PEGReg504x1008.alist")
    return q.Code.from_alist(p), O.Graph.from_alist(p)


def i32(a):
    return np.ascontiguousarray(a).view(np.int32)


def frames(rng, F, N, lo=0.03, hi=0.075):
    """per-frame crossover probabilities spread over the waterfall, so the iteration counts are spread too (and some frames fail)"""
    p = rng.uniform(lo, hi, F)
    y = (rng.random((F, N)) < p[:, None]).astype(np.uint8)
    return y, p


def run(q, torch, dec, llr=None, bits=None, mag=None, cls=None, synd=None):
    if llr is not None:
        dec.load_llr(torch.from_numpy(llr).cuda())
    else:
        dec.load_bits(torch.from_numpy(i32(q.pack_bits(bits))).cuda(), torch.from_numpy(mag).cuda(), None if cls is None else torch.from_numpy(cls).cuda())
    if synd is not None:
        dec.load_syndrome(torch.from_numpy(i32(q.pack_bits(synd))).cuda())
    dec.run()
    hard = q.unpack_bits(dec.fetch_packed().cpu().numpy().view(np.uint32), dec.N)
    it, ok = dec.fetch_status()
    return hard, it.cpu().numpy(), ok.cpu().numpy()


@pytest.mark.parametrize("dtype,V", [("f32", 1), ("f32", 2), ("f32", 4), ("f16", 0), ("i8", 0)])
@pytest.mark.parametrize("form", ["llr", "bits"])
def test_compacted_run_equals_oracle_and_uncompacted_run(q, O, torch, peg, dtype, V, form):
    code, og = peg
    rng = np.random.default_rng(40 + V)
    F, N, n_ite = 1100, 1008, 30
    y, p = frames(rng, F, N)
    mag = np.full(F, 2.6, np.float32)
    llr = np.where(y == 1, -mag[:, None], mag[:, None]).astype(np.float32)
    ref = O.decode(og, llr, "NMS", 0.75, n_ite, "flooding", True, 1, n_threads=8, msg_fp16=(dtype == "f16"), msg_i8=(dtype == "i8"))
    assert len(np.unique(ref["iters"])) >= 6 and 0 < (ref["synd_ok"] == 0).sum() < F // 2      # a spread of iteration counts, both outcomes
    kw = dict(llr=llr) if form == "llr" else dict(bits=y, mag=mag)
    out = {}
    for mode in ("off", "on"):
        dec = q.Decoder(code, N, n_ite, rule="NMS", rule_param=0.75, n_frames=F, frames_per_lane=V, msg_dtype=dtype, compact=mode)
        hard, it, ok = run(q, torch, dec, **kw)
        st = dec.last_run_stats()
        assert (hard == ref["hard"]).all(), mode
        assert (it == ref["iters"]).all(), mode
        assert (ok == ref["synd_ok"]).all(), mode
        out[mode] = st
        if mode == "on":
            assert st["compactions"] >= 2, st
            with pytest.raises(q.QldpcError) as e:
                dec.fetch_post()
            assert e.value.status == -7
            # the decoder is reusable: the same batch again (no reload), then a smaller one
            dec.run()
            assert (q.unpack_bits(dec.fetch_packed().cpu().numpy().view(np.uint32), N) == ref["hard"]).all()
            h2, it2, ok2 = run(q, torch, dec, **({"llr": llr[:333]} if form == "llr" else {"bits": y[:333], "mag": mag[:333]}))
            assert (h2 == ref["hard"][:333]).all() and (it2 == ref["iters"][:333]).all() and (ok2 == ref["synd_ok"][:333]).all()
    # compaction is what it is for: fewer lane-iterations for the same frames
    assert out["on"]["lane_iterations"] < out["off"]["lane_iterations"], out
    assert out["off"]["compactions"] == 0


def test_fetch_info_and_decode_siho_after_compaction(q, O, torch, peg):
    """decode_siho's output layout (one int per information bit at info_bits_pos) is gathered from every generation"""
    code, og = peg
    rng = np.random.default_rng(77)
    F, N = 500, 1008
    y, _ = frames(rng, F, N)
    llr = np.where(y == 1, -2.6, 2.6).astype(np.float32)
    pos = np.arange(504, 1008)
    ref = O.decode(og, llr, "OMS", 0.3, 25, "flooding", True, 1, n_threads=8)
    dec = q.Decoder(code, 504, 25, info_bits_pos=pos, rule="OMS", rule_param=0.3, n_frames=F, compact="on")
    V = dec.decode_siho(llr)
    assert dec.last_run_stats()["compactions"] >= 1
    assert (V == ref["hard"][:, pos]).all()


@pytest.mark.parametrize("dtype", ["f32", "i8"])
def test_syndrome_form_shortening_and_pinned_classes_move_with_their_frames(q, O, torch, dtype):
    """everything per-frame follows the frame: target syndrome, |LLR|, shortening length; the VN classes are shared"""
    rng = np.random.default_rng(5)
    code = q.Code.ira(2048, 1536, 0.2, 8, 3, 11)
    var, chk = code.edges()
    og = O.Graph.from_edges(code.N, code.M, var, chk)
    F, N, K, n_ite = 900, code.N, 1536, 30
    x = rng.integers(0, 2, (F, N)).astype(np.uint8)
    qber = rng.uniform(0.005, 0.04, F).astype(np.float32)
    y = x ^ (rng.random((F, N)) < qber[:, None])
    mag = np.array([q.bsc_llr(float(p)) for p in qber], np.float32)
    cls = np.zeros(N, np.uint8)
    cls[K + 20:] = q.VN_PINNED
    cls[K:K + 20] = q.VN_PUNCTURED
    y[:, K + 20:] = x[:, K + 20:]
    s = np.stack([og.syndrome(xx)[1] for xx in x])
    llr = np.where(y == 1, -mag[:, None], mag[:, None]).astype(np.float32)
    llr[:, K + 20:] = np.where(x[:, K + 20:] == 1, -np.float32(q.CONFIRMED_BIT_LLR), np.float32(q.CONFIRMED_BIT_LLR))
    llr[:, K:K + 20] = 0.0
    ref = O.decode(og, llr, "NMS", 0.8125, n_ite, "flooding", True, 1, n_threads=8, target=s, msg_i8=(dtype == "i8"))
    assert len(np.unique(ref["iters"])) >= 5
    dec = q.Decoder(code, N, n_ite, rule="NMS", rule_param=0.8125, n_frames=F, msg_dtype=dtype, compact="on")
    hard, it, ok = run(q, torch, dec, bits=y, mag=mag, cls=cls, synd=s)
    assert dec.last_run_stats()["compactions"] >= 1
    assert (hard == ref["hard"]).all() and (it == ref["iters"]).all() and (ok == ref["synd_ok"]).all()


def test_auto_mode_and_refusals(q, torch, peg):
    code, _ = peg
    # auto: small batches (< 4 groups) are left alone; frozen messages and fixed-iteration runs never compact
    llr = torch.full((100, 1008), 2.0, device="cuda")
    for kw in (dict(n_frames=100), dict(n_frames=600, freeze_messages=True, compact="on"), dict(n_frames=600, enable_syndrome=False, compact="on")):
        dec = q.Decoder(code, 1008, 8, rule="NMS", rule_param=0.75, **kw)
        dec.load_llr(llr)
        dec.run()
        assert dec.last_run_stats()["compactions"] == 0
        dec.fetch_post()


@pytest.mark.parametrize("depth", [2, 3])
def test_syndrome_depth_counters_move_with_their_frames(q, O, torch, peg, depth):
    """AFF3CT's syndrome_depth > 1 keeps a per-frame count of consecutive zero syndromes (cur_syndrome_depth); a frame that is dealt
    into another group in the middle of such a run must take its count along"""
    code, og = peg
    rng = np.random.default_rng(50 + depth)
    F, N = 700, 1008
    y, _ = frames(rng, F, N)
    llr = np.where(y == 1, -2.6, 2.6).astype(np.float32)
    ref = O.decode(og, llr, "NMS", 0.75, 30, "flooding", True, depth, n_threads=8)
    dec = q.Decoder(code, N, 30, rule="NMS", rule_param=0.75, n_frames=F, syndrome_depth=depth, compact="on")
    hard, it, ok = run(q, torch, dec, llr=llr)
    assert dec.last_run_stats()["compactions"] >= 2
    assert (hard == ref["hard"]).all() and (it == ref["iters"]).all() and (ok == ref["synd_ok"]).all()
