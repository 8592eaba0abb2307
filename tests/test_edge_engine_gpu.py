"""GPU suite: the low-batch edge-parallel engine (LDS-staged check groups + wavefront-shuffle fold) vs the oracle.

Min-sum family: bit-exact (hard decisions, iteration counts, success flags, posteriors) -- the butterfly
all-reduce of (min1, min2, sign) is order-independent.  SPA reduces its product as a tree: tolerance class.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bsc(rng, F, N, p, mag):
    return np.where(rng.random((F, N)) < p, -mag, mag).astype(np.float32)


def run(q, torch, dec, llr, post=True):
    dec.load_llr(torch.from_numpy(llr).cuda())
    dec.run()
    hard = q.unpack_bits(dec.fetch_packed().cpu().numpy().view(np.uint32), dec.N)
    it, ok = dec.fetch_status()
    return hard, it.cpu().numpy(), ok.cpu().numpy(), (dec.fetch_post().cpu().numpy() if post else None)


@pytest.fixture(scope="module")
def torch():
    import torch
    return torch


@pytest.fixture(scope="module")
def peg(q, O, gold):
    p = os.path.join(gold, "PEGReg504x1008.alist")
    return q.Code.from_alist(p), O.Graph.from_alist(p)


@pytest.mark.parametrize("rule,param", [("MS", 0.0), ("OMS", 0.35), ("NMS", 0.75)])
@pytest.mark.parametrize("F", [1, 5, 16])
def test_edge_engine_bit_exact_peg(q, O, torch, peg, rule, param, F):
    code, og = peg
    llr = bsc(np.random.default_rng(100 + F), F, 1008, 0.055, 2.7)
    ref = O.decode(og, llr, rule, param, 25, n_threads=4)
    dec = q.Decoder(code, 1008, 25, rule=rule, rule_param=param, n_frames=F, engine="edges")
    hard, it, ok, post = run(q, torch, dec, llr)
    assert (hard == ref["hard"]).all() and (it == ref["iters"]).all() and (ok == ref["synd_ok"]).all()
    assert (post.view(np.uint32) == ref["post"].view(np.uint32)).all()


@pytest.mark.parametrize("synd,depth", [(False, 1), (True, 2), (True, 3)])
def test_edge_engine_fixed_and_depth(q, O, torch, peg, synd, depth):
    code, og = peg
    llr = bsc(np.random.default_rng(7), 8, 1008, 0.05, 2.9)
    ref = O.decode(og, llr, "NMS", 0.75, 14, "flooding", synd, depth, n_threads=4)
    dec = q.Decoder(code, 1008, 14, rule="NMS", rule_param=0.75, n_frames=8, enable_syndrome=synd, syndrome_depth=depth, engine="edges")
    hard, it, ok, post = run(q, torch, dec, llr)
    assert (hard == ref["hard"]).all() and (it == ref["iters"]).all() and (ok == ref["synd_ok"]).all()
    assert (post.view(np.uint32) == ref["post"].view(np.uint32)).all()


def test_edge_engine_is_the_default_for_one_block_and_passes_the_kat(q, peg, gold):
    """auto engine with n_frames = 1 (the daemon's case) is the edge engine; SPA through it still gives the KAT word."""
    code, _ = peg
    kat = json.load(open(os.path.join(gold, "kat_peg504x1008.json")))
    for engine in ("auto", "edges", "frames"):
        dec = q.Decoder(code, 504, 10, info_bits_pos=np.arange(504, 1008), rule="SPA", n_frames=1, engine=engine)
        assert (dec.decode_siho(np.array(kat["llrs"], np.float32))[0] == np.array(kat["decoded"])).all(), engine


def test_edge_engine_full_size_block(q, O, torch):
    """one 65 536-VN block (config 2 code), QBER 2 %, the way the daemon decodes it"""
    code = q.Code.ira(65536, 52429, 0.125, 11, 3, 7)
    enc = q.Encoder(code, "IRA")
    rng = np.random.default_rng(9)
    F = 3
    cw = enc.encode(rng.integers(0, 2, (F, enc.K)))
    mag = np.float32(q.bsc_llr(0.02))
    noisy = cw.copy()
    noisy[:, :enc.K] ^= rng.random((F, enc.K)) < 0.02
    llr = np.where(noisy == 1, -mag, mag).astype(np.float32)
    llr[:, enc.K:] = np.where(cw[:, enc.K:] == 1, -np.float32(q.CONFIRMED_BIT_LLR), np.float32(q.CONFIRMED_BIT_LLR))
    var, chk = code.edges()
    ref = O.decode(O.Graph.from_edges(code.N, code.M, var, chk), llr, "NMS", 0.75, 50, n_threads=4)
    dec = q.Decoder(code, enc.K, 50, rule="NMS", rule_param=0.75, n_frames=F, engine="edges")
    hard, it, ok, post = run(q, torch, dec, llr)
    assert (hard == ref["hard"]).all() and (it == ref["iters"]).all() and (ok == 1).all() and (hard == cw).all()
    assert (post.view(np.uint32) == ref["post"].view(np.uint32)).all()
    assert dec.last_run_iterations <= int(it.max()) + 9            # host stopped launching soon after convergence


@pytest.mark.parametrize("name,kind", [("NR_2_3_112.qc", "qc"), ("1998.5.3.2665.alist", "alist"), ("20.alist", "alist")])
def test_edge_engine_other_matrices(q, O, torch, gold, name, kind):
    p = os.path.join(gold, name)
    code = q.Code.from_qc(p) if kind == "qc" else q.Code.from_alist(p)
    og = O.Graph.from_qc(p) if kind == "qc" else O.Graph.from_alist(p)
    llr = bsc(np.random.default_rng(3), 6, code.N, 0.03, 3.4)
    ref = O.decode(og, llr, "OMS", 0.3, 15, n_threads=4)
    dec = q.Decoder(code, code.N, 15, rule="OMS", rule_param=0.3, n_frames=6, engine="edges")
    hard, it, ok, post = run(q, torch, dec, llr)
    assert (hard == ref["hard"]).all() and (it == ref["iters"]).all() and (ok == ref["synd_ok"]).all()
    assert (post.view(np.uint32) == ref["post"].view(np.uint32)).all()


def test_edge_engine_spa_tolerance(q, O, torch, peg):
    code, og = peg
    F = 16
    llr = bsc(np.random.default_rng(5), F, 1008, 0.06, 2.75)
    ref = O.decode(og, llr, "SPA", 0.0, 20, n_threads=4)
    dec = q.Decoder(code, 1008, 20, rule="SPA", n_frames=F, engine="edges")
    hard, it, ok, _ = run(q, torch, dec, llr, post=False)
    conv = ref["synd_ok"] == 1
    assert conv.sum() >= F // 2
    assert ((hard == ref["hard"]).all(axis=1))[conv].mean() >= 0.9        # tolerance: tree-ordered product
    assert np.abs(it[conv] - ref["iters"][conv]).max() <= 1


def test_edge_engine_bits_path_and_info_fetch(q, O, torch, peg):
    code, og = peg
    rng = np.random.default_rng(12)
    enc = q.Encoder(code, "IDENTITY")
    cw = enc.encode(rng.integers(0, 2, (4, enc.K)))
    qber = np.array([0.02, 0.03, 0.04, 0.05], np.float32)
    noisy = cw ^ (rng.random((4, 1008)) < qber[:, None])
    mag = np.array([q.bsc_llr(p) for p in qber], np.float32)
    llr = np.where(noisy == 1, -mag[:, None], mag[:, None]).astype(np.float32)
    ref = O.decode(og, llr, "NMS", 0.75, 30, n_threads=4)
    dec = q.Decoder(code, enc.K, 30, info_bits_pos=enc.info_bits_pos, rule="NMS", rule_param=0.75, n_frames=4, engine="edges")
    dec.load_bits(torch.from_numpy(q.pack_bits(noisy).astype(np.int64).astype(np.uint32).view(np.int32)).cuda(), torch.from_numpy(mag).cuda())
    dec.run()
    hard = q.unpack_bits(dec.fetch_packed().cpu().numpy().view(np.uint32), 1008)
    assert (hard == ref["hard"]).all()
    assert (dec.fetch_info().cpu().numpy() == ref["hard"][:, enc.info_bits_pos]).all()


def test_edge_engine_unsupported_requests_say_so(q, peg):
    code, _ = peg
    for kw in (dict(rule="LSPA"), dict(schedule="hlayered", rule="MS")):
        with pytest.raises(q.QldpcError) as e:
            q.Decoder(code, 1008, 5, n_frames=1, engine="edges", **kw)
        assert e.value.status == -7
    big = q.Code.ira(4096, 3850, 0.4, 14, 4, 3)                           # check degree > 64? no: > 40 only
    if big.max_cn_degree > 64:
        with pytest.raises(q.QldpcError):
            q.Decoder(big, big.N, 5, rule="MS", n_frames=1, engine="edges")


@pytest.mark.parametrize("wgs", ["", "100000"])
def test_one_launch_decode_one_xcd_per_block(q, O, torch, peg, monkeypatch, wgs):
    """QLDPC_EDGE_PERSIST=1 (opt-in, csrc/qldpc_kernels_edge.h qe_xcd): the whole decode in one launch, every block on the XCD that
    claimed it, grid barriers on that XCD's L2, early exit decided on the device -- the same bits, iteration counts and flags as the
    launch-per-pass path and the oracle.  With more workgroups per block asked for than an XCD can hold the barriers time out (bounded
    waits: the grid drains), the decoder says so and repeats the decode with a launch per pass: still the same answers."""
    code, og = peg
    monkeypatch.setenv("QLDPC_EDGE_PERSIST", "1")
    if wgs:
        monkeypatch.setenv("QLDPC_EDGE_WGS", wgs)
    for F, rule, param in ((1, "NMS", 0.75), (5, "OMS", 0.35), (8, "MS", 0.0)):
        llr = bsc(np.random.default_rng(200 + F), F, 1008, 0.055, 2.7)
        ref = O.decode(og, llr, rule, param, 25, n_threads=4)
        dec = q.Decoder(code, 1008, 25, rule=rule, rule_param=param, n_frames=F, engine="edges")
        for _ in range(2):
            hard, it, ok, post = run(q, torch, dec, llr)
            assert (hard == ref["hard"]).all() and (it == ref["iters"]).all() and (ok == ref["synd_ok"]).all()
            assert (post.view(np.uint32) == ref["post"].view(np.uint32)).all()
