"""GPU parity suite (-m gpu): the HIP path, called through the C ABI, against the CPU oracle.

Bar: MS/OMS/NMS (and the AMS<min> member) fp32 -- identical hard decisions,
iteration counts, success flags AND posteriors (bit-exact floats) on every frame.  SPA / LSPA / min*:
identical on the reference's known-answer vector; on random frames >= 99 % identical frame verdicts
(device tanh/atanh/log/exp differ from glibc in the last ulp).
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EXACT = [("MS", 0.0), ("OMS", 0.35), ("NMS", 0.75), ("AMS_MIN", 0.0)]
SOFT = [("SPA", 0.0), ("LSPA", 0.0), ("AMS_MINSTAR", 0.0), ("AMS_MINSTAR_L2", 0.0)]


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


@pytest.fixture(scope="module")
def kat(gold):
    return json.load(open(os.path.join(gold, "kat_peg504x1008.json")))


@pytest.fixture(scope="module")
def peg(q, O, gold):
    p = os.path.join(gold, "PEGReg504x1008.alist")
    return q.Code.from_alist(p), O.Graph.from_alist(p)


def bsc_frames(rng, F, N, p, mag):
    return np.where(rng.random((F, N)) < p, -mag, mag).astype(np.float32)


def staged(q, torch, dec, llr, want_post=True):
    t = torch.from_numpy(llr).cuda()
    dec.load_llr(t)
    dec.run()
    hard = q.unpack_bits(dec.fetch_packed().cpu().numpy().view(np.uint32), dec.N)
    it, ok = dec.fetch_status()
    post = dec.fetch_post().cpu().numpy() if want_post else None
    return hard, it.cpu().numpy(), ok.cpu().numpy(), post


def test_kat_through_the_c_abi(q, peg, kat):
    code, _ = peg
    dec = q.Decoder(code, 504, 10, info_bits_pos=np.arange(504, 1008), rule="SPA", n_frames=1)
    V = dec.decode_siho(np.array(kat["llrs"], np.float32))
    assert (V[0] == np.array(kat["decoded"])).all()
    dec.reset()
    V2 = dec.decode_siho(np.array(kat["llrs"], np.float32))     # reset() -> same answer again (BS/src/main.cpp:389)
    assert (V2 == V).all()


@pytest.mark.parametrize("sched", ["flooding", "hlayered"])
@pytest.mark.parametrize("rule,param", EXACT)
@pytest.mark.parametrize("V", [1, 2, 4])
def test_minsum_family_bit_exact(q, O, torch, peg, rule, param, sched, V):
    code, og = peg
    F = 300                                                     # ragged: not a multiple of 64*V
    llr = bsc_frames(np.random.default_rng(10 + V), F, 1008, 0.065 if rule in ("NMS", "OMS") else 0.045, 2.6)
    if sched == "hlayered":
        order, _, _ = code.layer_order()
        var, chk = og.edges()
        inv = np.empty(code.M, np.int32); inv[order] = np.arange(code.M, dtype=np.int32)
        og = O.Graph.from_edges(code.N, code.M, *_reorder(var, chk, inv))
    ref = O.decode(og, llr, rule, param, 25, sched, True, 1, n_threads=8)
    dec = q.Decoder(code, 1008, 25, rule=rule, rule_param=param, n_frames=F, schedule=sched, frames_per_lane=V, freeze_messages=True)
    hard, it, ok, post = staged(q, torch, dec, llr)
    assert (hard == ref["hard"]).all()
    assert (it == ref["iters"]).all()
    assert (ok == ref["synd_ok"]).all()
    assert (post.view(np.uint32) == ref["post"].view(np.uint32)).all()     # bit-exact floats
    if rule in ("NMS", "OMS"):
        assert 0 < (ref["synd_ok"] == 0).sum() < F                          # both outcomes are exercised


def _reorder(var, chk, inv):
    """edge list of the row-permuted H: check c becomes inv[c]; keeps each check's edge order"""
    newc = inv[chk]
    idx = np.argsort(newc, kind="stable")
    return var[idx], newc[idx]


@pytest.mark.parametrize("rule,param", EXACT[:3])
def test_fixed_iterations_no_syndrome(q, O, torch, peg, rule, param):
    code, og = peg
    llr = bsc_frames(np.random.default_rng(5), 130, 1008, 0.06, 2.75)
    ref = O.decode(og, llr, rule, param, 12, "flooding", False, 1, n_threads=8)
    dec = q.Decoder(code, 1008, 12, rule=rule, rule_param=param, n_frames=130, enable_syndrome=False)
    hard, it, ok, post = staged(q, torch, dec, llr)
    assert (hard == ref["hard"]).all() and (it == 12).all() and (ok == ref["synd_ok"]).all()
    assert (post.view(np.uint32) == ref["post"].view(np.uint32)).all()


@pytest.mark.parametrize("depth", [2, 3])
def test_syndrome_depth(q, O, torch, peg, depth):
    code, og = peg
    llr = bsc_frames(np.random.default_rng(6), 64, 1008, 0.05, 2.9)
    ref = O.decode(og, llr, "NMS", 0.75, 30, "flooding", True, depth, n_threads=8)
    dec = q.Decoder(code, 1008, 30, rule="NMS", rule_param=0.75, n_frames=64, syndrome_depth=depth)
    hard, it, ok, _ = staged(q, torch, dec, llr, want_post=False)
    assert (hard == ref["hard"]).all() and (it == ref["iters"]).all()


@pytest.mark.parametrize("sched", ["flooding", "hlayered"])
@pytest.mark.parametrize("rule,param", SOFT)
def test_transcendental_rules_tolerance(q, O, torch, peg, rule, param, sched):
    code, og = peg
    if sched == "hlayered":
        order, _, _ = code.layer_order()
        var, chk = og.edges()
        inv = np.empty(code.M, np.int32); inv[order] = np.arange(code.M, dtype=np.int32)
        og = O.Graph.from_edges(code.N, code.M, *_reorder(var, chk, inv))
    F = 256
    llr = bsc_frames(np.random.default_rng(7), F, 1008, 0.07, 2.59)
    ref = O.decode(og, llr, rule, param, 20, sched, True, 1, n_threads=8)
    dec = q.Decoder(code, 1008, 20, rule=rule, rule_param=param, n_frames=F, schedule=sched, freeze_messages=True)
    hard, it, ok, post = staged(q, torch, dec, llr)
    same = (hard == ref["hard"]).all(axis=1)
    conv_ref = ref["synd_ok"] == 1
    # tolerance (stated): of the frames the oracle converges on, >= 99 % get the identical word; frames that
    # never converge are chaotic under last-ulp differences of the device tanh/atanh/log/exp, so for them only
    # the FER is compared: |FER_gpu - FER_oracle| within 3 binomial sigma.
    assert conv_ref.sum() > F // 2
    assert same[conv_ref].mean() >= 0.99, same[conv_ref].mean()
    fer_ref, fer_gpu = 1.0 - conv_ref.mean(), 1.0 - (ok == 1).mean()
    sigma = max(np.sqrt(fer_ref * (1 - fer_ref) / F), 1.0 / F)
    assert abs(fer_gpu - fer_ref) <= 3 * sigma
    conv = conv_ref & (ok == 1) & same
    assert np.abs(it[conv] - ref["iters"][conv]).max() <= 1
    good = conv & (it == ref["iters"])
    rel = np.abs(post[good] - ref["post"][good]) / (1.0 + np.abs(ref["post"][good]))
    # tolerance on posteriors of identically-converged frames: SPA median 1e-5, 99.9 % within 5e-2, max 0.25; LSPA (log domain) median 1e-4, max 0.25;
    # 5e-2; the min* rules (parity unpinned, recollected AFF3CT semantics) amplify last-ulp exp/log differences
    # over the iterations (a few frames take another trajectory to the same word): median within 1e-4 and
    # >= 75 % of the frames within 2e-3 everywhere (layered min* is the most sensitive: ~88 % measured)
    if rule == "SPA":
        # hardware exp/log/rcp (qk_tanh_half / qk_2atanh): messages near the 1 - eps clamp are the sensitive ones
        assert np.median(rel) < 1e-5 and np.quantile(rel, 0.999) < 5e-2 and rel.max() < 0.25
    elif rule == "LSPA":
        assert np.median(rel) < 1e-4 and rel.max() < 0.25
    else:
        per_frame = rel.max(axis=1)
        assert np.median(rel) < 1e-4 and (per_frame < 2e-3).mean() >= 0.75, (per_frame < 2e-3).mean()


@pytest.mark.parametrize("F", [1, 63, 64, 65, 129])
def test_ragged_batches(q, O, torch, peg, F):
    code, og = peg
    llr = bsc_frames(np.random.default_rng(F), F, 1008, 0.06, 2.75)
    ref = O.decode(og, llr, "NMS", 0.8, 20, n_threads=8)
    dec = q.Decoder(code, 1008, 20, rule="NMS", rule_param=0.8, n_frames=max(F, 70))     # capacity > load
    hard, it, ok, _ = staged(q, torch, dec, llr, want_post=False)
    assert hard.shape == (F, 1008) and (hard == ref["hard"]).all() and (it == ref["iters"]).all()


def test_load_more_than_capacity_is_an_error(q, torch, peg):
    code, _ = peg
    dec = q.Decoder(code, 1008, 5, rule="MS", n_frames=8)
    with pytest.raises(q.QldpcError) as e:
        dec.load_llr(torch.zeros((9, 1008), device="cuda"))
    assert e.value.status == -6
    with pytest.raises(q.QldpcError) as e:
        dec.run()
    assert e.value.status == -8


def test_qkd_frame_formation_from_packed_bits(q, O, torch, peg, kat):
    """BS/src/main.cpp:348-362: channel bits -> +-ln((1-p)/p), parity VNs pinned +-23.03, punctured -> 0."""
    code, og = peg
    rng = np.random.default_rng(11)
    F, N = 70, 1008
    enc = q.Encoder(code, "IDENTITY")
    cw = enc.encode(rng.integers(0, 2, (F, enc.K)))
    qber = rng.uniform(0.02, 0.05, F).astype(np.float32)
    noisy = cw ^ (rng.random((F, N)) < qber[:, None])
    cls = np.zeros(N, np.uint8)
    par = np.setdiff1d(np.arange(N), enc.info_bits_pos)
    cls[par] = q.VN_PINNED
    cls[par[:40]] = q.VN_PUNCTURED
    noisy[:, par] = cw[:, par]                                   # Alice discloses the parity bits
    mag = np.array([q.bsc_llr(p) for p in qber], np.float32)
    llr = np.where(noisy == 1, -mag[:, None], mag[:, None]).astype(np.float32)
    llr[:, par] = np.where(cw[:, par] == 1, -np.float32(q.CONFIRMED_BIT_LLR), np.float32(q.CONFIRMED_BIT_LLR))
    llr[:, par[:40]] = 0.0
    ref = O.decode(og, llr, "NMS", 0.75, 30, n_threads=8)
    dec = q.Decoder(code, enc.K, 30, info_bits_pos=enc.info_bits_pos, rule="NMS", rule_param=0.75, n_frames=F, freeze_messages=True)
    bits = torch.from_numpy(q.pack_bits(noisy).astype(np.int64).astype(np.uint32).view(np.int32)).cuda()
    dec.load_bits(bits, torch.from_numpy(mag).cuda(), torch.from_numpy(cls).cuda())
    dec.run()
    hard = q.unpack_bits(dec.fetch_packed().cpu().numpy().view(np.uint32), N)
    assert (hard == ref["hard"]).all()
    info = dec.fetch_info().cpu().numpy()
    assert (info == ref["hard"][:, enc.info_bits_pos]).all()
    assert (dec.fetch_post().cpu().numpy().view(np.uint32) == ref["post"].view(np.uint32)).all()
    okf = ref["synd_ok"] == 1
    assert okf.mean() > 0.8 and (hard[okf] == cw[okf]).all()    # decoded word == Alice's codeword


def test_encoder_identity_matches_kat(q, peg, kat):
    """Encoder_LDPC_from_H(..., "IDENTITY", ...) pin: VAR/main.cpp (alist-v1.0.1):445,447."""
    code, _ = peg
    enc = q.Encoder(code, "IDENTITY")
    assert enc.K == 504 and (enc.info_bits_pos == np.arange(504, 1008)).all()
    X = enc.encode(np.array(kat["data"]))
    assert (X[0] == np.array(kat["encoded"])).all()


def test_encoder_lu_dec_and_qc_methods(q, peg, kat, gold):
    """The harness' other encoder constructions (VAR/main.cpp (alist-v1.0.1):135-145 G_method "LU_DEC", (qc):145 Encoder_LDPC_from_QC):
    systematic codewords with H x = 0 and the information bits where get_info_bits_pos says.  No reference vector exists for either
    (parity unpinned: AFF3CT's LU_DEC column permutation may differ); the contract the harness relies on is what is checked."""
    code, _ = peg
    rng = np.random.default_rng(8)
    lu = q.Encoder(code, "LU_DEC")
    assert lu.K == 504 and len(set(lu.info_bits_pos.tolist())) == 504 and (lu.info_bits_pos != q.Encoder(code, "IDENTITY").info_bits_pos).any()
    u = rng.integers(0, 2, (9, 504))
    X = lu.encode(u)
    assert (X[:, lu.info_bits_pos] == u).all() and all(code.syndrome(x)[0] == 0 for x in X)
    assert np.mean(lu.info_bits_pos < 504) > 0.9                                  # parity at the end wherever H allows
    # the same codeword space as IDENTITY: re-encoding IDENTITY's codeword from its LU_DEC information positions returns it
    Xi = q.Encoder(code, "IDENTITY").encode(np.array(kat["data"]))
    assert (lu.encode(Xi[:, lu.info_bits_pos]) == Xi).all()
    for name in ("NR_1_0_2.qc", "NR_2_6_52_rm_half.qc", "test.qc"):            # 5G base graphs of the reference: H2 (last M columns) invertible or refused
        c = q.Code.from_qc(os.path.join(gold, name))
        try:
            e = q.Encoder(c, "QC")
        except q.QldpcError as ex:
            assert "not invertible" in str(ex)
            continue
        assert e.K == c.N - c.M and (e.info_bits_pos == np.arange(e.K)).all()
        u = rng.integers(0, 2, (5, e.K))
        X = e.encode(u)
        assert (X[:, :e.K] == u).all() and all(c.syndrome(x)[0] == 0 for x in X)
    with pytest.raises(q.QldpcError):
        q.Encoder(code, "CHOLESKY")


def test_encoder_ira_and_roundtrip_full_size(q, O, torch):
    """config 2 code at full size: encode -> BSC 2 % -> decode == codeword; 64 frames vs the oracle."""
    code = q.Code.ira(65536, 52429, 0.125, 11, 3, 7)
    enc = q.Encoder(code, "IRA")
    rng = np.random.default_rng(21)
    F = 64
    cw = enc.encode(rng.integers(0, 2, (F, enc.K)))
    for f in range(0, F, 16):
        assert code.syndrome(cw[f])[0] == 0
    mag = np.float32(q.bsc_llr(0.02))
    noisy = cw.copy()
    noisy[:, :enc.K] ^= rng.random((F, enc.K)) < 0.02
    llr = np.where(noisy == 1, -mag, mag).astype(np.float32)
    llr[:, enc.K:] = np.where(cw[:, enc.K:] == 1, -np.float32(q.CONFIRMED_BIT_LLR), np.float32(q.CONFIRMED_BIT_LLR))
    var, chk = code.edges()
    og = O.Graph.from_edges(code.N, code.M, var, chk)
    ref = O.decode(og, llr, "NMS", 0.75, 50, n_threads=8)
    dec = q.Decoder(code, enc.K, 50, rule="NMS", rule_param=0.75, n_frames=F)
    hard, it, ok, _ = staged(q, torch, dec, llr, want_post=False)
    assert (hard == ref["hard"]).all() and (it == ref["iters"]).all() and (ok == ref["synd_ok"]).all()
    assert (ok == 1).all() and (hard == cw).all()
    assert 5 <= it.mean() <= 20


@pytest.mark.parametrize("name,kind", [("NR_2_3_112.qc", "qc"), ("20.alist", "alist"), ("1998.5.3.2665.alist", "alist")])
def test_other_matrices(q, O, torch, gold, name, kind):
    p = os.path.join(gold, name)
    code = q.Code.from_qc(p) if kind == "qc" else q.Code.from_alist(p)
    og = O.Graph.from_qc(p) if kind == "qc" else O.Graph.from_alist(p)
    llr = bsc_frames(np.random.default_rng(3), 96, code.N, 0.03, 3.4)
    for sched in ("flooding", "hlayered"):
        g2 = og
        if sched == "hlayered":
            order, _, _ = code.layer_order()
            var, chk = og.edges()
            inv = np.empty(code.M, np.int32); inv[order] = np.arange(code.M, dtype=np.int32)
            g2 = O.Graph.from_edges(code.N, code.M, *_reorder(var, chk, inv))
        ref = O.decode(g2, llr, "OMS", 0.3, 15, sched, n_threads=8)
        dec = q.Decoder(code, code.N, 15, rule="OMS", rule_param=0.3, n_frames=96, schedule=sched, freeze_messages=True)
        hard, it, ok, post = staged(q, torch, dec, llr)
        assert (hard == ref["hard"]).all() and (it == ref["iters"]).all()
        assert (post.view(np.uint32) == ref["post"].view(np.uint32)).all()


def test_natural_layer_order_equals_plain_sequential_sweep(q, O, torch, gold):
    """For a QC code the level schedule must equal AFF3CT's c = 0..M-1 sweep on the UNpermuted H."""
    p = os.path.join(gold, "NR_2_3_112.qc")
    code, og = q.Code.from_qc(p), O.Graph.from_qc(p)
    assert code.layer_order()[2]
    llr = bsc_frames(np.random.default_rng(4), 64, code.N, 0.04, 3.1)
    ref = O.decode(og, llr, "NMS", 0.8, 10, "hlayered", n_threads=8)
    dec = q.Decoder(code, code.N, 10, rule="NMS", rule_param=0.8, n_frames=64, schedule="hlayered", freeze_messages=True)
    hard, it, _, post = staged(q, torch, dec, llr)
    assert (hard == ref["hard"]).all() and (it == ref["iters"]).all()
    assert (post.view(np.uint32) == ref["post"].view(np.uint32)).all()


def test_high_degree_checks_use_the_generic_kernel(q, O, torch):
    code = q.Code.ira(4096, 3850, 0.4, 14, 4, 3)               # rate 0.94 -> check degree > 40
    assert code.max_cn_degree > 40
    var, chk = code.edges()
    og = O.Graph.from_edges(code.N, code.M, var, chk)
    llr = bsc_frames(np.random.default_rng(8), 64, code.N, 0.004, 5.5)
    ref = O.decode(og, llr, "NMS", 0.75, 12, n_threads=8)
    dec = q.Decoder(code, code.N, 12, rule="NMS", rule_param=0.75, n_frames=64, freeze_messages=True)
    hard, it, _, post = staged(q, torch, dec, llr)
    assert (hard == ref["hard"]).all() and (it == ref["iters"]).all()
    assert (post.view(np.uint32) == ref["post"].view(np.uint32)).all()


def test_zero_llr_and_all_converged_inputs(q, O, torch, peg):
    code, og = peg
    llr = np.zeros((3, 1008), np.float32)                       # everything erased: stays at 0, never "fails" to parse
    llr[1] = 4.0                                                # trivially converged frame
    llr[2, ::2] = -4.0
    ref = O.decode(og, llr, "MS", 0.0, 8, n_threads=1)
    dec = q.Decoder(code, 1008, 8, rule="MS", n_frames=3)
    hard, it, ok, post = staged(q, torch, dec, llr)
    assert (hard == ref["hard"]).all() and (it == ref["iters"]).all() and (ok == ref["synd_ok"]).all()
    assert (post.view(np.uint32) == ref["post"].view(np.uint32)).all()


def test_profile_hooks_report_algorithmic_bytes(q, torch, peg):
    code, _ = peg
    dec = q.Decoder(code, 1008, 6, rule="NMS", rule_param=0.75, n_frames=128, enable_syndrome=False)
    dec.profile(True)
    dec.load_llr(torch.from_numpy(bsc_frames(np.random.default_rng(2), 128, 1008, 0.05, 2.9)).cuda())
    dec.run()
    st = {s["name"]: s for s in dec.profile_read()}
    assert st["cn_update"]["launches"] == 6 and st["vn_update"]["launches"] == 7       # FIRST + 5 + POST
    assert st["cn_update"]["alg_bytes"] == 6 * 2 * 3024 * 4 * 128
    assert st["cn_update"]["total_ms"] > 0


@pytest.mark.parametrize("sched", ["flooding", "hlayered"])
@pytest.mark.parametrize("V", [1, 4])
def test_default_mode_freezes_decisions_not_messages(q, O, torch, peg, sched, V):
    """freeze_messages = 0 (default, full-row stores): a converged frame's hard decision, iteration count and success
    flag are still exactly the reference's; only its posterior keeps evolving with the rest of its 64-frame group."""
    code, og = peg
    F = 300
    llr = bsc_frames(np.random.default_rng(77), F, 1008, 0.062, 2.65)
    if sched == "hlayered":
        order, _, _ = code.layer_order()
        var, chk = og.edges()
        inv = np.empty(code.M, np.int32); inv[order] = np.arange(code.M, dtype=np.int32)
        og = O.Graph.from_edges(code.N, code.M, *_reorder(var, chk, inv))
    ref = O.decode(og, llr, "NMS", 0.75, 25, sched, True, 1, n_threads=8)
    # compact="off": posteriors are read back below, which a run that compacted its active frames refuses (tests/test_compaction_gpu.py)
    dec = q.Decoder(code, 1008, 25, rule="NMS", rule_param=0.75, n_frames=F, schedule=sched, frames_per_lane=V, engine="frames", compact="off")
    hard, it, ok, post = staged(q, torch, dec, llr)
    assert (hard == ref["hard"]).all() and (it == ref["iters"]).all() and (ok == ref["synd_ok"]).all()
    ran_all = it == dec.last_run_iterations            # frames that were never frozen keep exact posteriors
    assert ran_all.any() and (post[ran_all].view(np.uint32) == ref["post"][ran_all].view(np.uint32)).all()
    assert 0 < (ref["synd_ok"] == 0).sum() < F


@pytest.mark.parametrize("rule,param", [("NMS", 0.75), ("NMS", 0.7), ("OMS", 0.35), ("MS", 0.0)])
@pytest.mark.parametrize("synd", [False, True])
def test_fp16_packed_check_node_kernel_is_bit_identical(q, O, torch, gold, rule, param, synd):
    """Default fp16 configuration (two frames per lane, no message freezing) runs the packed binary16 fold of
    qldpc_kernels_h16.h; it must give what the fp32-widening kernel gives, i.e. the rounding oracle -- posteriors included when
    every frame runs all iterations.  Irregular code, so several degree buckets (and the generic kernel for dc > 40) mix."""
    code = q.Code.ira(4096, 3400, 0.3, 9, 3, 5)
    var, chk = code.edges()
    og = O.Graph.from_edges(code.N, code.M, var, chk)
    F = 300
    llr = bsc_frames(np.random.default_rng(77), F, code.N, 0.018, 4.0)
    ref = O.decode(og, llr, rule, param, 14, "flooding", synd, 1, n_threads=8, msg_fp16=True)
    dec = q.Decoder(code, code.N, 14, rule=rule, rule_param=param, n_frames=F, enable_syndrome=synd, msg_dtype="f16")
    hard, it, ok, post = staged(q, torch, dec, llr, want_post=not synd)
    assert (hard == ref["hard"]).all() and (it == ref["iters"]).all() and (ok == ref["synd_ok"]).all()
    if not synd:
        assert (post.view(np.uint32) == ref["post"].view(np.uint32)).all()


@pytest.mark.parametrize("V", [1, 2, 4])
@pytest.mark.parametrize("rule,param", [("NMS", 0.75), ("OMS", 0.35), ("MS", 0.0)])
def test_fp16_message_storage_is_bit_exact_against_the_rounding_oracle(q, O, torch, peg, rule, param, V):
    """msg_dtype = f16: messages rounded to binary16 (RNE) when stored, fp32 arithmetic.  Against the oracle run with the
    same rounding: identical hard decisions, iteration counts, success flags and posteriors."""
    code, og = peg
    F = 200
    llr = bsc_frames(np.random.default_rng(31 + V), F, 1008, 0.06 if rule != "MS" else 0.045, 2.7)
    ref = O.decode(og, llr, rule, param, 25, "flooding", True, 1, n_threads=8, msg_fp16=True)
    dec = q.Decoder(code, 1008, 25, rule=rule, rule_param=param, n_frames=F, frames_per_lane=V, msg_dtype="f16", freeze_messages=True)
    hard, it, ok, post = staged(q, torch, dec, llr)
    assert (hard == ref["hard"]).all() and (it == ref["iters"]).all() and (ok == ref["synd_ok"]).all()
    assert (post.view(np.uint32) == ref["post"].view(np.uint32)).all()
    ref32 = O.decode(og, llr, rule, param, 25, "flooding", True, 1, n_threads=8)
    if rule != "MS":
        # FER tolerance vs fp32: 5 points on 200 frames.  (Plain MS is excluded: on a BSC all |LLR| are equal and its ties
        # stall the fp32 decoder -- FER 0.77 here -- while the binary16 rounding noise breaks them: FER 0.19.)
        assert abs((ref["synd_ok"] == 0).mean() - (ref32["synd_ok"] == 0).mean()) < 0.05


def test_fp16_message_storage_full_size_fer_and_requests(q, O, torch):
    code = q.Code.ira(65536, 52429, 0.125, 11, 3, 7)
    enc = q.Encoder(code, "IRA")
    rng = np.random.default_rng(8)
    F = 128
    cw = enc.encode(rng.integers(0, 2, (F, enc.K)))
    mag = np.float32(q.bsc_llr(0.02))
    y = cw.copy()
    y[:, :enc.K] ^= rng.random((F, enc.K)) < 0.02
    llr = np.where(y == 1, -mag, mag).astype(np.float32)
    llr[:, enc.K:] = np.where(cw[:, enc.K:] == 1, -np.float32(q.CONFIRMED_BIT_LLR), np.float32(q.CONFIRMED_BIT_LLR))
    var, chk = code.edges()
    og = O.Graph.from_edges(code.N, code.M, var, chk)
    ref = O.decode(og, llr[:32], "NMS", 0.75, 50, n_threads=8, msg_fp16=True)
    dec = q.Decoder(code, enc.K, 50, rule="NMS", rule_param=0.75, n_frames=F, msg_dtype="f16")
    hard, it, ok, _ = staged(q, torch, dec, llr, want_post=False)
    assert (hard[:32] == ref["hard"]).all() and (it[:32] == ref["iters"]).all()
    assert (ok == 1).all() and (hard == cw).all()                                        # FER 0/128 at QBER 2 %, as fp32
    # half the message bytes; (the fp32 LLR array is allocated on first use: `dec` has one, the fresh fp32 decoder does not)
    assert dec.device_bytes < 0.70 * q.Decoder(code, enc.K, 50, rule="NMS", rule_param=0.75, n_frames=F, frames_per_lane=2).device_bytes
    for kw in (dict(schedule="hlayered"), dict(engine="edges")):
        with pytest.raises(q.QldpcError) as e:
            q.Decoder(code, enc.K, 5, rule="NMS", n_frames=4, msg_dtype="f16", **kw)
        assert e.value.status == -7


def test_generated_qc_peg_code_decodes_bit_exact(q, O, torch):
    """a code from qldpc_code_qc_peg (the reference's psd-peg.py as a C generator): weight-1 parity columns, regular dv = 3
    information part; flooding and layered NMS against the oracle, and the frames are actually corrected."""
    code = q.Code.qc_peg(12, 6, 3, 211, seed=5)
    var, chk = code.edges()
    og = O.Graph.from_edges(code.N, code.M, var, chk)
    F = 192
    llr = bsc_frames(np.random.default_rng(12), F, code.N, 0.02, 3.9)
    llr[:, 12 * 211:] = 23.02585                                 # disclosed parity of the all-zero word
    for sched in ("flooding", "hlayered"):
        g2 = og
        if sched == "hlayered":
            order, _, _ = code.layer_order()
            inv = np.empty(code.M, np.int32); inv[order] = np.arange(code.M, dtype=np.int32)
            g2 = O.Graph.from_edges(code.N, code.M, *_reorder(var, chk, inv))
        ref = O.decode(g2, llr, "NMS", 0.75, 30, sched, n_threads=8)
        dec = q.Decoder(code, code.N, 30, rule="NMS", rule_param=0.75, n_frames=F, schedule=sched)
        hard, it, ok, _ = staged(q, torch, dec, llr, want_post=False)
        assert (hard == ref["hard"]).all() and (it == ref["iters"]).all() and (ok == ref["synd_ok"]).all()
        assert ok.mean() > 0.9 and (hard[ok == 1] == 0).all()


@pytest.mark.parametrize("variant", ["frames-bits", "frames-llr", "frames-f16", "frames-i8", "edges", "hlayered", "frames-bits-compact"])
def test_per_frame_erasures(q, O, torch, variant):
    """qldpc_load_erasures_dev: the harness's `LLRs[pattern[i]] = 0` (BS/src/main.cpp:359-362) with a pattern per frame -- every
    engine, message width and LLR form must give what the oracle gives on LLRs with those entries zeroed."""
    rng = np.random.default_rng(21)
    code = q.Code.ira(2048, 1536, 0.2, 8, 3, 11)
    var, chk = code.edges()
    og = O.Graph.from_edges(code.N, code.M, var, chk)
    N, K = code.N, 1536
    F = 8 if variant == "edges" else 330
    enc = q.Encoder(code, "IRA")
    info = rng.integers(0, 2, (F, K)).astype(np.uint8)
    cw = q.unpack_bits(enc.encode_packed(torch.from_numpy(q.pack_bits(info).view(np.int32)).cuda()).cpu().numpy().view(np.uint32), N)
    qber = rng.uniform(0.005, 0.03, F).astype(np.float32)
    y = cw.copy()
    y[:, :K] ^= (rng.random((F, K)) < qber[:, None]).astype(np.uint8)
    mag = np.array([q.bsc_llr(float(p)) for p in qber], np.float32)
    cls = np.zeros(N, np.uint8)
    cls[K:] = q.VN_PINNED
    erase = np.zeros((F, N), np.uint8)
    for f in range(F):                                            # a different number of evenly spaced punctured parity VNs per frame
        p = int(rng.integers(0, 200))
        j = np.arange(N - K, dtype=np.int64)
        erase[f, K:] = ((j + 1) * p // (N - K) > j * p // (N - K))
    llr = np.where(y == 1, -mag[:, None], mag[:, None]).astype(np.float32)
    llr[:, K:] = np.where(cw[:, K:] == 1, -np.float32(q.CONFIRMED_BIT_LLR), np.float32(q.CONFIRMED_BIT_LLR))
    llr[erase == 1] = 0.0
    kw = dict(rule="NMS", rule_param=0.75, n_frames=F)
    okw = {}
    sched = "flooding"
    if variant == "frames-f16":
        kw["msg_dtype"] = "f16"; okw["msg_fp16"] = True
    if variant == "frames-i8":
        kw["msg_dtype"] = "i8"; okw["msg_i8"] = True
    if variant == "edges":
        kw["engine"] = "edges"
    if variant == "frames-bits-compact":
        kw["compact"] = "on"
    g2 = og
    if variant == "hlayered":
        sched = "hlayered"; kw["schedule"] = "hlayered"
        order, _, _ = code.layer_order()
        inv = np.empty(code.M, np.int32); inv[order] = np.arange(code.M, dtype=np.int32)
        g2 = O.Graph.from_edges(code.N, code.M, *_reorder(var, chk, inv))
    ref = O.decode(g2, llr, "NMS", 0.75, 30, sched, True, 1, n_threads=8, **okw)
    dec = q.Decoder(code, N, 30, **kw)
    if variant == "frames-llr":
        clean = llr.copy()
        clean[erase == 1] = np.where(cw[erase == 1] == 1, -np.float32(q.CONFIRMED_BIT_LLR), np.float32(q.CONFIRMED_BIT_LLR))      # un-erased; the erasure comes from the call
        dec.load_llr(torch.from_numpy(clean).cuda())
    else:
        dec.load_bits(torch.from_numpy(q.pack_bits(y).view(np.int32)).cuda(), torch.from_numpy(mag).cuda(), torch.from_numpy(cls).cuda())
    dec.load_erasures(torch.from_numpy(q.pack_bits(erase).view(np.int32)).cuda())
    dec.run()
    hard = q.unpack_bits(dec.fetch_packed().cpu().numpy().view(np.uint32), N)
    it, ok = dec.fetch_status()
    assert (hard == ref["hard"]).all() and (it.cpu().numpy() == ref["iters"]).all() and (ok.cpu().numpy() == ref["synd_ok"]).all()
    assert (ref["synd_ok"] == 1).mean() > 0.5


@pytest.mark.parametrize("engine", ["frames", "edges"])
def test_spa_tolerance_at_scale(q, O, torch, peg, engine):
    """SURVEY.md section 8c, the tolerance of the float-LLR SPA variant, in the driver's suite: over >= 10^4 frames inside the waterfall
    >= 99.9 % of the frame verdicts (reconciled / not) are the oracle's and the GPU's FER lies inside the binomial 95 % interval around
    the oracle's (device exp / log / rcp differ from glibc's tanh / atanh in the last ulp; the edge engine also multiplies in tree
    order).  tests/spa_verdicts.py is the same check on the full-size config-2 code (10 240 frames, QBER 3.0 %: all verdicts identical)."""
    code, og = peg
    F, N = 10240, 1008
    rng = np.random.default_rng(31)
    enc = q.Encoder(code, "IDENTITY")
    cw = enc.encode(rng.integers(0, 2, (F, enc.K)))
    y = cw ^ (rng.random((F, N)) < 0.07)
    llr = np.where(y == 1, -2.5866892, 2.5866892).astype(np.float32)   # ln((1 - p) / p) at p = 0.07: FER 0.15, inside the waterfall
    ref = O.decode(og, llr, "SPA", 0.0, 30, "flooding", True, 1, n_threads=min(16, os.cpu_count() or 8))
    o_good = (ref["hard"] == cw).all(axis=1) & (ref["synd_ok"] == 1)
    hard = np.empty((F, N), np.uint8)
    okg = np.empty(F, np.int32)
    B = 2048
    dec = q.Decoder(code, N, 30, rule="SPA", n_frames=B, engine=engine)
    for lo in range(0, F, B):
        dec.load_llr(torch.from_numpy(llr[lo:lo + B]).cuda())
        dec.run()
        hard[lo:lo + B] = q.unpack_bits(dec.fetch_packed().cpu().numpy().view(np.uint32), N)
        okg[lo:lo + B] = dec.fetch_status()[1].cpu().numpy()
    g_good = (hard == cw).all(axis=1) & (okg == 1)
    p = 1.0 - o_good.mean()
    assert 0.02 < p < 0.6, p                                             # inside the waterfall: both outcomes are plentiful
    assert (g_good == o_good).mean() >= 0.999, (g_good == o_good).mean()
    ci = 1.96 * np.sqrt(p * (1 - p) / F)
    assert abs((1.0 - g_good.mean()) - p) <= ci, (1.0 - g_good.mean(), p, ci)
    same = (hard == ref["hard"]).all(axis=1)
    assert same[o_good].mean() >= 0.999                                  # and on frames the oracle reconciles it is the same word


@pytest.mark.parametrize("rule,param", EXACT[:3])
@pytest.mark.parametrize("frames", [5, 64, 200])
def test_one_launch_layered_sweep_is_bit_exact(q, O, torch, peg, rule, param, frames):
    """Round 3 (qldpc_kernels_chain.h): a layered sweep as ONE launch -- a check waits for the earlier checks on its own VNs (per-VN
    version counters, agent-coherent posterior rows) instead of for the whole layer before it.  The order of the updates on every VN is
    that of the launch-per-layer sweep, so hard decisions, iteration counts and success flags must be those of the oracle given the
    code's row order, and in fixed-iteration mode the posteriors bit for bit: one ragged group, one full group, several groups."""
    code, og = peg
    order, _, _ = code.layer_order()
    var, chk = og.edges()
    inv = np.empty(code.M, np.int32); inv[order] = np.arange(code.M, dtype=np.int32)
    og = O.Graph.from_edges(code.N, code.M, *_reorder(var, chk, inv))
    llr = bsc_frames(np.random.default_rng(40 + frames), frames, 1008, 0.065 if rule in ("NMS", "OMS") else 0.045, 2.6)
    for synd in (True, False):
        ref = O.decode(og, llr, rule, param, 14, "hlayered", synd, 1, n_threads=8)
        dec = q.Decoder(code, 1008, 14, rule=rule, rule_param=param, n_frames=frames, schedule="hlayered", enable_syndrome=synd, layer_chain="on")
        hard, it, ok, post = staged(q, torch, dec, llr, want_post=not synd)
        assert (hard == ref["hard"]).all() and (it == ref["iters"]).all() and (ok == ref["synd_ok"]).all()
        if not synd:
            assert (post.view(np.uint32) == ref["post"].view(np.uint32)).all()     # bit-exact floats


@pytest.mark.parametrize("rule,param", EXACT)
@pytest.mark.parametrize("frames", [3, 64, 130])
def test_layered_min_sum_on_the_compressed_check_state_is_bit_exact(q, O, torch, monkeypatch, rule, param, frames):
    """Round 3 (qldpc_kernels_cst.h): layered MS / OMS / NMS / AMS sweeps keep {cst1, cst2} and two dc-bit masks per check and frame instead of the dc
    messages and rebuild each message from them.  Same floats in, same floats out: hard decisions, iteration counts, success flags and (fixed
    iterations) posteriors must be those of the oracle, with the state on (the default for these rules) and with explicit messages
    (QLDPC_LAYER_CST = 0), on an irregular code whose checks fall into several degree buckets; and the syndrome (coset) form through load_bits."""
    code = q.Code.ira(4096, 3277, 0.125, 11, 3, 7)
    assert 8 < code.max_cn_degree <= 32
    order, _, _ = code.layer_order()
    var, chk = code.edges()
    inv = np.empty(code.M, np.int32); inv[order] = np.arange(code.M, dtype=np.int32)
    og = O.Graph.from_edges(code.N, code.M, *_reorder(var, chk, inv))
    llr = bsc_frames(np.random.default_rng(70 + frames), frames, code.N, 0.03, 2.9)
    for synd in (True, False):
        ref = O.decode(og, llr, rule, param, 12, "hlayered", synd, 1, n_threads=8)
        got = {}
        for cst in ("1", "0"):
            monkeypatch.setenv("QLDPC_LAYER_CST", cst)
            dec = q.Decoder(code, code.N, 12, rule=rule, rule_param=param, n_frames=frames, schedule="hlayered", enable_syndrome=synd, layer_chain="off")
            dec.profile(True)
            hard, it, ok, post = staged(q, torch, dec, llr, want_post=not synd)
            assert (hard == ref["hard"]).all() and (it == ref["iters"]).all() and (ok == ref["synd_ok"]).all(), (cst, synd)
            if not synd:
                assert (post.view(np.uint32) == ref["post"].view(np.uint32)).all(), cst     # bit-exact floats
            lay = {k["name"]: k for k in dec.profile_read()}["layer_update"]
            got[cst] = lay["moved_bytes"] / lay["alg_bytes"]
        assert got["0"] == 1.0 and got["1"] < 0.75      # the state path really ran: 2 E + 8 M rows instead of 4 E


@pytest.mark.parametrize("rule,param,cst", [("NMS", 0.75, "1"), ("NMS", 0.75, "0"), ("OMS", 0.25, "0")])
@pytest.mark.parametrize("frames", [5, 130])
def test_layer_records_and_the_index_walk_give_the_same_floats(q, O, torch, monkeypatch, rule, param, cst, frames):
    """Round 3, third session (bucket::d_rec): the layer kernels learn {check, first edge, degree, VNs} from one aligned record per list
    entry instead of walking list -> cn_ptr -> cn_var.  Only where the indices come from changes: with the records (the default) and with
    the index walk (QLDPC_LAYER_REC = 0), on the compressed check state and on explicit messages, on a code whose checks fall into several
    degree buckets (several record strides), hard decisions, iteration counts, success flags and posteriors are the oracle's bit for bit."""
    code = q.Code.ira(4096, 3277, 0.125, 11, 3, 7)
    order, _, _ = code.layer_order()
    var, chk = code.edges()
    inv = np.empty(code.M, np.int32); inv[order] = np.arange(code.M, dtype=np.int32)
    og = O.Graph.from_edges(code.N, code.M, *_reorder(var, chk, inv))
    llr = bsc_frames(np.random.default_rng(170 + frames), frames, code.N, 0.03, 2.9)
    monkeypatch.setenv("QLDPC_LAYER_CST", cst)
    for synd in (True, False):
        ref = O.decode(og, llr, rule, param, 10, "hlayered", synd, 1, n_threads=8)
        for rec in ("1", "0"):
            monkeypatch.setenv("QLDPC_LAYER_REC", rec)
            dec = q.Decoder(code, code.N, 10, rule=rule, rule_param=param, n_frames=frames, schedule="hlayered", enable_syndrome=synd, layer_chain="off")
            hard, it, ok, post = staged(q, torch, dec, llr, want_post=not synd)
            assert (hard == ref["hard"]).all() and (it == ref["iters"]).all() and (ok == ref["synd_ok"]).all(), (rec, synd)
            if not synd:
                assert (post.view(np.uint32) == ref["post"].view(np.uint32)).all(), rec
