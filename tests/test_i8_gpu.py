"""GPU parity of the 8-bit fixed-point variant (qldpc_decoder_cfg.msg_dtype = 2) against the oracle's integer decoder.

Integer work, so the bar is bit-exactness (prompt section 3): hard decisions, iteration counts, success flags and -- with a
fixed iteration count, where no frame converges before its group stops -- the integer posteriors.  Against the float
decoder the variant is FER-tolerance class ("FER within 1.5x of fp32, reported not asserted");
parity unpinned against AFF3CT's own fixed-point build (no vector of it in the reference).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RULES = [("MS", 0.0), ("OMS", 0.5), ("NMS", 0.75), ("NMS", 0.8125)]


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


def i32(a):
    return np.ascontiguousarray(a).view(np.int32)


def bsc_frames(rng, F, N, p, mag):
    return np.where(rng.random((F, N)) < p, -mag, mag).astype(np.float32)


def run(q, torch, dec, llr, want_post):
    dec.load_llr(torch.from_numpy(llr).cuda())
    dec.run()
    hard = q.unpack_bits(dec.fetch_packed().cpu().numpy().view(np.uint32), dec.N)
    it, ok = dec.fetch_status()
    post = dec.fetch_post().cpu().numpy() if want_post else None
    return hard, it.cpu().numpy(), ok.cpu().numpy(), post


@pytest.mark.parametrize("rule,param", RULES)
@pytest.mark.parametrize("F", [1, 257, 600])
def test_fixed_iterations_bit_exact_incl_posteriors(q, O, torch, gold, rule, param, F):
    p = os.path.join(gold, "PEGReg504x1008.alist")
    code, og = q.Code.from_alist(p), O.Graph.from_alist(p)
    llr = bsc_frames(np.random.default_rng(F), F, 1008, 0.06, 2.75)
    ref = O.decode(og, llr, rule, param, 12, enable_syndrome=False, n_threads=8, msg_i8=True, quant_scale=4.0)
    dec = q.Decoder(code, 1008, 12, rule=rule, rule_param=param, n_frames=F, enable_syndrome=False, msg_dtype="i8", quant_scale=4.0)
    hard, it, ok, post = run(q, torch, dec, llr, True)
    assert (post == ref["post"]).all()                                  # integer posteriors, exactly
    assert (hard == ref["hard"]).all() and (it == ref["iters"]).all() and (ok == ref["synd_ok"]).all()


@pytest.mark.parametrize("rule,param,scale", [("NMS", 0.75, 4.0), ("OMS", 0.4, 6.0), ("NMS", 0.875, 2.5)])
def test_early_exit_bit_exact(q, O, torch, gold, rule, param, scale):
    p = os.path.join(gold, "PEGReg504x1008.alist")
    code, og = q.Code.from_alist(p), O.Graph.from_alist(p)
    F = 700
    llr = bsc_frames(np.random.default_rng(5), F, 1008, 0.065, 2.67)
    ref = O.decode(og, llr, rule, param, 30, n_threads=8, msg_i8=True, quant_scale=scale)
    dec = q.Decoder(code, 1008, 30, rule=rule, rule_param=param, n_frames=F, msg_dtype="i8", quant_scale=scale)
    hard, it, ok, _ = run(q, torch, dec, llr, False)
    assert (it == ref["iters"]).all() and (ok == ref["synd_ok"]).all() and (hard == ref["hard"]).all()
    assert 0.2 < ok.mean() < 1.0 or ok.all()                             # a mix of outcomes is what makes this a test
    # syndrome_depth 2 exercises the per-frame depth counter with four frames per lane
    ref2 = O.decode(og, llr, rule, param, 30, syndrome_depth=2, n_threads=8, msg_i8=True, quant_scale=scale)
    dec2 = q.Decoder(code, 1008, 30, rule=rule, rule_param=param, n_frames=F, syndrome_depth=2, msg_dtype="i8", quant_scale=scale)
    hard, it, ok, _ = run(q, torch, dec2, llr, False)
    assert (it == ref2["iters"]).all() and (ok == ref2["synd_ok"]).all() and (hard == ref2["hard"]).all()


def test_irregular_code_all_degree_buckets(q, O, torch):
    """dv up to 14 (register buckets 4 / 12 and the any-degree loop), dc above 40 (any-degree check loop)."""
    code = q.Code.ira(4096, 3850, 0.4, 14, 4, 3)
    assert code.max_cn_degree > 40 and code.max_vn_degree > 12
    var, chk = code.edges()
    og = O.Graph.from_edges(code.N, code.M, var, chk)
    llr = bsc_frames(np.random.default_rng(8), 300, code.N, 0.004, 5.5)
    for synd in (False, True):
        ref = O.decode(og, llr, "NMS", 0.75, 10, enable_syndrome=synd, n_threads=8, msg_i8=True, quant_scale=8.0)      # 8 = the default
        dec = q.Decoder(code, code.N, 10, rule="NMS", rule_param=0.75, n_frames=300, enable_syndrome=synd, msg_dtype="i8")
        hard, it, ok, post = run(q, torch, dec, llr, not synd)
        assert (hard == ref["hard"]).all() and (it == ref["iters"]).all() and (ok == ref["synd_ok"]).all()
        if not synd:
            assert (post == ref["post"]).all()


def test_saturation_and_pinned_llrs(q, O, torch, gold):
    """+-23.03 pinned VNs quantise to +-92, a scale of 8 drives them (and the sums) into the +-127 clamps."""
    p = os.path.join(gold, "PEGReg504x1008.alist")
    code, og = q.Code.from_alist(p), O.Graph.from_alist(p)
    rng = np.random.default_rng(2)
    llr = bsc_frames(rng, 256, 1008, 0.05, 2.94)
    llr[:, ::7] = np.where(llr[:, ::7] > 0, 23.02585, -23.02585)
    llr[:, 5::11] = 0.0                                                  # punctured
    for scale in (4.0, 8.0, 40.0):
        ref = O.decode(og, llr, "OMS", 0.5, 9, enable_syndrome=False, n_threads=8, msg_i8=True, quant_scale=scale)
        dec = q.Decoder(code, 1008, 9, rule="OMS", rule_param=0.5, n_frames=256, enable_syndrome=False, msg_dtype="i8", quant_scale=scale)
        hard, it, ok, post = run(q, torch, dec, llr, True)
        assert (post == ref["post"]).all() and (hard == ref["hard"]).all() and (ok == ref["synd_ok"]).all()
    assert np.abs(ref["post"]).max() > 127                               # the posterior itself is not clamped


def test_syndrome_form_and_packed_bits(q, O, torch, gold):
    p = os.path.join(gold, "PEGReg504x1008.alist")
    code, og = q.Code.from_alist(p), O.Graph.from_alist(p)
    rng = np.random.default_rng(4)
    F, qber = 300, 0.045
    x = rng.integers(0, 2, (F, 1008)).astype(np.uint8)
    s = np.stack([og.syndrome(xx)[1] for xx in x])
    y = x ^ (rng.random((F, 1008)) < qber)
    mag = np.float32(q.bsc_llr(qber))
    llr = np.where(y == 1, -mag, mag).astype(np.float32)
    ref = O.decode(og, llr, "NMS", 0.75, 30, n_threads=8, target=s, msg_i8=True, quant_scale=8.0)
    dec = q.Decoder(code, 1008, 30, rule="NMS", rule_param=0.75, n_frames=F, msg_dtype="i8")
    dec.load_bits(torch.from_numpy(i32(q.pack_bits(y))).cuda(), torch.full((F,), float(mag), device="cuda"))
    dec.load_syndrome(torch.from_numpy(i32(q.pack_bits(s))).cuda())
    dec.run()
    hard = q.unpack_bits(dec.fetch_packed().cpu().numpy().view(np.uint32), 1008)
    it, ok = dec.fetch_status()
    it, ok = it.cpu().numpy(), ok.cpu().numpy()
    assert (hard == ref["hard"]).all() and (it == ref["iters"]).all() and (ok == ref["synd_ok"]).all()
    good = ok == 1
    assert good.mean() > 0.8 and (hard[good] == x[good]).all()


def test_config2_code_sample_and_fer_class(q, O, torch):
    """256 frames of the headline code (N 65 536, rate 0.8, QBER 2 %): exact against the integer oracle; FER and iteration
    count in the class of the float decoder (reported, and loosely bounded)."""
    code = q.Code.ira(65536, 52429)
    enc = q.Encoder(code, "IRA")
    rng = np.random.default_rng(11)
    F = 256
    info = rng.integers(0, 2, (F, enc.K)).astype(np.uint8)
    cw = enc.encode(info)
    mag = np.float32(q.bsc_llr(0.02))
    noisy = cw.copy()
    noisy[:, :enc.K] ^= rng.random((F, enc.K)) < 0.02
    llr = np.where(noisy == 1, -mag, mag).astype(np.float32)
    llr[:, enc.K:] = np.where(cw[:, enc.K:] == 1, -np.float32(q.CONFIRMED_BIT_LLR), np.float32(q.CONFIRMED_BIT_LLR))
    var, chk = code.edges()
    og = O.Graph.from_edges(code.N, code.M, var, chk)
    ref = O.decode(og, llr, "NMS", 0.75, 50, n_threads=8, msg_i8=True, quant_scale=8.0)
    dec = q.Decoder(code, enc.K, 50, rule="NMS", rule_param=0.75, n_frames=F, msg_dtype="i8")
    hard, it, ok, _ = run(q, torch, dec, llr, False)
    assert (hard == ref["hard"]).all() and (it == ref["iters"]).all() and (ok == ref["synd_ok"]).all()
    f32 = q.Decoder(code, enc.K, 50, rule="NMS", rule_param=0.75, n_frames=F)
    h2, it2, ok2, _ = run(q, torch, f32, llr, False)
    print("i8: FER %.4f avg it %.2f | f32: FER %.4f avg it %.2f" % (1 - (hard == cw).all(1).mean(), it.mean(), 1 - (h2 == cw).all(1).mean(), it2.mean()))
    assert (hard == cw).all(1).mean() >= (h2 == cw).all(1).mean() - 0.05
    assert abs(it.mean() - it2.mean()) < 3.0


def test_unsupported_combinations_are_refused(q, gold):
    code = q.Code.from_alist(os.path.join(gold, "PEGReg504x1008.alist"))
    for kw in (dict(rule="SPA"), dict(rule="NMS", rule_param=0.75, engine="edges"),
               dict(rule="NMS", rule_param=0.75, freeze_messages=True), dict(rule="NMS", rule_param=0.75, frames_per_lane=2)):
        with pytest.raises(q.QldpcError) as e:
            q.Decoder(code, 1008, 10, n_frames=4, msg_dtype="i8", **kw)
        assert e.value.status == -7                                         # QLDPC_EUNSUPPORTED


def _layer_graph(O, code, og):
    """the oracle sweeps checks 0..M-1; give it H with its rows in the product's layer order"""
    order, _, _ = code.layer_order()
    var, chk = og.edges()
    inv = np.empty(code.M, np.int32); inv[order] = np.arange(code.M, dtype=np.int32)
    newc = inv[chk]; idx = np.argsort(newc, kind="stable")
    return O.Graph.from_edges(code.N, code.M, var[idx], newc[idx]), order


@pytest.mark.parametrize("rule,param", [("OMS", 0.5), ("NMS", 0.75), ("MS", 0.0)])
@pytest.mark.parametrize("synd", [False, True])
def test_layered_fixed_point_bit_exact(q, O, torch, gold, rule, param, synd):
    """horizontal layered with 8-bit posteriors and messages (the recursion of the reference's BPSK_nrldpc_sim_RM_FP.m)."""
    p = os.path.join(gold, "PEGReg504x1008.alist")
    code, og = q.Code.from_alist(p), O.Graph.from_alist(p)
    og2, _ = _layer_graph(O, code, og)
    F = 700
    llr = bsc_frames(np.random.default_rng(21), F, 1008, 0.07, 2.6)
    ref = O.decode(og2, llr, rule, param, 20, "hlayered", enable_syndrome=synd, n_threads=8, msg_i8=True, quant_scale=8.0)
    dec = q.Decoder(code, 1008, 20, rule=rule, rule_param=param, n_frames=F, schedule="hlayered", enable_syndrome=synd, msg_dtype="i8")
    hard, it, ok, post = run(q, torch, dec, llr, not synd)
    assert (hard == ref["hard"]).all() and (it == ref["iters"]).all() and (ok == ref["synd_ok"]).all()
    if not synd:
        assert (post == ref["post"]).all() and np.abs(post).max() <= 127


def test_layered_fixed_point_irregular_and_syndrome_form(q, O, torch):
    code = q.Code.ira(4096, 3850, 0.4, 14, 4, 3)               # dc > 40: any-degree loop; colour-class layers
    var, chk = code.edges()
    og = O.Graph.from_edges(code.N, code.M, var, chk)
    og2, order = _layer_graph(O, code, og)
    rng = np.random.default_rng(9)
    F = 260
    x = rng.integers(0, 2, (F, code.N)).astype(np.uint8)
    s = np.stack([og.syndrome(xx)[1] for xx in x])
    y = x ^ (rng.random((F, code.N)) < 0.004)
    mag = np.float32(q.bsc_llr(0.004))
    llr = np.where(y == 1, -mag, mag).astype(np.float32)
    ref = O.decode(og2, llr, "NMS", 0.75, 12, "hlayered", n_threads=8, target=s[:, order], msg_i8=True, quant_scale=8.0)
    dec = q.Decoder(code, code.N, 12, rule="NMS", rule_param=0.75, n_frames=F, schedule="hlayered", msg_dtype="i8")
    dec.load_bits(torch.from_numpy(i32(q.pack_bits(y))).cuda(), torch.full((F,), float(mag), device="cuda"))
    dec.load_syndrome(torch.from_numpy(i32(q.pack_bits(s))).cuda())
    dec.run()
    hard = q.unpack_bits(dec.fetch_packed().cpu().numpy().view(np.uint32), code.N)
    it, ok = dec.fetch_status()
    assert (hard == ref["hard"]).all() and (it.cpu().numpy() == ref["iters"]).all() and (ok.cpu().numpy() == ref["synd_ok"]).all()
    good = ok.cpu().numpy() == 1
    assert good.mean() > 0.2 and (hard[good] == x[good]).all()          # all N VNs are channel VNs here: a hard point for rate 0.94


def test_layered_fixed_point_runs_the_references_matlab_experiment(q, O, torch, gold):
    """The reference's fixed-point MATLAB decoder experiment (tests/matlab_fp.py; sim_results.m:8-13) on the GPU: NR_2_6_52 at
    rate 1/2, 6-bit quantised AWGN LLRs, layered offset-min-sum, offset 2, 20 sweeps.  Bit-exact against the integer oracle
    (which equals a literal restatement of the MATLAB loop, tests/test_oracle.py), and the published frame-error rate at
    2.35 dB (54 / 20 000) comes out of 20 000 frames here."""
    import matlab_fp
    name = "NR_2_6_52"
    path = os.path.join(gold, matlab_fp.PINS[name]["qc"])
    code, og = q.Code.from_qc(path), O.Graph.from_qc(path)
    assert code.layer_order()[2]                                         # natural order: the base rows are the layers, as in the script
    dec = q.Decoder(code, code.N, matlab_fp.MAX_ITRS, rule="OMS", rule_param=matlab_fp.OFFSET, n_frames=20000, schedule="hlayered",
                    enable_syndrome=False, msg_dtype="i8", quant_scale=1.0)
    x, rq, k = matlab_fp.frames(name, 1.5, 700, np.random.default_rng(8))
    ref = O.decode(og, rq, "OMS", matlab_fp.OFFSET, matlab_fp.MAX_ITRS, "hlayered", enable_syndrome=False, n_threads=8, msg_i8=True, quant_scale=1.0)
    hard, _, _, post = run(q, torch, dec, rq, True)
    assert (post == ref["post"]).all() and (hard == ref["hard"]).all()
    x, rq, k = matlab_fp.frames(name, 2.35, 20000, np.random.default_rng(9))
    hard, _, _, _ = run(q, torch, dec, rq, False)
    errs = int((hard[:, :k] != 0).any(1).sum())
    lo, hi = matlab_fp.band(name, 2.35, 20000)
    print("NR_2_6_52 @ 2.35 dB: %d frame errors of 20000 (published 54 of 20000)" % errs)
    assert lo <= errs / 20000.0 <= hi
