"""CPU suite: the oracle against every pin the reference holds for the LDPC path.

Pins: the commented-out known-answer vectors in
VAR/main.cpp (alist-v1.0.1):445,447,456,460 on BS/matrices/H/PEGReg504x1008.alist, decoder
'BP flooding SPA', n_ite = 10, syndrome on, depth 1 (:39-41,:454).
"""
import json
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def kat(gold):
    return json.load(open(os.path.join(gold, "kat_peg504x1008.json")))


@pytest.fixture(scope="module")
def peg(O, gold):
    return O.Graph.from_alist(os.path.join(gold, "PEGReg504x1008.alist"))


def test_alist_shape(peg):
    assert (peg.N, peg.M, peg.E, peg.max_dv, peg.max_dc) == (1008, 504, 3024, 3, 8)


def test_kat_encoded_is_codeword(peg, kat):
    w, _ = peg.syndrome(kat["encoded"])
    assert w == 0
    assert kat["data"] == kat["encoded"][504:]          # info_bits_pos = 504..1007


def test_kat_flooding_spa_exact(O, peg, kat):
    r = O.decode(peg, np.array(kat["llrs"], np.float32), "SPA", 0.0, 10, "flooding", True, 1)
    assert r["iters"][0] == 6                            # converges at the 6th iteration, before n_ite
    assert r["synd_ok"][0] == 1
    assert (r["hard"][0][504:] == np.array(kat["decoded"])).all()
    raw = (np.array(kat["llrs"]) < 0).astype(int)[504:]
    assert (raw != np.array(kat["decoded"])).sum() == 32  # a non-trivial decode: 32 channel errors fixed


@pytest.mark.parametrize("rule,param,sched", [("NMS", 0.75, "flooding"), ("OMS", 0.5, "flooding"), ("MS", 0.0, "flooding"),
                                              ("MS", 0.0, "hlayered"), ("NMS", 0.75, "hlayered"), ("SPA", 0.0, "hlayered"),
                                              ("LSPA", 0.0, "flooding"), ("AMS_MINSTAR", 0.0, "flooding"),
                                              ("AMS_MINSTAR_L2", 0.0, "flooding"), ("AMS_MIN", 0.0, "flooding")])
def test_kat_other_rules_reach_same_word(O, peg, kat, rule, param, sched):
    r = O.decode(peg, np.array(kat["llrs"], np.float32), rule, param, 50, sched)
    assert r["synd_ok"][0] == 1 and r["iters"][0] < 50
    assert (r["hard"][0][504:] == np.array(kat["decoded"])).all()


def test_syndrome_disabled_runs_all_iterations(O, peg, kat):
    r = O.decode(peg, np.array(kat["llrs"], np.float32), "SPA", 0.0, 10, "flooding", False, 1)
    assert r["iters"][0] == 10
    assert (r["hard"][0][504:] == np.array(kat["decoded"])).all()


def test_syndrome_depth_two_needs_one_more_iteration(O, peg, kat):
    r1 = O.decode(peg, np.array(kat["llrs"], np.float32), "SPA", 0.0, 10, "flooding", True, 1)
    r2 = O.decode(peg, np.array(kat["llrs"], np.float32), "SPA", 0.0, 10, "flooding", True, 2)
    assert r2["iters"][0] == r1["iters"][0] + 1


def test_last_iteration_is_not_checked(O, peg, kat):
    # AFF3CT skips the syndrome test on ite == n_ite-1: with n_ite = 6 the loop runs out instead of breaking
    r = O.decode(peg, np.array(kat["llrs"], np.float32), "SPA", 0.0, 6, "flooding", True, 1)
    assert r["iters"][0] == 6 and r["synd_ok"][0] == 1


def test_qc_convention(O, gold):
    # test2.qc: 18 x 6 blocks of Z = 7 (EC/ldpc_examples/README.md:7); shift s puts row r's 1 at column (r+s)%Z
    g = O.Graph.from_qc(os.path.join(gold, "test2.qc"))
    assert (g.N, g.M) == (126, 42)
    ex = g.export()
    row0 = sorted(ex["cn_var"][ex["cn_ptr"][0]:ex["cn_ptr"][1]].tolist())
    # first base row: 564 -1 276 -1 522 -1 404 -1 579 -1 332 -1 0 -1 ...  -> columns j*7 + (0 + s) % 7
    exp = sorted([0 * 7 + 564 % 7, 2 * 7 + 276 % 7, 4 * 7 + 522 % 7, 6 * 7 + 404 % 7, 8 * 7 + 579 % 7, 10 * 7 + 332 % 7, 12 * 7 + 0])
    assert row0 == exp


@pytest.mark.parametrize("name,shape", [("20.alist", (504, 252)), ("1998.5.3.2665.alist", (1998, 222))])
def test_other_alists_parse(O, gold, name, shape):
    g = O.Graph.from_alist(os.path.join(gold, name))
    assert (g.N, g.M) == shape
    ex = g.export()
    assert ex["vn_ptr"][-1] == g.E and ex["cn_ptr"][-1] == g.E
    assert sorted(ex["transpose"].tolist()) == list(range(g.E))


def test_all_zero_codeword_all_rules(O, peg):
    llr = np.full((2, 1008), 2.0, np.float32)
    for rule in O.RULES:
        for sched in O.SCHEDULES:
            r = O.decode(peg, llr, rule, 0.75 if rule == "NMS" else 0.0, 5, sched)
            assert (r["hard"] == 0).all() and (r["iters"] == 1).all()


def test_threads_do_not_change_results(O, peg):
    rng = np.random.default_rng(3)
    llr = np.where(rng.random((16, 1008)) < 0.07, -2.59, 2.59).astype(np.float32)
    a = O.decode(peg, llr, "NMS", 0.75, 20, n_threads=1)
    b = O.decode(peg, llr, "NMS", 0.75, 20, n_threads=4)
    assert (a["hard"] == b["hard"]).all() and (a["iters"] == b["iters"]).all() and (a["post"] == b["post"]).all()


def test_fp16_rounding_emulation_matches_numpy_float16(O):
    import ctypes as C
    L = O.lib()
    L.orc_round_fp16.restype = C.c_float
    L.orc_round_fp16.argtypes = [C.c_float]
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.normal(0, 30, 4000), rng.normal(0, 1e-3, 1000), rng.normal(0, 1e-6, 1000), rng.normal(0, 1e-8, 1000),
                           [0.0, -0.0, 65504, 65519.9, 65520, 7e4, -7e4, 6.1e-5, 5.96e-8, 2.9802322e-8, 2.99e-8, 1e-10, 8.9e-8]]).astype(np.float32)
    with np.errstate(over="ignore"):
        exp = vals.astype(np.float16).astype(np.float32)
    got = np.array([L.orc_round_fp16(float(v)) for v in vals], np.float32)
    assert (got.view(np.uint32) == exp.view(np.uint32)).all()


def test_fp16_message_storage_changes_little(O, peg):
    rng = np.random.default_rng(2)
    llr = np.where(rng.random((32, 1008)) < 0.05, -2.9, 2.9).astype(np.float32)
    a = O.decode(peg, llr, "NMS", 0.75, 20)
    b = O.decode(peg, llr, "NMS", 0.75, 20, msg_fp16=True)
    assert (a["synd_ok"] == b["synd_ok"]).mean() > 0.9 and np.abs(a["iters"] - b["iters"]).max() <= 3


@pytest.mark.parametrize("sched,n_ite", [("flooding", 4), ("hlayered", 2)])
@pytest.mark.parametrize("rule,param", [("MS", 0.0), ("OMS", 1.0), ("NMS", 1.0)])
def test_fixed_point_decoder_equals_the_float_one_while_nothing_saturates(O, peg, rule, param, sched, n_ite):
    """The integer restatement (orc_decode_i8, what the 8-bit GPU variant is tested against) is pinned to the float decoder
    (itself pinned by the KAT): on integer-valued LLRs, with rules that stay in the integers (offset 1, factor 1) and few
    enough iterations that neither the +-127 (flooding) nor the +-31 (layered) message clamp is reached, both compute the
    same numbers exactly."""
    rng = np.random.default_rng(5)
    llr = np.where(rng.random((40, 1008)) < 0.06, -3.0, 3.0).astype(np.float32)
    a = O.decode(peg, llr, rule, param, n_ite, sched, enable_syndrome=False)
    b = O.decode(peg, llr, rule, param, n_ite, sched, enable_syndrome=False, msg_i8=True, quant_scale=1.0)
    lim = 127 if sched == "flooding" else 31
    assert np.abs(a["post"]).max() < lim                                 # the premise: no clamp was active
    assert (a["post"] == b["post"]).all() and (a["hard"] == b["hard"]).all() and (a["synd_ok"] == b["synd_ok"]).all()


def test_fixed_point_quantiser_and_rules(O, peg):
    llr = np.zeros((1, 1008), np.float32)
    llr[0, :8] = [0.124, 0.126, -0.374, 31.7, 31.9, -40.0, 1e9, -1e9]   # scale 4: 0.496 -> 0, 0.504 -> 1, -1.496 -> -1, 126.8 -> 127, clamp
    r = O.decode(peg, llr, "MS", 0.0, 0, enable_syndrome=False, msg_i8=True, quant_scale=4.0)     # zero iterations: posterior = quantised LLR
    assert r["post"][0, :8].tolist() == [0, 1, -1, 127, 127, -127, 127, -127]
    # NMS factor is rint(alpha * 128) / 128 applied with a floor: a lone-error frame shows (m * 96) >> 7 on its first messages
    llr = np.full((1, 1008), 2.5, np.float32)                             # -> 10
    llr[0, 17] = -2.5
    r = O.decode(peg, llr, "NMS", 0.75, 1, enable_syndrome=False, msg_i8=True, quant_scale=4.0)
    m = (10 * 96) >> 7                                                    # 7
    assert set(np.unique(r["post"][0]).tolist()) == {10 + 3 * m, 10 + m, -10 + 3 * m}   # untouched VNs, neighbours of the flipped VN's checks, the flipped VN


@pytest.mark.parametrize("ebno,F", [(1.0, 600), (1.5, 1500), (2.0, 4000)])
def test_fixed_point_layered_decoder_reproduces_the_references_matlab_fer(O, gold, ebno, F):
    """Statistical pin for the integer layered decoder (the 8-bit GPU variant's oracle): the frame-error rates the reference
    publishes for its fixed-point layered offset-min-sum MATLAB decoder (sim_results.m:8-13: NR_2_6_52 at rate 1/2, 20
    iterations, offset 2, 6-bit messages / 8-bit posteriors) on the same matrix, channel, puncturing and quantiser
    (tests/matlab_fp.py says which message choice reproduces the table)."""
    import matlab_fp
    name = "NR_2_6_52"
    g = O.Graph.from_qc(os.path.join(gold, matlab_fp.PINS[name]["qc"]))
    x, rq, k = matlab_fp.frames(name, ebno, F, np.random.default_rng(int(ebno * 10) + F))
    r = O.decode(g, rq, "OMS", matlab_fp.OFFSET, matlab_fp.MAX_ITRS, "hlayered", enable_syndrome=False, n_threads=8, msg_i8=True, quant_scale=1.0)
    fer = float((r["hard"][:, :k] != x[:, :k]).any(1).mean())
    lo, hi = matlab_fp.band(name, ebno, F)
    print("%s Eb/N0 %.2f dB: FER %.4f (published %.4f, band %.4f..%.4f)" % (name, ebno, fer, matlab_fp.published(name, ebno)["fer"], lo, hi))
    assert lo <= fer <= hi
    assert np.abs(r["post"]).max() <= 127


def test_fixed_point_layered_decoder_equals_a_literal_restatement_of_the_matlab_loop(O, gold):
    """The same recursion written the way the MATLAB script writes it (block rows, mul_sh rotations, R storage per base
    entry, [-32, 31] / [-128, 127] clamps) in numpy: identical posteriors, so the symmetric clamps of the integer decoder
    change nothing on this workload."""
    import matlab_fp
    name = "NR_2_6_52"
    p = matlab_fp.PINS[name]
    z, nb, mb = p["z"], p["nb_rm"], p["mb_rm"]
    B = np.array([l.split() for l in open(os.path.join(gold, p["qc"])) if l.strip()][1:], dtype=int)
    x, rq, k = matlab_fp.frames(name, 1.5, 120, np.random.default_rng(3))
    F = rq.shape[0]
    rot = lambda a, s: np.concatenate([a[:, s % z:], a[:, :s % z]], axis=1)      # mul_sh
    L = rq.astype(np.int64).copy()
    R = np.zeros((int((B != -1).sum()), F, z), np.int64)
    for _ in range(matlab_fp.MAX_ITRS):
        Ri = 0
        for lyr in range(mb):
            cols = [c for c in range(nb) if B[lyr, c] != -1]
            treg = []
            for c in cols:
                L[:, c * z:(c + 1) * z] -= R[Ri]
                treg.append(np.clip(rot(L[:, c * z:(c + 1) * z], B[lyr, c]), -32, 31))
                Ri += 1
            T = np.stack(treg)
            A = np.abs(T)
            pos = A.argmin(0)
            min1 = A.min(0)
            A2 = A.copy()
            np.put_along_axis(A2, pos[None], 10 ** 9, 0)
            min2 = A2.min(0)
            S = np.where(T >= 0, 1, -1)
            out = np.broadcast_to(np.maximum(min1 - 2, 0), T.shape).copy()
            np.put_along_axis(out, pos[None], np.maximum(min2 - 2, 0)[None], 0)
            out = S.prod(0)[None] * S * out
            Ri -= len(cols)
            for ti, c in enumerate(cols):
                R[Ri] = rot(out[ti], z - B[lyr, c])
                L[:, c * z:(c + 1) * z] = np.clip(L[:, c * z:(c + 1) * z] + R[Ri], -128, 127)
                Ri += 1
    g = O.Graph.from_qc(os.path.join(gold, p["qc"]))
    r = O.decode(g, rq, "OMS", 2.0, matlab_fp.MAX_ITRS, "hlayered", enable_syndrome=False, n_threads=8, msg_i8=True, quant_scale=1.0)
    assert (r["post"] == L).all()


def test_float_layered_min_sum_follows_the_references_matlab_recursion(O, gold):
    """decode_hlayered (the float horizontal-layered schedule) against a numpy restatement, in double precision, of the loop
    the reference keeps in MATLAB (BPSK_nrldpc_sim.m:29-69: L = L - R, min1 / min2 / parity per row, R = new, L = L + R,
    one base row per layer, all-zero word over AWGN, 8 sweeps): same decisions, posteriors equal to float32 accuracy."""
    import matlab_fp
    p = matlab_fp.PINS["NR_2_6_52"]
    z, nb, mb = p["z"], p["nb_rm"], p["mb_rm"]
    B = np.array([l.split() for l in open(os.path.join(gold, p["qc"])) if l.strip()][1:], dtype=int)
    rng = np.random.default_rng(12)
    F, n = 150, nb * z
    r = (1.0 + 0.75 * rng.standard_normal((F, n))).astype(np.float32)
    rot = lambda a, s: np.concatenate([a[:, s % z:], a[:, :s % z]], axis=1)
    L = r.astype(np.float64).copy()
    R = np.zeros((int((B != -1).sum()), F, z))
    for _ in range(8):
        Ri = 0
        for lyr in range(mb):
            cols = [c for c in range(nb) if B[lyr, c] != -1]
            treg = []
            for c in cols:
                L[:, c * z:(c + 1) * z] -= R[Ri]
                treg.append(rot(L[:, c * z:(c + 1) * z], B[lyr, c]))
                Ri += 1
            T = np.stack(treg)
            A = np.abs(T)
            pos = A.argmin(0)
            min1 = A.min(0)
            A2 = A.copy()
            np.put_along_axis(A2, pos[None], np.inf, 0)
            min2 = A2.min(0)
            S = np.sign(T)
            out = np.broadcast_to(min1, T.shape).copy()
            np.put_along_axis(out, pos[None], min2[None], 0)
            out = S.prod(0)[None] * S * out
            Ri -= len(cols)
            for ti, c in enumerate(cols):
                R[Ri] = rot(out[ti], z - B[lyr, c])
                L[:, c * z:(c + 1) * z] += R[Ri]
                Ri += 1
    g = O.Graph.from_qc(os.path.join(gold, p["qc"]))
    ref = O.decode(g, r, "MS", 0.0, 8, "hlayered", enable_syndrome=False, n_threads=8)
    assert np.allclose(ref["post"], L, rtol=2e-4, atol=2e-4)
    assert ((ref["hard"] == 1) == (L < 0)).mean() > 0.9999
