"""CPU suite: host logic of libqldpc (graph layer, scalar helpers, C ABI surface). No GPU compute."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(q):
    hdr = open(os.path.join(ROOT, "include", "qldpc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(qldpc_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 45
    lib = ctypes.CDLL(q.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_version_and_strerror(q):
    assert q.version() == 100
    assert q._L.qldpc_strerror(-6) == b"size mismatch"


def test_scalar_helpers_follow_the_harness_macros(q):
    # BS/src/main.cpp:19-34
    assert abs(q.llr_from_ber(0.02) - (-np.log(0.02 / 0.98))) < 1e-6
    assert abs(q.bsc_llr(0.02) - np.log(np.float32(0.98) / np.float32(0.02))) < 1e-6
    assert abs(q.CONFIRMED_BIT_LLR - (-np.log(1e-10 / (1 - 1e-10)))) < 1e-12
    h = -0.02 * np.log2(0.02) - 0.98 * np.log2(0.98)
    assert abs(q.binary_entropy(0.02) - h) < 1e-6
    assert abs(q.min_code_rate(0.02, 1.4) - 1 / (1 + 1.4 * h)) < 1e-6
    # parity_bits_to_punct(INFO_B, TTL_B, GOAL_CR) = -((INFO_B) - GOAL_CR*TTL_B)/GOAL_CR, truncated
    assert q.parity_bits_to_punct(64800, 48600, 0.8) == int(-(48600 - 0.8 * 64800) / 0.8)
    # README_LDPC.md:900-935 table row: QBER 2 % -> Shannon CR 0.8761
    assert abs(q.min_code_rate(0.02, 1.0) - 0.8761) < 5e-4


def test_rate_and_efficiency_helpers_reproduce_the_references_table(q):
    """errorcorrection/README_LDPC.md:901-935, the reconciliation-efficiency table: the "Shannon Limit" rows are CR = min_cr(q, 1)
    and K = 1 - 2 h(q); the DVB rows are f = ((1 - R) / R) / h(q) and K = (1 - FER) (1 - (1 - R) / R - h(q)).  (The two DVB rows printed under
    BER 0.01 repeat the BER 0.02 values -- 37/45 at 1 % would be f = 2.68 -- and are left out.)"""
    shannon = [(0.01, 0.9252, 0.8384), (0.02, 0.8761, 0.7171), (0.03, 0.8372, 0.6112), (0.04, 0.8050, 0.5154), (0.05, 0.7774, 0.4272),
               (0.06, 0.7533, 0.3451), (0.07, 0.7321, 0.2682), (0.08, 0.7132, 0.1956), (0.09, 0.6962, 0.1271), (0.10, 0.6807, 0.0620), (0.11, 0.6667, 0.0002)]
    for p, cr, k in shannon:
        assert abs(q.min_code_rate(p, 1.0) - cr) < 6e-5, p
        assert abs(1.0 - 2.0 * q.binary_entropy(p) - k) < 6e-5, p
    dvb = [(0.02, 37 / 45, 1.5287, 0.6423, 0), (0.03, 7 / 9, 1.4698, 0.5199, 0), (0.04, 59 / 81, 1.5390, 0.3848, 0), (0.05, 59 / 81, 1.3020, 0.2271, 10 / 30),
           (0.05, 2 / 3, 1.7458, 0.2136, 0), (0.06, 2 / 3, 1.5270, 0.1726, 0), (0.07, 2 / 3, 1.3664, 0.1341, 0), (0.08, 2 / 3, 1.2432, 0.0978, 0),
           (0.09, 2 / 3, 1.1456, 0.0233, 19 / 30), (0.09, 3 / 5, 1.5274, -0.1031, 0), (0.10, 3 / 5, 1.4215, -0.1357, 0)]
    for p, R, f, k, fer in dvb:      # fer: the frame-error rate the table quotes next to the row; its K is then (1 - FER) K
        ratio, h = (1.0 - R) / R, q.binary_entropy(p)
        assert abs(ratio / h - f) < 6e-4, (p, R)
        assert abs((1.0 - ratio - h) * (1.0 - fer) - k) < 6e-4, (p, R)
        # and the table's own logic: a code can be used at QBER p with efficiency f iff its rate is <= min_cr(p, f)
        assert R <= q.min_code_rate(p, f) + 1e-4


def test_puncturing_arithmetic_reproduces_the_references_5g_table(q):
    """errorcorrection/README_LDPC.md:937-974 (NR_1_0_2.qc: N = 136, K = 44, parity bits punctured to reach efficiency 1): for every
    QBER row the printed number of punctured bits is parity_bits_to_punct(K, N, min_cr(QBER, 1)) rounded down, and the printed code
    rate, reconciliation efficiency and key rate follow from it."""
    N, K = 136, 44
    rows = [(0.01, 88, 1.12521, 0.828298, 0.916667), (0.02, 85, 1.12479, 0.699469, 0.862745), (0.03, 83, 1.05223, 0.601063, 0.830189),
            (0.04, 81, 1.03181, 0.507708, 0.8), (0.05, 79, 1.03163, 0.418149, 0.77193), (0.06, 77, 1.04112, 0.331646, 0.745763),
            (0.07, 75, 1.05586, 0.247713, 0.721311), (0.08, 74, 1.01719, 0.18873, 0.709677), (0.09, 72, 1.04141, 0.108985, 0.6875),
            (0.10, 71, 1.01765, 0.0537318, 0.676923), (0.11, 70, 1.00017, 8.40425e-05, 0.666667)]
    for p, punct, f, key_rate, cr in rows:
        n = q.parity_bits_to_punct(N, K, q.min_code_rate(p, 1.0))
        assert n == punct, (p, n)
        rate = K / (N - n)
        ratio, h = (1.0 - rate) / rate, q.binary_entropy(p)
        assert abs(rate - cr) < 2e-6 and abs(ratio / h - f) < 2e-4 and abs(1.0 - ratio - h - key_rate) < 2e-5, p


def test_alist_graph_matches_oracle_graph(q, O, gold):
    p = os.path.join(gold, "PEGReg504x1008.alist")
    c, g = q.Code.from_alist(p), O.Graph.from_alist(p)
    assert (c.N, c.M, c.E, c.max_cn_degree, c.max_vn_degree) == (g.N, g.M, g.E, g.max_dc, g.max_dv)
    var, chk = c.edges()
    ovar, ochk = g.edges()
    assert (var == ovar).all() and (chk == ochk).all()


@pytest.mark.parametrize("name", ["test2.qc", "NR_2_3_112.qc", "NR_1_0_2.qc"])
def test_qc_graph_matches_oracle_graph(q, O, gold, name):
    p = os.path.join(gold, name)
    c, g = q.Code.from_qc(p), O.Graph.from_qc(p)
    var, chk = c.edges()
    ovar, ochk = g.edges()
    assert (var == ovar).all() and (chk == ochk).all()


def test_bad_files_return_codes(q, tmp_path):
    with pytest.raises(q.QldpcError) as e:
        q.Code.from_alist(str(tmp_path / "nope.alist"))
    assert e.value.status == -3
    bad = tmp_path / "bad.alist"
    bad.write_text("4 2\n2 2\n1 1 1 1\n2 2\n1\n2\n1\n9\n1 3\n2 4\n")
    with pytest.raises(q.QldpcError):
        q.Code.from_alist(str(bad))
    with pytest.raises(q.QldpcError) as e:
        q.Code.from_edges(4, 2, [0, 0], [0, 0])          # duplicate edge
    assert e.value.status == -1


def test_ira_config2_shape(q):
    # SURVEY.md section 8d, config 2
    c = q.Code.ira(65536, 52429, 0.125, 11, 3, 7)
    assert (c.N, c.M, c.E) == (65536, 13107, 235925)
    assert c.max_cn_degree == 18 and c.max_vn_degree == 11 and c.is_ira
    var, chk = c.edges()
    dv = np.bincount(var, minlength=c.N)
    dc = np.bincount(chk, minlength=c.M)
    assert (dv[:6553] == 11).all() and (dv[6553] == 4) and (dv[6554:52429] == 3).all()
    assert (dv[52429:-1] == 2).all() and dv[-1] == 1
    assert dc[0] == 17 and (dc[1:] == 18).all()
    # no duplicate edges, deterministic in the seed
    assert len(set(zip(var.tolist(), chk.tolist()))) == c.E
    v2, c2 = q.Code.ira(65536, 52429, 0.125, 11, 3, 7).edges()
    assert (v2 == var).all() and (c2 == chk).all()
    v3, _ = q.Code.ira(65536, 52429, 0.125, 11, 3, 8).edges()
    assert (v3 != var).any()


@pytest.mark.parametrize("rate", [0.5, 0.7, 0.8, 0.9])
def test_ira_multi_rate_set(q, rate):
    N = 8192
    K = int(round(N * rate))
    c = q.Code.ira(N, K, 0.125, 11, 3, 7)
    var, chk = c.edges()
    dc = np.bincount(chk, minlength=c.M)
    assert c.is_ira and dc[1:].min() == dc[1:].max()
    assert len(set(zip(var.tolist(), chk.tolist()))) == c.E


def test_layer_order_is_a_valid_schedule(q, gold):
    for c in (q.Code.from_alist(os.path.join(gold, "PEGReg504x1008.alist")), q.Code.from_qc(os.path.join(gold, "NR_2_3_112.qc")),
              q.Code.ira(4096, 3277)):
        order, ptr, natural = c.layer_order()
        assert sorted(order.tolist()) == list(range(c.M)) and ptr[0] == 0 and ptr[-1] == c.M
        var, chk = c.edges()
        vs = [set() for _ in range(c.M)]
        for v, m in zip(var.tolist(), chk.tolist()):
            vs[m].add(v)
        for l in range(c.n_layers):            # checks of a layer share no variable node
            seen = set()
            for m in order[ptr[l]:ptr[l + 1]]:
                assert not (seen & vs[m])
                seen |= vs[m]
        if natural:                            # level schedule: conflicting checks keep their c-order
            where = np.empty(c.M, int)
            for l in range(c.n_layers):
                where[order[ptr[l]:ptr[l + 1]]] = l
            last = {}
            for m in range(c.M):
                for v in vs[m]:
                    if v in last:
                        assert where[last[v]] < where[m]
                    last[v] = m


def test_colour_classes_by_dsatur_are_fewer_fuller_and_thread_safe(q, monkeypatch):
    """Round 3: when the natural order has no parallelism (IRA: the dual-diagonal chain) the layers are colour classes of the check-conflict graph, coloured by
    DSATUR instead of first-fit in index order: fewer classes, none of them a handful of checks, still VN-disjoint; the same code built on several threads at
    once (as qldpc_recon_create(preload) does) gets the same order as built alone -- the routine keeps no state outside its arrays."""
    import threading

    def layers(code):
        order, ptr, natural = code.layer_order()
        assert not natural and sorted(order.tolist()) == list(range(code.M))
        var, chk = code.edges()
        lay = np.empty(code.M, np.int64)
        for l in range(code.n_layers):
            lay[order[ptr[l]:ptr[l + 1]]] = l
        key = var.astype(np.int64) * 4096 + lay[chk]
        assert len(np.unique(key)) == len(key)          # no VN twice in one layer
        return order.copy(), np.diff(ptr)

    order, sizes = layers(q.Code.ira(65536, 52429, 0.125, 11, 3, 7))
    assert len(sizes) <= 24 and sizes.min() >= 64 and sizes[:20].min() > 400      # first-fit: 30 classes, the last five of 166, 109, 63, 15 and 5 checks
    monkeypatch.setenv("QLDPC_FIRST_FIT_LAYERS", "1")
    _, sizes_ff = layers(q.Code.ira(65536, 52429, 0.125, 11, 3, 7))
    monkeypatch.delenv("QLDPC_FIRST_FIT_LAYERS")
    assert len(sizes_ff) == 30 and sizes_ff.min() < 16
    got = [None] * 6

    def build(i):
        got[i] = q.Code.ira(65536, 52429, 0.125, 11, 3, 7).layer_order()[0].copy() if i % 2 == 0 else q.Code.ira(16384, 13107, 0.125, 11, 3, 7).layer_order()[0].copy()
    th = [threading.Thread(target=build, args=(i,)) for i in range(6)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert all((got[i] == order).all() for i in (0, 2, 4)) and (got[1] == got[3]).all() and (got[3] == got[5]).all()


def test_qc_natural_layers_are_block_rows(q, gold):
    c = q.Code.from_qc(os.path.join(gold, "NR_2_3_112.qc"))     # 42 block rows, Z = 112
    _, _, natural = c.layer_order()
    assert natural and c.n_layers <= 42


def test_syndrome_host_and_kat(q, gold):
    import json
    kat = json.load(open(os.path.join(gold, "kat_peg504x1008.json")))
    c = q.Code.from_alist(os.path.join(gold, "PEGReg504x1008.alist"))
    assert c.syndrome(kat["encoded"])[0] == 0
    x = list(kat["encoded"])
    x[7] ^= 1
    assert c.syndrome(x)[0] == 3                     # a degree-3 VN flips 3 checks


def test_pack_unpack_is_msb_first(q):
    # helpers.h:65-70: bit i <-> word[i/32] & (1 << (31 - i%32))
    bits = np.zeros(70, np.uint8)
    bits[[0, 33, 69]] = 1
    w = q.pack_bits(bits)
    assert w.tolist() == [1 << 31, 1 << 30, 1 << (31 - 5)]
    assert (q.unpack_bits(w, 70) == bits).all()


def test_decoder_without_gpu_fails_loudly(q, gold):
    if q.device_count() > 0:
        pytest.skip("GPU present")
    c = q.Code.from_alist(os.path.join(gold, "PEGReg504x1008.alist"))
    with pytest.raises(q.QldpcError) as e:
        q.Decoder(c, 504, 10)
    assert e.value.status == -5 and "no CPU fallback" in str(e.value)
    with pytest.raises(q.QldpcError):
        q.Encoder(c, "IDENTITY")


def test_decoder_argument_checks_come_before_the_device(q, gold):
    c = q.Code.from_alist(os.path.join(gold, "PEGReg504x1008.alist"))
    for kw, status in ((dict(K=0), -6), (dict(K=2000), -6), (dict(n_ite=0), -1), (dict(n_frames=0), -1), (dict(syndrome_depth=0), -1),
                       (dict(frames_per_lane=3), -1)):
        args = dict(K=504, n_ite=10, n_frames=1, syndrome_depth=1, frames_per_lane=0)
        args.update(kw)
        with pytest.raises(q.QldpcError) as e:
            q.Decoder(c, args["K"], args["n_ite"], n_frames=args["n_frames"], syndrome_depth=args["syndrome_depth"],
                      frames_per_lane=args["frames_per_lane"])
        assert e.value.status == status, kw


def test_peg_construction_has_no_4_cycles_and_keeps_the_ira_shape(q):
    """qldpc_code_ira_peg: depth 2 closes no 4-cycle, check degrees stay concentrated,
    the parity part is still the dual diagonal, and both ends build the same code from the same seed."""
    from collections import defaultdict
    N, K = 4096, 3277

    def shared_pairs(code):
        var, chk = code.edges()
        rows = defaultdict(list)
        for v, c in zip(var.tolist(), chk.tolist()):
            rows[c].append(v)
        seen = defaultdict(int)
        for vs in rows.values():
            for i in range(len(vs)):
                for j in range(i + 1, len(vs)):
                    seen[(min(vs[i], vs[j]), max(vs[i], vs[j]))] += 1
        return sum(1 for n in seen.values() if n >= 2)

    peg, rnd = q.Code.ira_peg(N, K, depth=2, seed=7), q.Code.ira(N, K, seed=7)
    assert shared_pairs(peg) == 0 and shared_pairs(rnd) > 100
    assert peg.is_ira and peg.N == N and peg.M == N - K
    var, chk = peg.edges()
    dv, dc = np.bincount(var, minlength=N), np.bincount(chk, minlength=N - K)
    assert (dv[:409] == 11).all() and (dv[409:K] == 3).all() and dc.max() - dc.min() <= 3
    v2, c2 = q.Code.ira_peg(N, K, depth=2, seed=7).edges()
    assert (v2 == var).all() and (c2 == chk).all()
    v3, _ = q.Code.ira_peg(N, K, depth=2, seed=8).edges()
    assert (v3 != var).any()
    with pytest.raises(q.QldpcError):
        q.Code.ira_peg(N, K, depth=9)


def test_qc_peg_generator(q, O, tmp_path):
    """qldpc_code_qc_peg: the reference's psd-peg.py (usage example `psd-peg.py 12 6 3 600 ...`) as a seeded C generator.
    Shape and degrees as the script produces them (H = [lift(P) | I]), no 4-cycle in the lifted graph, 6-cycles of the base
    graph broken by the shifts, the written .qc file reads back (through both readers) as the same code."""
    import scipy.sparse as sp
    n, m, dv, Z = 12, 6, 3, 601
    path = str(tmp_path / "peg.qc")
    code = q.Code.qc_peg(n, m, dv, Z, seed=3, qc_path=path)
    assert (code.N, code.M, code.E) == ((n + m) * Z, m * Z, (n * dv + m) * Z)
    var, chk = code.edges()
    dvs, dcs = np.bincount(var, minlength=code.N), np.bincount(chk, minlength=code.M)
    assert (dvs[:n * Z] == dv).all() and (dvs[n * Z:] == 1).all()
    assert dcs.max() - dcs.min() <= 1 and dcs.sum() == code.E                # PEG keeps the rows balanced: 12 * 3 / 6 = 6 (+1 identity)
    H = sp.csr_matrix((np.ones(code.E, np.int32), (chk, var)), shape=(code.M, code.N))
    A = (H.T @ H).tolil()
    A.setdiag(0)
    assert A.tocsr().max() <= 1                                              # two columns never share two rows: girth >= 6
    assert code.base_girth in (4, 6)                                         # 12 columns of degree 3 on 6 rows must close short base cycles
    # girth 8 in the lifted graph when Z is large enough for the tree-path condition: count 6-cycles through H H^T cubes is
    # expensive; check the base-cycle condition directly on the file instead
    lines = [l.split() for l in open(path).read().splitlines() if l.strip()]
    assert lines[0] == [str(n + m), str(m), str(Z)]
    B = np.array(lines[1:], dtype=np.int64)
    assert B.shape == (m, n + m) and ((B[:, n:] >= 0) == np.eye(m, dtype=bool)).all() and (B[:, n:][np.eye(m, dtype=bool)] == 0).all()
    assert ((B[:, :n] >= 0).sum(0) == dv).all() and B.max() < Z
    for j in range(n):                                                       # every base 4-cycle: alternating shift sum != 0 mod Z
        for k in range(j + 1, n):
            rows = [i for i in range(m) if B[i, j] >= 0 and B[i, k] >= 0]
            for x in range(len(rows)):
                for y in range(x + 1, len(rows)):
                    a, c = rows[x], rows[y]
                    assert (B[a, j] - B[a, k] + B[c, k] - B[c, j]) % Z != 0
    back = q.Code.from_qc(path)
    v2, c2 = back.edges()
    assert (v2 == var).all() and (c2 == chk).all()
    og = O.Graph.from_qc(path)
    v3, c3 = og.edges()
    assert (np.asarray(v3) == var).all() and (np.asarray(c3) == chk).all()
    v4, _ = q.Code.qc_peg(n, m, dv, Z, seed=3).edges()
    v5, _ = q.Code.qc_peg(n, m, dv, Z, seed=4).edges()
    assert (v4 == var).all() and (v5 != var).any()
    for bad in ((12, 6, 7, 601), (12, 6, 3, 1), (0, 6, 3, 601)):
        with pytest.raises(q.QldpcError):
            q.Code.qc_peg(*bad)


@pytest.mark.parametrize("order,name", [(0, "IDENTITY"), (1, "LU_DEC"), (2, "QC")])
def test_gf2_elimination_orders_give_systematic_generators(q, gold, order, name):
    """The host half of Encoder_LDPC_from_H's G_methods and Encoder_LDPC_from_QC (VAR/main.cpp (alist-v1.0.1):135-145, (qc):145):
    whatever the pivot order, x_parity = A x_info satisfies H x = 0 and the positions partition 0..N-1."""
    import ctypes as C
    code = q.Code.from_alist(os.path.join(gold, "PEGReg504x1008.alist")) if order < 2 else q.Code.from_qc(os.path.join(gold, "NR_1_0_2.qc"))
    f = q._L.qldpc_gf2_systematic_ord
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_int)]
    piv, fr, A, wpr = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_uint64)(), C.c_int()
    r = f(code._h, order, C.byref(piv), C.byref(fr), C.byref(A), C.byref(wpr))
    if order == 2 and r < 0:
        assert b"not invertible" in q._L.qldpc_last_error()
        return
    assert r > 0
    N, K = code.N, code.N - r
    pv, fp = np.array(piv[:r]), np.array(fr[:K])
    assert (np.diff(pv) > 0).all() and sorted(pv.tolist() + fp.tolist()) == list(range(N))
    if order == 0:
        assert (fp == np.arange(504, 1008)).all()          # the KAT's info_bits_pos (VAR/main.cpp (alist-v1.0.1):445-460)
    if order == 2:
        assert (fp == np.arange(K)).all()
    Ab = np.array(A[:r * wpr.value], dtype=np.uint64).reshape(r, wpr.value)
    bits = ((Ab[:, :, None] >> np.arange(64, dtype=np.uint64)[None, None, :]) & np.uint64(1)).reshape(r, -1)[:, :K].astype(np.uint8)
    rng = np.random.default_rng(order)
    u = rng.integers(0, 2, K).astype(np.uint8)
    x = np.zeros(N, np.uint8)
    x[fp] = u
    x[pv] = (bits @ u) & 1
    assert code.syndrome(x)[0] == 0
    libc = C.CDLL(None)
    for p in (piv, fr, A):
        libc.free(C.cast(p, C.c_void_p))


def test_chunked_crc_equals_the_bytewise_crc(q):
    """The device verification of the sessions (rk_verify / rk_crc in qldpc_recon.hip) computes CRC-32 as a fold of per-lane partial
    registers with x^len multipliers mod the CRC polynomial; qldpc_crc32_words_chunked is that arithmetic on the host.  It must
    equal the byte-wise table CRC (and zlib) for every length and every lane count, tail bits masked."""
    import zlib
    rng = np.random.default_rng(0)
    for nb in (1, 31, 32, 33, 64, 100, 1000, 8191, 8192, 52429, 65535, 65536, 120002, 262144):
        w = rng.integers(0, 2 ** 32, (nb + 31) // 32, dtype=np.uint64).astype(np.uint32)
        ref = q.crc32_words(w, nb)
        masked = w.copy()
        if nb & 31:
            masked[-1] &= np.uint32((0xFFFFFFFF << (32 - (nb & 31))) & 0xFFFFFFFF)
        assert ref == zlib.crc32(masked.astype(">u4").tobytes())
        for lanes in (1, 2, 8, 64, 256):
            assert q.crc32_words(w, nb, lanes) == ref, (nb, lanes)


def test_peg_edge_list_cache_returns_the_built_code_and_survives_a_damaged_file(q, tmp_path, monkeypatch):
    """QLDPC_CODE_CACHE: a PEG code is built once and read back afterwards -- the same graph edge for edge, layers included; a file whose
    checksum fails is ignored and rebuilt; Alice and Bob (no shared directory) still derive the same code."""
    monkeypatch.delenv("QLDPC_CODE_CACHE", raising=False)
    ref = q.Code.ira_peg(4096 + 1024, 4096, 0.125, 11, 3, 2, 7)
    monkeypatch.setenv("QLDPC_CODE_CACHE", str(tmp_path))
    built = q.Code.ira_peg(4096 + 1024, 4096, 0.125, 11, 3, 2, 7)
    files = list(tmp_path.iterdir())
    assert len(files) == 1 and files[0].name.startswith("ira_peg_N5120_K4096_") and files[0].suffix == ".edges"
    cached = q.Code.ira_peg(4096 + 1024, 4096, 0.125, 11, 3, 2, 7)
    for c in (built, cached):
        assert (c.N, c.M, c.E, c.n_layers) == (ref.N, ref.M, ref.E, ref.n_layers)
        for x, y in zip(c.edges(), ref.edges()):
            assert (x == y).all()
    raw = bytearray(files[0].read_bytes())
    raw[len(raw) // 2] ^= 0x40
    files[0].write_bytes(bytes(raw))
    again = q.Code.ira_peg(4096 + 1024, 4096, 0.125, 11, 3, 2, 7)       # checksum fails -> rebuilt, and the file is whole again
    assert all((x == y).all() for x, y in zip(again.edges(), ref.edges()))
    assert files[0].read_bytes() != bytes(raw)
    other = q.Code.ira_peg(4096 + 1024, 4096, 0.125, 11, 3, 2, 8)       # another seed: another file, another code
    assert len(list(tmp_path.iterdir())) == 2 and not all((x == y).all() for x, y in zip(other.edges(), ref.edges()))
