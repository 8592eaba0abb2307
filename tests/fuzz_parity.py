#!/usr/bin/env python3
"""Differential fuzzing of the HIP decoder against the CPU oracle (test infrastructure; run on the GPU box).

Random IRA / QC-PEG codes, frame counts, QBER, update rules, schedules, engines, frames-per-lane, message widths, early exit
on/off, syndrome form on/off.  For every case the bit-exact class (min-sum family) must agree with the oracle on hard
decisions, iteration counts and success flags -- and on the posteriors when every frame runs all iterations.
usage: fuzz_parity.py [seconds] [seed]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # tests/ is the only place allowed to use oracle/
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import _qldpc_loader  # noqa: E402
import oracle as O  # noqa: E402

q = _qldpc_loader.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)


def i32(a):
    return np.ascontiguousarray(a).view(np.int32)


def make_code():
    kind = rng.integers(0, 4)
    if kind == 3:      # high rate: check degree above every register bucket, VN degree above 12
        return q.Code.ira(4096, 3850, 0.4, 14, 4, int(rng.integers(1, 99))), "ira(4096,3850,dv14)"
    if kind == 0:
        N = int(rng.integers(8, 40)) * 64
        rate = rng.choice([0.5, 0.7, 0.8, 0.9])
        K = int(N * rate)
        return q.Code.ira(N, K, float(rng.choice([0.1, 0.3])), int(rng.integers(6, 15)), int(rng.integers(2, 5)), int(rng.integers(1, 99))), "ira(%d,%d)" % (N, K)
    if kind == 1:
        n, m, dv, Z = int(rng.integers(6, 20)), int(rng.integers(4, 9)), int(rng.integers(2, 5)), int(rng.choice([37, 53, 101]))
        return q.Code.qc_peg(n, m, min(dv, m), Z, seed=int(rng.integers(1, 99))), "qc_peg(%d,%d,%d,%d)" % (n, m, dv, Z)
    N = int(rng.integers(16, 64)) * 32
    return q.Code.ira_peg(N, int(N * 0.75), depth=2, seed=int(rng.integers(1, 99))), "ira_peg(%d)" % N


def layer_graph(code, var, chk):
    order, _, _ = code.layer_order()
    inv = np.empty(code.M, np.int32)
    inv[order] = np.arange(code.M, dtype=np.int32)
    newc = inv[chk]
    idx = np.argsort(newc, kind="stable")
    return O.Graph.from_edges(code.N, code.M, var[idx], newc[idx]), order


t0, cases, fails = time.time(), 0, 0
t_say = t0
while time.time() - t0 < budget:
    if time.time() - t_say > 60:      # a long run must keep talking (the GPU box takes 7 silent minutes for a hang)
        t_say = time.time()
        print("fuzz: %d cases so far, %d mismatches" % (cases, fails), flush=True)
    try:
        code, cname = make_code()
    except q.QldpcError:
        continue
    var, chk = code.edges()
    og = O.Graph.from_edges(code.N, code.M, var, chk)
    for _ in range(6):
        F = int(rng.choice([1, 3, 8, 9, 65, 130, 257, 600]))
        qber = float(rng.choice([0.005, 0.02, 0.04, 0.07]))
        rule, param = [("MS", 0.0), ("OMS", 0.3), ("NMS", 0.75), ("NMS", 0.8125), ("AMS_MIN", 0.0)][rng.integers(0, 5)]
        sched = str(rng.choice(["flooding", "hlayered"]))
        dtype = str(rng.choice(["f32", "f32", "f16", "i8"]))
        synd = bool(rng.integers(0, 2))
        coset = bool(rng.integers(0, 2))
        n_ite = int(rng.integers(1, 25))
        V = int(rng.choice([0, 1, 2, 4]))
        engine = "auto"
        if dtype == "i8":
            V = 0
            if rule == "AMS_MIN":
                rule, param = "NMS", 0.75
        if dtype == "f16":
            sched = "flooding"
        if dtype == "f32" and sched == "flooding" and F <= 8 and rule != "AMS_MIN":
            engine = str(rng.choice(["auto", "frames", "edges"]))
        if engine == "edges":
            V = 0
        freeze = bool(rng.integers(0, 2)) and dtype != "i8"
        chain = str(rng.choice(["auto", "on", "on", "off"]))        # only acts on layered fp32 runs of the frames engine with 64-frame groups, messages not frozen
        compact = str(rng.choice(["auto", "on", "on", "off"]))      # only acts with early exit on the flooding schedule of the frames engine, messages not frozen
        x = rng.integers(0, 2, (F, code.N)).astype(np.uint8) if coset else np.zeros((F, code.N), np.uint8)
        s = np.stack([og.syndrome(xx)[1] for xx in x]) if coset else None
        y = x ^ (rng.random((F, code.N)) < qber)
        mag = np.float32(q.bsc_llr(qber))
        llr = np.where(y == 1, -mag, mag).astype(np.float32)
        g2, tgt = og, s
        if sched == "hlayered":
            g2, order = layer_graph(code, var, chk)
            tgt = s[:, order] if coset else None
        ref = O.decode(g2, llr, rule, param, n_ite, sched, synd, 1, n_threads=8, target=tgt, msg_fp16=(dtype == "f16"), msg_i8=(dtype == "i8"))
        desc = "%s F=%d q=%.3f %s(%g) %s %s synd=%d coset=%d ite=%d V=%d eng=%s freeze=%d compact=%s chain=%s" % (cname, F, qber, rule, param, sched, dtype, synd, coset, n_ite, V, engine, freeze, compact, chain)
        dec = None
        try:
            dec = q.Decoder(code, code.N, n_ite, rule=rule, rule_param=param, n_frames=F, schedule=sched, enable_syndrome=synd, frames_per_lane=V,
                            engine=engine, freeze_messages=freeze, msg_dtype=dtype, compact=compact, layer_chain=chain)
            dec.load_bits(torch.from_numpy(i32(q.pack_bits(y))).cuda(), torch.full((F,), float(mag), device="cuda"))
            if coset:
                dec.load_syndrome(torch.from_numpy(i32(q.pack_bits(s))).cuda())
            dec.run()
            hard = q.unpack_bits(dec.fetch_packed().cpu().numpy().view(np.uint32), code.N)
            it, ok = dec.fetch_status()
            good = (hard == ref["hard"]).all() and (it.cpu().numpy() == ref["iters"]).all() and (ok.cpu().numpy() == ref["synd_ok"]).all()
            if good and (not synd or freeze) and not (synd and engine != "frames" and dtype == "f32" and F <= 8):
                post = dec.fetch_post().cpu().numpy()
                good = bool((post.view(np.uint32) == ref["post"].view(np.uint32)).all())
                if not good:
                    desc += " [posterior]"
        except q.QldpcError as ex:
            if ex.status == -7:      # QLDPC_EUNSUPPORTED: a combination the library refuses by design (e.g. edge engine, check degree > 64)
                continue
            good = False
            desc += " EXC %s" % ex
        except Exception as ex:      # noqa: BLE001
            good = False
            desc += " EXC %s" % ex
        cases += 1
        if not good:
            fails += 1
            print("MISMATCH:", desc, flush=True)
        del dec
print("fuzz: %d cases, %d mismatches, %.0f s, seed %d" % (cases, fails, time.time() - t0, seed), flush=True)
sys.exit(1 if fails else 0)
