#!/usr/bin/env python3
"""BASELINE config 3 as a measurement: a stream of sifted-key epochs with per-epoch QBER ~ U[0.5 %, 6 %] (seed 42),
rate picked per epoch from the {0.5, 0.7, 0.8, 0.9} table (f = 1.4), blocks of one plan batched per launch.
Host buffers in and out (the daemon's situation): the figure includes PCIe copies, CRC and host packing."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _qldpc_loader  # noqa: E402

q = _qldpc_loader.load()

EPOCHS = int(os.environ.get("EPOCHS", "512"))
KEY_BITS = int(os.environ.get("KEY_BITS", "52429"))
BATCH = int(os.environ.get("BATCH", "64"))
rng = np.random.default_rng(int(os.environ.get("SEED", "42")))
qbers = rng.uniform(0.005, 0.06, EPOCHS).astype(np.float32)
alice = rng.integers(0, 2, (EPOCHS, KEY_BITS)).astype(np.uint8)
bob = alice ^ (rng.random((EPOCHS, KEY_BITS)) < qbers[:, None])
aw, bw = q.pack_bits(alice), q.pack_bits(bob)

SCHED = os.environ.get("SCHEDULE", "flooding")
GAP = float(os.environ["RATE_GAP"]) if os.environ.get("RATE_GAP") else None
ra, rb = q.Recon(max_blocks=BATCH, rate_gap=GAP), q.Recon(max_blocks=BATCH, schedule=SCHED, rate_gap=GAP)
plans = [ra.plan(KEY_BITS, p) for p in qbers]
groups = {}
for i, m in enumerate(plans):
    groups.setdefault((m.rate_index, m.code_k, m.code_m), []).append(i)

# Alice's side: all epochs in one call (grouped by plan inside, launches of up to BATCH blocks); a first call warms the
# per-plan code / encoder cache, the second one is timed
ra.encode_blocks([aw[i] for i in range(EPOCHS)], [KEY_BITS] * EPOCHS, qbers)
t0 = time.perf_counter()
msgs, pars = ra.encode_blocks([aw[i] for i in range(EPOCHS)], [KEY_BITS] * EPOCHS, qbers)
t_enc = time.perf_counter() - t0

# warm the per-plan decoder cache
for key, idx in groups.items():
    j = idx[:1]
    rb.decode_batch(bw[j], KEY_BITS, qbers[j], [msgs[j[0]]], [pars[j[0]]])

t0 = time.perf_counter()
ok = np.zeros(EPOCHS, bool)
iters = np.zeros(EPOCHS, int)
for key, idx in groups.items():
    for lo in range(0, len(idx), BATCH):
        j = idx[lo:lo + BATCH]
        st, fixed, co, it = rb.decode_batch(bw[j], KEY_BITS, qbers[j], [msgs[k] for k in j], [pars[k] for k in j])
        good = st == 0
        ok[j] = good
        iters[j] = it
        assert (fixed[good] == aw[j][good]).all()
# second round for the blocks that did not decode (the ecd2 plugin's verdict 2): the withheld parity bits, decode again at the mother rate
first_fail = int((~ok).sum())
leaked = np.array([q.Recon.leaked_bits(m) for m in msgs])
if os.environ.get("SECOND_ROUND", "1") != "0":
    for i in np.nonzero(~ok)[0]:
        if msgs[i].n_punct == 0:
            continue
        m2, p2 = ra.encode_planned(aw[i], KEY_BITS, msgs[i], 0)
        st, fixed, co, it = rb.decode_batch(bw[[i]], KEY_BITS, qbers[[i]], [m2], [p2])
        leaked[i] = q.Recon.leaked_bits(m2)
        if st[0] == 0:
            assert (fixed[0] == aw[i]).all()
            ok[i] = True
dt = time.perf_counter() - t0
leak = int(leaked[ok].sum())
print("  first round: %d of %d blocks failed; after the second round: %d" % (first_fail, EPOCHS, int((~ok).sum())))
print("config 3 stream: %d epochs x %d bits, QBER U[0.5%%, 6%%], batch <= %d, Bob decodes %s" % (EPOCHS, KEY_BITS, BATCH, SCHED))
for key, idx in sorted(groups.items()):
    print("  rate %.1f (K %d, M %d): %3d epochs, %3d reconciled, mean iterations %.1f" % (ra.rates[key[0]], key[1], key[2], len(idx), int(ok[idx].sum()), iters[idx].mean()))
print("  Bob decode: %.1f ms total -> %.1f Mbit/s of sifted key, FER %.3f, leaked fraction %.3f | Alice encode %.1f ms (%.2f ms/epoch)" % (
    dt * 1e3, ok.sum() * KEY_BITS / dt / 1e6, 1 - ok.mean(), leak / max(1, ok.sum() * KEY_BITS), t_enc * 1e3, t_enc / EPOCHS * 1e3))
