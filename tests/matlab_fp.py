"""The experiment of the reference's fixed-point MATLAB decoder (BPSK_nrldpc_sim_RM_FP.m:23-40), restated for the tests:
BPSK over AWGN at a given Eb/N0, first 2 z bits punctured, received values quantised to 6 bits (floor(r / rmax * maxqr),
clamped to [-32, 31]), layered offset-min-sum with offset 2 and 20 iterations.

Which words were sent matters, because floor() is biased towards negative values: with the ALL-ZERO message (every symbol
+1) this decoder gives FER 0.844 / 0.359 / 0.030 / 0.0019 at 1 / 1.5 / 2 / 2.35 dB on NR_2_6_52 -- the table the reference
publishes for exactly this script and matrix (sim_results.m:8-13: 0.84 / 0.335 / 0.0366 / 0.0027, bit-error rates agree as
well); with random codewords the same decoder is about 8x better (0.04 at 1.5 dB).  The script draws random messages, so
the table was evidently produced with zero messages (or an encoder returning them); the tests pin the all-zero case, the
one that reproduces the published numbers.  The second table (NR_1_1_24, sim_results.m:1-6) is reproduced by neither
choice (0.39 / 0.065 against 0.173 at 1.5 dB) -- its run parameters are not recorded in the tree -- and is kept as data only.
`coset=True` sends a random word x and asks the decoder for H x^ = H x, which has the statistics of random codewords."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PINS = json.load(open(os.path.join(GOLD, "matlab_fp_fer.json")))
RMAX, MAXQR, OFFSET, MAX_ITRS = 3.0, 31, 2.0, 20


def frames(name, ebno_db, F, rng, coset=False):
    p = PINS[name]
    z, kb, nb = p["z"], p["kb"], p["nb_rm"]
    n, k = nb * z, kb * z
    x = rng.integers(0, 2, (F, n)).astype(np.uint8) if coset else np.zeros((F, n), np.uint8)
    sigma = np.sqrt(1.0 / (2.0 * (k / (n - 2 * z)) * 10.0 ** (ebno_db / 10.0)))
    r = (1.0 - 2.0 * x) + sigma * rng.standard_normal((F, n))
    r[:, :2 * z] = 0.0
    rq = np.clip(np.floor(r / RMAX * MAXQR), -(MAXQR + 1), MAXQR)
    return x, rq.astype(np.float32), k


def published(name, ebno_db):
    for row in PINS[name]["rows"]:
        if abs(row[0] - ebno_db) < 1e-9:
            return dict(fer=row[1], errors=row[3], blocks=row[5])
    raise KeyError(ebno_db)


def band(name, ebno_db, F):
    """acceptance band for a frame-error count out of F frames: the published estimate +- 4 sigma of both estimates, with 20 % slack
    for the small differences in the clamps (ours are symmetric: +-31 / +-127 instead of [-32, 31] / [-128, 127])"""
    pub = published(name, ebno_db)
    f = pub["fer"]
    s = 4.0 * np.sqrt(f * (1 - f) / F + f * (1 - f) / pub["blocks"])
    return max(0.0, f * 0.8 - s), min(1.0, f * 1.2 + s)
