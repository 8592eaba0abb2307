#!/usr/bin/env python3
"""Developer probe: what the planning choices cost or win INSIDE the daemon, where the QBER is a sampled estimate: one daemon pair
(oracle/_ref/ecd2_ldpc_urandom: seeds from /dev/urandom), N blocks of 12 000 - 32 000 bits at QBER 1 - 5 %, for a list of -L settings.
Prints per setting: blocks with identical keys, second rounds, cascade fallbacks, final key bits in all / sifted bits in.
usage: daemon_yield.py <blocks> <-L items> [<-L items> ...]      e.g. daemon_yield.py 96 m0 m20 G1,m0 G1,m20"""
import os
import pathlib
import re
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ecd2_loopback import run_loopback  # noqa: E402

binary = os.path.join(ROOT, "oracle", "_ref", "ecd2_ldpc_urandom")
nblk = int(sys.argv[1])
rng = np.random.default_rng(99)
sizes = []
while len(sizes) < 2 * nblk:
    x, y = int(rng.integers(6000, 16000)), int(rng.integers(6000, 16000))
    if (x + y) % 32:
        sizes += [x, y]
a = [rng.integers(0, 2, n).astype(np.uint8) for n in sizes]
b = [x ^ (rng.random(x.size) < rng.uniform(0.01, 0.05)) for x in a]
for items in sys.argv[2:]:
    d = pathlib.Path(tempfile.mkdtemp())
    out = run_loopback(binary, d, a, b, extra_args=["-T", "2", "-L", "1,g," + items], blocks=[2] * nblk, timeout=60 + 2 * nblk, cmd_gaps=(60.0, 0.12))
    ok = [v for v in out["finals"].values() if v[0] is not None and v[1] is not None and v[0]["nbits"] == v[1]["nbits"] and (v[0]["words"] == v[1]["words"]).all()]
    print("-L 1,g,%-10s %3d / %d blocks with identical keys, %3d second rounds, %3d cascade fallbacks, %2d ended by the reference; final key %7d bits of %d sifted (%.4f)" % (
        items, len(ok), nblk, out["b_log"].count("asking for the"), out["b_log"].count("falling back to cascade"), out["b_log"].count("Reply mode out of bounds") + out["a_log"].count("Reply mode out of bounds"),
        sum(v[0]["nbits"] for v in ok), sum(sizes), sum(v[0]["nbits"] for v in ok) / sum(sizes)), flush=True)
