#!/usr/bin/env python3
"""Developer loop: the multi-block two-daemon loopback many times, to chase intermittent failures (run on the GPU box).
usage: ecd2_loop.py <binary name under oracle/_ref> <runs> [ENV=VAL ...]"""
import os
import pathlib
import re
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))      # lives in tests/: it runs the reference daemons built under oracle/_ref
from ecd2_loopback import run_loopback  # noqa: E402

binary = os.path.join(ROOT, "oracle", "_ref", sys.argv[1])
if sys.argv[2].startswith("single="):
    # ONE daemon pair, many blocks of different length: every block must end with identical keys, and the number of
    # (code, encoder, decoder) sets the follower has built must be the same after the last block as after ldpc_init
    nblk = int(sys.argv[2].split("=", 1)[1])
    rng = np.random.default_rng(99)
    sizes = []
    while len(sizes) < 2 * nblk:
        x, y = int(rng.integers(6000, 16000)), int(rng.integers(6000, 16000))      # (below ~10 000 bits the reference ends blocks whose sample shows few errors)
        if (x + y) % 32:
            sizes += [x, y]
    a = [rng.integers(0, 2, n).astype(np.uint8) for n in sizes]
    b = [x ^ (rng.random(x.size) < rng.uniform(0.01, 0.05)) for x in a]
    d = pathlib.Path(tempfile.mkdtemp())
    # usage: ecd2_loop.py <binary> single=<blocks> [more -L items, e.g. b4,w50,p2000]; -T 2: the reference's own "terminate this block" verdicts must not end the daemons
    out = run_loopback(binary, d, a, b, extra_args=["-T", "2", "-L", ",".join(["1,g"] + sys.argv[3:])], blocks=[2] * nblk, timeout=40 + 2 * nblk, cmd_gaps=(60.0, 0.12))
    okb = sum(1 for v in out["finals"].values() if v[0] is not None and v[1] is not None and v[0]["nbits"] == v[1]["nbits"] and (v[0]["words"] == v[1]["words"]).all())
    built = [int(x) for x in re.findall(r"code sets built so far: (\d+)", out["b_log"])]
    ready = re.search(r"engine ready, (\d+) code", out["b_log"])
    fell = out["b_log"].count("falling back to cascade")
    ended = out["b_log"].count("Reply mode out of bounds") + out["a_log"].count("Reply mode out of bounds")      # the reference's REPLYMODE_TERMINATE (qber_estim.c:28-36)
    print("one daemon pair, %d blocks of %d..%d bits: %d with identical final keys (%d via the cascade fallback), %d ended by the reference's QBER estimation, code sets at init %s, after the last block %s (min %s), batches %s" % (
        nblk, min(sizes) * 2, max(sizes) * 2, okb, fell, ended, ready.group(1) if ready else "?", built[-1] if built else "?", min(built) if built else "?",
        re.findall(r"decoded a batch of (\d+)", out["b_log"])[:12]), flush=True)
    nblk -= ended
    if okb < nblk:
        for side in "ab":
            open(os.path.join(ROOT, "gpurun_out", "loop_single_%s.log" % side), "w").write(out[side + "_log"])
    sys.exit(0 if okb >= nblk and built and ready and built[-1] == int(ready.group(1)) else 1)
runs = int(sys.argv[2])
env = {"ECD2_LDPC": "1"}
for kv in sys.argv[3:]:
    k, v = kv.split("=", 1)
    env[k] = v
bad = 0
for trial in range(runs):
    rng = np.random.default_rng(17)
    sizes = [3001, 3410, 3107, 3311, 3005, 3502, 3203, 3057] * 2      # block sizes off multiples of 32, see tests/test_ecd2_integration.py:epochs
    a = [rng.integers(0, 2, n).astype(np.uint8) for n in sizes]
    b = [x ^ (rng.random(x.size) < 0.03) for x in a]
    d = pathlib.Path(tempfile.mkdtemp())
    out = run_loopback(binary, d, a, b, env_extra=env, blocks=[2] * 8, timeout=40)
    missing = [hex(k) for k, v in out["finals"].items() if v[0] is None or v[1] is None]
    missing += ["%x differs" % k for k, v in out["finals"].items() if v[0] is not None and v[1] is not None and
                (v[0]["nbits"] != v[1]["nbits"] or not (v[0]["words"] == v[1]["words"]).all())]
    if missing:
        bad += 1
        first = [l for l in out["b_log"].splitlines() if "qber_processReceivedQberEstBits" in l][-1]
        print(trial, "missing", missing, "|", re.sub(r"\s+", " ", first), flush=True)
        for side in "ab":
            open(os.path.join(ROOT, "gpurun_out", "loop_fail_%s.log" % side), "w").write(out[side + "_log"])
print("%s %s: %d of %d runs failed" % (sys.argv[1], env, bad, runs), flush=True)
