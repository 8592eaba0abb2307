"""Two-process FIFO loopback of the reference's ecd2 daemon (SURVEY.md section 4's recipe).

Alice (-s a2b -r b2a) and Bob (-s b2a -r a2b) talk over named pipes without transferd (it forwards EC
packets unchanged, remotecrypto/transferd.c:766-768).  Inputs are synthetic stream-3 epoch files
(packetheaders/pkt_header_3.h:4-12: {tag 3, epoch, length in bits, bitsperentry 1} + MSB-first words);
outputs are the stream-7 final key files of both sides.  Test infrastructure only.
"""
import errno
import os
import struct
import subprocess
import time

import numpy as np


def write_stream3(path, epoch, bits):
    n = len(bits)
    pad = (-n) % 32
    b = np.concatenate([np.asarray(bits, np.uint8), np.zeros(pad, np.uint8)])
    words = np.packbits(b).view(">u4").astype("<u4")
    with open(path, "wb") as f:
        f.write(struct.pack("<iIIi", 3, epoch, n, 1))
        f.write(words.tobytes())


def read_stream7(path):
    raw = open(path, "rb").read()
    tag, epoch, nepochs, nbits = struct.unpack("<iIIi", raw[:16])
    words = np.frombuffer(raw[16:], "<u4")
    return dict(tag=tag, epoch=epoch, nepochs=nepochs, nbits=nbits, words=words)


def run_loopback(binary, workdir, alice_bits, bob_bits, epoch0=0xb0b80000, env_extra=None, timeout=180, blocks=None, cmd_gaps=(60.0, 0.25), extra_args=None,
                 extra_args_a=None, extra_args_b=None):
    """alice_bits / bob_bits: lists of per-epoch 0/1 arrays. Returns dict with both final keys + logs.
    blocks: epochs per command (default: one command = one block of all epochs); with several commands written at once
    several blocks are in flight, and out["finals"] maps each block's first epoch to its (alice, bob) stream-7 files."""
    d = str(workdir)
    for side in "ab":
        for sub in ("raw", "final"):
            os.makedirs(os.path.join(d, side, sub), exist_ok=True)
    fifos = ["a_cmd", "b_cmd", "a2b", "b2a", "a_q", "b_q"]
    for f in fifos:
        p = os.path.join(d, f)
        if not os.path.exists(p):
            os.mkfifo(p)
    for i, (a, b) in enumerate(zip(alice_bits, bob_bits)):
        write_stream3(os.path.join(d, "a", "raw", "%08x" % (epoch0 + i)), epoch0 + i, a)
        write_stream3(os.path.join(d, "b", "raw", "%08x" % (epoch0 + i)), epoch0 + i, b)
    env = dict(os.environ)
    env.update(env_extra or {})

    def daemon(side, send, recv):
        args = [binary, "-c", side + "_cmd", "-s", send, "-r", recv, "-d", side + "/raw", "-f", side + "/final",
                "-l", side + "/notify", "-q", side + "/resp", "-Q", side + "_q", "-V", "5"] + list(extra_args or []) + \
            list((extra_args_a if side == "a" else extra_args_b) or [])
        log = open(os.path.join(d, side + ".log"), "w")
        return subprocess.Popen(args, cwd=d, env=env, stdout=log, stderr=subprocess.STDOUT), log

    pa, la = daemon("a", "a2b", "b2a")
    pb, lb = daemon("b", "b2a", "a2b")
    try:
        time.sleep(0.5)
        blocks_ = blocks or [len(alice_bits)]
        starts, e = [], epoch0
        for nb in blocks_:
            starts.append(e)
            e += nb
        # a daemon that refused its options never opens its command pipe: a blocking open() here would wait for ever
        fd, tw = None, time.time()
        while fd is None:
            try:
                fd = os.open(os.path.join(d, "a_cmd"), os.O_WRONLY | os.O_NONBLOCK)
            except OSError as e:
                if e.errno != errno.ENXIO or pa.poll() is not None or time.time() - tw > 30:
                    break
                time.sleep(0.05)
        if fd is not None:
            os.set_blocking(fd, True)
        with (os.fdopen(fd, "w") if fd is not None else open(os.devnull, "w")) as f:
            # one line per write, spaced out: the daemon parses ONE command per wake-up of its command pipe, so lines that
            # pile up while a handler is busy (the first block pays for GPU start-up and code construction) would be lost
            for i, (st, nb) in enumerate(zip(starts, blocks_)):
                f.write("0x%08x %d\n" % (st, nb))
                f.flush()
                if len(blocks_) > 1 and i == 0:      # first block: wait until it is through (GPU start-up, code construction), at most cmd_gaps[0]
                    tw = time.time()
                    while time.time() - tw < cmd_gaps[0] and not all(os.path.exists(x) for x in
                                                                     (os.path.join(d, "a", "final", "%08x" % st), os.path.join(d, "b", "final", "%08x" % st))):
                        time.sleep(0.05)
                elif len(blocks_) > 1:
                    time.sleep(cmd_gaps[1])
        fa = os.path.join(d, "a", "final", "%08x" % epoch0)
        fb = os.path.join(d, "b", "final", "%08x" % epoch0)
        finals = [(os.path.join(d, "a", "final", "%08x" % st), os.path.join(d, "b", "final", "%08x" % st)) for st in starts]
        t0 = time.time()
        while time.time() - t0 < timeout:
            if all(os.path.exists(x) and os.path.getsize(x) >= 16 for pair in finals for x in pair):      # 16 = a header with 0 bits
                time.sleep(0.3)
                break
            if pa.poll() is not None or pb.poll() is not None:
                break
            time.sleep(0.1)
    finally:
        for p in (pa, pb):
            if p.poll() is None:
                p.terminate()
        for p in (pa, pb):
            try:
                p.wait(10)
            except subprocess.TimeoutExpired:
                p.kill()
        la.close()
        lb.close()
    out = dict(a_log=open(os.path.join(d, "a.log")).read(), b_log=open(os.path.join(d, "b.log")).read(),
               elapsed=time.time() - t0)
    out["a_final"] = read_stream7(fa) if os.path.exists(fa) else None
    out["b_final"] = read_stream7(fb) if os.path.exists(fb) else None
    out["finals"] = {st: (read_stream7(x) if os.path.exists(x) else None, read_stream7(y) if os.path.exists(y) else None)
                     for st, (x, y) in zip(starts, finals)}
    for side in "ab":
        p = os.path.join(d, side, "notify")
        out[side + "_notify"] = open(p).read() if os.path.exists(p) else ""
    return out
