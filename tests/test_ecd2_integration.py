"""Integration: the LDPC handlers inside the reference's own ecd2 daemon.

oracle/_ref/ecd2_cascade and oracle/_ref/ecd2_ldpc are built by oracle/build_ref_ecd2.sh from the
reference's sources where they lie (only in the build container; the binaries travel to the GPU box,
the sources do not).  Skipped when the binaries are absent.
"""
import os
import subprocess

import numpy as np
import pytest

from ecd2_loopback import run_loopback

import json
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")

# The test daemons are built with the reference's own -DFIXED_RNG_SEED (oracle/build_ref_ecd2.sh; subcomponents/rnd.c:145-166): the QBER
# sample positions, the estimate, the plan and the privacy-amplification hash of a run are reproducible, so the tests pin EXACT key
# lengths.  The expected values live in tests/golden/ecd2_expected.json (data: observed once on an MI355X box with ECD2_RECORD=1, which
# writes gpurun_out/ecd2_observed.json instead of asserting); a value that is missing there fails the test.
GOLD_PATH = os.path.join(ROOT, "tests", "golden", "ecd2_expected.json")
GOLD = json.load(open(GOLD_PATH)) if os.path.exists(GOLD_PATH) else {}


@pytest.fixture(scope="module")
def warm_code_cache(q):
    """the daemons build 32 PEG mother codes in ldpc_init; one preloading session fills QLDPC_CODE_CACHE (tests/conftest.py) first, so that
    every daemon of this module reads them back in milliseconds instead of two daemons growing them side by side in every test"""
    q.Recon(preload=True)


def expect(key, value):
    if os.environ.get("ECD2_RECORD"):
        path = os.path.join(ROOT, "gpurun_out", "ecd2_observed.json")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        seen = json.load(open(path)) if os.path.exists(path) else {}
        seen[key] = value
        json.dump(seen, open(path, "w"), indent=1, sort_keys=True)
        return
    assert key in GOLD, "no pinned value for %r (record with ECD2_RECORD=1 on the GPU box, commit tests/golden/ecd2_expected.json)" % key
    assert GOLD[key] == value, (key, GOLD[key], value)


def pa_account(log):
    """privAmp_doPrivAmp's printout (priv_amp.c:175-181) of the LAST block in a daemon log: workbits, corrected errors, sneakloss,
    leakageBits - correctedErrors, final key bits"""
    g = lambda pat: int(re.findall(pat, log)[-1])
    return dict(workbits=g(r" workbits: (\d+)"), corrected=g(r"corrected errors: (\d+)"), sneakloss=g(r" sneakloss: (-?\d+)"),
                leak_minus_credit=g(r" leakageBits: (-?\d+)"), final=g(r" finakeybits: (-?\d+)"))


def epochs(seed, n_epochs, bits_per_epoch, qber):
    """Per-epoch sifted keys.  Block sizes are kept off multiples of 32 bits: the reference's QBER sampling loops accept the
    position `initialBits` itself (`if (bipo > processBlock->initialBits) continue;`, subcomponents/comms.c:80 and
    qber_estim.c:190 -- should be >=); when initialBits is a multiple of 32 that position's marker word is the first word of
    the uninitialised permuteIndex array (processblock_mgmt.c:146-156 clears only `newindex` words), so Alice and Bob may or
    may not skip it depending on heap junk and their sample positions desynchronise (seen as 20-40 % "QBER" on later blocks
    of a daemon that has reused heap memory).  Not ours to fix; the tests steer around it."""
    assert (n_epochs * bits_per_epoch) % 32 != 0
    rng = np.random.default_rng(seed)
    a = [rng.integers(0, 2, bits_per_epoch).astype(np.uint8) for _ in range(n_epochs)]
    b = [x ^ (rng.random(bits_per_epoch) < qber) for x in a]
    return a, b


# Tests that assert "this block went through LDPC" plan with two sigma of margin on the sampled QBER (`-L m20`): each daemon run draws its own
# sample, and at m0 a few per cent of short blocks get a plan the true error rate exceeds -- they fall back to cascade, correctly, but a
# test that greps for "decoded" would then fail one run in twenty.
LDPC = "1,m20"


def need(binary):
    p = os.path.join(REF_DIR, binary)
    if not os.path.exists(p):
        pytest.skip("%s not built (oracle/build_ref_ecd2.sh needs /root/reference)" % binary)
    return p


def test_plugin_compiles_against_the_reference_headers():
    ref = "/root/reference/errorcorrection"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present on this machine")
    src = os.path.join(ROOT, "qcrypto-ldpc_amd", "host", "ldpc_reconcile.c")
    r = subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Werror", "-c", "-I" + ref, "-I" + os.path.join(ROOT, "include"),
                        "-o", "/dev/null", src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_reference_cascade_loopback_is_the_integration_oracle(tmp_path):
    """pristine reference daemon (fixed RNG seed): both sides end with the same stream-7 key, one attempt, exact length."""
    binary = need("ecd2_cascade")
    a, b = epochs(1, 4, 4001, 0.02)
    out = run_loopback(binary, tmp_path, a, b, timeout=60)
    assert out["a_final"] is not None and out["b_final"] is not None, out["a_log"][-2000:] + out["b_log"][-2000:]
    assert out["a_final"]["tag"] == 7 and out["a_final"]["nbits"] == out["b_final"]["nbits"] == 10486      # reproducible: -DFIXED_RNG_SEED
    assert (out["a_final"]["words"] == out["b_final"]["words"]).all()
    acc = pa_account(out["b_log"])
    assert acc["final"] == 10486 == acc["workbits"] - acc["leak_minus_credit"] - acc["sneakloss"]      # priv_amp.c:166 with cascade's credit


@pytest.mark.gpu
@pytest.mark.parametrize("n_epochs,bits,qber", [(4, 4001, 0.02), (4, 15001, 0.02), (2, 9000, 0.03)])
def test_ldpc_handlers_inside_ecd2(warm_code_cache, tmp_path, n_epochs, bits, qber):
    """`-L 1` (ecd2's new option): Bob (QBER follower) picks ALG_LDPC_CONTINUE_ROLES; one parity packet + one verdict instead of
    ~55 cascade packets each way; both daemons write identical final keys."""
    binary = need("ecd2_ldpc")
    a, b = epochs(2, n_epochs, bits, qber)
    out = run_loopback(binary, tmp_path, a, b, extra_args=["-L", LDPC])
    assert out["a_final"] is not None and out["b_final"] is not None, out["a_log"][-3000:] + "\n----\n" + out["b_log"][-3000:]
    assert out["a_final"]["nbits"] == out["b_final"]["nbits"] > 0, out["b_log"][-2500:]
    assert (out["a_final"]["words"] == out["b_final"]["words"]).all()
    expect("ldpc_handlers[%d,%d,%g].nbits" % (n_epochs, bits, qber), int(out["a_final"]["nbits"]))
    # the leakage account of an LDPC block: every disclosed parity bit + the 32 CRC bits count in full.  privAmp_doPrivAmp's credit of one
    # bit per corrected error (priv_amp.c:86-90,166) is cascade's -- a parity bit made redundant by its binary search -- and the handlers
    # cancel it on both sides: final key = workbits - (disclosed + 32) - sneakloss, the same on Alice's and Bob's side
    leaked = int(re.search(r"decoded \d+ key bits in \d+ iterations, \d+ errors corrected, (\d+) bits leaked", out["b_log"]).group(1))
    disclosed = int(re.search(r"(\d+) bits disclosed", out["a_log"]).group(1))
    assert leaked == disclosed
    for log in (out["a_log"], out["b_log"]):
        acc = pa_account(log)
        assert acc["final"] == acc["workbits"] - leaked - acc["sneakloss"] == out["a_final"]["nbits"], acc
        assert acc["leak_minus_credit"] == leaked
    # the exchange really was LDPC: subtype 9 / 10 in the logs, no cascade subtypes 4..7
    assert "ldpc: epoch b0b80000: sent parity" in out["a_log"] and "ldpc: epoch b0b80000: decoded" in out["b_log"]
    for cascade_subtype in (4, 5, 6, 7):
        assert "Prep to send pkt subtype %d\n" % cascade_subtype not in out["a_log"] + out["b_log"]
    errs = sum(int((x != y).sum()) for x, y in zip(a, b))
    m = re.search(r"(\d+) errors corrected", out["b_log"])
    assert m and 0 < int(m.group(1)) <= errs                   # sample bits revealed in QBER estimation are not counted
    note = out["b_notify"] + out["a_notify"]
    assert "final bit number" in note


@pytest.mark.gpu
def test_full_size_block_ldpc_plus_gpu_privacy_amplification(warm_code_cache, tmp_path):
    """4 epochs x 15 000 bits (the block of SURVEY.md section 4 / 6): LDPC reconciliation and the PA hash both on the GPU.
    Alice hashes on her GPU, Bob on his: equal stream-7 files prove the GPU hash equals itself across processes AND,
    through the CPU run below with the SAME daemons, that it is the reference's hash."""
    binary = need("ecd2_ldpc")
    a, b = epochs(5, 4, 15001, 0.02)
    gpu = run_loopback(binary, tmp_path / "gpu", a, b, env_extra={"ECD2_LDPC": "1", "ECD2_GPU_PA": "1"}, extra_args=["-L", "m20"])
    assert gpu["a_final"] is not None and (gpu["a_final"]["words"] == gpu["b_final"]["words"]).all()
    expect("full_size_gpu_pa.nbits", int(gpu["a_final"]["nbits"]))
    # Bob on the GPU hash, Alice on the reference's CPU loop: the two final keys must still be identical
    d = tmp_path / "mixed"
    import subprocess, time, os
    from ecd2_loopback import write_stream3, read_stream7
    os.makedirs(d / "a" / "raw"); os.makedirs(d / "a" / "final"); os.makedirs(d / "b" / "raw"); os.makedirs(d / "b" / "final")
    for f in ("a_cmd", "b_cmd", "a2b", "b2a", "a_q", "b_q"):
        os.mkfifo(d / f)
    for i, (x, y) in enumerate(zip(a, b)):
        write_stream3(d / "a" / "raw" / ("%08x" % (0xb0b80000 + i)), 0xb0b80000 + i, x)
        write_stream3(d / "b" / "raw" / ("%08x" % (0xb0b80000 + i)), 0xb0b80000 + i, y)

    def start(side, send, recv, env):
        e = dict(os.environ); e.update(env)
        return subprocess.Popen([binary, "-c", side + "_cmd", "-s", send, "-r", recv, "-d", side + "/raw", "-f", side + "/final", "-l", side + "/notify",
                                 "-q", side + "/resp", "-Q", side + "_q", "-V", "5", "-L", "m20"], cwd=d, env=e, stdout=open(d / (side + ".log"), "w"), stderr=subprocess.STDOUT)
    pa = start("a", "a2b", "b2a", {"ECD2_LDPC": "1"})                          # CPU privacy amplification (reference loop)
    pb = start("b", "b2a", "a2b", {"ECD2_LDPC": "1", "ECD2_GPU_PA": "1"})     # GPU privacy amplification
    try:
        time.sleep(0.5)
        with open(d / "a_cmd", "w") as f:
            f.write("0xb0b80000 4\n")
        t0 = time.time()
        fa, fb = d / "a" / "final" / "b0b80000", d / "b" / "final" / "b0b80000"
        while time.time() - t0 < 120 and not (fa.exists() and fb.exists() and fa.stat().st_size > 16 and fb.stat().st_size > 16):
            time.sleep(0.2)
        time.sleep(0.3)
    finally:
        for p in (pa, pb):
            p.terminate()
        for p in (pa, pb):
            p.wait(10)
    ka, kb = read_stream7(fa), read_stream7(fb)
    assert ka["nbits"] == kb["nbits"] > 20000 and (ka["words"] == kb["words"]).all()
    # same epochs, same fixed seed: the mixed CPU / GPU run must give the very key the all-GPU run gave
    assert ka["nbits"] == gpu["a_final"]["nbits"] and (ka["words"] == gpu["a_final"]["words"]).all()


@pytest.mark.gpu
def test_ldpc_and_cascade_daemons_agree_on_key_length_order(warm_code_cache, tmp_path):
    """same epochs through both daemons: both reconcile; LDPC leaks M+32 bits, cascade its parity count."""
    a, b = epochs(3, 4, 6001, 0.02)
    o1 = run_loopback(need("ecd2_cascade"), tmp_path / "c", a, b)
    o2 = run_loopback(need("ecd2_ldpc"), tmp_path / "l", a, b, env_extra={"ECD2_LDPC": "1"}, extra_args=["-L", "m20"])
    for o in (o1, o2):
        assert o["a_final"] is not None and (o["a_final"]["words"] == o["b_final"]["words"]).all()
    assert o1["a_final"]["nbits"] > 0 and o2["a_final"]["nbits"] > 0
    expect("cascade_vs_ldpc.cascade_nbits", int(o1["a_final"]["nbits"]))
    expect("cascade_vs_ldpc.ldpc_nbits", int(o2["a_final"]["nbits"]))


@pytest.mark.gpu
def test_decode_failure_falls_back_to_cascade(warm_code_cache, tmp_path):
    """SURVEY.md section 8(f)1 "fallback to cascade on decode failure": Alice's parity packet is corrupted on purpose
    (ECD2_LDPC_FAULT flips disclosed parity bits), Bob finds no verified codeword, says so in the verdict, and both
    daemons finish the block with the reference's own cascade exchange. The final keys are identical, and the wasted
    parity + CRC bits stay in leakageBits."""
    binary = need("ecd2_ldpc")
    a, b = epochs(7, 4, 6001, 0.02)
    clean = run_loopback(binary, tmp_path / "clean", a, b, env_extra={"ECD2_LDPC": "1"}, extra_args=["-L", "m20"])
    out = run_loopback(binary, tmp_path / "fault", a, b, extra_args=["-L", "1,x600,r0"])      # r0: no second round, straight to cascade
    assert out["a_final"] is not None and out["b_final"] is not None, out["a_log"][-3000:] + "\n----\n" + out["b_log"][-3000:]
    assert "no verified codeword" in out["b_log"]
    assert "falling back to cascade as EC follower" in out["b_log"] and "falling back to cascade as EC initiator" in out["a_log"]
    assert "Prep to send pkt subtype 4\n" in out["a_log"]            # the cascade parity list really went out
    assert out["a_final"]["nbits"] == out["b_final"]["nbits"] > 0
    assert (out["a_final"]["words"] == out["b_final"]["words"]).all()
    # the bits LDPC disclosed stay in the leakage account: what PA subtracts (printed by privAmp_doPrivAmp as leakageBits -
    # correctedErrors) exceeds M + 32 by cascade's own parities
    expect("fallback.clean_nbits", int(clean["a_final"]["nbits"]))
    expect("fallback.fault_nbits", int(out["a_final"]["nbits"]))
    assert out["a_final"]["nbits"] < clean["a_final"]["nbits"]      # same sample, same estimate (fixed seed): the failed attempt's bits are what is missing
    disclosed = int(re.search(r"sent parity in \d+ packet\(s\), \d+ key bits, rate index \d+, K \d+, M \d+, \d+ punctured, (\d+) bits disclosed", out["a_log"]).group(1))
    for log in (out["a_log"], out["b_log"]):
        corr = int(re.findall(r"corrected errors: (\d+)", log)[-1])
        leak = int(re.findall(r"leakageBits: (-?\d+)", log)[-1])
        assert leak + corr > disclosed + 100, (leak, corr, disclosed)
    assert clean["a_final"] is not None and clean["a_final"]["nbits"] > 0
    # dropping instead of falling back is still available
    drop = run_loopback(binary, tmp_path / "drop", a, b, extra_args=["-L", "1,x600,r0,f0"], timeout=12)
    assert drop["a_final"] is None and drop["b_final"] is None


@pytest.mark.gpu
def test_batched_ingest_several_blocks_in_one_decode_call(warm_code_cache, tmp_path):
    """eight commands are written at once, so eight blocks are in flight; with ECD2_LDPC_BATCH=4 Bob
    queues the parity packets and ldpc_tick() decodes them four at a time in one qldpc_recon_decode_blocks call (blocks of
    different length share a plan).  Every block ends with identical keys on both sides, equal to what the unbatched
    daemons produce for the same epochs."""
    binary = need("ecd2_ldpc")
    rng = np.random.default_rng(17)
    # 16 epochs, 2 per block -> 8 blocks of 6 260 .. 6 507 bits, none a multiple of 32 (see epochs()); QBER 3 % so that the
    # reference's first 411-bit sample practically never shows so few errors that it asks for more test bits than the block
    # has and terminates it (at 2 % that happens to 1 block in 100)
    sizes = [3001, 3410, 3107, 3311, 3005, 3502, 3203, 3057] * 2
    assert all((sizes[i] + sizes[i + 1]) % 32 for i in range(0, 16, 2))
    a = [rng.integers(0, 2, n).astype(np.uint8) for n in sizes]
    b = [x ^ (rng.random(x.size) < 0.03) for x in a]
    one = run_loopback(binary, tmp_path / "one", a, b, extra_args=["-L", LDPC + ",g"], blocks=[2] * 8, timeout=90)
    bat = run_loopback(binary, tmp_path / "batch", a, b, extra_args=["-L", LDPC + ",g,b4,w1500"], blocks=[2] * 8, timeout=90)
    # the mother codes were all built in ldpc_init: eight blocks of different length later, still the same 32 entries
    assert "ldpc: engine ready" in bat["b_log"]
    for name, o in (("one", one), ("batch", bat)):
        missing = [hex(st) for st, (x, y) in o["finals"].items() if x is None or y is None]
        if missing:      # keep the daemons' logs where gpurun brings them back
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            for side in "ab":
                open(os.path.join(ROOT, "gpurun_out", "ecd2_stuck_%s_%s.log" % (name, side)), "w").write(o[side + "_log"])
        assert not missing and o["elapsed"] < 60, (name, missing, o["elapsed"], o["a_log"][-1500:], o["b_log"][-1500:])
    assert len(bat["finals"]) == 8
    for st, (fa, fb) in bat["finals"].items():
        assert fa is not None and fb is not None, bat["b_log"][-3000:]
        assert fa["nbits"] == fb["nbits"] > 0 and (fa["words"] == fb["words"]).all()
        ref_a, ref_b = one["finals"][st]
        # fixed seed: the batched daemons must write the very keys the unbatched ones write, block for block
        assert ref_a is not None and ref_b is not None and (ref_a["words"] == ref_b["words"]).all()
        assert ref_a["nbits"] == fa["nbits"] and (ref_a["words"] == fa["words"]).all()
    expect("batched_ingest.nbits", [int(bat["finals"][st][0]["nbits"]) for st in sorted(bat["finals"])])
    batches = [int(x) for x in re.findall(r"decoded a batch of (\d+) blocks in one call", bat["b_log"])]
    if sum(batches) != 8:      # keep the daemons' logs where gpurun brings them back
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        for side in "ab":
            open(os.path.join(ROOT, "gpurun_out", "ecd2_batch_%s.log" % side), "w").write(bat[side + "_log"])
    assert sum(batches) == 8 and max(batches) >= 2, batches
    assert "decoded a batch" not in one["b_log"]


@pytest.mark.gpu
def test_block_above_65536_bits_with_fragmented_parity(warm_code_cache, tmp_path):
    """SURVEY.md 8f #4 second half / VERDICT r1 #4a, #8: two 60 001-bit epochs are loaded as ONE block of 120 002 bits (the LDPC build
    raises MAX_BITS_PER_PROCESSBLOCK, processblock_mgmt.c:94-95; cascade with its unsigned short indices could not take it), at
    QBER 4.5 % the plan discloses 42 000 - 50 000 parity bits; `-L p3000` makes every parity packet at most 3 000 bytes so the
    payload travels in several fragments, which Bob reassembles before decoding.  Identical final keys on both sides.  (At 6 % the
    daemon's privacy amplification leaves no key at all when the sampled error rate comes out high: seen once in ten runs.)"""
    binary = need("ecd2_ldpc")
    a, b = epochs(11, 2, 60001, 0.045)
    out = run_loopback(binary, tmp_path, a, b, extra_args=["-L", LDPC + ",g,p3000"], timeout=150)
    assert out["a_final"] is not None and out["b_final"] is not None, out["a_log"][-3000:] + "\n----\n" + out["b_log"][-3000:]
    m = re.search(r"sent parity in (\d+) packet\(s\), (\d+) key bits", out["a_log"])
    assert m and int(m.group(1)) >= 2 and int(m.group(2)) > 100000, out["a_log"][-2000:]      # 42 000 .. 50 000 disclosed bits at <= 3 000 bytes a packet
    assert "decoded" in out["b_log"]
    assert out["a_final"]["nbits"] == out["b_final"]["nbits"] > 10000
    assert (out["a_final"]["words"] == out["b_final"]["words"]).all()
    expect("block_120002.nbits", int(out["a_final"]["nbits"]))


@pytest.mark.gpu
def test_sample_without_errors_and_repeated_parity_packets(warm_code_cache, tmp_path):
    """(1) identical keys -> the QBER sample has no error, localError = 0 (qber_estim.c:26); the plan clamps it instead
    of refusing, the block reconciles (nothing to correct) and yields a key.  (2) a parity packet that arrives twice (`-L d1`
    makes Alice send every parity packet twice) must not queue the block twice: with the batched ingest the second copy used to
    decode a block that privacy amplification had already freed."""
    binary = need("ecd2_ldpc")
    rng = np.random.default_rng(23)
    a = [rng.integers(0, 2, 5001).astype(np.uint8) for _ in range(4)]
    out = run_loopback(binary, tmp_path / "clean", a, [x.copy() for x in a], extra_args=["-L", "1"], timeout=90)
    assert out["a_final"] is not None and out["b_final"] is not None, out["a_log"][-2000:] + "\n----\n" + out["b_log"][-2000:]
    assert out["a_final"]["nbits"] == out["b_final"]["nbits"] > 10000 and (out["a_final"]["words"] == out["b_final"]["words"]).all()
    assert "0 errors corrected" in out["b_log"]
    expect("error_free_sample.nbits", int(out["a_final"]["nbits"]))
    a2, b2 = epochs(29, 4, 5003, 0.02)
    dup = run_loopback(binary, tmp_path / "dup", a2, b2, extra_args=["-L", LDPC + ",b2,w200"], extra_args_a=["-L", LDPC + ",b2,w200,d1"], timeout=90)
    assert dup["a_final"] is not None and dup["b_final"] is not None, dup["a_log"][-2000:] + "\n----\n" + dup["b_log"][-2000:]
    assert (dup["a_final"]["words"] == dup["b_final"]["words"]).all()
    expect("repeated_packets.nbits", int(dup["a_final"]["nbits"]))
    assert dup["b_log"].count("decoded 1") + dup["b_log"].count(": decoded ") >= 1 and "Segmentation" not in dup["b_log"]


@pytest.mark.gpu
def test_refused_parity_header_gets_a_failed_verdict_and_the_block_falls_back(warm_code_cache, tmp_path):
    """a parity header that is not what the follower's own rate table gives for the block (`-L y1`: Alice claims the next
    rate index with the dimensions of the real one) is refused BEFORE anything is allocated for its payload, answered with a failed
    verdict -- the initiator is not left waiting -- and the block is reconciled by cascade in the same daemons."""
    binary = need("ecd2_ldpc")
    a, b = epochs(31, 4, 5003, 0.02)
    out = run_loopback(binary, tmp_path, a, b, extra_args=["-L", "1"], extra_args_a=["-L", "1,y1"], timeout=90)
    assert out["a_final"] is not None and out["b_final"] is not None, out["a_log"][-2000:] + "\n----\n" + out["b_log"][-2000:]
    assert "parity header refused" in out["b_log"] and "falling back to cascade as EC follower" in out["b_log"]
    assert "falling back to cascade as EC initiator" in out["a_log"]
    assert out["a_final"]["nbits"] == out["b_final"]["nbits"] > 0 and (out["a_final"]["words"] == out["b_final"]["words"]).all()
    expect("refused_header.nbits", int(out["a_final"]["nbits"]))


@pytest.mark.gpu
def test_planning_margin_discloses_more_and_still_reconciles(warm_code_cache, tmp_path):
    """`-L m<n>` on the initiator: the code is planned for q + n/10 sigma of the sampled estimate (short blocks: the estimate is noisy and
    the plan sits 0.02 - 0.05 from capacity).  More parity bits go out, the header carries the plan, the follower needs no option."""
    binary = need("ecd2_ldpc")
    a, b = epochs(5, 2, 6001, 0.03)
    base = run_loopback(binary, tmp_path / "m10", a, b, extra_args=["-L", "1"], extra_args_a=["-L", "1,m10"], timeout=90)
    marg = run_loopback(binary, tmp_path / "m70", a, b, extra_args=["-L", "1"], extra_args_a=["-L", "1,m70"], timeout=90)
    d = []
    for out in (base, marg):
        assert out["a_final"] is not None and out["b_final"] is not None, out["a_log"][-2000:] + "\n----\n" + out["b_log"][-2000:]
        assert out["a_final"]["nbits"] == out["b_final"]["nbits"] and (out["a_final"]["words"] == out["b_final"]["words"]).all()
        d.append(int(re.search(r"(\d+) bits disclosed", out["a_log"]).group(1)))
    # same sample and same estimate in both runs (fixed seed): the margin alone moves the plan
    assert base["a_final"]["nbits"] > 0 and d[1] > d[0]      # six more sigma of a ~1 000-bit sample at 3 %: about +3 % of QBER
    assert marg["a_final"]["nbits"] == max(0, base["a_final"]["nbits"] - (d[1] - d[0]))      # and privacy amplification removes exactly what was disclosed on top
    expect("planning_margin.disclosed", d)
    expect("planning_margin.nbits", [int(base["a_final"]["nbits"]), int(marg["a_final"]["nbits"])])
    import subprocess
    args = [binary, "-c", "c", "-s", "s", "-r", "r", "-d", "d", "-f", "f", "-l", "l", "-q", "q", "-Q", "Q"]
    bad = subprocess.run(args + ["-L", "1,m500"], cwd=str(tmp_path), capture_output=True, text=True, timeout=20)      # out of range: refused while parsing options
    assert bad.returncode != 0 and "engine ready" not in bad.stdout


@pytest.mark.gpu
def test_block_beyond_the_rate_table_goes_to_cascade_from_the_start(warm_code_cache, tmp_path):
    """The rate table ends at 0.5: for an estimated QBER of about 10.5 % and more the plan has no code, and the reference itself ends blocks
    at 15 % (USELESS_ERRORBOUND, qber_estim.c:28-31).  In between, the QBER follower's per-block choice (`ldpc_selectedFor`) hands the block
    to cascade instead of ending the daemon with error 88; the next block, at 2 %, is LDPC again in the same daemon pair.  Each run draws
    its own QBER sample with the fixed seed, so which way the first block goes is reproducible and pinned."""
    binary = need("ecd2_ldpc")
    rng = np.random.default_rng(41)
    a = [rng.integers(0, 2, 9001).astype(np.uint8) for _ in range(4)]
    b = [x ^ (rng.random(x.size) < p) for x, p in zip(a, (0.125, 0.125, 0.02, 0.02))]
    out = run_loopback(binary, tmp_path, a, b, extra_args=["-T", "2", "-L", LDPC], blocks=[2, 2], timeout=90, cmd_gaps=(45.0, 0.5))
    logs = out["a_log"] + out["b_log"]
    assert "Segmentation" not in logs
    # ("QBER too high for the LDPC rate table" is printed by an initiator that cannot plan a block the follower chose LDPC for -- it then
    #  tells the follower with a no-plan header and both go on with cascade; test_initiator_without_a_plan... forces that path)
    first, second = out["finals"][0xb0b80000], out["finals"][0xb0b80002]
    took = ("ldpc" if "ldpc: epoch b0b80000: sent parity" in out["a_log"] else
            "cascade" if "Prep to send pkt subtype 4" in logs else "ended by the reference" if "Reply mode out of bounds" in logs else "?")
    print("first block (QBER 12.5 %):", took)
    assert took != "?", out["a_log"][-1500:] + "\n----\n" + out["b_log"][-1500:]
    expect("beyond_rate_table.first_block_took", took)
    if took != "ended by the reference":
        assert first[0] is not None and first[1] is not None, out["a_log"][-1500:] + "\n----\n" + out["b_log"][-1500:]
        assert first[0]["nbits"] == first[1]["nbits"] and (first[0]["words"] == first[1]["words"]).all()
    assert second[0] is not None and second[1] is not None, out["a_log"][-1500:] + "\n----\n" + out["b_log"][-1500:]
    assert "ldpc: epoch b0b80002: sent parity" in out["a_log"] and (second[0]["words"] == second[1]["words"]).all()


@pytest.mark.gpu
def test_failed_first_decode_gets_the_withheld_parity_bits_instead_of_cascade(warm_code_cache, tmp_path):
    """Second round (incremental redundancy): Alice's first parity message is corrupted (`-L x600`), Bob finds no verified codeword and
    answers with verdict 2; Alice sends the parity bits her plan had punctured (the whole parity of the same codeword, header nPunct = 0),
    Bob decodes at the mother code's rate.  One more packet instead of the cascade exchange; the block's leak is M + 32 bits."""
    binary = need("ecd2_ldpc")
    a, b = epochs(7, 4, 6001, 0.02)
    out = run_loopback(binary, tmp_path, a, b, extra_args=["-L", "1,x600"])
    assert out["a_final"] is not None and out["b_final"] is not None, out["a_log"][-3000:] + "\n----\n" + out["b_log"][-3000:]
    assert "asking for the" in out["b_log"] and "second round, sent the" in out["a_log"]
    assert "falling back to cascade" not in out["a_log"] + out["b_log"] and "Prep to send pkt subtype 4\n" not in out["a_log"]
    assert out["a_final"]["nbits"] == out["b_final"]["nbits"] > 0 and (out["a_final"]["words"] == out["b_final"]["words"]).all()
    m = re.search(r"M (\d+), (\d+) punctured", out["a_log"])
    M, p = int(m.group(1)), int(m.group(2))
    assert p > 0
    leaked = int(re.search(r"decoded \d+ key bits in \d+ iterations, \d+ errors corrected, (\d+) bits leaked", out["b_log"]).group(1))
    assert leaked == M + 32                                          # not (M - p + 32) + (M + 32): the first message's bits are among the second's
    expect("second_round.nbits", int(out["a_final"]["nbits"]))
    for log in (out["a_log"], out["b_log"]):
        acc = pa_account(log)
        assert acc["final"] == acc["workbits"] - (M + 32) - acc["sneakloss"], acc


@pytest.mark.gpu
def test_initiator_without_a_plan_tells_the_follower_and_both_use_cascade(warm_code_cache, tmp_path):
    """the main loop drops a handler's return value (ecd2.c:524), so `return LDPC_ERR_RATE` from an initiator that has no code
    for a block the follower chose LDPC for used to leave both daemons waiting.  `-L n1` (test hook) makes Alice's plan come back
    "unsupported": she must send the no-plan header, the follower answers with a failed verdict, and the block is reconciled by
    cascade in the same daemons."""
    binary = need("ecd2_ldpc")
    # (inputs the reference's cascade reconciles with its FIXED seed: with every seed the same constant its BICONF rounds repeat their
    #  subsets, and the pristine daemon leaves residual errors on some inputs -- e.g. epochs(48, 4, 9001, 0.02), or these at 3 %)
    a, b = epochs(43, 4, 5003, 0.02)
    out = run_loopback(binary, tmp_path, a, b, extra_args=["-L", "1"], extra_args_a=["-L", "1,n1"], timeout=90)
    assert out["a_final"] is not None and out["b_final"] is not None, out["a_log"][-2000:] + "\n----\n" + out["b_log"][-2000:]
    assert "telling the follower" in out["a_log"] and "the initiator has no code for this block" in out["b_log"]
    assert "falling back to cascade as EC follower" in out["b_log"] and "falling back to cascade as EC initiator" in out["a_log"]
    assert out["a_final"]["nbits"] == out["b_final"]["nbits"] > 0 and (out["a_final"]["words"] == out["b_final"]["words"]).all()
    expect("no_plan.nbits", int(out["a_final"]["nbits"]))


@pytest.mark.gpu
def test_malformed_parity_fragment_gets_a_failed_verdict(warm_code_cache, tmp_path):
    """a fragment whose word offset does not match its index (`-L z1`: Alice shifts the offset of fragment 1) is not placed
    anywhere; the follower says so, answers with a failed verdict and the block goes to cascade instead of stalling both daemons."""
    binary = need("ecd2_ldpc")
    a, b = epochs(47, 4, 9001, 0.02)      # (an input the fixed-seed cascade reconciles, see the test above; <= 500-byte packets: the parity travels in fragments)
    out = run_loopback(binary, tmp_path, a, b, extra_args=["-L", "1,p500"], extra_args_a=["-L", "1,p500,z1"], timeout=90)
    assert int(re.search(r"sent parity in (\d+) packet", out["a_log"]).group(1)) >= 2
    assert out["a_final"] is not None and out["b_final"] is not None, out["a_log"][-2000:] + "\n----\n" + out["b_log"][-2000:]
    assert "malformed parity packet" in out["b_log"] and "falling back to cascade as EC follower" in out["b_log"]
    assert out["a_final"]["nbits"] == out["b_final"]["nbits"] and (out["a_final"]["words"] == out["b_final"]["words"]).all()
    expect("malformed_fragment.nbits", int(out["a_final"]["nbits"]))


def test_device_choice_function_round_robin(tmp_path):
    """`-L D<n>` (daemon side: replicas only): blocks go to device epoch % n.  The function is compiled on its own
    from ldpc_reconcile.h; unmeasured on multi-GPU hardware."""
    src = tmp_path / "t.c"
    src.write_text('#define LDPC_DEVICE_CHOICE_ONLY\n#include "ldpc_reconcile.h"\n#include <stdio.h>\n'
                   'int main(void){unsigned e;int n;for(n=1;n<=8;n++){int c[8]={0};for(e=0xb0b80000u;e<0xb0b80000u+64;e++){int d=ldpc_deviceOf(e,n);'
                   'if(d<0||d>=n)return 1;c[d]++;}for(e=0;e<(unsigned)n;e++)if(c[e]!=64/n&&c[e]!=64/n+1)return 2;}'
                   'if(ldpc_deviceOf(7,0)!=0||ldpc_deviceOf(0xffffffffu,8)!=7)return 3;puts("ok");return 0;}\n')
    exe = tmp_path / "t"
    r = subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "qcrypto-ldpc_amd", "host"), "-o", str(exe), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert subprocess.run([str(exe)], capture_output=True, text=True).stdout.strip() == "ok"


def test_test_hooks_are_not_in_the_maintainer_build():
    """the fault-injection letters of -L exist only under -DLDPC_TEST_HOOKS."""
    ref = "/root/reference/errorcorrection"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present on this machine")
    src = os.path.join(ROOT, "qcrypto-ldpc_amd", "host", "ldpc_reconcile.c")
    out = {}
    for name, flags in (("plain", []), ("hooks", ["-DLDPC_TEST_HOOKS"])):
        r = subprocess.run(["gcc", "-std=gnu11", "-E", "-P", "-I" + ref, "-I" + os.path.join(ROOT, "include")] + flags + [src], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        out[name] = r.stdout
    for sym in ("g_opt_fault", "g_opt_dup", "g_opt_badhdr", "g_opt_noplan", "g_opt_badfrag"):
        assert sym in out["hooks"] and sym not in out["plain"], sym
