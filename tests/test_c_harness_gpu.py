"""GPU suite: the C host harness (qcrypto-ldpc_amd/host/qldpc_sim.c), i.e. BS/src/main.cpp's loop in C over the C ABI."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM = os.path.join(ROOT, "qcrypto-ldpc_amd", "host", "qldpc_sim")


def rows(out):
    r = []
    for line in out.splitlines():
        if line.startswith("#") or "|" not in line:
            continue
        f = [x.strip() for x in line.split("|")]
        r.append(dict(ep=float(f[0]), fra=int(f[1]), be=int(f[2]), fe=int(f[3]), ber=float(f[4]), fer=float(f[5]), thr=float(f[6])))
    return r


def run(*args):
    if not os.path.exists(SIM):
        subprocess.check_call(["make", "-C", os.path.dirname(SIM)])
    p = subprocess.run([SIM] + list(args), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    return p.stdout


def test_rate08_threshold_matches_the_reference_tables():
    """BS/data_dvb/data3 (DVB S2)/DVB_S2_N_64800_K_51840_CR_0.8.txt:31-36: rate 0.8 is clean up to 3 % and dead at
    3.5 % and beyond (SPA); the DVB-like IRA code shows the same band."""
    out = run("-N", "16384", "-K", "13107", "-r", "SPA", "-i", "50", "-f", "64", "-b", "64", "-s", "0.01:0.05:0.01")
    r = {round(x["ep"], 2): x for x in rows(out)}
    assert "Decoder_LDPC_BP_flooding_Update_rule_SPA" in out and "Code rate  (R) = 0.79" in out
    assert r[0.01]["fe"] == 0 and r[0.02]["fe"] == 0 and r[0.01]["fra"] == 64
    assert r[0.05]["fe"] == 64                         # far beyond the threshold: every frame fails
    assert r[0.01]["thr"] > 0


def test_alist_layered_minsum_runs():
    out = run("-a", os.path.join(ROOT, "tests", "golden", "PEGReg504x1008.alist"), "-r", "NMS", "-p", "0.75", "-l", "-i", "20", "-f", "200", "-b", "100",
              "-s", "0.02:0.02:0.01")
    r = rows(out)
    assert len(r) == 1 and r[0]["fra"] == 200 and r[0]["fe"] <= 2
    assert "horizontal_layered" in out and "Info. bits (K) = 504" in out


def test_puncturing_to_a_target_efficiency():
    """main.cpp (parity bit puncture pattern finder): mother code rate 0.7, parity bits punctured (LLR 0) until the rate is
    min_cr(QBER, f); at f = 1.8 and QBER 2 % the punctured code still decodes, the header reports rate and efficiency."""
    out = run("-N", "16384", "-K", "11469", "-r", "SPA", "-i", "50", "-f", "64", "-b", "64", "-s", "0.02:0.02:0.01", "-e", "1.8")
    r = rows(out)
    assert len(r) == 1 and r[0]["fra"] == 64 and r[0]["fe"] <= 3
    line = [l for l in out.splitlines() if "puncturing" in l][0]
    import re
    m = re.search(r"puncturing (\d+) of (\d+) parity bits -> rate ([0-9.]+), efficiency f = ([0-9.]+)", line)
    n_p, n_par, rate, f = int(m.group(1)), int(m.group(2)), float(m.group(3)), float(m.group(4))
    assert 0 < n_p < n_par and 0.70 < rate < 0.81 and 1.75 < f < 1.85


def test_puncture_pattern_search_and_message_widths(tmp_path):
    """-R: the reference's random pattern search (one shuffled pattern per simulation round, BS/src/main.cpp:321-333) keeps
    the best pattern and writes it out; -Q 8 / 16 run the same harness on the narrow message variants."""
    pat = tmp_path / "best.txt"
    out = run("-N", "16384", "-K", "11469", "-r", "NMS", "-p", "0.75", "-i", "50", "-f", "256", "-b", "64", "-s", "0.02:0.02:0.01", "-e", "1.6", "-R",
              "-o", str(pat), "-Q", "8")
    assert "8-bit messages" in out
    pats = [l for l in out.splitlines() if l.startswith("#   pattern")]
    assert len(pats) == 4                                                # 256 frames / 64 per batch
    best = [l for l in out.splitlines() if "best of 4 patterns" in l]
    assert len(best) == 1
    import re
    fes = [int(re.search(r"FE (\d+) /", l).group(1)) for l in pats]
    assert int(re.search(r"FE (\d+),", best[0]).group(1)) == min(fes)
    idx = [int(l) for l in pat.read_text().splitlines() if not l.startswith("#")]
    n_p = int(re.search(r"puncturing (\d+) of", out).group(1))
    assert len(idx) == n_p == len(set(idx)) and min(idx) >= 11469 and max(idx) < 16384      # parity VNs only
    out16 = run("-N", "8192", "-K", "6554", "-r", "NMS", "-p", "0.75", "-f", "128", "-b", "128", "-s", "0.02:0.02:0.01", "-Q", "16")
    assert "16-bit messages" in out16 and rows(out16)[0]["fe"] == 0


def test_encoder_construction_option():
    """`-G` = p.G_method of the harness (VAR/main.cpp (alist-v1.0.1):135-145) / Encoder_LDPC_from_QC ((qc):145): the loop runs with each
    construction and decodes (the information positions differ, the code is the same)."""
    alist = os.path.join(ROOT, "tests", "golden", "PEGReg504x1008.alist")
    for g in ("IDENTITY", "LU_DEC"):
        out = run("-a", alist, "-G", g, "-r", "NMS", "-p", "0.75", "-i", "30", "-f", "100", "-b", "100", "-s", "0.02:0.02:0.01")
        r = rows(out)
        assert len(r) == 1 and r[0]["fra"] == 100 and r[0]["fe"] <= 2 and "Info. bits (K) = 504" in out
    out = run("-q", os.path.join(ROOT, "tests", "golden", "NR_1_0_2.qc"), "-G", "QC", "-r", "SPA", "-i", "50", "-f", "100", "-b", "100", "-s", "0.01:0.01:0.01")
    r = rows(out)
    assert len(r) == 1 and r[0]["fra"] == 100 and r[0]["fe"] < 60      # N = 136: short code, FER band of README_LDPC.md:937-974 without puncturing is far below
    p = subprocess.run([SIM, "-a", alist, "-G", "CHOLESKY"], capture_output=True, text=True, timeout=60)
    assert p.returncode != 0


def test_config3_stream_timed_from_c():
    """host/qldpc_stream.c: BASELINE config 3 with nothing but the C ABI in the timed region -- Alice's encode_blocks, ONE decode_blocks call for
    the whole stream on Bob's side (lanes, device-side verification).  A small stream here: every epoch reconciled, Bob's words equal Alice's
    (the tool compares them itself), leak in the band of the plan, all four table rates in use."""
    import json
    exe = os.path.join(ROOT, "qcrypto-ldpc_amd", "host", "qldpc_stream")
    if not os.path.exists(exe):
        pytest.skip("qldpc_stream not built")
    p = subprocess.run([exe, "-e", "96", "-k", "20011", "-b", "64", "-r", "2", "-S", "5", "-p"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["reconciled"] == d["epochs"] == 96 and sum(d["epochs_per_rate"]) == 96 and sum(1 for x in d["epochs_per_rate"] if x) >= 3
    assert 0.25 < d["leaked_fraction"] < 0.36 and d["failed_per_rate"] == [0, 0, 0, 0]
    assert d["ms_best"] > 0 and d["kernels"]["layer_update"]["launches"] > 0 and "cn_update" not in d["kernels"]      # batches decode on the layered schedule by default
    p = subprocess.run([exe, "-e", "96", "-k", "20011", "-b", "64", "-r", "2", "-S", "5", "-p", "-f"], capture_output=True, text=True, timeout=300)      # -f: flooding, round 2's schedule
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    f = json.loads(p.stdout.strip().splitlines()[-1])
    assert f["reconciled"] == 96 and f["kernels"]["cn_update"]["launches"] > 0 and f["leaked_fraction"] == d["leaked_fraction"] and f["avg_iterations"] > 1.5 * d["avg_iterations"]
