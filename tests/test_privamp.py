"""Privacy amplification: oracle pinned by the reference's own rnd.c; GPU hash bit-exact vs the oracle."""
import ctypes as C

import numpy as np
import pytest


def test_oracle_lfsr_equals_the_reference_rnd_c(O):
    R = O.ref_rnd()
    if R is None:
        pytest.skip("oracle/_ref/librefrnd.so not built (needs /root/reference)")
    for seed in (1, 0xdeadbeef, 0x80000000, 0xe0000200, 12345):
        st = C.c_uint32(seed)
        ref = np.array([R.rnd_getPrngValue2_32(C.byref(st)) for _ in range(200)], np.uint32)
        assert (O.lfsr32_stream(seed, 200) == ref).all()


def test_oracle_privamp_small_known_structure(O):
    # one key word = all ones, workbits 32: output bit i = parity of the i-th LFSR word
    seed = 0x12345678
    out = O.privamp(np.array([0xFFFFFFFF], np.uint32), 32, seed, 64)
    words = O.lfsr32_stream(seed, 64)
    exp = np.array([bin(int(w)).count("1") & 1 for w in words], np.uint8)
    got = np.unpackbits(out.view(">u4").astype(">u4").view(np.uint8)) if False else np.array(
        [(int(out[i // 32]) >> (31 - i % 32)) & 1 for i in range(64)], np.uint8)
    assert (got == exp).all()


@pytest.mark.gpu
@pytest.mark.parametrize("workbits,final_bits,seed", [(32, 64, 1), (1000, 700, 0xdeadbeef), (20000, 12000, 0x8badf00d),
                                                     (4097, 1, 7), (65535, 300, 0xffffffff), (33, 33, 0x80000001)])
def test_gpu_privamp_bit_exact(q, O, workbits, final_bits, seed):
    rng = np.random.default_rng(workbits)
    key = q.pack_bits(rng.integers(0, 2, workbits))
    key[-1] |= np.uint32((1 << ((-workbits) % 32)) - 1)          # garbage past workbits must be ignored (priv_amp.c:196-198)
    assert (q.privamp(key, workbits, seed, final_bits) == O.privamp(key, workbits, seed, final_bits)).all()


@pytest.mark.gpu
def test_gpu_privamp_full_block_and_speed(q, O):
    """the block of SURVEY.md section 3.5: 56 880 work bits -> 41 935 final bits (CPU: ~8 s in the daemon)"""
    import time
    rng = np.random.default_rng(1)
    key = q.pack_bits(rng.integers(0, 2, 56880))
    q.privamp(key, 56880, 1, 64)                                # warm-up
    t0 = time.perf_counter()
    got = q.privamp(key, 56880, 0xb0b80000, 41935)
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref = O.privamp(key, 56880, 0xb0b80000, 41935)
    t_cpu = time.perf_counter() - t0
    assert (got == ref).all()
    print("privamp 56880 -> 41935 bits: GPU %.2f ms (incl. copies), CPU oracle %.2f s" % (t_gpu * 1e3, t_cpu))
    assert t_gpu < 0.25


@pytest.mark.gpu
def test_privamp_argument_checks(q):
    with pytest.raises(q.QldpcError):
        q.privamp(np.zeros(4, np.uint32), 0, 1, 10)
    with pytest.raises(q.QldpcError):
        q.privamp(np.zeros(4, np.uint32), 128, 1, 1 << 17)
    assert q.privamp(np.zeros(4, np.uint32), 128, 1, 0).size == 0
