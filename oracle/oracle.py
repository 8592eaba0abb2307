"""ctypes wrapper around the CPU oracle (oracle/libqldpc_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libqldpc_oracle.so")

RULES = {"MS": 0, "OMS": 1, "NMS": 2, "SPA": 3, "LSPA": 4, "AMS_MIN": 5, "AMS_MINSTAR_L2": 6, "AMS_MINSTAR": 7}
SCHEDULES = {"flooding": 0, "hlayered": 1}


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "qldpc_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libqldpc_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        ip = C.POINTER(C.c_int)
        fp = C.POINTER(C.c_float)
        L.orc_graph_from_edges.restype = C.c_void_p
        L.orc_graph_from_edges.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip]
        L.orc_graph_from_alist.restype = C.c_void_p
        L.orc_graph_from_alist.argtypes = [C.c_char_p]
        L.orc_graph_from_qc.restype = C.c_void_p
        L.orc_graph_from_qc.argtypes = [C.c_char_p]
        L.orc_graph_free.argtypes = [C.c_void_p]
        for n in ("N", "M", "E", "max_cn_degree", "max_vn_degree"):
            f = getattr(L, "orc_graph_" + n)
            f.restype = C.c_int
            f.argtypes = [C.c_void_p]
        L.orc_graph_export.argtypes = [C.c_void_p, ip, ip, ip, ip, ip]
        L.orc_decode.restype = C.c_int
        L.orc_decode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, fp, C.c_int,
                                 fp, ip, ip, ip, C.c_int]
        L.orc_decode_coset.restype = C.c_int
        L.orc_decode_coset.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, fp, ip, C.c_int,
                                       fp, ip, ip, ip, C.c_int]
        L.orc_decode_i8.restype = C.c_int
        L.orc_decode_i8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, fp, ip, C.c_int,
                                    fp, ip, ip, ip, C.c_int]
        L.orc_syndrome.restype = C.c_int
        L.orc_syndrome.argtypes = [C.c_void_p, ip, ip]
        up = C.POINTER(C.c_uint32)
        L.orc_lfsr32.restype = C.c_uint32
        L.orc_lfsr32.argtypes = [up]
        L.orc_privamp.restype = None
        L.orc_privamp.argtypes = [up, C.c_int, C.c_uint32, C.c_int, up]
        _lib = L
    return _lib


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Graph:
    """Tanner graph in AFF3CT orientation/order (see qldpc_oracle.c)."""

    def __init__(self, handle):
        if not handle:
            raise ValueError("oracle: could not build graph")
        self._h = C.c_void_p(handle)
        L = lib()
        self.N, self.M, self.E = L.orc_graph_N(self._h), L.orc_graph_M(self._h), L.orc_graph_E(self._h)
        self.max_dc, self.max_dv = L.orc_graph_max_cn_degree(self._h), L.orc_graph_max_vn_degree(self._h)

    @classmethod
    def from_alist(cls, path):
        return cls(lib().orc_graph_from_alist(os.fsencode(path)))

    @classmethod
    def from_qc(cls, path):
        return cls(lib().orc_graph_from_qc(os.fsencode(path)))

    @classmethod
    def from_edges(cls, N, M, var, chk):
        var = np.ascontiguousarray(var, dtype=np.int32)
        chk = np.ascontiguousarray(chk, dtype=np.int32)
        return cls(lib().orc_graph_from_edges(N, M, len(var), _ip(var), _ip(chk)))

    def export(self):
        cn_ptr = np.empty(self.M + 1, np.int32)
        cn_var = np.empty(self.E, np.int32)
        vn_ptr = np.empty(self.N + 1, np.int32)
        vn_chk = np.empty(self.E, np.int32)
        tr = np.empty(self.E, np.int32)
        lib().orc_graph_export(self._h, _ip(cn_ptr), _ip(cn_var), _ip(vn_ptr), _ip(vn_chk), _ip(tr))
        return dict(cn_ptr=cn_ptr, cn_var=cn_var, vn_ptr=vn_ptr, vn_chk=vn_chk, transpose=tr)

    def edges(self):
        """(var, chk) pairs, CN-major -- rebuilding from these reproduces the same graph."""
        ex = self.export()
        chk = np.repeat(np.arange(self.M, dtype=np.int32), np.diff(ex["cn_ptr"]))
        return ex["cn_var"].copy(), chk

    def syndrome(self, x):
        x = np.ascontiguousarray(x, dtype=np.int32)
        s = np.empty(self.M, np.int32)
        w = lib().orc_syndrome(self._h, _ip(x), _ip(s))
        return w, s

    def __del__(self):
        try:
            lib().orc_graph_free(self._h)
        except Exception:
            pass


def decode(graph, llr, rule="SPA", param=0.0, n_ite=10, schedule="flooding", enable_syndrome=True,
           syndrome_depth=1, n_threads=1, msg_fp16=False, target=None, msg_i8=False, quant_scale=8.0):
    """decode_siho on llr[n_frames, N]; returns dict(post, hard, iters, synd_ok).
    msg_i8: the 8-bit fixed-point flooding min-sum (orc_decode_i8); post then holds the integer posteriors."""
    llr = np.ascontiguousarray(llr, dtype=np.float32)
    if llr.ndim == 1:
        llr = llr[None, :]
    F, N = llr.shape
    assert N == graph.N
    post = np.empty((F, N), np.float32)
    hard = np.empty((F, N), np.int32)
    iters = np.empty(F, np.int32)
    ok = np.empty(F, np.int32)
    tgt = None
    if target is not None:
        tgt = np.ascontiguousarray(target, dtype=np.int32).reshape(F, graph.M)
    if msg_i8:
        assert not msg_fp16
        rc = lib().orc_decode_i8(graph._h, SCHEDULES[schedule], RULES[rule], float(param), float(quant_scale), int(n_ite), int(enable_syndrome), int(syndrome_depth),
                                 _fp(llr), _ip(tgt) if tgt is not None else None, F, _fp(post), _ip(hard), _ip(iters), _ip(ok), int(n_threads))
        if rc != 0:
            raise RuntimeError("orc_decode_i8 failed: %d" % rc)
        return dict(post=post, hard=hard, iters=iters, synd_ok=ok)
    rc = lib().orc_decode_coset(graph._h, SCHEDULES[schedule], RULES[rule] | (0x100 if msg_fp16 else 0), float(param), int(n_ite),
                                int(enable_syndrome), int(syndrome_depth), _fp(llr), _ip(tgt) if tgt is not None else None, F, _fp(post),
                                _ip(hard), _ip(iters), _ip(ok), int(n_threads))
    if rc != 0:
        raise RuntimeError("orc_decode failed: %d" % rc)
    return dict(post=post, hard=hard, iters=iters, synd_ok=ok)


def lfsr32_stream(seed, n):
    """n successive outputs of the restated rnd_getPrngValue2_32."""
    st = C.c_uint32(seed)
    return np.array([lib().orc_lfsr32(C.byref(st)) for _ in range(n)], np.uint32)


def privamp(key_words, workbits, seed, final_bits):
    kw = np.ascontiguousarray(key_words, dtype=np.uint32)
    out = np.zeros((final_bits + 31) // 32, np.uint32)
    up = C.POINTER(C.c_uint32)
    lib().orc_privamp(kw.ctypes.data_as(up), int(workbits), int(seed), int(final_bits), out.ctypes.data_as(up))
    return out


def ref_rnd():
    """The reference's own subcomponents/rnd.c as built by oracle/build_ref_ecd2.sh (None when absent)."""
    p = os.path.join(_HERE, "_ref", "librefrnd.so")
    if not os.path.exists(p):
        return None
    R = C.CDLL(p)
    R.rnd_getPrngValue2_32.restype = C.c_uint32
    R.rnd_getPrngValue2_32.argtypes = [C.POINTER(C.c_uint32)]
    return R
