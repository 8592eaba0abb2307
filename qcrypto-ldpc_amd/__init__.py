"""qcrypto-ldpc_amd: host-side mirror (Python, ctypes) of the LDPC reconciliation path.

The product is ``libqldpc.so`` (hand-written HIP for gfx950 behind the C ABI of ``include/qldpc.h``).
This package only binds it: names and argument meaning follow the objects the reference harness
builds from AFF3CT (``Decoder_LDPC_BP_flooding(K, N, n_ite, H, info_bits_pos, rule, enable_syndrome,
syndrome_depth, n_frames)``, ``decode_siho``, ``reset``, ``encode`` -- BS/src/main.cpp:172-195,335-393).
PyTorch appears only as the owner of device buffers / streams handed to the C ABI as raw pointers.

There is NO CPU fallback: if the shared library is missing the import fails loudly.

The directory name has a hyphen, so load it with ``_qldpc_loader.load()`` (repo root) which
registers it as module ``qcrypto_ldpc_amd``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QLDPC_LIB") or os.path.join(_HERE, "libqldpc.so")      # QLDPC_LIB: A/B builds of the same library

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libqldpc.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` or "
        "`make -C qcrypto-ldpc_amd/csrc`. There is no CPU fallback." % LIB_PATH)

def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 (soname libamdhip64.so.7, requested by file name).
    If libqldpc pulled /opt/rocm's copy in first, a later `import torch` would load a SECOND HIP runtime and
    whichever initialises last sees no GPU.  Loading torch's copy first makes both share one runtime."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    for d in spec.submodule_search_locations:
        p = os.path.join(d, "lib", "libamdhip64.so")
        if os.path.exists(p):
            try:
                C.CDLL(p, mode=C.RTLD_GLOBAL)
            except OSError:
                pass
            return


_preload_torch_hip_runtime()
_L = C.CDLL(LIB_PATH)

RULES = {"MS": 0, "OMS": 1, "NMS": 2, "SPA": 3, "LSPA": 4, "AMS_MIN": 5, "AMS_MINSTAR_L2": 6, "AMS_MINSTAR": 7}
SCHEDULES = {"flooding": 0, "hlayered": 1}
VN_CHANNEL, VN_PINNED, VN_PUNCTURED = 0, 1, 2
CONFIRMED_BIT_LLR = 23.025850929840455

_ip = C.POINTER(C.c_int)
_fp = C.POINTER(C.c_float)
_vp = C.c_void_p


class DecoderCfg(C.Structure):
    _fields_ = [("schedule", C.c_int), ("rule", C.c_int), ("rule_param", C.c_float), ("n_ite", C.c_int),
                ("enable_syndrome", C.c_int), ("syndrome_depth", C.c_int), ("max_frames", C.c_int),
                ("device", C.c_int), ("frames_per_lane", C.c_int), ("engine", C.c_int), ("freeze_messages", C.c_int), ("msg_dtype", C.c_int), ("quant_scale", C.c_float), ("compact", C.c_int), ("layer_chain", C.c_int), ("reserved", C.c_int * 1)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_uint64), ("total_ms", C.c_double), ("alg_bytes", C.c_double), ("moved_bytes", C.c_double)]


def _sig(name, res, args):
    f = getattr(_L, name)
    f.restype = res
    f.argtypes = args
    return f


_sig("qldpc_version", C.c_int, [])
_sig("qldpc_strerror", C.c_char_p, [C.c_int])
_sig("qldpc_last_error", C.c_char_p, [])
_sig("qldpc_device_count", C.c_int, [])
_sig("qldpc_copy_probe", C.c_int, [C.c_int, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_double)])
_sig("qldpc_llr_from_ber", C.c_float, [C.c_float])
_sig("qldpc_bsc_llr", C.c_float, [C.c_float])
_sig("qldpc_binary_entropy", C.c_float, [C.c_float])
_sig("qldpc_min_code_rate", C.c_float, [C.c_float, C.c_float])
_sig("qldpc_parity_bits_to_punct", C.c_int, [C.c_int, C.c_int, C.c_float])
_sig("qldpc_code_from_alist", C.c_int, [C.c_char_p, C.POINTER(_vp)])
_sig("qldpc_code_from_qc", C.c_int, [C.c_char_p, C.POINTER(_vp)])
_sig("qldpc_code_from_edges", C.c_int, [C.c_int, C.c_int, C.c_int, _ip, _ip, C.POINTER(_vp)])
_sig("qldpc_code_ira", C.c_int, [C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_uint64, C.POINTER(_vp)])
_sig("qldpc_code_ira_peg", C.c_int, [C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_uint64, C.POINTER(_vp)])
_sig("qldpc_code_free", None, [_vp])
for _n in ("n", "m", "e", "max_cn_degree", "max_vn_degree", "is_ira", "layer_count"):
    _sig("qldpc_code_" + _n, C.c_int, [_vp])
_sig("qldpc_code_qc_peg", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_char_p, C.POINTER(_vp), _ip])
_sig("qldpc_code_export_edges", C.c_int, [_vp, _ip, _ip])
_sig("qldpc_code_layer_order", C.c_int, [_vp, _ip, _ip])
_sig("qldpc_code_syndrome_host", C.c_int, [_vp, _ip, _ip])
_sig("qldpc_decoder_cfg_default", None, [C.POINTER(DecoderCfg)])
_sig("qldpc_decoder_create", C.c_int, [_vp, C.c_int, _ip, C.POINTER(DecoderCfg), C.POINTER(_vp)])
_sig("qldpc_decoder_free", None, [_vp])
_sig("qldpc_decoder_set_stream", C.c_int, [_vp, _vp])
_sig("qldpc_decoder_reset", C.c_int, [_vp])
_sig("qldpc_decoder_device_bytes", C.c_size_t, [_vp])
_sig("qldpc_decoder_reserve", C.c_int, [_vp])
_sig("qldpc_decode_siho", C.c_int, [_vp, _fp, _ip, C.c_int])
_sig("qldpc_load_llr_dev", C.c_int, [_vp, _vp, C.c_int])
_sig("qldpc_load_bits_dev", C.c_int, [_vp, _vp, _vp, _vp, C.c_int])
_sig("qldpc_load_bits_short_dev", C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int])
_sig("qldpc_load_syndrome_dev", C.c_int, [_vp, _vp, C.c_int])
_sig("qldpc_load_erasures_dev", C.c_int, [_vp, _vp, C.c_int])
_sig("qldpc_syndrome_dev", C.c_int, [_vp, _vp, _vp, C.c_int])
_sig("qldpc_run", C.c_int, [_vp])
_sig("qldpc_fetch_packed_dev", C.c_int, [_vp, _vp])
_sig("qldpc_fetch_info_dev", C.c_int, [_vp, _vp])
_sig("qldpc_fetch_status_dev", C.c_int, [_vp, _vp, _vp])
_sig("qldpc_fetch_post_dev", C.c_int, [_vp, _vp])
_sig("qldpc_sync", C.c_int, [_vp])
_sig("qldpc_profile_enable", C.c_int, [_vp, C.c_int])
_sig("qldpc_profile_read", C.c_int, [_vp, C.POINTER(KernelStat), C.c_int])
_sig("qldpc_profile_clear", C.c_int, [_vp])
_sig("qldpc_last_run_iterations", C.c_int, [_vp])
_sig("qldpc_last_run_stats", C.c_int, [_vp, C.POINTER(C.c_longlong)])
_sig("qldpc_encoder_create", C.c_int, [_vp, C.c_char_p, C.c_int, C.POINTER(_vp)])
_sig("qldpc_encoder_free", None, [_vp])
_sig("qldpc_encoder_reserve", C.c_int, [_vp, C.c_int])
_sig("qldpc_encoder_k", C.c_int, [_vp])
_sig("qldpc_encoder_info_bits_pos", C.c_int, [_vp, _ip])
_sig("qldpc_encode", C.c_int, [_vp, _ip, _ip, C.c_int])
_sig("qldpc_encode_packed_dev", C.c_int, [_vp, _vp, _vp, C.c_int, _vp])


class QldpcError(RuntimeError):
    """Raised where the AFF3CT objects would throw tools::exception; carries the C status."""

    def __init__(self, status, where):
        self.status = status
        msg = _L.qldpc_last_error().decode(errors="replace")
        super().__init__("%s: %s (%d)%s" % (where, _L.qldpc_strerror(status).decode(), status, (": " + msg) if msg else ""))


def _chk(rc, where):
    if rc < 0:
        raise QldpcError(rc, where)
    return rc


def version():
    return _L.qldpc_version()


def device_count():
    return _L.qldpc_device_count()


def copy_probe(nbytes=1 << 30, reps=10, wide=False, device=0):
    """GB/s (bytes read + bytes written) this device copies at in the decoder's access shape (qldpc_copy_probe): a measured ceiling
    to read the kernels' rates against."""
    out = C.c_double(0.0)
    _chk(_L.qldpc_copy_probe(int(device), int(nbytes), int(reps), 1 if wide else 0, C.byref(out)), "copy_probe")
    return out.value


def llr_from_ber(p):
    """LLR(BER) macro, BS/src/main.cpp:20."""
    return _L.qldpc_llr_from_ber(float(p))


def bsc_llr(p):
    """|LLR| Modem_OOK_BSC gives a channel bit, BS/src/main.cpp:317,348."""
    return _L.qldpc_bsc_llr(float(p))


def binary_entropy(q):
    return _L.qldpc_binary_entropy(float(q))


def min_code_rate(qber, efficiency):
    return _L.qldpc_min_code_rate(float(qber), float(efficiency))


def parity_bits_to_punct(N, K, target_cr):
    return _L.qldpc_parity_bits_to_punct(int(N), int(K), float(target_cr))


def _np_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Code:
    """Parity-check matrix H (tools::Sparse_matrix as the harness holds it; rows = variable nodes)."""

    def __init__(self, handle):
        self._h = _vp(handle)
        self.N = _L.qldpc_code_n(self._h)
        self.M = _L.qldpc_code_m(self._h)
        self.E = _L.qldpc_code_e(self._h)
        self.max_cn_degree = _L.qldpc_code_max_cn_degree(self._h)
        self.max_vn_degree = _L.qldpc_code_max_vn_degree(self._h)
        self.is_ira = bool(_L.qldpc_code_is_ira(self._h))
        self.n_layers = _L.qldpc_code_layer_count(self._h)

    @classmethod
    def from_alist(cls, path):
        h = _vp()
        _chk(_L.qldpc_code_from_alist(os.fsencode(path), C.byref(h)), "Code.from_alist")
        return cls(h.value)

    @classmethod
    def from_qc(cls, path):
        h = _vp()
        _chk(_L.qldpc_code_from_qc(os.fsencode(path), C.byref(h)), "Code.from_qc")
        return cls(h.value)

    @classmethod
    def from_edges(cls, N, M, var, chk):
        var, chk = _np_i32(var), _np_i32(chk)
        if var.shape != chk.shape:
            raise QldpcError(-6, "Code.from_edges")
        h = _vp()
        _chk(_L.qldpc_code_from_edges(int(N), int(M), int(var.size), var.ctypes.data_as(_ip), chk.ctypes.data_as(_ip),
                                      C.byref(h)), "Code.from_edges")
        return cls(h.value)

    @classmethod
    def ira(cls, N, K, hi_frac=0.125, dv_hi=11, dv_lo=3, seed=7):
        h = _vp()
        _chk(_L.qldpc_code_ira(int(N), int(K), float(hi_frac), int(dv_hi), int(dv_lo), int(seed), C.byref(h)), "Code.ira")
        return cls(h.value)

    @classmethod
    def ira_peg(cls, N, K, hi_frac=0.125, dv_hi=11, dv_lo=3, depth=2, seed=7):
        """IRA profile with a progressive-edge-growth information part (depth 2: no 4-cycles)."""
        h = _vp()
        _chk(_L.qldpc_code_ira_peg(int(N), int(K), float(hi_frac), int(dv_hi), int(dv_lo), int(depth), int(seed), C.byref(h)), "Code.ira_peg")
        return cls(h.value)

    @classmethod
    def qc_peg(cls, n_cols, m_rows, dv, Z, seed=1, qc_path=None):
        """QC code, base graph by PEG with cycle-breaking circulant shifts (the reference's psd-peg.py), H = [lift(P) | I].
        The returned code carries .base_girth; qc_path also writes the AFF3CT .qc file."""
        h = _vp()
        g = C.c_int(0)
        _chk(_L.qldpc_code_qc_peg(int(n_cols), int(m_rows), int(dv), int(Z), int(seed), qc_path.encode() if qc_path else None, C.byref(h), C.byref(g)),
             "Code.qc_peg")
        c = cls(h.value)
        c.base_girth = g.value
        return c

    def edges(self):
        var = np.empty(self.E, np.int32)
        chk = np.empty(self.E, np.int32)
        _chk(_L.qldpc_code_export_edges(self._h, var.ctypes.data_as(_ip), chk.ctypes.data_as(_ip)), "Code.edges")
        return var, chk

    def layer_order(self):
        """(check_order[M], layer_ptr[n_layers+1], natural) of the horizontal-layered sweep."""
        order = np.empty(self.M, np.int32)
        ptr = np.empty(self.n_layers + 1, np.int32)
        nat = _chk(_L.qldpc_code_layer_order(self._h, order.ctypes.data_as(_ip), ptr.ctypes.data_as(_ip)), "Code.layer_order")
        return order, ptr, bool(nat)

    def syndrome(self, x):
        x = _np_i32(x)
        s = np.empty(self.M, np.int32)
        w = _chk(_L.qldpc_code_syndrome_host(self._h, x.ctypes.data_as(_ip), s.ctypes.data_as(_ip)), "Code.syndrome")
        return w, s

    def __del__(self):
        try:
            _L.qldpc_code_free(self._h)
        except Exception:
            pass


def _torch():
    import torch
    return torch


class Decoder:
    """module::Decoder_LDPC_BP_{flooding,horizontal_layered}<B,Q,Rule> as a batched HIP decoder.

    Decoder(code, K, n_ite, info_bits_pos, rule=("NMS", 0.75), enable_syndrome, syndrome_depth, n_frames)
    mirrors the AFF3CT ctor (VAR/main.cpp (alist-v1.0.1):203-237).
    """

    def __init__(self, code, K, n_ite, info_bits_pos=None, rule="SPA", rule_param=0.0, enable_syndrome=True,
                 syndrome_depth=1, n_frames=1, schedule="flooding", device=0, frames_per_lane=0, engine="auto",
                 freeze_messages=False, msg_dtype="f32", quant_scale=0.0, compact="auto", layer_chain="auto"):
        cfg = DecoderCfg()
        _L.qldpc_decoder_cfg_default(C.byref(cfg))
        cfg.schedule = SCHEDULES[schedule]
        cfg.rule = RULES[rule]
        cfg.rule_param = float(rule_param)
        cfg.n_ite = int(n_ite)
        cfg.enable_syndrome = int(bool(enable_syndrome))
        cfg.syndrome_depth = int(syndrome_depth)
        cfg.max_frames = int(n_frames)
        cfg.device = int(device)
        cfg.frames_per_lane = int(frames_per_lane)
        cfg.engine = {"auto": 0, "frames": 1, "edges": 2}[engine]
        cfg.freeze_messages = int(bool(freeze_messages))
        cfg.msg_dtype = {"f32": 0, "f16": 1, "i8": 2}[msg_dtype]
        cfg.quant_scale = float(quant_scale)
        cfg.compact = {"auto": 0, "on": 1, "off": 2}[compact]
        cfg.layer_chain = {"auto": 0, "on": 1, "off": 2}[layer_chain]
        pos = None
        if info_bits_pos is not None:
            pos = _np_i32(info_bits_pos)
            if pos.size != K:
                raise QldpcError(-6, "Decoder: len(info_bits_pos) != K")
        h = _vp()
        _chk(_L.qldpc_decoder_create(code._h, int(K), pos.ctypes.data_as(_ip) if pos is not None else None,
                                     C.byref(cfg), C.byref(h)), "Decoder")
        self._h = h
        self.code, self.K, self.N = code, int(K), code.N
        self.max_frames, self.device = int(n_frames), int(device)
        self.n_frames = 0

    # -- AFF3CT mirror (host vectors) --------------------------------------------------------
    def decode_siho(self, Y_N):
        """decode_siho(LLRs, dec_bits): Y_N[n_frames, N] float32 -> V_K[n_frames, K] int32 (numpy)."""
        Y = np.ascontiguousarray(Y_N, dtype=np.float32)
        if Y.ndim == 1:
            Y = Y[None, :]
        if Y.shape[1] != self.N:
            raise QldpcError(-6, "decode_siho: Y_N has %d columns, N = %d" % (Y.shape[1], self.N))
        V = np.empty((Y.shape[0], self.K), np.int32)
        _chk(_L.qldpc_decode_siho(self._h, Y.ctypes.data_as(_fp), V.ctypes.data_as(_ip), Y.shape[0]), "decode_siho")
        self.n_frames = Y.shape[0]
        return V

    def reset(self):
        _chk(_L.qldpc_decoder_reset(self._h), "reset")

    # -- staged HBM-resident path (torch tensors own the buffers) ----------------------------
    def set_stream(self, stream=None):
        torch = _torch()
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        _chk(_L.qldpc_decoder_set_stream(self._h, _vp(s.cuda_stream)), "set_stream")

    def load_llr(self, llr):
        torch = _torch()
        assert llr.is_cuda and llr.dtype == torch.float32 and llr.is_contiguous() and llr.dim() == 2 and llr.shape[1] == self.N
        self.n_frames = llr.shape[0]
        _chk(_L.qldpc_load_llr_dev(self._h, _vp(llr.data_ptr()), llr.shape[0]), "load_llr")

    def load_bits(self, bits, llr_mag, vn_class=None, n_channel=None):
        """QKD frame formation on the device; n_channel[f] (int32, optional): channel VNs at index >= n_channel[f] are known
        (shortened) bits of frame f"""
        torch = _torch()
        W = (self.N + 31) // 32
        assert bits.is_cuda and bits.dtype in (torch.int32, torch.uint32) and bits.is_contiguous() and bits.shape[1] == W
        assert llr_mag.is_cuda and llr_mag.dtype == torch.float32 and llr_mag.numel() == bits.shape[0]
        if vn_class is not None:
            assert vn_class.is_cuda and vn_class.dtype == torch.uint8 and vn_class.numel() == self.N
        self.n_frames = bits.shape[0]
        if n_channel is not None:
            assert n_channel.is_cuda and n_channel.dtype == torch.int32 and n_channel.numel() == bits.shape[0]
            _chk(_L.qldpc_load_bits_short_dev(self._h, _vp(bits.data_ptr()), _vp(llr_mag.data_ptr()),
                                              _vp(vn_class.data_ptr()) if vn_class is not None else None, _vp(n_channel.data_ptr()), bits.shape[0]), "load_bits")
            return
        _chk(_L.qldpc_load_bits_dev(self._h, _vp(bits.data_ptr()), _vp(llr_mag.data_ptr()),
                                    _vp(vn_class.data_ptr()) if vn_class is not None else None, bits.shape[0]), "load_bits")

    def load_erasures(self, erase_bits):
        """per-frame puncturing: packed masks [n_frames, ceil(N/32)], a set bit makes that VN of that frame an erasure (LLR 0)"""
        torch = _torch()
        assert erase_bits.is_cuda and erase_bits.dtype == torch.int32 and erase_bits.is_contiguous() and erase_bits.shape[1] == (self.N + 31) // 32
        _chk(_L.qldpc_load_erasures_dev(self._h, _vp(erase_bits.data_ptr()), erase_bits.shape[0]), "load_erasures")

    def load_syndrome(self, synd_bits):
        """syndrome form: packed target syndromes [n_frames, ceil(M/32)] for the frames just loaded"""
        torch = _torch()
        Wm = (self.code.M + 31) // 32
        assert synd_bits.is_cuda and synd_bits.dtype == torch.int32 and synd_bits.is_contiguous() and synd_bits.shape[1] == Wm
        _chk(_L.qldpc_load_syndrome_dev(self._h, _vp(synd_bits.data_ptr()), synd_bits.shape[0]), "load_syndrome")

    def syndrome_of(self, bits):
        """s = H x for packed words [F, ceil(N/32)] -> [F, ceil(M/32)] (device)"""
        torch = _torch()
        assert bits.is_cuda and bits.dtype == torch.int32 and bits.is_contiguous() and bits.shape[1] == (self.N + 31) // 32
        out = torch.empty((bits.shape[0], (self.code.M + 31) // 32), dtype=torch.int32, device=bits.device)
        _chk(_L.qldpc_syndrome_dev(self._h, _vp(bits.data_ptr()), _vp(out.data_ptr()), bits.shape[0]), "syndrome_of")
        return out

    def run(self):
        _chk(_L.qldpc_run(self._h), "run")

    def fetch_packed(self, out=None):
        torch = _torch()
        W = (self.N + 31) // 32
        if out is None:
            out = torch.empty((self.n_frames, W), dtype=torch.int32, device="cuda:%d" % self.device)
        _chk(_L.qldpc_fetch_packed_dev(self._h, _vp(out.data_ptr())), "fetch_packed")
        return out

    def fetch_info(self, out=None):
        torch = _torch()
        if out is None:
            out = torch.empty((self.n_frames, self.K), dtype=torch.int32, device="cuda:%d" % self.device)
        _chk(_L.qldpc_fetch_info_dev(self._h, _vp(out.data_ptr())), "fetch_info")
        return out

    def fetch_status(self):
        torch = _torch()
        it = torch.empty(self.n_frames, dtype=torch.int32, device="cuda:%d" % self.device)
        ok = torch.empty(self.n_frames, dtype=torch.int32, device="cuda:%d" % self.device)
        _chk(_L.qldpc_fetch_status_dev(self._h, _vp(it.data_ptr()), _vp(ok.data_ptr())), "fetch_status")
        return it, ok

    def fetch_post(self):
        torch = _torch()
        out = torch.empty((self.n_frames, self.N), dtype=torch.float32, device="cuda:%d" % self.device)
        _chk(_L.qldpc_fetch_post_dev(self._h, _vp(out.data_ptr())), "fetch_post")
        return out

    def sync(self):
        _chk(_L.qldpc_sync(self._h), "sync")

    @property
    def device_bytes(self):
        return _L.qldpc_decoder_device_bytes(self._h)

    @property
    def last_run_iterations(self):
        return _L.qldpc_last_run_iterations(self._h)

    def last_run_stats(self):
        """Early-exit bookkeeping of the last run: lane-iterations executed, compactions, groups in flight at the end, frames per group."""
        out = (C.c_longlong * 4)()
        _chk(_L.qldpc_last_run_stats(self._h, out), "last_run_stats")
        return dict(lane_iterations=int(out[0]), compactions=int(out[1]), final_groups=int(out[2]), frames_per_group=int(out[3]))

    # -- measurement ---------------------------------------------------------------------------
    def profile(self, on=True):
        _chk(_L.qldpc_profile_enable(self._h, int(on)), "profile")

    def profile_clear(self):
        _chk(_L.qldpc_profile_clear(self._h), "profile_clear")

    def profile_read(self):
        arr = (KernelStat * 16)()
        n = _chk(_L.qldpc_profile_read(self._h, arr, 16), "profile_read")
        return [dict(name=arr[i].name.decode(), launches=int(arr[i].launches), total_ms=float(arr[i].total_ms),
                     alg_bytes=float(arr[i].alg_bytes), moved_bytes=float(arr[i].moved_bytes)) for i in range(n)]

    def __del__(self):
        try:
            _L.qldpc_decoder_free(self._h)
        except Exception:
            pass


class Encoder:
    """m.encoder->encode (BS/src/main.cpp:341): method 'IRA', 'IDENTITY' / 'LU_DEC' (Encoder_LDPC_from_H's G_methods) or 'QC' (Encoder_LDPC_from_QC)."""

    def __init__(self, code, method="IDENTITY", device=0):
        h = _vp()
        _chk(_L.qldpc_encoder_create(code._h, method.encode(), int(device), C.byref(h)), "Encoder")
        self._h = h
        self.code, self.N, self.device = code, code.N, int(device)
        self.K = _L.qldpc_encoder_k(self._h)
        pos = np.empty(self.K, np.int32)
        _chk(_L.qldpc_encoder_info_bits_pos(self._h, pos.ctypes.data_as(_ip)), "Encoder.info_bits_pos")
        self.info_bits_pos = pos

    def encode(self, U_K):
        U = _np_i32(U_K)
        if U.ndim == 1:
            U = U[None, :]
        if U.shape[1] != self.K:
            raise QldpcError(-6, "encode: U_K has %d columns, K = %d" % (U.shape[1], self.K))
        X = np.empty((U.shape[0], self.N), np.int32)
        _chk(_L.qldpc_encode(self._h, U.ctypes.data_as(_ip), X.ctypes.data_as(_ip), U.shape[0]), "encode")
        return X

    def encode_packed(self, info, stream=None):
        torch = _torch()
        Wk, Wn = (self.K + 31) // 32, (self.N + 31) // 32
        assert info.is_cuda and info.dtype == torch.int32 and info.is_contiguous() and info.shape[1] == Wk
        out = torch.empty((info.shape[0], Wn), dtype=torch.int32, device=info.device)
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        _chk(_L.qldpc_encode_packed_dev(self._h, _vp(info.data_ptr()), _vp(out.data_ptr()), info.shape[0], _vp(s.cuda_stream)),
             "encode_packed")
        return out

    def __del__(self):
        try:
            _L.qldpc_encoder_free(self._h)
        except Exception:
            pass


# ---- packed-bit helpers (ProcessBlock.mainBufPtr layout, helpers.h:65-70) ----------------------

def pack_bits(bits):
    """bits[..., n] of 0/1 -> uint32 words [..., ceil(n/32)], bit i <-> word[i/32] & (1 << (31 - i%32))."""
    b = np.asarray(bits).astype(np.uint8)
    n = b.shape[-1]
    pad = (-n) % 32
    if pad:
        b = np.concatenate([b, np.zeros(b.shape[:-1] + (pad,), np.uint8)], axis=-1)
    by = np.packbits(b, axis=-1, bitorder="big")
    return by.reshape(b.shape[:-1] + (-1, 4)).astype(np.uint32) @ np.array([1 << 24, 1 << 16, 1 << 8, 1], np.uint32)


def unpack_bits(words, n):
    w = np.asarray(words).astype(np.uint32)
    by = np.stack([(w >> 24) & 255, (w >> 16) & 255, (w >> 8) & 255, w & 255], axis=-1).astype(np.uint8)
    bits = np.unpackbits(by.reshape(w.shape[:-1] + (-1,)), axis=-1, bitorder="big")
    return bits[..., :n]


# ---- reconciliation sessions (engine of the ecd2 LDPC handler) ---------------------------------

class ReconCfg(C.Structure):
    _fields_ = [("device", C.c_int), ("efficiency", C.c_float), ("n_rates", C.c_int), ("rates", C.c_float * 8),
                ("n_ite", C.c_int), ("rule", C.c_int), ("rule_param", C.c_float), ("key_quantum", C.c_int),
                ("max_blocks", C.c_int), ("seed", C.c_uint64), ("schedule", C.c_int), ("mother_step", C.c_int), ("mother_max", C.c_int),
                ("rate_gap", C.c_float), ("puncture", C.c_int), ("preload", C.c_int), ("peg_depth", C.c_int), ("gap_profile", C.c_int)]


class ReconMsg(C.Structure):
    _fields_ = [("rate_index", C.c_uint32), ("key_bits", C.c_uint32), ("code_k", C.c_uint32), ("code_m", C.c_uint32),
                ("crc32", C.c_uint32), ("n_punct", C.c_uint32)]


_up = C.POINTER(C.c_uint32)
_sig("qldpc_recon_cfg_default", None, [C.POINTER(ReconCfg)])
_sig("qldpc_recon_create", C.c_int, [C.POINTER(ReconCfg), C.POINTER(_vp)])
_sig("qldpc_recon_free", None, [_vp])
_sig("qldpc_recon_plan", C.c_int, [_vp, C.c_int, C.c_float, C.POINTER(ReconMsg)])
_sig("qldpc_recon_encode", C.c_int, [_vp, _up, C.c_int, C.c_float, C.POINTER(ReconMsg), _up, C.c_int])
_sig("qldpc_recon_encode_planned", C.c_int, [_vp, _up, C.c_int, C.POINTER(ReconMsg), _up, C.c_int])
_sig("qldpc_recon_decode", C.c_int, [_vp, _up, C.c_int, C.c_float, C.POINTER(ReconMsg), _up, _ip, _ip, _ip])
_sig("qldpc_recon_decode_batch", C.c_int, [_vp, C.c_int, _up, C.c_int, _fp, C.POINTER(ReconMsg), _up, _ip, _ip, _ip])
_sig("qldpc_recon_encode_blocks", C.c_int, [_vp, C.c_int, C.POINTER(_up), _ip, _fp, C.POINTER(ReconMsg), C.POINTER(_up), _ip])
_sig("qldpc_recon_decode_blocks", C.c_int, [_vp, C.c_int, C.POINTER(_up), _ip, _fp, C.POINTER(ReconMsg), C.POINTER(_up), _ip, _ip, _ip])
_sig("qldpc_crc32_words", C.c_uint32, [_up, C.c_int])
_sig("qldpc_crc32_words_chunked", C.c_uint32, [_up, C.c_int, C.c_int])
_sig("qldpc_recon_parity_words", C.c_int, [C.POINTER(ReconMsg)])
_sig("qldpc_recon_leaked_bits", C.c_int, [C.POINTER(ReconMsg)])
_sig("qldpc_recon_entries_created", C.c_long, [_vp])
_sig("qldpc_recon_check_header", C.c_int, [_vp, C.POINTER(ReconMsg), C.c_int])
_sig("qldpc_recon_profile_enable", C.c_int, [_vp, C.c_int])
_sig("qldpc_recon_profile_read", C.c_int, [_vp, C.POINTER(KernelStat), C.c_int])


_sig("qldpc_privamp", C.c_int, [C.c_int, _up, C.c_int, C.c_uint32, C.c_int, _up])
_sig("qldpc_privamp_dev", C.c_int, [_vp, C.c_int, C.c_uint32, C.c_int, _vp, _vp])


def privamp(key_words, workbits, seed, final_bits, device=0):
    """privAmp_doPrivAmp's hash (priv_amp.c:213-218) on the GPU: -> ceil(final_bits/32) words, MSB-first."""
    kw = np.ascontiguousarray(key_words, dtype=np.uint32)
    out = np.zeros((int(final_bits) + 31) // 32, np.uint32)
    _chk(_L.qldpc_privamp(int(device), kw.ctypes.data_as(_up), int(workbits), int(seed) & 0xFFFFFFFF, int(final_bits),
                          out.ctypes.data_as(_up)), "privamp")
    return out


def crc32_words(words, n_bits, lanes=0):
    """CRC-32 of the key bits; lanes > 0: the chunked fold the device verification uses (same value)"""
    w = np.ascontiguousarray(words, dtype=np.uint32)
    if lanes:
        return int(_L.qldpc_crc32_words_chunked(w.ctypes.data_as(_up), int(n_bits), int(lanes)))
    return int(_L.qldpc_crc32_words(w.ctypes.data_as(_up), int(n_bits)))


class Recon:
    """One side's reconciliation engine: what an ecd2 LDPC handler calls (qber_estim.c:337-340,420-423)."""

    def __init__(self, device=0, efficiency=1.4, rates=(0.5, 0.7, 0.8, 0.9), n_ite=None, rule=None, rule_param=None,
                 key_quantum=1024, max_blocks=1, seed=7, schedule="auto", mother_step=None, mother_max=None, rate_gap=None,
                 puncture=True, preload=False, peg_depth=None, gap_profile=0):
        cfg = ReconCfg()
        _L.qldpc_recon_cfg_default(C.byref(cfg))
        cfg.device, cfg.efficiency, cfg.n_rates = int(device), float(efficiency), len(rates)
        for i, r in enumerate(rates):
            cfg.rates[i] = float(r)
        if n_ite is not None:
            cfg.n_ite = int(n_ite)
        if rule is not None:
            cfg.rule, cfg.rule_param = RULES[rule], float(rule_param or 0.0)
        cfg.key_quantum, cfg.max_blocks, cfg.seed = int(key_quantum), int(max_blocks), int(seed)
        cfg.schedule = 2 if schedule == "auto" else SCHEDULES[schedule]      # QLDPC_RECON_SCHED_AUTO: layered for batches (max_blocks > 8), flooding (edge engine) below
        if mother_step is not None:
            cfg.mother_step = int(mother_step)
        if mother_max is not None:
            cfg.mother_max = int(mother_max)
        if rate_gap is not None:
            cfg.rate_gap = float(rate_gap)
        cfg.puncture, cfg.preload = (1 if puncture else 2), int(bool(preload))
        cfg.gap_profile = int(gap_profile)
        if peg_depth is not None:      # library default: 2 (mother codes by progressive edge growth)
            cfg.peg_depth = int(peg_depth)
        h = _vp()
        _chk(_L.qldpc_recon_create(C.byref(cfg), C.byref(h)), "Recon")
        self._h = h
        self.rates = tuple(rates)

    def plan(self, key_bits, qber):
        m = ReconMsg()
        _chk(_L.qldpc_recon_plan(self._h, int(key_bits), float(qber), C.byref(m)), "Recon.plan")
        return m

    @staticmethod
    def parity_words(msg):
        """words of disclosed parity a message carries"""
        return int(_L.qldpc_recon_parity_words(C.byref(msg)))

    @staticmethod
    def leaked_bits(msg):
        return int(_L.qldpc_recon_leaked_bits(C.byref(msg)))

    @property
    def entries_created(self):
        return int(_L.qldpc_recon_entries_created(self._h))

    def profile(self, on=True):
        _chk(_L.qldpc_recon_profile_enable(self._h, int(on)), "Recon.profile")

    def profile_read(self):
        arr = (KernelStat * 16)()
        n = _L.qldpc_recon_profile_read(self._h, arr, 16)
        _chk(n, "Recon.profile_read")
        return [dict(name=arr[i].name.decode(), launches=int(arr[i].launches), total_ms=float(arr[i].total_ms),
                     alg_bytes=float(arr[i].alg_bytes), moved_bytes=float(arr[i].moved_bytes)) for i in range(n)]

    def encode(self, key_words, key_bits, qber):
        """Alice: -> (msg, parity_words)."""
        kw = np.ascontiguousarray(key_words, dtype=np.uint32)
        m = self.plan(key_bits, qber)
        par = np.zeros(self.parity_words(m), np.uint32)
        _chk(_L.qldpc_recon_encode(self._h, kw.ctypes.data_as(_up), int(key_bits), float(qber), C.byref(m),
                                   par.ctypes.data_as(_up), par.size), "Recon.encode")
        return m, par

    def encode_planned(self, key_words, key_bits, msg, n_punct=0):
        """Alice, second round: the parity bits of the plan in `msg` with only `n_punct` of them withheld (0 = all) -> (msg2, parity_words)."""
        kw = np.ascontiguousarray(key_words, dtype=np.uint32)
        m = ReconMsg.from_buffer_copy(msg)
        m.n_punct = int(n_punct)
        par = np.zeros(self.parity_words(m), np.uint32)
        _chk(_L.qldpc_recon_encode_planned(self._h, kw.ctypes.data_as(_up), int(key_bits), C.byref(m), par.ctypes.data_as(_up), par.size), "Recon.encode_planned")
        return m, par

    def decode(self, key_words, key_bits, qber, msg, parity_words):
        """Bob: -> (ok, key_words, corrected_bits, leaked_bits, iterations); the returned copy is corrected."""
        kw = np.array(key_words, dtype=np.uint32, copy=True)
        par = np.ascontiguousarray(parity_words, dtype=np.uint32)
        c, l, it = C.c_int(0), C.c_int(0), C.c_int(0)
        rc = _L.qldpc_recon_decode(self._h, kw.ctypes.data_as(_up), int(key_bits), float(qber), C.byref(msg),
                                   par.ctypes.data_as(_up), C.byref(c), C.byref(l), C.byref(it))
        if rc not in (0, -9):
            _chk(rc, "Recon.decode")
        return rc == 0, kw, c.value, l.value, it.value

    def decode_batch(self, key_words, key_bits, qber, msgs, parity_words):
        """n blocks of one length on one code; parity_words: one uint32 array per block (they differ in length when the blocks
        are punctured differently) or a 2-D array with ceil(code_m / 32) columns"""
        kw = np.array(key_words, dtype=np.uint32, copy=True)
        n = kw.shape[0]
        Wm = (int(msgs[0].code_m) + 31) // 32
        par = np.zeros((n, Wm), np.uint32)
        for i in range(n):
            row = np.asarray(parity_words[i], dtype=np.uint32).ravel()
            par[i, :min(Wm, row.size)] = row[:Wm]
        qb = np.ascontiguousarray(qber, dtype=np.float32)
        arr = (ReconMsg * n)(*msgs)
        st, co, it = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.int32)
        _chk(_L.qldpc_recon_decode_batch(self._h, n, kw.ctypes.data_as(_up), int(key_bits), qb.ctypes.data_as(_fp), arr,
                                         par.ctypes.data_as(_up), st.ctypes.data_as(_ip), co.ctypes.data_as(_ip),
                                         it.ctypes.data_as(_ip)), "Recon.decode_batch")
        return st, kw, co, it

    def encode_blocks(self, keys, key_bits, qber):
        """Alice's side for a list of blocks: returns (msgs, parities), one ReconMsg / uint32 array per block"""
        n = len(keys)
        kws = [np.ascontiguousarray(k, dtype=np.uint32) for k in keys]
        kb = np.ascontiguousarray(key_bits, dtype=np.int32)
        qb = np.ascontiguousarray(qber, dtype=np.float32)
        plans = [self.plan(int(b), float(p)) for b, p in zip(kb, qb)]
        pars = [np.zeros(self.parity_words(m), np.uint32) for m in plans]
        caps = np.array([p.size for p in pars], np.int32)
        kp = (_up * n)(*[k.ctypes.data_as(_up) for k in kws])
        pp = (_up * n)(*[p.ctypes.data_as(_up) for p in pars])
        arr = (ReconMsg * n)()
        _chk(_L.qldpc_recon_encode_blocks(self._h, n, kp, kb.ctypes.data_as(_ip), qb.ctypes.data_as(_fp), arr, pp, caps.ctypes.data_as(_ip)),
             "Recon.encode_blocks")
        return [arr[i] for i in range(n)], pars

    def decode_blocks(self, keys, key_bits, qber, msgs, parities):
        """blocks of any mix of lengths / plans: lists of per-block uint32 arrays; returns (status[], corrected keys, corrected[], iterations[])"""
        n = len(keys)
        kws = [np.array(k, dtype=np.uint32, copy=True) for k in keys]
        pars = [np.ascontiguousarray(p, dtype=np.uint32) for p in parities]
        kp = (_up * n)(*[k.ctypes.data_as(_up) for k in kws])
        pp = (_up * n)(*[p.ctypes.data_as(_up) for p in pars])
        kb = np.ascontiguousarray(key_bits, dtype=np.int32)
        qb = np.ascontiguousarray(qber, dtype=np.float32)
        arr = (ReconMsg * n)(*msgs)
        st, co, it = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.int32)
        _chk(_L.qldpc_recon_decode_blocks(self._h, n, kp, kb.ctypes.data_as(_ip), qb.ctypes.data_as(_fp), arr, pp, st.ctypes.data_as(_ip),
                                          co.ctypes.data_as(_ip), it.ctypes.data_as(_ip)), "Recon.decode_blocks")
        return st, kws, co, it

    def prepare_decode(self, keys, key_bits, qber, msgs, parities):
        """decode_blocks with the argument marshalling done once: returns an object whose run() is the one C call
        (qldpc_recon_decode_blocks) and nothing else -- what a C caller's timed region contains"""
        return _PreparedDecode(self, keys, key_bits, qber, msgs, parities)

    def __del__(self):
        try:
            _L.qldpc_recon_free(self._h)
        except Exception:
            pass


class _PreparedDecode:
    def __init__(self, recon, keys, key_bits, qber, msgs, parities):
        self._r = recon
        self.n = n = len(keys)
        self._orig = [np.ascontiguousarray(k, dtype=np.uint32) for k in keys]
        self.keys = [k.copy() for k in self._orig]
        self._pars = [np.ascontiguousarray(p_, dtype=np.uint32) for p_ in parities]
        self._kp = (_up * n)(*[k.ctypes.data_as(_up) for k in self.keys])
        self._pp = (_up * n)(*[p_.ctypes.data_as(_up) for p_ in self._pars])
        self._kb = np.ascontiguousarray(key_bits, dtype=np.int32)
        self._qb = np.ascontiguousarray(qber, dtype=np.float32)
        self._msgs = (ReconMsg * n)(*msgs)
        self.status, self.corrected, self.iterations = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.int32)

    def reset(self):
        """Bob's uncorrected keys again (the decode works in place)"""
        for k, o in zip(self.keys, self._orig):
            k[:] = o

    def run(self):
        _chk(_L.qldpc_recon_decode_blocks(self._r._h, self.n, self._kp, self._kb.ctypes.data_as(_ip), self._qb.ctypes.data_as(_fp), self._msgs, self._pp,
                                          self.status.ctypes.data_as(_ip), self.corrected.ctypes.data_as(_ip), self.iterations.ctypes.data_as(_ip)),
             "Recon.decode_blocks")
