"""ctypes binding of libfhe_hip.so (C-ABI: include/fhe_hip.h).

Buffers are either torch CUDA tensors (dtype int64, bit pattern = uint64 limbs; device pointers are
passed straight through, no copies) or C-contiguous numpy uint64 arrays (host pointers, staged by
the library).  PyTorch is used for device memory and streams only.
"""
import ctypes as C
import os

import numpy as np

LIB_PATH = os.environ.get("FHE_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libfhe_hip.so")

MEM_DEVICE, MEM_HOST = 0, 1

_lib = None


class FheHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("fhe_hip error %d: %s" % (code, msg))
        self.code = code


def _load_hip_runtime():
    """libfhe_hip.so carries no NEEDED entry for the HIP runtime (built with -no-hip-rt): make the process's
    runtime visible first.  Inside a PyTorch process that is the libamdhip64 PyTorch ships, so tensors, streams
    and events are shared with torch; otherwise the system ROCm runtime."""
    candidates = []
    try:
        import torch
        candidates.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except Exception:
        pass
    candidates += ["/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"]
    for path in candidates:
        if os.path.isabs(path) and not os.path.exists(path):
            continue
        try:
            return C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            continue
    raise OSError("no HIP runtime (libamdhip64) found")


def load_library():
    """Load libfhe_hip.so; raises if it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            "%s not found: build it with `make -C fhe_amd/csrc` (or python -c 'import __graft_entry__ as g; g.build()')" % LIB_PATH)
    _load_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    vp, u32, u64, i, sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_size_t
    sig = {
        "fhe_hip_ctx_create": (i, [C.POINTER(vp), u32, u32, u32, vp, vp, i]),
        "fhe_hip_ctx_destroy": (None, [vp]),
        "fhe_hip_last_error": (C.c_char_p, [vp]),
        "fhe_hip_set_stream": (i, [vp, vp]),
        "fhe_hip_sync": (i, [vp]),
        "fhe_hip_ntt_forward": (i, [vp, vp, vp, sz, i]),
        "fhe_hip_ntt_backward": (i, [vp, vp, vp, sz, i]),
        "fhe_hip_apply_galois": (i, [vp, vp, u32, u64, sz, i]),
        "fhe_hip_apply_galois_non_ntt": (i, [vp, vp, u32, u64, sz, i]),
        "fhe_hip_mod_up": (i, [vp, vp, vp, u32, sz, i]),
        "fhe_hip_mod_down": (i, [vp, vp, vp, u32, sz, i]),
        "fhe_hip_rescale": (i, [vp, vp, vp, u32, sz, i]),
        "fhe_hip_conv_to_normal": (i, [vp, vp, vp, u32, sz, i]),
        "fhe_hip_conv_to_special": (i, [vp, vp, vp, u32, u32, sz, i]),
        "fhe_hip_mem_alloc": (i, [vp, C.POINTER(vp), sz]),
        "fhe_hip_mem_free": (i, [vp, vp]),
        "fhe_hip_mem_copy": (i, [vp, vp, vp, sz, i]),
        "fhe_hip_encode": (i, [vp, vp, vp, sz, u32, C.c_double, sz, i]),
        "fhe_hip_decode": (i, [vp, vp, vp, u32, C.c_double, sz, i]),
        "fhe_hip_load_mult_key": (i, [vp, vp, vp, i]),
        "fhe_hip_load_rot_key": (i, [vp, u64, vp, vp, i]),
        "fhe_hip_add": (i, [vp, vp, vp, u32, sz, i]),
        "fhe_hip_multiply_plain": (i, [vp, vp, vp, u32, sz, i]),
        "fhe_hip_cross_mult": (i, [vp, vp, vp, vp, u32, sz, i]),
        "fhe_hip_relin_only": (i, [vp, vp, vp, u32, sz, i]),
        "fhe_hip_relinearize": (i, [vp, vp, vp, u32, sz, i]),
        "fhe_hip_mul_relin_rescale": (i, [vp, vp, vp, vp, u32, sz, i]),
        "fhe_hip_rotate": (i, [vp, u64, vp, u32, sz, i]),
        "fhe_hip_rotate_non_ntt": (i, [vp, u64, vp, u32, sz, i]),
        "fhe_hip_rotate_to": (i, [vp, u64, vp, vp, u32, sz, i]),
        "fhe_hip_load_mixed_rot_key": (i, [vp, u64, vp, vp, i]),
        "fhe_hip_rotate_mixed": (i, [vp, u64, vp, u32, sz, i]),
        "fhe_hip_decrypt": (i, [vp, vp, vp, vp, u32, sz, i]),
        "fhe_hip_encrypt": (i, [vp, vp, vp, vp, vp, vp, vp, u32, sz, i]),
        "fhe_hip_batch_run": (i, [C.POINTER(vp), u32, i, vp, vp, vp, u64, u32, sz, i]),
        "fhe_hip_mul_relin_rescale_noalloc": (i, [vp, vp, vp, vp, u32, sz]),
        "fhe_hip_set_tuning": (i, [vp, i, u64]),
        "fhe_hip_debug_violations": (i, [vp, C.POINTER(u64), C.POINTER(u32), C.POINTER(i)]),
        "fhe_hip_profile_enable": (i, [vp, i]),
        "fhe_hip_profile_reset": (i, [vp]),
        "fhe_hip_profile_get": (i, [vp, C.POINTER(C.c_char_p), C.POINTER(u64), C.POINTER(C.c_double), i]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "fhe_hip_ctx_create", "fhe_hip_ctx_destroy", "fhe_hip_last_error", "fhe_hip_set_stream", "fhe_hip_sync",
    "fhe_hip_ntt_forward", "fhe_hip_ntt_backward", "fhe_hip_apply_galois", "fhe_hip_apply_galois_non_ntt",
    "fhe_hip_mod_up", "fhe_hip_mod_down", "fhe_hip_rescale", "fhe_hip_load_mult_key", "fhe_hip_load_rot_key",
    "fhe_hip_add", "fhe_hip_multiply_plain", "fhe_hip_cross_mult", "fhe_hip_relin_only", "fhe_hip_relinearize",
    "fhe_hip_mul_relin_rescale", "fhe_hip_rotate", "fhe_hip_rotate_non_ntt", "fhe_hip_load_mixed_rot_key", "fhe_hip_rotate_mixed", "fhe_hip_decrypt", "fhe_hip_encrypt", "fhe_hip_profile_enable", "fhe_hip_profile_reset",
    "fhe_hip_profile_get", "fhe_hip_batch_run", "fhe_hip_conv_to_normal", "fhe_hip_mul_relin_rescale_noalloc", "fhe_hip_set_tuning", "fhe_hip_debug_violations", "fhe_hip_rotate_to",
    "fhe_hip_conv_to_special", "fhe_hip_mem_alloc", "fhe_hip_mem_free", "fhe_hip_mem_copy", "fhe_hip_encode", "fhe_hip_decode",
]
OP_MUL_RELIN_RESCALE, OP_ROTATE, OP_ROTATE_MIXED, OP_ROTATE_NON_NTT = 1, 2, 3, 4
TUNE_RELIN_GROUP = 1
TUNE_NTT_CHUNK = 2
TUNE_ONEPASS = 3
TUNE_NTT_STREAMS = 4
TUNE_RELIN_KERNEL = 5
TUNE_BCONV = 6
NOALLOC_SCRATCH_BYTES = 256 << 20


def _buf(x):
    """-> (pointer, mem_kind, keepalive) for a torch CUDA int64 tensor or a numpy uint64 array."""
    if isinstance(x, np.ndarray):
        if x.dtype != np.uint64 or not x.flags["C_CONTIGUOUS"]:
            raise TypeError("host buffers must be C-contiguous numpy uint64 arrays")
        return x.ctypes.data, MEM_HOST
    import torch
    if isinstance(x, torch.Tensor):
        if not x.is_cuda or x.dtype != torch.int64 or not x.is_contiguous():
            raise TypeError("device buffers must be contiguous CUDA int64 tensors")
        return x.data_ptr(), MEM_DEVICE
    raise TypeError("unsupported buffer type %r" % type(x))


def _cbuf(x):
    """-> (pointer, mem_kind) for complex slot values: a C-contiguous numpy complex128 array or a contiguous CUDA complex128 tensor."""
    if isinstance(x, np.ndarray):
        if x.dtype != np.complex128 or not x.flags["C_CONTIGUOUS"]:
            raise TypeError("host slot buffers must be C-contiguous numpy complex128 arrays")
        return x.ctypes.data, MEM_HOST
    import torch
    if isinstance(x, torch.Tensor):
        if not x.is_cuda or x.dtype != torch.complex128 or not x.is_contiguous():
            raise TypeError("device slot buffers must be contiguous CUDA complex128 tensors")
        return x.data_ptr(), MEM_DEVICE
    raise TypeError("unsupported buffer type %r" % type(x))


def _same_kind(*bufs):
    ptrs, kinds = zip(*[_buf(b) for b in bufs])
    if len(set(kinds)) != 1:
        raise TypeError("all buffers of one call must live on the same side (all host or all device)")
    return ptrs, kinds[0]


def _count(x):
    return x.size if isinstance(x, np.ndarray) else x.numel()


def _need(**bufs):
    """name=(buffer, words): the C-ABI takes raw pointers, so every buffer is checked here against the number of 64-bit
    words the call will touch (a short buffer would be read or written out of bounds by the kernels and copies)."""
    for name, (x, words) in bufs.items():
        if _count(x) < words:
            raise ValueError("%s holds %d words, the call touches %d" % (name, _count(x), words))


class HipContext:
    """fHE::context made runtime (reference include/fHE/context.hpp:6-25): N = 2**logN, L normal and
    K special primes; owns the device tables, keys and workspaces of one GPU."""

    def __init__(self, logN, L, K, primes, psi, device=0):
        self.lib = load_library()
        self.logN, self.L, self.K, self.N = logN, L, K, 1 << logN
        pr = np.ascontiguousarray(primes, np.uint64)
        ps = np.ascontiguousarray(psi, np.uint64)
        if pr.shape != (L + K,) or ps.shape != (L + K,):
            raise ValueError("need L+K primes and roots")
        h = C.c_void_p()
        rc = self.lib.fhe_hip_ctx_create(C.byref(h), logN, L, K, pr.ctypes.data, ps.ctypes.data, device)
        if rc:
            raise FheHipError(rc, self.lib.fhe_hip_last_error(None).decode())
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.fhe_hip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise FheHipError(rc, self.lib.fhe_hip_last_error(self.h).decode())

    def set_stream(self, stream_ptr):
        self._chk(self.lib.fhe_hip_set_stream(self.h, stream_ptr))

    def sync(self):
        self._chk(self.lib.fhe_hip_sync(self.h))

    # ---- NTT --------------------------------------------------------------------------------
    def ntt_forward(self, limbs, moduli_idx):
        idx = np.ascontiguousarray(moduli_idx, np.uint32)
        (p,), kind = _same_kind(limbs)
        _need(limbs=(limbs, idx.size * self.N))
        self._chk(self.lib.fhe_hip_ntt_forward(self.h, p, idx.ctypes.data, idx.size, kind))
        return limbs

    def ntt_backward(self, limbs, moduli_idx):
        idx = np.ascontiguousarray(moduli_idx, np.uint32)
        (p,), kind = _same_kind(limbs)
        _need(limbs=(limbs, idx.size * self.N))
        self._chk(self.lib.fhe_hip_ntt_backward(self.h, p, idx.ctypes.data, idx.size, kind))
        return limbs

    # ---- automorphisms ----------------------------------------------------------------------
    def apply_galois(self, polys, nlimbs, galois, batch=1):
        (p,), kind = _same_kind(polys)
        _need(polys=(polys, batch * nlimbs * self.N))
        self._chk(self.lib.fhe_hip_apply_galois(self.h, p, nlimbs, galois, batch, kind))
        return polys

    def apply_galois_non_ntt(self, polys, nlimbs, galois, batch=1):
        (p,), kind = _same_kind(polys)
        _need(polys=(polys, batch * nlimbs * self.N))
        self._chk(self.lib.fhe_hip_apply_galois_non_ntt(self.h, p, nlimbs, galois, batch, kind))
        return polys

    # ---- base converter ---------------------------------------------------------------------
    def mod_up(self, rop, op, Li, batch=1):
        (r, o), kind = _same_kind(rop, op)
        _need(rop=(rop, batch * (Li + self.K) * self.N), op=(op, batch * Li * self.N))
        self._chk(self.lib.fhe_hip_mod_up(self.h, r, o, Li, batch, kind))
        return rop

    def mod_down(self, rop, op, L0, batch=1):
        (r, o), kind = _same_kind(rop, op)
        _need(rop=(rop, batch * L0 * self.N), op=(op, batch * (L0 + self.K) * self.N))
        self._chk(self.lib.fhe_hip_mod_down(self.h, r, o, L0, batch, kind))
        return rop

    def rescale(self, rop, op, n, batch=1):
        (r, o), kind = _same_kind(rop, op)
        _need(rop=(rop, batch * max(n - 1, 0) * self.N), op=(op, batch * n * self.N))
        self._chk(self.lib.fhe_hip_rescale(self.h, r, o, n, batch, kind))
        return rop

    def conv_to_normal(self, rop, sp, L0, batch=1):
        """BaseConverter::neg_approx_convert_to_normal_basis for targets 0..L0-1 (src/base_converter.cpp:297-311):
        sp = [batch][K][N] special limbs, power basis -> rop = [batch][L0][N], power basis."""
        (r, o), kind = _same_kind(rop, sp)
        _need(rop=(rop, batch * L0 * self.N), sp=(sp, batch * self.K * self.N))
        self._chk(self.lib.fhe_hip_conv_to_normal(self.h, r, o, L0, batch, kind))
        return rop

    def conv_to_special(self, rop, op, Li, sp_index, batch=1):
        """BaseConverter::approx_convert_to_special_basis for one special prime (src/base_converter.cpp:276-295):
        op = [batch][Li][N] power basis -> rop = [batch][N] under P[L + sp_index]."""
        (r, o), kind = _same_kind(rop, op)
        _need(rop=(rop, batch * self.N), op=(op, batch * Li * self.N))
        self._chk(self.lib.fhe_hip_conv_to_special(self.h, r, o, Li, sp_index, batch, kind))
        return rop

    # ---- encoder ---------------------------------------------------------
    def encode(self, msg, values, nlimbs, scale, batch=1):
        """Encoder::encode (src/encoder.cpp:84-157): values = [batch][nvals] complex128 -> msg = [batch][nlimbs][N], power basis."""
        vp_, vk = _cbuf(values)
        (m,), kind = _same_kind(msg)
        if vk != kind:
            raise TypeError("all buffers of one call must live on the same side (all host or all device)")
        nvals = _count(values) // batch
        if nvals * batch != _count(values):
            raise ValueError("values must hold batch * nvals entries")
        _need(msg=(msg, batch * nlimbs * self.N))
        self._chk(self.lib.fhe_hip_encode(self.h, m, vp_, nvals, nlimbs, float(scale), batch, kind))
        return msg

    def decode(self, slots, msg, nlimbs, scale, batch=1):
        """Encoder::decode (src/encoder.cpp:159-200): msg = [batch][nlimbs][N] power basis -> slots = [batch][N/2] complex128."""
        sp_, sk = _cbuf(slots)
        (m,), kind = _same_kind(msg)
        if sk != kind:
            raise TypeError("all buffers of one call must live on the same side (all host or all device)")
        if _count(slots) < batch * (self.N // 2):
            raise ValueError("slots holds %d values, the call writes %d" % (_count(slots), batch * (self.N // 2)))
        _need(msg=(msg, batch * nlimbs * self.N))
        self._chk(self.lib.fhe_hip_decode(self.h, sp_, m, nlimbs, float(scale), batch, kind))
        return slots

    # ---- keys -------------------------------------------------------------------------------
    def load_mult_key(self, beta, alpha):
        (b, a), kind = _same_kind(beta, alpha)
        _need(beta=(beta, self.L * self.L * self.N), alpha=(alpha, self.L * self.L * self.N))
        self._chk(self.lib.fhe_hip_load_mult_key(self.h, b, a, kind))

    def load_rot_key(self, galois, beta, alpha):
        (b, a), kind = _same_kind(beta, alpha)
        _need(beta=(beta, (self.L + self.K) * self.N), alpha=(alpha, (self.L + self.K) * self.N))
        self._chk(self.lib.fhe_hip_load_rot_key(self.h, galois, b, a, kind))

    # ---- evaluator --------------------------------------------------------------------------
    def add(self, rop, op, nlimbs, batch=1):
        (r, o), kind = _same_kind(rop, op)
        _need(rop=(rop, batch * nlimbs * self.N), op=(op, batch * nlimbs * self.N))
        self._chk(self.lib.fhe_hip_add(self.h, r, o, nlimbs, batch, kind))
        return rop

    def multiply_plain(self, ct, pt, nlimbs, batch=1):
        (c, p), kind = _same_kind(ct, pt)
        _need(ct=(ct, batch * 2 * nlimbs * self.N), pt=(pt, nlimbs * self.N))
        self._chk(self.lib.fhe_hip_multiply_plain(self.h, c, p, nlimbs, batch, kind))
        return ct

    def cross_mult(self, out3, c0, c1, Li, batch=1):
        (o, a, b), kind = _same_kind(out3, c0, c1)
        _need(out3=(out3, batch * 3 * Li * self.N), c0=(c0, batch * 2 * Li * self.N), c1=(c1, batch * 2 * Li * self.N))
        self._chk(self.lib.fhe_hip_cross_mult(self.h, o, a, b, Li, batch, kind))
        return out3

    def relin_only(self, out, ct3, Li, batch=1):
        (o, a), kind = _same_kind(out, ct3)
        _need(out=(out, batch * 2 * Li * self.N), ct3=(ct3, batch * 3 * Li * self.N))
        self._chk(self.lib.fhe_hip_relin_only(self.h, o, a, Li, batch, kind))
        return out

    def relinearize(self, out, ct3, Li, batch=1):
        (o, a), kind = _same_kind(out, ct3)
        _need(out=(out, batch * 2 * max(Li - 1, 0) * self.N), ct3=(ct3, batch * 3 * Li * self.N))
        self._chk(self.lib.fhe_hip_relinearize(self.h, o, a, Li, batch, kind))
        return out

    def mul_relin_rescale(self, out, c0, c1, Li, batch=1):
        (o, a, b), kind = _same_kind(out, c0, c1)
        _need(out=(out, batch * 2 * max(Li - 1, 0) * self.N), c0=(c0, batch * 2 * Li * self.N), c1=(c1, batch * 2 * Li * self.N))
        self._chk(self.lib.fhe_hip_mul_relin_rescale(self.h, o, a, b, Li, batch, kind))
        return out

    def mul_relin_rescale_noalloc(self, out, c0, c1, Li, batch=1):
        """fhe_hip_mul_relin_rescale_noalloc: same result, device buffers only, workspaces in the library's static scratch
        array (no allocation of any kind inside the call)."""
        (o, a, b), kind = _same_kind(out, c0, c1)
        if kind != MEM_DEVICE:
            raise TypeError("the no-alloc entry point takes device buffers only")
        _need(out=(out, batch * 2 * max(Li - 1, 0) * self.N), c0=(c0, batch * 2 * Li * self.N), c1=(c1, batch * 2 * Li * self.N))
        self._chk(self.lib.fhe_hip_mul_relin_rescale_noalloc(self.h, o, a, b, Li, batch))
        return out

    def debug_violations(self):
        """(count, first site, enabled): residue-range violations counted by a debug build since the last call."""
        n, site, en = C.c_uint64(0), C.c_uint32(0), C.c_int(0)
        self._chk(self.lib.fhe_hip_debug_violations(self.h, C.byref(n), C.byref(site), C.byref(en)))
        return int(n.value), int(site.value), bool(en.value)

    def set_tuning(self, knob, value):
        self._chk(self.lib.fhe_hip_set_tuning(self.h, knob, value))

    def rotate(self, galois, ct, Li, batch=1):
        (c,), kind = _same_kind(ct)
        _need(ct=(ct, batch * 2 * Li * self.N))
        self._chk(self.lib.fhe_hip_rotate(self.h, galois, c, Li, batch, kind))
        return ct

    def rotate_to(self, galois, out, ct, Li, batch=1):
        """fhe_hip_rotate_to: the rotation out of place (out != ct; ct is left untouched)."""
        (o, c), kind = _same_kind(out, ct)
        _need(out=(out, batch * 2 * Li * self.N), ct=(ct, batch * 2 * Li * self.N))
        self._chk(self.lib.fhe_hip_rotate_to(self.h, galois, o, c, Li, batch, kind))
        return out

    def rotate_non_ntt(self, galois, ct, Li, batch=1):
        """Evaluator::rotate_slots_non_ntt (src/evaluator.cpp:138-146, 327-339, 520-574): ct in the power basis, in place."""
        (c,), kind = _same_kind(ct)
        _need(ct=(ct, batch * 2 * Li * self.N))
        self._chk(self.lib.fhe_hip_rotate_non_ntt(self.h, galois, c, Li, batch, kind))
        return ct

    # ---- MixedRotKey variant -----------------------------------------
    def load_mixed_rot_key(self, galois, beta, alpha):
        (b, a), kind = _same_kind(beta, alpha)
        _need(beta=(beta, self.L * (self.L + 1) * self.N), alpha=(alpha, self.L * (self.L + 1) * self.N))
        self._chk(self.lib.fhe_hip_load_mixed_rot_key(self.h, galois, b, a, kind))

    def rotate_mixed(self, galois, ct, Li, batch=1):
        (c,), kind = _same_kind(ct)
        _need(ct=(ct, batch * 2 * Li * self.N))
        self._chk(self.lib.fhe_hip_rotate_mixed(self.h, galois, c, Li, batch, kind))
        return ct

    # ---- the steps either side of the path ----------------------------
    def decrypt(self, out, ct, sx, Li, batch=1):
        (o, c, s), kind = _same_kind(out, ct, sx)
        _need(out=(out, batch * Li * self.N), ct=(ct, batch * 2 * Li * self.N), sx=(sx, Li * self.N))
        self._chk(self.lib.fhe_hip_decrypt(self.h, o, c, s, Li, batch, kind))
        return out

    def encrypt(self, ct, u, e0m, e1, pk_b, pk_a, Li, batch=1):
        ptrs, kind = _same_kind(ct, u, e0m, e1, pk_b, pk_a)
        pw = batch * Li * self.N
        _need(ct=(ct, 2 * pw), u=(u, pw), e0m=(e0m, pw), e1=(e1, pw), pk_b=(pk_b, Li * self.N), pk_a=(pk_a, Li * self.N))
        self._chk(self.lib.fhe_hip_encrypt(self.h, *ptrs, Li, batch, kind))
        return ct

    # ---- measurement ------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._chk(self.lib.fhe_hip_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        self._chk(self.lib.fhe_hip_profile_reset(self.h))

    def profile_get(self):
        cap = 64
        names = (C.c_char_p * cap)()
        launches = (C.c_uint64 * cap)()
        ms = (C.c_double * cap)()
        n = self.lib.fhe_hip_profile_get(self.h, names, launches, ms, cap)
        if n < 0:
            self._chk(n)
        return {names[k].decode(): (int(launches[k]), float(ms[k])) for k in range(min(n, cap))}


def batch_run(ctxs, op, out, in0=None, in1=None, galois=0, Li=0, batch=1):
    """fhe_hip_batch_run: single-process multi-GPU mode (include/fhe_hip.h).  `ctxs` = HipContext objects, normally one per
    GPU, each already holding the keys the op needs; the batch is split contiguously over them, one host thread per
    context.  Host (numpy uint64) buffers only."""
    ctxs = list(ctxs)
    if not ctxs:
        raise ValueError("batch_run needs at least one context")
    bufs = [b for b in (out, in0, in1) if b is not None]
    ptrs, kind = _same_kind(*bufs)
    ptrs = list(ptrs) + [None] * (3 - len(ptrs))
    handles = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    rc = ctxs[0].lib.fhe_hip_batch_run(handles, len(ctxs), op, ptrs[0], ptrs[1], ptrs[2], galois, Li, batch, kind)
    if rc:
        msgs = [c.lib.fhe_hip_last_error(c.h).decode() for c in ctxs]
        raise FheHipError(rc, next((m for m in msgs if m), ""))
    return out
