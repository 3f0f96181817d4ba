"""ctypes binding of oracle/liboracle.so (the CPU restatement in fhe_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by fhe_amd/.  PARITY UNPINNED at the yell boundary (see fhe_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORACLE_LIB selects another build of the same source (tools/run_sanitized_tests.sh: liboracle_san.so, ASan + UBSan)
_LIB = os.environ.get("ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    if not os.path.exists(_LIB):
        build()
    lib = C.CDLL(_LIB)
    u64p, u32, u64, vp, i = C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_int
    sig = {
        "orc_gen_primes": (None, [u32, u32, u64p, u64p]),
        "orc_is_prime": (i, [u64]),
        "orc_ctx_create": (vp, [u32, u32, u32, u64p, u64p]),
        "orc_ctx_destroy": (None, [vp]),
        "orc_prime": (u64, [vp, u32]),
        "orc_mulmod": (u64, [u64, u64, u64]),
        "orc_powmod": (u64, [u64, u64, u64]),
        "orc_invmod": (u64, [u64, u64]),
        "orc_shoupify": (u64, [u64, u64]),
        "orc_mulmod_shoup": (u64, [u64, u64, u64, u64]),
        "orc_ntt_forward": (None, [vp, u64p, u32]),
        "orc_ntt_backward": (None, [vp, u64p, u32]),
        "orc_ntt_forward_schoolbook": (None, [vp, u64p, u64p, u32]),
        "orc_apply_galois": (None, [vp, u64p, u32, u64]),
        "orc_apply_galois_non_ntt": (None, [vp, u64p, u32, u64]),
        "orc_rotation_galois": (u64, [vp, i, i]),
        "orc_approx_convert_to_special_basis": (None, [vp, u64p, u64p, u32, u32]),
        "orc_neg_approx_convert_to_normal_basis": (None, [vp, u64p, u64p, u32]),
        "orc_approx_convert_to_special_basis": (None, [vp, u64p, u64p, u32, u32]),
        "orc_neg_approx_convert_to_normal_basis": (None, [vp, u64p, u64p, u32]),
        "orc_mod_up": (i, [vp, u64p, u64p, u32]),
        "orc_mod_down": (i, [vp, u64p, u64p, u32]),
        "orc_rescale": (i, [vp, u64p, u64p, u32]),
        "orc_bconv_table": (C.POINTER(C.c_uint64), [vp, i, C.POINTER(C.c_size_t)]),
        "orc_cross_mult": (None, [vp] + [u64p] * 7 + [u32]),
        "orc_relinearize": (None, [vp] + [u64p] * 5 + [u32]),
        "orc_rescale_ntt": (None, [vp] + [u64p] * 4 + [u32]),
        "orc_mul_relin_rescale": (i, [vp] + [u64p] * 6 + [u32, u64p, u64p]),
        "orc_add": (None, [vp, u64p, u64p, u32]),
        "orc_multiply_plain": (None, [vp, u64p, u64p, u64p, u32]),
        "orc_rotate": (i, [vp, u64p, u64p, u32, u64, u64p, u64p]),
        "orc_rotate_non_ntt": (i, [vp, u64p, u64p, u32, u64, u64p, u64p]),
        "orc_uniform_poly": (None, [vp, u64p, u32, u32, u64]),
        "orc_keygen_sk": (None, [vp, u64p, u64p, u32, u64]),
        "orc_keygen_pk": (None, [vp, u64p, u64p, u64p, u64]),
        "orc_keygen_mult": (None, [vp, u64p, u64p, u64p, u64]),
        "orc_keygen_rot": (None, [vp, u64p, u64p, u64p, u64p, u64, u64]),
        "orc_encrypt": (None, [vp, u64p, u64p, u64p, u32, u64p, u64p, u64]),
        "orc_decrypt": (None, [vp, u64p, u64p, u64p, u32, u64p]),
        "orc_rotate_mixed": (i, [vp, u64p, u64p, u32, u64, u64p, u64p]),
        "orc_keygen_mixed_rot": (None, [vp, u64p, u64p, u64p, u64p, u64, u64]),
        "orc_encrypt_core": (None, [vp, u64p, u64p, u64p, u64p, u64p, u32, u64p, u64p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def gen_primes(logN, count):
    """The repo's deterministic prime table: (primes[count], psi[count]) as uint64 arrays."""
    pr = np.zeros(count, np.uint64)
    ps = np.zeros(count, np.uint64)
    lib.orc_gen_primes(logN, count, _p(pr), _p(ps))
    return pr, ps


class Oracle:
    """CPU oracle context: N = 2**logN, L normal + K special primes."""

    def __init__(self, logN, L, K, primes=None, psi=None):
        if primes is None:
            primes, psi = gen_primes(logN, L + K)
        self.logN, self.L, self.K, self.N = logN, L, K, 1 << logN
        self.primes = np.ascontiguousarray(primes, np.uint64)
        self.psi = np.ascontiguousarray(psi, np.uint64)
        self.h = lib.orc_ctx_create(logN, L, K, _p(self.primes), _p(self.psi))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:  # lib is already gone at interpreter shutdown
            lib.orc_ctx_destroy(self.h)
            self.h = None

    # --- NTT -------------------------------------------------------------------------------
    def ntt_forward(self, limbs, idx):
        """limbs: [n][N] uint64 (modified in place); idx: per-limb global modulus index."""
        limbs = limbs.reshape(-1, self.N)
        for k, m in enumerate(idx):
            lib.orc_ntt_forward(self.h, limbs[k].ctypes.data, int(m))
        return limbs

    def ntt_backward(self, limbs, idx):
        limbs = limbs.reshape(-1, self.N)
        for k, m in enumerate(idx):
            lib.orc_ntt_backward(self.h, limbs[k].ctypes.data, int(m))
        return limbs

    def ntt_forward_schoolbook(self, limb, idx):
        out = np.zeros(self.N, np.uint64)
        lib.orc_ntt_forward_schoolbook(self.h, _p(out), _p(limb), idx)
        return out

    # --- automorphisms ---------------------------------------------------------------------
    def apply_galois(self, poly, g):
        lib.orc_apply_galois(self.h, _p(poly), poly.shape[0], g)
        return poly

    def apply_galois_non_ntt(self, poly, g):
        lib.orc_apply_galois_non_ntt(self.h, _p(poly), poly.shape[0], g)
        return poly

    def rotation_galois(self, offset, right=False):
        return int(lib.orc_rotation_galois(self.h, offset, 1 if right else 0))

    # --- base converter --------------------------------------------------------------------
    def bconv_table(self, which):
        n = C.c_size_t(0)
        ptr = lib.orc_bconv_table(self.h, which, C.byref(n))
        return np.ctypeslib.as_array(ptr, shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint64)

    def approx_convert_to_special_basis(self, op, k):  # base_converter.cpp:276-295
        rop = np.zeros(self.N, np.uint64)
        lib.orc_approx_convert_to_special_basis(self.h, _p(rop), _p(op), op.shape[0], k)
        return rop

    def neg_approx_convert_to_normal_basis(self, sp, j):  # base_converter.cpp:297-311; sp = [K][N] power basis
        rop = np.zeros(self.N, np.uint64)
        lib.orc_neg_approx_convert_to_normal_basis(self.h, _p(rop), _p(sp), j)
        return rop

    def mod_up(self, op):
        Li = op.shape[0]
        rop = np.zeros((Li + self.K, self.N), np.uint64)
        rc = lib.orc_mod_up(self.h, _p(rop), _p(op), Li)
        if rc:
            raise ValueError("mod_up rc=%d" % rc)
        return rop

    def mod_down(self, op):
        L0 = op.shape[0] - self.K
        rop = np.zeros((L0, self.N), np.uint64)
        rc = lib.orc_mod_down(self.h, _p(rop), _p(op), L0)
        if rc:
            raise ValueError("mod_down rc=%d" % rc)
        return rop

    def rescale(self, op):
        n = op.shape[0]
        rop = np.zeros((n - 1, self.N), np.uint64)
        rc = lib.orc_rescale(self.h, _p(rop), _p(op), n)
        if rc:
            raise ValueError("rescale rc=%d" % rc)
        return rop

    # --- evaluator -------------------------------------------------------------------------
    def cross_mult(self, b0, a0, b1, a1):
        Li = b0.shape[0]
        d = [np.zeros((Li, self.N), np.uint64) for _ in range(3)]
        lib.orc_cross_mult(self.h, _p(d[0]), _p(d[1]), _p(d[2]), _p(b0), _p(a0), _p(b1), _p(a1), Li)
        return d

    def relinearize(self, d0, d1, d2, beta, alpha):
        d0, d1 = d0.copy(), d1.copy()
        lib.orc_relinearize(self.h, _p(d0), _p(d1), _p(d2), _p(beta), _p(alpha), d0.shape[0])
        return d0, d1

    def rescale_ntt(self, ct0, ct1):
        Li = ct0.shape[0]
        ct0, ct1 = ct0.copy(), ct1.copy()
        r0 = np.zeros((Li - 1, self.N), np.uint64)
        r1 = np.zeros((Li - 1, self.N), np.uint64)
        lib.orc_rescale_ntt(self.h, _p(r0), _p(r1), _p(ct0), _p(ct1), Li)
        return r0, r1

    def mul_relin_rescale(self, c0b, c0a, c1b, c1a, beta, alpha):
        Li = c0b.shape[0]
        ob = np.zeros((Li - 1, self.N), np.uint64)
        oa = np.zeros((Li - 1, self.N), np.uint64)
        rc = lib.orc_mul_relin_rescale(self.h, _p(ob), _p(oa), _p(c0b), _p(c0a), _p(c1b), _p(c1a), Li, _p(beta), _p(alpha))
        if rc:
            raise ValueError("mul_relin_rescale rc=%d" % rc)
        return ob, oa

    def add(self, rop, op):
        rop = rop.copy()
        lib.orc_add(self.h, _p(rop), _p(op), rop.shape[0])
        return rop

    def multiply_plain(self, bx, ax, pt):
        bx, ax = bx.copy(), ax.copy()
        lib.orc_multiply_plain(self.h, _p(bx), _p(ax), _p(pt), bx.shape[0])
        return bx, ax

    def rotate(self, bx, ax, galois, rk_beta, rk_alpha):
        bx, ax = bx.copy(), ax.copy()
        rc = lib.orc_rotate(self.h, _p(bx), _p(ax), bx.shape[0], galois, _p(rk_beta), _p(rk_alpha))
        if rc:
            raise ValueError("rotate rc=%d" % rc)
        return bx, ax

    def rotate_non_ntt(self, bx, ax, galois, rk_beta, rk_alpha):
        bx, ax = bx.copy(), ax.copy()
        rc = lib.orc_rotate_non_ntt(self.h, _p(bx), _p(ax), bx.shape[0], galois, _p(rk_beta), _p(rk_alpha))
        if rc:
            raise ValueError("rotate_non_ntt rc=%d" % rc)
        return bx, ax

    def rotate_mixed(self, bx, ax, galois, mk_beta, mk_alpha):
        bx, ax = bx.copy(), ax.copy()
        rc = lib.orc_rotate_mixed(self.h, _p(bx), _p(ax), bx.shape[0], galois, _p(mk_beta), _p(mk_alpha))
        if rc:
            raise ValueError("rotate_mixed rc=%d" % rc)
        return bx, ax

    def keygen_mixed_rot(self, sx, ext_sx, galois, seed):
        beta = np.zeros((self.L, self.L + 1, self.N), np.uint64)
        alpha = np.zeros((self.L, self.L + 1, self.N), np.uint64)
        lib.orc_keygen_mixed_rot(self.h, _p(beta), _p(alpha), _p(sx), _p(ext_sx), galois, seed)
        return beta, alpha

    # --- plumbing --------------------------------------------------------------------------
    def uniform_poly(self, nlimbs, seed, first_idx=0):
        poly = np.zeros((nlimbs, self.N), np.uint64)
        lib.orc_uniform_poly(self.h, _p(poly), first_idx, nlimbs, seed)
        return poly

    def keygen_sk(self, seed, hwt=32):
        sx = np.zeros((self.L, self.N), np.uint64)
        ext = np.zeros((self.K, self.N), np.uint64)
        lib.orc_keygen_sk(self.h, _p(sx), _p(ext), hwt, seed)
        return sx, ext

    def keygen_pk(self, sx, seed):
        b = np.zeros((self.L, self.N), np.uint64)
        a = np.zeros((self.L, self.N), np.uint64)
        lib.orc_keygen_pk(self.h, _p(b), _p(a), _p(sx), seed)
        return b, a

    def keygen_mult(self, sx, seed):
        beta = np.zeros((self.L, self.L, self.N), np.uint64)
        alpha = np.zeros((self.L, self.L, self.N), np.uint64)
        lib.orc_keygen_mult(self.h, _p(beta), _p(alpha), _p(sx), seed)
        return beta, alpha

    def keygen_rot(self, sx, ext_sx, galois, seed):
        M = self.L + self.K
        beta = np.zeros((M, self.N), np.uint64)
        alpha = np.zeros((M, self.N), np.uint64)
        lib.orc_keygen_rot(self.h, _p(beta), _p(alpha), _p(sx), _p(ext_sx), galois, seed)
        return beta, alpha

    def encrypt(self, msg, pk_b, pk_a, seed):
        Li = msg.shape[0]
        bx = np.zeros((Li, self.N), np.uint64)
        ax = np.zeros((Li, self.N), np.uint64)
        lib.orc_encrypt(self.h, _p(bx), _p(ax), _p(msg), Li, _p(pk_b), _p(pk_a), seed)
        return bx, ax

    def encrypt_core(self, u, e0m, e1, pk_b, pk_a):
        Li = u.shape[0]
        bx = np.zeros((Li, self.N), np.uint64)
        ax = np.zeros((Li, self.N), np.uint64)
        lib.orc_encrypt_core(self.h, _p(bx), _p(ax), _p(u), _p(e0m), _p(e1), Li, _p(pk_b), _p(pk_a))
        return bx, ax

    def decrypt(self, bx, ax, sx):
        Li = bx.shape[0]
        rop = np.zeros((Li, self.N), np.uint64)
        lib.orc_decrypt(self.h, _p(rop), _p(bx), _p(ax), Li, _p(sx))
        return rop
