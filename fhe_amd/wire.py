"""Limb-dump wire format, version 1.  The reference has no serialization; this is the simple
format the survey asks to define first, shared with the C++ mirror (fhe_amd/host/include/fHE/wire.hpp) and laid out in
INTEGRATION.md.  One record:

    offset  size  field
    0       4     magic "fHEw"
    4       2     version (1)
    6       2     kind: 1 polynomial, 2 ciphertext (bx, ax[, mul_aux]), 3 key material
    8       4     logN
    12      4     n_limbs   limbs per polynomial
    16      4     n_polys   polynomials in the record
    20      4     flags     bit 0: NTT domain, bit 1: ciphertext carries mul_aux (not canonical)
    24      8     scale     float64 (ciphertexts; 1.0 otherwise)
    32      8     galois    uint64 (rotation keys; 0 otherwise)
    40      ...   n_polys * n_limbs * 2^logN little-endian uint64 limbs, limb-major (the C-ABI's layout)
    end-8   8     checksum  sum_k word[k] * ((k+1) * 0x9E3779B97F4A7C15 | 1)  mod 2^64

Pure host-side plumbing (numpy), no GPU involved."""
import struct

import numpy as np

MAGIC = b"fHEw"
VERSION = 1
POLY, CIPHER, KEY = 1, 2, 3
NTT_DOMAIN, HAS_MUL_AUX = 1, 2
_HDR = struct.Struct("<4sHHIIIIdQ")
assert _HDR.size == 40


def checksum(words):
    """Position-weighted word sum mod 2^64 over a flat uint64 array."""
    w = np.ascontiguousarray(words, dtype="<u8").reshape(-1)
    with np.errstate(over="ignore"):
        k = np.arange(1, w.size + 1, dtype=np.uint64)
        weights = (k * np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1)
        return int((w * weights).sum(dtype=np.uint64))


def dump(f, limbs, kind=POLY, logN=None, flags=NTT_DOMAIN, scale=1.0, galois=0):
    """Write one record.  `limbs` is uint64 [n_polys][n_limbs][N] (or [n_limbs][N] for a single polynomial)."""
    a = np.ascontiguousarray(limbs, dtype="<u8")
    if a.ndim == 2:
        a = a[None]
    if a.ndim != 3 or a.size == 0:
        raise ValueError("wire: limbs must be [n_polys][n_limbs][N], non-empty")
    n_polys, n_limbs, N = a.shape
    if N & (N - 1) or (logN is not None and N != 1 << logN):
        raise ValueError("wire: N must be 2^logN")
    if kind == CIPHER and n_polys != (3 if flags & HAS_MUL_AUX else 2):
        raise ValueError("wire: a ciphertext record holds bx, ax and, with HAS_MUL_AUX, mul_aux")
    if not scale > 0:
        raise ValueError("wire: scale must be positive")
    f.write(_HDR.pack(MAGIC, VERSION, kind, N.bit_length() - 1, n_limbs, n_polys, flags, float(scale), int(galois)))
    f.write(a.tobytes())
    f.write(struct.pack("<Q", checksum(a)))


def load(f):
    """Read one record; returns (meta dict, uint64 array [n_polys][n_limbs][N]).  Raises ValueError on a bad record."""
    raw = f.read(_HDR.size)
    if len(raw) != _HDR.size:
        raise ValueError("wire: truncated header")
    magic, version, kind, logN, n_limbs, n_polys, flags, scale, galois = _HDR.unpack(raw)
    if magic != MAGIC:
        raise ValueError("wire: bad magic")
    if version != VERSION:
        raise ValueError("wire: unsupported version %d" % version)
    if not (0 < logN <= 17 and 0 < n_limbs < 255 and 0 < n_polys <= 4096) or kind not in (POLY, CIPHER, KEY):
        raise ValueError("wire: bad shape")
    nbytes = n_polys * n_limbs * (8 << logN)
    body = f.read(nbytes)
    tail = f.read(8)
    if len(body) != nbytes or len(tail) != 8:
        raise ValueError("wire: truncated payload")
    a = np.frombuffer(body, dtype="<u8").reshape(n_polys, n_limbs, 1 << logN).copy()
    if struct.unpack("<Q", tail)[0] != checksum(a):
        raise ValueError("wire: checksum mismatch")
    meta = {"kind": kind, "logN": logN, "n_limbs": n_limbs, "n_polys": n_polys, "ntt": bool(flags & NTT_DOMAIN),
            "has_mul_aux": bool(flags & HAS_MUL_AUX), "scale": scale, "galois": galois}
    return meta, a
