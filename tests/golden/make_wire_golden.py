"""Writes tests/golden/wire_v1.bin: one ciphertext record and one key record of the limb-dump wire format v1
(fhe_amd/wire.py) with a closed-form payload, so that a later change of the byte layout shows up as a test failure.
Run from the repository root:  python tests/golden/make_wire_golden.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fhe_amd import wire  # noqa: E402


def payload(n_polys, n_limbs, logN, salt):
    k = np.arange(n_polys * n_limbs * (1 << logN), dtype=np.uint64)
    with np.errstate(over="ignore"):
        return ((k + np.uint64(salt)) * np.uint64(0xD6E8FEB86659FD93) >> np.uint64(2)).reshape(n_polys, n_limbs, 1 << logN)


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wire_v1.bin")
    with open(out, "wb") as f:
        wire.dump(f, payload(3, 2, 4, 1), kind=wire.CIPHER, flags=wire.NTT_DOMAIN | wire.HAS_MUL_AUX, scale=2.0 ** 62)
        wire.dump(f, payload(2, 4, 4, 2), kind=wire.KEY, galois=3)
    print("wrote", out, os.path.getsize(out), "bytes")
