"""Limb-dump wire format v1 (no counterpart in the reference): Python <-> C++ mirror agreement,
the committed byte-layout fixture, and rejection of damaged records.  CPU only."""
import io
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from fhe_amd import wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELFTEST = os.path.join(ROOT, "fhe_amd", "host", "bin", "wire_selftest")
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_wire_golden import payload  # noqa: E402


def _record(limbs, **kw):
    b = io.BytesIO()
    wire.dump(b, limbs, **kw)
    return b.getvalue()


def test_python_round_trip_and_layout():
    rng = np.random.default_rng(7)
    a = rng.integers(0, 1 << 62, size=(2, 3, 64), dtype=np.uint64)
    raw = _record(a, kind=wire.CIPHER, scale=2.0 ** 40)
    assert len(raw) == 40 + a.size * 8 + 8 and raw[:4] == b"fHEw"
    assert struct.unpack_from("<HHIIII", raw, 4) == (1, wire.CIPHER, 6, 3, 2, wire.NTT_DOMAIN)
    meta, back = wire.load(io.BytesIO(raw))
    assert np.array_equal(back, a)
    assert meta == {"kind": wire.CIPHER, "logN": 6, "n_limbs": 3, "n_polys": 2, "ntt": True, "has_mul_aux": False,
                    "scale": 2.0 ** 40, "galois": 0}
    # a single polynomial may be passed as [n_limbs][N]
    meta, back = wire.load(io.BytesIO(_record(a[0], flags=0)))
    assert meta["kind"] == wire.POLY and not meta["ntt"] and np.array_equal(back[0], a[0])


def test_checksum_is_the_documented_sum():
    w = np.array([5, 1 << 63, 12345678901234567], dtype=np.uint64)
    want = sum(int(x) * (((k + 1) * 0x9E3779B97F4A7C15) % (1 << 64) | 1) for k, x in enumerate(w)) % (1 << 64)
    assert wire.checksum(w) == want


def test_golden_fixture_bytes():
    with open(os.path.join(ROOT, "tests", "golden", "wire_v1.bin"), "rb") as f:
        m1, a1 = wire.load(f)
        m2, a2 = wire.load(f)
        assert f.read() == b""
    assert m1["kind"] == wire.CIPHER and m1["has_mul_aux"] and m1["scale"] == 2.0 ** 62 and m1["logN"] == 4
    assert np.array_equal(a1, payload(3, 2, 4, 1))
    assert m2["kind"] == wire.KEY and m2["galois"] == 3 and np.array_equal(a2, payload(2, 4, 4, 2))


@pytest.mark.parametrize("damage", ["magic", "version", "payload", "truncated", "shape"])
def test_damaged_records_are_rejected(damage):
    raw = bytearray(_record(np.arange(2 * 2 * 16, dtype=np.uint64).reshape(2, 2, 16), kind=wire.CIPHER))
    if damage == "magic":
        raw[0] ^= 1
    elif damage == "version":
        raw[4] = 9
    elif damage == "payload":
        raw[40 + 17] ^= 0x40
    elif damage == "truncated":
        raw = raw[:-9]
    elif damage == "shape":
        raw[12:16] = struct.pack("<I", 0)
    with pytest.raises(ValueError):
        wire.load(io.BytesIO(bytes(raw)))


def test_dump_rejects_bad_arguments():
    with pytest.raises(ValueError):
        wire.dump(io.BytesIO(), np.zeros((2, 3, 48), dtype=np.uint64))  # N not a power of two
    with pytest.raises(ValueError):
        wire.dump(io.BytesIO(), np.zeros((3, 2, 16), dtype=np.uint64), kind=wire.CIPHER)  # 3 polys without HAS_MUL_AUX
    with pytest.raises(ValueError):
        wire.dump(io.BytesIO(), np.zeros((2, 2, 16), dtype=np.uint64), kind=wire.CIPHER, scale=0.0)


def test_cpp_mirror_reads_and_writes_the_same_bytes(tmp_path):
    assert os.path.exists(SELFTEST), "run __graft_entry__.build() (make -C fhe_amd/host)"
    rng = np.random.default_rng(11)
    for n_polys, flags in ((2, wire.NTT_DOMAIN), (3, wire.NTT_DOMAIN | wire.HAS_MUL_AUX)):
        a = rng.integers(0, 1 << 62, size=(n_polys, 3, 256), dtype=np.uint64)
        src, dst = tmp_path / "in.bin", tmp_path / "out.bin"
        src.write_bytes(_record(a, kind=wire.CIPHER, flags=flags, scale=3.5e18))
        r = subprocess.run([SELFTEST, "8", str(src), str(dst)], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        assert "limbs=3 canonical=%d" % (n_polys == 2) in r.stdout
        out = dst.read_bytes()
        assert out[:len(src.read_bytes())] == src.read_bytes()  # C++ writes the identical record
        with open(dst, "rb") as f:
            wire.load(f)
            meta, bx = wire.load(f)
        assert meta["kind"] == wire.POLY and np.array_equal(bx[0], a[0])
    # the C++ reader rejects a damaged payload, a wrong ring degree and a non-ciphertext record
    bad = bytearray(src.read_bytes()); bad[100] ^= 1
    (tmp_path / "bad.bin").write_bytes(bytes(bad))
    r = subprocess.run([SELFTEST, "8", str(tmp_path / "bad.bin"), str(dst)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "checksum" in r.stderr
    r = subprocess.run([SELFTEST, "9", str(src), str(dst)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "ring degree" in r.stderr
    (tmp_path / "key.bin").write_bytes(_record(a[:2], kind=wire.KEY))
    r = subprocess.run([SELFTEST, "8", str(tmp_path / "key.bin"), str(dst)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "not a ciphertext" in r.stderr
