"""CPU checks of the encoder restatement (oracle/encoder.py, reference src/encoder.cpp): properties the canonical embedding
must have, independent of any implementation."""
import numpy as np

from oracle import oracle as orc
from oracle.encoder import EncoderOracle


def _centered(row, p):
    return np.array([int(v) - p if int(v) > p // 2 else int(v) for v in row], dtype=object)


def test_constant_vector_encodes_to_constant_polynomial():
    logN = 7
    pr, _ = orc.gen_primes(logN, 2)
    e = EncoderOracle(logN)
    m = e.encode(np.full(e.slots, 3.0), 2.0 ** 30, pr)
    c = _centered(m[0], int(pr[0]))
    assert abs(int(c[0]) - 3 * (1 << 30)) <= 2 and max(abs(int(v)) for v in c[1:]) <= 2


def test_round_trip_and_slotwise_product():
    logN = 6
    pr, _ = orc.gen_primes(logN, 3)
    e = EncoderOracle(logN)
    rng = np.random.default_rng(3)
    a = rng.uniform(-4, 4, e.slots) + 1j * rng.uniform(-4, 4, e.slots)
    b = rng.uniform(-4, 4, e.slots)
    sc = 2.0 ** 25
    ma, mb = e.encode(a, sc, pr), e.encode(b, sc, pr)
    assert np.abs(e.decode(ma, pr, sc) - a).max() < 1e-5
    # negacyclic product of the encodings decodes to the slot-wise product at scale^2
    N = e.N
    prod = np.zeros((3, N), np.uint64)
    for l in range(3):
        p = int(pr[l])
        x, y = [int(v) for v in ma[l]], [int(v) for v in mb[l]]
        r = [0] * N
        for i in range(N):
            if x[i] == 0:
                continue
            for j in range(N):
                k = i + j
                if k < N:
                    r[k] = (r[k] + x[i] * y[j]) % p
                else:
                    r[k - N] = (r[k - N] - x[i] * y[j]) % p
        prod[l] = np.array(r, dtype=np.uint64)
    assert np.abs(e.decode(prod, pr, sc * sc) - a * b).max() < 1e-4


def test_rotation_by_galois_three_shifts_slots_left():
    logN = 6
    pr, _ = orc.gen_primes(logN, 2)
    e = EncoderOracle(logN)
    v = np.arange(e.slots, dtype=np.float64) + 1
    sc = 2.0 ** 30
    m = e.encode(v, sc, pr)
    # X -> X^3 on coefficients (src/yell.cpp:29-53)
    N = e.N
    rot = np.zeros_like(m)
    for l in range(2):
        p = int(pr[l])
        for d in range(N):
            t = int(m[l][d])
            k = (d * 3) % (2 * N)
            if k < N:
                rot[l][k] = t
            else:
                rot[l][k - N] = (p - t) % p
    got = e.decode(rot, pr, sc).real
    assert np.abs(got - np.roll(v, -1)).max() < 1e-5
