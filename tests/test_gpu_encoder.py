"""GPU: the CKKS encoder on the device (fhe_hip_encode / fhe_hip_decode) against the numpy restatement of
the reference's Encoder (oracle/encoder.py <- src/encoder.cpp).  Floating point (complex double on both sides); tolerances:
  encode: coefficients are integers trunc(x * scale / N); the two sides differ by the rounding of a 16-stage double FFT, i.e. by
          at most max(2, 2^-40 * scale * max|v|) per coefficient (compared as centred integers);
  decode: slot values within 1e-9 * (1 + max|v|).
PARITY UNPINNED like the rest of the encoder: the reference holds no fixture for it.  Full size (N = 2^16) is covered by what the
embedding MEANS: decode(encode(v)) = v, and the negacyclic product of two encodings decodes to the slot-wise product."""
import numpy as np
import pytest

import fhe_amd
from oracle import oracle as orc
from oracle.encoder import EncoderOracle

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _centred(row, p):
    r = row.astype(object)
    return np.array([int(v) - p if int(v) > p // 2 else int(v) for v in r], dtype=object)


# (L = 30 runs the register-resident Garner lift at its full unrolled depth, L = 35 the LDS form for more than 32 limbs)
@pytest.mark.parametrize("logN,L,scale_bits,sp", [(10, 3, 30, True), (12, 4, 50, True), (13, 3, 62, True), (11, 3, 40, False),
                                                  (10, 30, 45, True), (10, 35, 45, True), (10, 33, 50, False)])
def test_encode_decode_against_the_oracle(logN, L, scale_bits, sp):
    if sp:
        pr, ps = orc.gen_primes(logN, L + 1)
    else:
        from test_gpu_parity import generic_primes
        pr, ps = generic_primes(logN, L + 1)
    g = fhe_amd.HipContext(logN, L, 1, pr, ps)
    e = EncoderOracle(logN)
    rng = np.random.default_rng(logN)
    scale = 2.0 ** scale_bits
    B = 2
    vals = rng.uniform(-8, 8, (B, e.slots)) + 1j * rng.uniform(-8, 8, (B, e.slots))
    vals[1, e.slots // 2:] = 0           # a short vector, zero padded
    msg = np.zeros((B, L, e.N), np.uint64)
    g.encode(msg, np.ascontiguousarray(vals), L, scale, batch=B)          # host buffers
    dmsg = torch.zeros((B, L, e.N), dtype=torch.int64, device="cuda")
    g.encode(dmsg, torch.from_numpy(vals).cuda(), L, scale, batch=B)      # device buffers
    assert np.array_equal(dmsg.cpu().numpy().view(np.uint64), msg)
    tol = max(2, int(2.0 ** -40 * scale * 8 * np.sqrt(2)))
    for b in range(B):
        want = e.encode(vals[b], scale, pr[:L])
        for l in range(L):
            p = int(pr[l])
            diff = _centred(msg[b, l], p) - _centred(want[l], p)
            assert max(abs(int(x)) for x in diff) <= tol, (b, l)
        # limbs of one coefficient are residues of ONE integer: every limb agrees with limb 0 after centring (values far below p)
        if scale_bits < 55:
            assert all(np.array_equal(_centred(msg[b, l], int(pr[l])), _centred(msg[b, 0], int(pr[0]))) for l in range(1, L))
    # decode: the oracle's encodings through the device decoder, and the device's own encodings back
    slots = np.zeros((B, e.slots), np.complex128)
    g.decode(slots, msg, L, scale, batch=B)
    assert np.abs(slots - vals).max() < 1e-9 * 9 if scale_bits >= 40 else np.abs(slots - vals).max() < 1e-6
    want_msg = np.stack([e.encode(vals[b], scale, pr[:L]) for b in range(B)])
    want_slots = np.stack([e.decode(want_msg[b], pr[:L], scale) for b in range(B)])
    got = torch.zeros((B, e.slots), dtype=torch.complex128, device="cuda")
    g.decode(got, torch.from_numpy(want_msg.view(np.int64)).cuda(), L, scale, batch=B)
    assert np.abs(got.cpu().numpy() - want_slots).max() < 1e-9 * 9
    # fewer limbs than the context holds, negative values exercise the centred lift
    one = np.zeros((1, 1, e.N), np.uint64)
    g.encode(one, np.ascontiguousarray(-vals[:1]), 1, 2.0 ** 30, batch=1)
    back = np.zeros((1, e.slots), np.complex128)
    g.decode(back, one, 1, 2.0 ** 30, batch=1)
    assert np.abs(back + vals[:1]).max() < 1e-5
    g.close()


def test_encoder_argument_errors_and_large_magnitudes():
    logN, L = 10, 3
    pr, ps = orc.gen_primes(logN, L + 1)
    g = fhe_amd.HipContext(logN, L, 1, pr, ps)
    N = 1 << logN
    msg = np.zeros((1, L, N), np.uint64)
    v = np.ones((1, N // 2), np.complex128)
    for bad_scale in (0.0, -1.0, float("inf")):
        with pytest.raises(fhe_amd.FheHipError) as ei:
            g.encode(msg, v, L, bad_scale)
        assert ei.value.code == -1
    with pytest.raises(fhe_amd.FheHipError) as ei:
        g.encode(np.zeros((1, L + 1, N), np.uint64), v, L + 1, 2.0 ** 30)
    assert ei.value.code == -2
    # more values than slots: the first N/2 are used, like the reference (encoder.cpp:87-91)
    many = np.concatenate([v, 5 * v], axis=1)
    g.encode(msg, many, L, 2.0 ** 30)
    only = np.zeros_like(msg)
    g.encode(only, v, L, 2.0 ** 30)
    assert np.array_equal(msg, only)
    # |x * scale / N| beyond 2^63: the mantissa path of trunc(x) mod p (mpz_set_d semantics).  A constant vector goes through the
    # butterflies exactly (sums of equal values, differences zero), so coefficient 0 is exactly 3 * 2^120 and the rest exactly 0
    big = np.full((1, N // 2), 3.0, np.complex128)
    scale = 2.0 ** 120
    g.encode(msg, big, L, scale)
    want = EncoderOracle(logN).encode(big[0], scale, pr[:L])
    assert np.array_equal(msg[0], want)
    for l in range(L):
        assert int(msg[0, l, 0]) == (3 << 120) % int(pr[l]) and not msg[0, l, 1:].any()
    g.encode(msg, -big, L, scale)
    for l in range(L):
        assert int(msg[0, l, 0]) == (-(3 << 120)) % int(pr[l])
    g.close()


def test_full_size_round_trip_and_slotwise_product():
    """N = 2^16: decode(encode(v)) = v, and INTT(NTT(encode a) * NTT(encode b)) decodes at scale^2 to a * b (the canonical embedding
    is a ring homomorphism): the encoder and the limb NTT pinned by each other's mathematics, no second implementation involved."""
    logN, L = 16, 3
    pr, ps = orc.gen_primes(logN, L + 1)
    g = fhe_amd.HipContext(logN, L, 1, pr, ps)
    N = 1 << logN
    gen = torch.Generator(device="cuda").manual_seed(16)
    a = (torch.rand((1, N // 2), dtype=torch.float64, device="cuda", generator=gen) * 4 - 2).to(torch.complex128)
    a = a + 1j * (torch.rand((1, N // 2), dtype=torch.float64, device="cuda", generator=gen) * 4 - 2)
    b = (torch.rand((1, N // 2), dtype=torch.float64, device="cuda", generator=gen) * 4 - 2).to(torch.complex128)
    scale = 2.0 ** 40
    ma = torch.zeros((1, L, N), dtype=torch.int64, device="cuda")
    mb = torch.zeros_like(ma)
    g.encode(ma, a, L, scale)
    g.encode(mb, b, L, scale)
    back = torch.zeros((1, N // 2), dtype=torch.complex128, device="cuda")
    g.decode(back, ma, L, scale)
    assert float((back - a).abs().max()) < 1e-8
    idx = list(range(L))
    g.ntt_forward(ma.view(L, N), idx)
    g.ntt_forward(mb.view(L, N), idx)
    g.multiply_plain(torch.stack([ma.view(L, N), ma.view(L, N)])[None].contiguous(), mb.view(L, N), L)  # warm path: ct-shaped operand
    prod = torch.stack([ma.view(L, N), ma.view(L, N)])[None].contiguous()
    g.multiply_plain(prod, mb.view(L, N), L)
    p0 = prod[0, 0].contiguous()
    g.ntt_backward(p0, idx)
    g.decode(back, p0.view(1, L, N), L, scale * scale)
    assert float((back - a * b).abs().max()) < 1e-6
    g.close()
