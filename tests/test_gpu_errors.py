"""GPU: error conventions and degenerate inputs of the C-ABI (reference: return false / std::invalid_argument)."""
import numpy as np
import pytest

import fhe_amd
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _ctx(logN=10, L=3, K=2):
    pr, ps = orc.gen_primes(logN, L + K)
    return orc.Oracle(logN, L, K, pr, ps), fhe_amd.HipContext(logN, L, K, pr, ps), pr, ps


def test_context_creation_rejects_bad_tables():
    pr, ps = orc.gen_primes(10, 3)
    bad = pr.copy(); bad[1] = bad[1] - 2  # not 1 mod 2N (and composite or not, the root check fails)
    with pytest.raises(fhe_amd.FheHipError) as e:
        fhe_amd.HipContext(10, 2, 1, bad, ps)
    assert e.value.code in (-1, -2)
    badpsi = ps.copy(); badpsi[0] = 1  # not a primitive 2N-th root
    with pytest.raises(fhe_amd.FheHipError):
        fhe_amd.HipContext(10, 2, 1, pr, badpsi)
    with pytest.raises(fhe_amd.FheHipError):
        fhe_amd.HipContext(17, 2, 1, pr, ps)  # logN outside [10,16]
    big = pr.copy(); big[0] = np.uint64((1 << 62) + 1)
    with pytest.raises(fhe_amd.FheHipError):
        fhe_amd.HipContext(10, 2, 1, big, ps)  # prime >= 2^62
    dup = pr.copy(); dup[2] = dup[0]
    dps = ps.copy(); dps[2] = dps[0]
    with pytest.raises(fhe_amd.FheHipError) as e:
        fhe_amd.HipContext(10, 2, 1, dup, dps)  # duplicate prime: no inverse modulo itself
    assert e.value.code == -2
    # a table whose primes differ by a factor >= 4 breaks the lazy-reduction bound across primes
    small_p = 12289 * 2048 * 4 + 1  # search a prime = 1 mod 2048 near 2^27
    while not orc.lib.orc_is_prime(small_p):
        small_p += 2048
    x = 2
    while pow(x, (small_p - 1) // 2, small_p) != small_p - 1:
        x += 1
    wide = pr.copy(); wide[2] = np.uint64(small_p)
    wps = ps.copy(); wps[2] = np.uint64(pow(x, (small_p - 1) >> 11, small_p))
    with pytest.raises(fhe_amd.FheHipError) as e:
        fhe_amd.HipContext(10, 2, 1, wide, wps)
    assert e.value.code == -2


def test_mod_up_without_special_primes_copies_the_normal_limbs():
    pr, ps = orc.gen_primes(10, 3)
    g = fhe_amd.HipContext(10, 3, 0, pr, ps)
    x = torch.randint(0, 1 << 60, (2, 3, 1 << 10), dtype=torch.int64, device="cuda")
    rop = torch.zeros_like(x)
    g.mod_up(rop, x, 3, batch=2)
    g.sync()
    assert torch.equal(rop, x)
    hx = x.cpu().numpy().view(np.uint64).copy()
    hr = np.zeros_like(hx)
    g.mod_up(hr, hx, 3, batch=2)
    assert np.array_equal(hr, hx)


def test_python_binding_checks_buffer_sizes():
    o, g, pr, ps = _ctx()
    N = o.N
    short = torch.zeros((2, N), dtype=torch.int64, device="cuda")
    with pytest.raises(ValueError):
        g.ntt_forward(short, [0, 1, 2])           # three limbs asked for, two present
    rop = torch.zeros((1, 3, N), dtype=torch.int64, device="cuda")
    with pytest.raises(ValueError):
        g.mod_up(rop, rop, 3, batch=1)            # rop must hold Li + K limbs
    out = torch.zeros((1, 2, 2, N), dtype=torch.int64, device="cuda")
    ct = torch.zeros((1, 2, 3, N), dtype=torch.int64, device="cuda")
    with pytest.raises(ValueError):
        g.mul_relin_rescale(out, ct, ct, 3, batch=2)  # batch larger than the buffers


def test_empty_batches_are_no_ops_and_keys_are_required():
    o, g, pr, ps = _ctx()
    N, Li = o.N, 3
    z = torch.zeros((0, N), dtype=torch.int64, device="cuda")
    g.ntt_forward(z, [])
    ct = torch.randint(0, 1 << 60, (1, 2, Li, N), dtype=torch.int64, device="cuda")
    before = ct.clone()
    g.add(ct, ct, Li, batch=0)
    g.multiply_plain(ct, ct[0, 0], Li, batch=0)
    assert torch.equal(ct, before)
    out = torch.zeros((1, 2, Li - 1, N), dtype=torch.int64, device="cuda")
    with pytest.raises(fhe_amd.FheHipError) as e:
        g.mul_relin_rescale(out, ct, ct, Li, batch=1)  # no MultKey loaded
    assert e.value.code == -3
    with pytest.raises(fhe_amd.FheHipError) as e:
        g.rotate(3, ct, Li, batch=1)  # no RotKey loaded
    assert e.value.code == -3


def test_limb_count_and_galois_validation():
    o, g, pr, ps = _ctx()
    N = o.N
    ct = torch.zeros((1, 2, 4, N), dtype=torch.int64, device="cuda")
    with pytest.raises(fhe_amd.FheHipError) as e:
        g.add(ct, ct, 6, batch=1)  # more limbs than the context has moduli
    assert e.value.code == -2
    one = torch.zeros((1, 1, N), dtype=torch.int64, device="cuda")
    with pytest.raises(fhe_amd.FheHipError) as e:
        g.rescale(one, one, 1, batch=1)  # "Already in the last moduli" (base_converter.cpp:250)
    assert e.value.code == -2
    with pytest.raises(fhe_amd.FheHipError):
        g.apply_galois(ct[0, 0], 3, 4, batch=1)  # even galois element
    with pytest.raises(fhe_amd.FheHipError):
        g.apply_galois(ct[0, 0], 3, 2 * N + 1, batch=1)  # >= 2N
    with pytest.raises(fhe_amd.FheHipError):
        g.ntt_forward(ct[0, 0], [0, 1, 9])  # modulus index out of range
