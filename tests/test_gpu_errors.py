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


def test_debug_build_counts_range_violations_and_sees_none_on_valid_inputs():
    """libfhe_hip_debug.so (make -C fhe_amd/csrc DEBUG=1): residue-range checks inside the kernels.  Loaded through FHE_HIP_LIB in
    a child process (the library path is read once per process): a full multiply+relin+rescale, a rotation and NTT round trips on
    valid inputs raise no violation and still match the oracle; an inverse NTT fed values >= 2p is counted, not trapped."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "fhe_amd", "lib", "libfhe_hip_debug.so")
    assert os.path.exists(lib), "run __graft_entry__.build() (make -C fhe_amd/csrc DEBUG=1)"
    code = textwrap.dedent('''
        import numpy as np, torch, fhe_amd
        from oracle import oracle as orc
        logN, L, K, Li = 12, 4, 4, 4
        pr, ps = orc.gen_primes(logN, L + K)
        o, g = orc.Oracle(logN, L, K, pr, ps), fhe_amd.HipContext(logN, L, K, pr, ps)
        assert g.debug_violations()[2], "not a debug build"
        rng = np.random.default_rng(5)
        poly = lambda n=Li: np.stack([rng.integers(0, int(pr[i]), o.N, dtype=np.uint64) for i in range(n)])
        dev = lambda a: torch.from_numpy(a.view(np.int64)).cuda()
        host = lambda t: t.cpu().numpy().view(np.uint64)
        x = poly(); idx = list(range(Li))
        d = dev(x); g.ntt_forward(d, idx); g.ntt_backward(d, idx)
        assert np.array_equal(host(d), x)
        sx, ext = o.keygen_sk(seed=1)
        beta, alpha = o.keygen_mult(sx, seed=2)
        g.load_mult_key(dev(beta), dev(alpha))
        c0 = np.stack([poly(), poly()]); c1 = np.stack([poly(), poly()])
        wb, wa = o.mul_relin_rescale(c0[0], c0[1], c1[0], c1[1], beta, alpha)
        out = torch.zeros((1, 2, Li - 1, o.N), dtype=torch.int64, device="cuda")
        g.mul_relin_rescale(out, dev(c0[None]), dev(c1[None]), Li, batch=1)
        h = host(out)
        assert np.array_equal(h[0, 0], wb) and np.array_equal(h[0, 1], wa)
        gal = o.rotation_galois(1)
        rkb, rka = o.keygen_rot(sx, ext, gal, seed=3)
        g.load_rot_key(gal, dev(rkb), dev(rka))
        ct = dev(c0[None].copy()); g.rotate(gal, ct, Li, batch=1)
        rb, ra = o.rotate(c0[0], c0[1], gal, rkb, rka)
        assert np.array_equal(host(ct)[0, 0], rb) and np.array_equal(host(ct)[0, 1], ra)
        n, site, en = g.debug_violations()
        assert n == 0, (n, site)
        bad = np.full((1, o.N), (1 << 64) - 5, np.uint64)   # >= 2p: outside the inverse NTT's input contract
        db = dev(bad); g.ntt_backward(db, [0]); g.sync()
        n, site, en = g.debug_violations()
        assert n > 0 and site == 4, (n, site)
        assert g.debug_violations()[0] == 0   # reading resets
        print("debug build OK", n)
    ''')
    env = dict(os.environ, FHE_HIP_LIB=lib)
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0 and "debug build OK" in r.stdout, r.stdout + r.stderr


def test_ordinary_build_reports_debug_checks_disabled():
    o, g, pr, ps = _ctx()
    assert g.debug_violations() == (0, 0, False)
