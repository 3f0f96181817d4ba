"""Pins the CPU oracle (oracle/fhe_oracle.c) with independent mathematics.

The reference ships no golden vectors and its arithmetic library (yell) is absent, so parity at
the yell boundary is unpinned (see oracle/fhe_oracle.h); these tests anchor the restatement on
O(N^2) schoolbook evaluation, Python big-integer CRT and decrypt round trips instead.
"""
import numpy as np
import pytest

from oracle import oracle as orc


def brv(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2)


def negacyclic(a, b, p):
    n = len(a)
    r = [0] * n
    for i in range(n):
        if a[i] == 0:
            continue
        for j in range(n):
            k = i + j
            if k < n:
                r[k] = (r[k] + a[i] * b[j]) % p
            else:
                r[k - n] = (r[k - n] - a[i] * b[j]) % p
    return r


def test_prime_table_contract():
    pr, ps = orc.gen_primes(16, 60)
    assert len(set(pr.tolist())) == 60
    prev = 1 << 62
    for p, s in zip(pr.tolist(), ps.tolist()):
        assert p < prev and p < (1 << 62) and p > (1 << 61)
        assert p % (1 << 32) == 1
        assert orc.lib.orc_is_prime(p) == 1
        assert pow(s, 1 << 16, p) == p - 1  # psi^N = -1: primitive 2N-th root
        prev = p
    # psi for a smaller ring is a power of the larger one's (same non-residue rule)
    pr13, ps13 = orc.gen_primes(13, 4)
    assert pr13.tolist() == pr[:4].tolist()
    assert [pow(int(s), 8, int(p)) for s, p in zip(ps[:4], pr[:4])] == ps13.tolist()


def test_scalar_ops_vs_python_ints():
    rng = np.random.default_rng(1)
    pr, _ = orc.gen_primes(13, 3)
    for p in pr.tolist():
        for _ in range(200):
            a, b = int(rng.integers(0, p)), int(rng.integers(0, p))
            assert orc.lib.orc_mulmod(a, b, p) == a * b % p
            ys = orc.lib.orc_shoupify(b, p)
            assert ys == (b << 64) // p
            x = int(rng.integers(0, 1 << 63)) * 2 + 1  # any 64-bit x
            assert orc.lib.orc_mulmod_shoup(x, b, ys, p) == x * b % p
            if a:
                assert orc.lib.orc_invmod(a, p) * a % p == 1


@pytest.mark.parametrize("logN", [4, 6, 8])
def test_ntt_forward_matches_schoolbook_definition(logN):
    o = orc.Oracle(logN, 2, 1)
    rng = np.random.default_rng(logN)
    for idx in range(3):
        p = int(o.primes[idx])
        x = rng.integers(0, p, o.N, dtype=np.uint64)
        want = o.ntt_forward_schoolbook(x, idx)
        # independent python evaluation at psi^(2*brv(d)+1)
        psi = int(o.psi[idx])
        for d in (0, 1, o.N - 1):
            root = pow(psi, 2 * brv(d, logN) + 1, p)
            assert int(want[d]) == sum(int(x[k]) * pow(root, k, p) for k in range(o.N)) % p
        got = o.ntt_forward(x.copy(), [idx])[0]
        assert np.array_equal(got, want)


@pytest.mark.parametrize("logN", [10, 13, 14, 15, 16])
def test_ntt_roundtrip_and_foreign_residues(logN):
    o = orc.Oracle(logN, 2, 1)
    rng = np.random.default_rng(logN)
    p0, p1 = int(o.primes[0]), int(o.primes[1])
    x = rng.integers(0, p0, o.N, dtype=np.uint64)
    y = o.ntt_forward(x.copy(), [0])
    assert int(y.max()) < p0
    assert np.array_equal(o.ntt_backward(y.copy(), [0])[0], x)
    # residues of another prime (>= p allowed): NTT_1(x) == NTT_1(x mod p1)   (evaluator.cpp:667-668)
    a = o.ntt_forward(x.copy(), [1])[0]
    b = o.ntt_forward((x % np.uint64(p1)).copy(), [1])[0]
    assert np.array_equal(a, b) and int(a.max()) < p1


def test_ntt_is_negacyclic_convolution():
    o = orc.Oracle(6, 1, 1)
    p = int(o.primes[0])
    rng = np.random.default_rng(7)
    a = rng.integers(0, p, o.N, dtype=np.uint64)
    b = rng.integers(0, p, o.N, dtype=np.uint64)
    fa, fb = o.ntt_forward(a.copy(), [0])[0], o.ntt_forward(b.copy(), [0])[0]
    prod = np.array([int(u) * int(v) % p for u, v in zip(fa, fb)], dtype=np.uint64)
    got = o.ntt_backward(prod, [0])[0]
    assert got.tolist() == negacyclic(a.tolist(), b.tolist(), p)


@pytest.mark.parametrize("g", [3, 9, 5, 2 * 64 - 1])
def test_galois_ntt_domain_equals_coefficient_domain(g):
    o = orc.Oracle(6, 2, 1)
    rng = np.random.default_rng(g)
    x = np.stack([rng.integers(0, int(o.primes[i]), o.N, dtype=np.uint64) for i in range(2)])
    # coefficient-domain automorphism, checked against the definition f(X) -> f(X^g) mod X^N+1
    c = o.apply_galois_non_ntt(x.copy(), g)
    for i in range(2):
        p = int(o.primes[i])
        want = [0] * o.N
        for d in range(o.N):
            e = d * g % (2 * o.N)
            if e < o.N:
                want[e] = int(x[i, d])
            else:
                want[e - o.N] = (p - int(x[i, d])) % p
        assert c[i].tolist() == want
    # NTT(sigma_g(x)) == apply_galois(NTT(x))   (src/yell.cpp:4-25 vs :29-53)
    lhs = o.ntt_forward(c.copy(), [0, 1])
    rhs = o.apply_galois(o.ntt_forward(x.copy(), [0, 1]).copy(), g)
    assert np.array_equal(lhs, rhs)


def test_rotation_galois_rule():
    o = orc.Oracle(6, 2, 1)
    # LEFT rotation by k -> 3^k mod 2N ; RIGHT by k -> 3^(N/2-k)  (rot_key.cpp:6-24)
    assert o.rotation_galois(1) == 3
    assert o.rotation_galois(5) == pow(3, 5, 128)
    assert o.rotation_galois(1, right=True) == pow(3, 31, 128)


def _crt(res, mods):
    Q = 1
    for m in mods:
        Q *= m
    x = 0
    for r, m in zip(res, mods):
        h = Q // m
        x += int(r) * pow(h, -1, m) * h
    return x % Q, Q


@pytest.mark.parametrize("L,K,Li", [(4, 4, 4), (4, 4, 2), (5, 2, 3), (3, 1, 3)])
def test_bconv_tables_and_mod_up_vs_bigint(L, K, Li):
    o = orc.Oracle(5, L, K)
    q = [int(v) for v in o.primes[:L]]
    sp = [int(v) for v in o.primes[L:]]
    P = 1
    for v in sp:
        P *= v
    assert o.bconv_table(3).tolist() == [P % m for m in q]
    assert o.bconv_table(2).tolist() == [pow(P, -1, m) for m in q]
    assert o.bconv_table(0).tolist() == [pow(P // sp[k], -1, sp[k]) for k in range(K)]
    assert o.bconv_table(1).reshape(L, K).tolist() == [[(m - (P // sp[k]) % m) for k in range(K)] for m in q]
    rng = np.random.default_rng(L * 10 + K)
    x = np.stack([rng.integers(0, q[j], o.N, dtype=np.uint64) for j in range(Li)])
    up = o.mod_up(x)
    assert np.array_equal(up[:Li], x)
    Q = 1
    for m in q[:Li]:
        Q *= m
    for d in range(o.N):
        # uncorrected fast basis extension: y = sum_j [x_j * hat_j^-1]_qj * hat_j  (no mod Q!)
        y = sum((int(x[j, d]) * pow(Q // q[j], -1, q[j]) % q[j]) * (Q // q[j]) for j in range(Li))
        exact, _ = _crt(x[:, d], q[:Li])
        assert (y - exact) % Q == 0 and 0 <= (y - exact) // Q < Li  # the alpha*Q overflow stays in
        for k in range(K):
            assert int(up[Li + k, d]) == y % sp[k]


def test_mod_up_rejects_single_limb():
    o = orc.Oracle(5, 3, 2)
    with pytest.raises(ValueError):
        o.mod_up(np.zeros((1, o.N), np.uint64))


@pytest.mark.parametrize("L,K,L0", [(4, 4, 4), (4, 4, 2), (3, 2, 3)])
def test_mod_down_vs_bigint(L, K, L0):
    o = orc.Oracle(5, L, K)
    q = [int(v) for v in o.primes[:L]]
    sp = [int(v) for v in o.primes[L:]]
    P = 1
    for v in sp:
        P *= v
    rng = np.random.default_rng(3)
    normal_pw = np.stack([rng.integers(0, q[j], o.N, dtype=np.uint64) for j in range(L0)])
    special_pw = np.stack([rng.integers(0, sp[k], o.N, dtype=np.uint64) for k in range(K)])
    op = np.concatenate([o.ntt_forward(normal_pw.copy(), list(range(L0))), special_pw])
    got = o.ntt_backward(o.mod_down(op).copy(), list(range(L0)))
    for d in range(o.N):
        y = sum((int(special_pw[k, d]) * pow(P // sp[k], -1, sp[k]) % sp[k]) * (P // sp[k]) for k in range(K))
        for j in range(L0):
            want = (int(normal_pw[j, d]) - y) * pow(P, -1, q[j]) % q[j]
            assert int(got[j, d]) == want


def test_rescale_vs_bigint_floor():
    o = orc.Oracle(5, 4, 1)
    q = [int(v) for v in o.primes[:4]]
    rng = np.random.default_rng(4)
    pw = np.stack([rng.integers(0, q[j], o.N, dtype=np.uint64) for j in range(4)])
    op = pw.copy()
    o.ntt_forward(op[:3], [0, 1, 2])  # last limb stays in the power basis (base_converter.cpp:246)
    got = o.ntt_backward(o.rescale(op).copy(), [0, 1, 2])
    for d in range(o.N):
        x, Q = _crt(pw[:, d], q)
        want = (x - int(pw[3, d])) // q[3]  # floor-style: no rounding term
        assert (x - int(pw[3, d])) % q[3] == 0
        for j in range(3):
            assert int(got[j, d]) == want % q[j]


def _centered_crt(limbs, mods):
    out = []
    for d in range(limbs.shape[1]):
        x, Q = _crt(limbs[:, d], mods)
        out.append(x - Q if x > Q // 2 else x)
    return out


def _encode_small(o, coeffs, Li):
    m = np.zeros((Li, o.N), np.uint64)
    for i in range(Li):
        p = int(o.primes[i])
        m[i] = np.array([c % p for c in coeffs], dtype=np.uint64)
    return m


def test_encrypt_multiply_relin_rescale_decrypt_roundtrip():
    """BASELINE config 1 at the integer-plaintext level: decrypt(mul(enc a, enc b)) ~ a*b / q_last."""
    logN, L, K = 6, 4, 4
    o = orc.Oracle(logN, L, K)
    q = [int(v) for v in o.primes[:L]]
    sx, ext = o.keygen_sk(seed=11, hwt=8)
    pkb, pka = o.keygen_pk(sx, seed=12)
    beta, alpha = o.keygen_mult(sx, seed=13)
    rng = np.random.default_rng(5)
    scale = 1 << 40
    a = [int(v) * scale for v in rng.integers(-8, 9, o.N)]
    b = [int(v) * scale for v in rng.integers(-8, 9, o.N)]
    ca = o.encrypt(_encode_small(o, a, L), pkb, pka, seed=21)
    cb = o.encrypt(_encode_small(o, b, L), pkb, pka, seed=22)
    # fresh decrypt is message + small noise
    da = _centered_crt(o.decrypt(ca[0], ca[1], sx), q)
    assert max(abs(x - y) for x, y in zip(da, a)) < 1 << 12
    ob, oa = o.mul_relin_rescale(ca[0], ca[1], cb[0], cb[1], beta, alpha)
    assert ob.shape == (L - 1, o.N)
    dec = _centered_crt(o.decrypt(ob, oa, sx), q[: L - 1])
    Qfull = 1
    for m in q:
        Qfull *= m
    prod = negacyclic([x % Qfull for x in a], [x % Qfull for x in b], Qfull)
    prod = [x - Qfull if x > Qfull // 2 else x for x in prod]
    err = max(abs(d * q[L - 1] - w) for d, w in zip(dec, prod))
    assert err < (1 << 62) * (1 << 30)  # relin + rescale noise, far below the 2^80 signal


def test_rotate_keyswitch_decrypts_to_rotated_plaintext():
    logN, L, K = 6, 3, 3
    o = orc.Oracle(logN, L, K)
    q = [int(v) for v in o.primes[:L]]
    sx, ext = o.keygen_sk(seed=31, hwt=8)
    pkb, pka = o.keygen_pk(sx, seed=32)
    g = o.rotation_galois(1)
    rkb, rka = o.keygen_rot(sx, ext, g, seed=33)
    rng = np.random.default_rng(6)
    scale = 1 << 50
    m = [int(v) * scale for v in rng.integers(-8, 9, o.N)]
    cb, ca = o.encrypt(_encode_small(o, m, L), pkb, pka, seed=34)
    rb, ra = o.rotate(cb, ca, g, rkb, rka)
    dec = _centered_crt(o.decrypt(rb, ra, sx), q)
    want = [0] * o.N
    for d in range(o.N):
        e = d * g % (2 * o.N)
        if e < o.N:
            want[e] = m[d]
        else:
            want[e - o.N] = -m[d]
    assert max(abs(x - y) for x, y in zip(dec, want)) < 1 << 30
    # galois == 0 is a no-op (evaluator.cpp:303-304)
    nb, na = o.rotate(cb, ca, 0, rkb, rka)
    assert np.array_equal(nb, cb) and np.array_equal(na, ca)


def test_relinearize_equals_reference_loop_in_python():
    """evaluator.cpp:649-693 restated independently with Python ints at a tiny size."""
    logN, L, K = 4, 3, 1
    o = orc.Oracle(logN, L, K)
    q = [int(v) for v in o.primes[:L]]
    rng = np.random.default_rng(8)
    d0, d1, d2 = (np.stack([rng.integers(0, q[j], o.N, dtype=np.uint64) for j in range(L)]) for _ in range(3))
    beta = np.stack([np.stack([rng.integers(0, q[i], o.N, dtype=np.uint64) for i in range(L)]) for _ in range(L)])
    alpha = np.stack([np.stack([rng.integers(0, q[i], o.N, dtype=np.uint64) for i in range(L)]) for _ in range(L)])
    r0, r1 = o.relinearize(d0, d1, d2, beta, alpha)
    pw = o.ntt_backward(d2.copy(), list(range(L)))
    for i in range(L):
        acc0 = [int(v) for v in d0[i]]
        acc1 = [int(v) for v in d1[i]]
        for j in range(L):
            t = o.ntt_forward(pw[j].copy(), [i])[0]
            for d in range(o.N):
                acc0[d] += int(beta[j, i, d]) * int(t[d])
                acc1[d] += int(alpha[j, i, d]) * int(t[d])
        assert r0[i].tolist() == [v % q[i] for v in acc0]
        assert r1[i].tolist() == [v % q[i] for v in acc1]


def test_lazy_sum_is_exact_beyond_16_terms():
    """128-bit lazy sums wrap above 16 terms; the oracle folds so the residue is exact."""
    logN, L, K = 4, 20, 1
    o = orc.Oracle(logN, L, K)
    q = [int(v) for v in o.primes[:L]]
    x = np.stack([np.full(o.N, q[j] - 1, dtype=np.uint64) for j in range(L)])
    up = o.mod_up(x)
    Q = 1
    for m in q:
        Q *= m
    y = sum(((q[j] - 1) * pow(Q // q[j], -1, q[j]) % q[j]) * (Q // q[j]) for j in range(L))
    assert int(up[L, 0]) == y % int(o.primes[L])


def test_mixed_rot_key_rotation_decrypts_to_rotated_plaintext():
    """MixedRotKey key switch (evaluator.cpp:403-516) in the oracle, pinned by decryption."""
    logN, L, K = 6, 3, 1
    o = orc.Oracle(logN, L, K)
    q = [int(v) for v in o.primes[:L]]
    sx, ext = o.keygen_sk(seed=41, hwt=8)
    pkb, pka = o.keygen_pk(sx, seed=42)
    g = o.rotation_galois(2)
    mkb, mka = o.keygen_mixed_rot(sx, ext, g, seed=43)
    rng = np.random.default_rng(9)
    scale = 1 << 50
    m = [int(v) * scale for v in rng.integers(-8, 9, o.N)]
    for Li in (3, 2):
        cb, ca = o.encrypt(_encode_small(o, m, Li), pkb[:Li].copy(), pka[:Li].copy(), seed=44)
        rb, ra = o.rotate_mixed(cb, ca, g, mkb, mka)
        dec = _centered_crt(o.decrypt(rb, ra, sx), q[:Li])
        want = [0] * o.N
        for d in range(o.N):
            e = d * g % (2 * o.N)
            if e < o.N:
                want[e] = m[d]
            else:
                want[e - o.N] = -m[d]
        assert max(abs(x - y) for x, y in zip(dec, want)) < 1 << 30


def test_rotate_non_ntt_is_the_literal_composition_of_the_reference_steps():
    """evaluator.cpp:327-339, 520-574 step by step from the oracle's primitives; also records that this path, as written in the
    reference, does not decrypt to the rotated plaintext (its mod_down mixes NTT-domain and power-basis limbs)."""
    logN, L, K = 6, 3, 2
    o = orc.Oracle(logN, L, K)
    sx, ext = o.keygen_sk(seed=41, hwt=8)
    g = o.rotation_galois(1)
    rkb, rka = o.keygen_rot(sx, ext, g, seed=43)
    rng = np.random.default_rng(8)
    Li = 3
    bx = np.stack([rng.integers(0, int(o.primes[i]), o.N, dtype=np.uint64) for i in range(Li)])
    ax = np.stack([rng.integers(0, int(o.primes[i]), o.N, dtype=np.uint64) for i in range(Li)])
    rb, ra = o.rotate_non_ntt(bx, ax, g, rkb, rka)
    gb = o.apply_galois_non_ntt(bx.copy(), g)
    ga = o.apply_galois_non_ntt(ax.copy(), g)
    idx = [cm if cm < Li else L + cm - Li for cm in range(Li + K)]
    e0 = o.ntt_forward(o.mod_up(ga), idx)
    e1 = e0.copy()
    for cm, m in enumerate(idx):
        p = int(o.primes[m])
        e0[cm] = np.array([int(x) * int(k) % p for x, k in zip(e0[cm], rkb[m])], dtype=np.uint64)
        e1[cm] = np.array([int(x) * int(k) % p for x, k in zip(e1[cm], rka[m])], dtype=np.uint64)
    e0 = o.ntt_backward(e0, idx)
    e1 = o.ntt_backward(e1, idx)
    wb = o.add(o.mod_down(e0), gb)
    wa = o.mod_down(e1)
    assert np.array_equal(rb, wb) and np.array_equal(ra, wa)
    nb, na = o.rotate_non_ntt(bx, ax, 0, rkb, rka)  # galois == 0: no-op (:331-332)
    assert np.array_equal(nb, bx) and np.array_equal(na, ax)
    with pytest.raises(ValueError):
        o.rotate_non_ntt(bx[:1], ax[:1], g, rkb, rka)  # Li <= 1: mod_up throws (base_converter.cpp:284-285)


# ---- full-size pins by mathematics (N = 2^14 .. 2^16): sampled outputs recomputed by Horner's rule in Python integers ----
from horner import horner_eval, ntt_value_at, sample_positions  # noqa: E402


@pytest.mark.parametrize("logN", [14, 15, 16])
def test_full_size_forward_ntt_at_sampled_points(logN):
    """Above logN = 8 the O(N^2) schoolbook check is out of reach; one output is N mulmods: X[d] = x(psi^(2 brv(d)+1))."""
    o = orc.Oracle(logN, 2, 1)
    rng = np.random.default_rng(1000 + logN)
    for idx in (0, 2):  # a normal and the special prime
        p, psi = int(o.primes[idx]), int(o.psi[idx])
        x = rng.integers(0, p, o.N, dtype=np.uint64)
        got = o.ntt_forward(x.copy()[None], [idx])[0]
        coeffs = [int(v) for v in x]
        for d in sample_positions(logN, rng):
            assert int(got[d]) == ntt_value_at(coeffs, d, logN, psi, p), (logN, idx, d)
        # and the inverse: INTT(got) must be x (forward at sampled points + exact round trip pin backward too)
        assert np.array_equal(o.ntt_backward(got.copy()[None], [idx])[0], x)


def test_full_size_relinearisation_at_sampled_coefficients():
    """N = 2^16, Li = 3: one output coefficient of Mul::reLinearize (reference src/evaluator.cpp:663-691) is
    sum_j NTT_i(INTT_j(d2_j))[d] * beta_j[i][d] + d0[i][d]; the digit polynomials INTT_j(d2_j) are pinned by evaluating them
    back at sampled points, the sum is recomputed in Python integers."""
    logN, L, K, Li = 16, 3, 1, 3
    o = orc.Oracle(logN, L, K)
    rng = np.random.default_rng(77)
    poly = lambda: np.stack([rng.integers(0, int(o.primes[i]), o.N, dtype=np.uint64) for i in range(Li)])  # noqa: E731
    d0, d1, d2 = poly(), poly(), poly()
    beta, alpha = np.stack([poly() for _ in range(L)]), np.stack([poly() for _ in range(L)])
    out0, out1 = o.relinearize(d0, d1, d2, beta, alpha)
    digits = o.ntt_backward(d2.copy(), range(Li))  # c_j = INTT_j(d2_j), coefficients below q_j
    dig = [[int(v) for v in digits[j]] for j in range(Li)]
    for j in range(Li):  # pin the digits: their forward NTT under q_j at sampled points is d2_j
        pj, sj = int(o.primes[j]), int(o.psi[j])
        for d in (0, 12345, o.N - 1):
            assert ntt_value_at(dig[j], d, logN, sj, pj) == int(d2[j][d])
    for i, d in ((0, 0), (1, 40000), (2, o.N - 1), (1, 1)):
        pi, si = int(o.primes[i]), int(o.psi[i])
        vals = [ntt_value_at(dig[j], d, logN, si, pi) for j in range(Li)]  # NTT_i of the residues of another prime, j == i included
        assert vals[i] == int(d2[i][d])
        w0 = (sum(vals[j] * int(beta[j][i][d]) for j in range(Li)) + int(d0[i][d])) % pi
        w1 = (sum(vals[j] * int(alpha[j][i][d]) for j in range(Li)) + int(d1[i][d])) % pi
        assert int(out0[i][d]) == w0 and int(out1[i][d]) == w1, (i, d)
