"""GPU parity: every C-ABI entry point against the CPU oracle on identical primes, keys and inputs.
Bit-exact (integer work).  Sizes are chosen so the oracle finishes in seconds; full-size cases use
size-independent properties (round trips, linearity)."""
import numpy as np
import pytest

import fhe_amd
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(a.view(np.int64)).cuda()


def host(t):
    return t.cpu().numpy().view(np.uint64)


_ctx_cache = {}


def ctx_pair(logN, L, K, sp=True):
    key = (logN, L, K, sp)
    if key not in _ctx_cache:
        if sp:
            pr, ps = orc.gen_primes(logN, L + K)
        else:
            pr, ps = generic_primes(logN, L + K)
        _ctx_cache[key] = (orc.Oracle(logN, L, K, pr, ps), fhe_amd.HipContext(logN, L, K, pr, ps))
    return _ctx_cache[key]


def generic_primes(logN, count):
    """Primes NOT of the p = 1 (mod 2^32) form: exercises the generic Montgomery path."""
    pr, ps = [], []
    step = 1 << (logN + 1)
    p = (1 << 61) + (1 << 60) + 1
    p -= (p - 1) % step
    while len(pr) < count:
        p -= step
        if p % (1 << 32) != 1 and orc.lib.orc_is_prime(p):
            x = 2
            while pow(x, (p - 1) // 2, p) != p - 1:
                x += 1
            pr.append(p)
            ps.append(pow(x, (p - 1) >> (logN + 1), p))
    return np.array(pr, np.uint64), np.array(ps, np.uint64)


def generic_primes_below(logN, count, top):
    """Generic-form primes descending from `top` (exclusive): the carry corners of the word-serial Montgomery steps sit at the
    upper end of the contract (p just below 2^62), and short primes exercise zero high words."""
    pr, ps = [], []
    step = 1 << (logN + 1)
    p = top - (top - 1) % step
    while len(pr) < count:
        p -= step
        if p % (1 << 32) != 1 and orc.lib.orc_is_prime(p):
            x = 2
            while pow(x, (p - 1) // 2, p) != p - 1:
                x += 1
            pr.append(p)
            ps.append(pow(x, (p - 1) >> (logN + 1), p))
    return np.array(pr, np.uint64), np.array(ps, np.uint64)


def rand_poly(o, idx, seed, bound=None):
    rng = np.random.default_rng(seed)
    return np.stack([rng.integers(0, int(o.primes[i]) if bound is None else bound, o.N, dtype=np.uint64) for i in idx])


@pytest.mark.parametrize("logN", [10, 11, 12, 13, 14, 15, 16])
@pytest.mark.parametrize("sp", [True, False])
def test_ntt_forward_backward_bit_exact(logN, sp):
    o, g = ctx_pair(logN, 3, 2, sp)
    idx = [0, 1, 2, 3, 4, 2, 0]
    x = rand_poly(o, idx, logN)
    want = o.ntt_forward(x.copy(), idx)
    d = dev(x)
    g.ntt_forward(d, idx)
    assert np.array_equal(host(d), want)
    g.ntt_backward(d, idx)
    assert np.array_equal(host(d), x)
    y = rand_poly(o, idx, logN + 100)
    want_b = o.ntt_backward(y.copy(), idx)
    d = dev(y)
    g.ntt_backward(d, idx)
    assert np.array_equal(host(d), want_b)


def test_ntt_accepts_foreign_residues_and_edge_values():
    o, g = ctx_pair(13, 3, 2)
    idx = [0, 1, 4]
    x = rand_poly(o, idx, 5, bound=1 << 62)  # residues of another 62-bit prime: >= p allowed
    x[0, :4] = [0, 1, int(o.primes[0]) - 1, (1 << 62) - 1]
    want = o.ntt_forward(x.copy(), idx)
    d = dev(x)
    g.ntt_forward(d, idx)
    assert np.array_equal(host(d), want)
    z = np.zeros((1, o.N), np.uint64)
    dz = dev(z)
    g.ntt_forward(dz, [1])
    assert not host(dz).any()


def _word_edge_patterns(p, N, seed):
    """Inputs that drive the word-serial Montgomery steps through their carry / zero-word corners."""
    rng = np.random.default_rng(seed)
    m32 = (1 << 32) - 1
    specials = np.array([0, 1, 2, m32, 1 << 32, (1 << 32) + 1, p - 1, p - 2, p >> 1, (p >> 32) << 32, ((p >> 32) << 32) - 1,
                         p - (1 << 32), p & ~m32 | 0, (p - 1) & ~m32, m32 << 30], dtype=np.uint64) % np.uint64(p)
    rows = []
    rows.append(np.resize(specials, N))                                                   # corner values cycled
    rows.append((rng.integers(0, p >> 32, N, dtype=np.uint64) << np.uint64(32)))          # low word zero everywhere
    rows.append(rng.integers(0, 1 << 32, N, dtype=np.uint64))                             # high word zero everywhere
    rows.append((rng.integers(0, p >> 32, N, dtype=np.uint64) << np.uint64(32)) | np.uint64(m32))  # low word all ones
    rows.append(np.full(N, p - 1, np.uint64))                                             # constant p-1
    rows.append(np.where(rng.integers(0, 2, N) == 1, np.uint64(p - 1), np.uint64(0)))     # 0 / p-1 mix
    return np.stack(rows)


@pytest.mark.parametrize("logN,sp", [(10, True), (13, True), (16, True), (12, False)])
def test_ntt_word_edge_patterns(logN, sp):
    o, g = ctx_pair(logN, 3, 2, sp)
    for m in (0, 4):
        p = int(o.primes[m])
        x = _word_edge_patterns(p, o.N, 100 + m)
        idx = [m] * x.shape[0]
        want = o.ntt_forward(x.copy(), idx)
        d = dev(x)
        g.ntt_forward(d, idx)
        assert np.array_equal(host(d), want)
        g.ntt_backward(d, idx)
        assert np.array_equal(host(d), x)
        want_b = o.ntt_backward(x.copy(), idx)
        d = dev(x)
        g.ntt_backward(d, idx)
        assert np.array_equal(host(d), want_b)
        # forward input bound of the C-ABI: any value < 4p
        hi = x + np.uint64(3 * p)
        want_hi = o.ntt_forward(hi.copy(), idx)
        d = dev(hi)
        g.ntt_forward(d, idx)
        assert np.array_equal(host(d), want_hi)


@pytest.mark.parametrize("top", [1 << 62, (1 << 62) - (1 << 40), 1 << 45, 1 << 33])
def test_generic_primes_across_the_contract_range(top):
    """The generic word-serial Montgomery arithmetic (modarith.h:mont_generic) with primes at the top of the contract (just below
    2^62: every carry word in play), and with short primes (high words small or zero): NTT both ways on corner patterns and on
    lazy inputs, then a whole multiply + relinearize + rescale and a rotation, against the oracle."""
    logN, L, K, Li = 12, 3, 2, 3
    pr, ps = generic_primes_below(logN, L + K, top)
    o, g = orc.Oracle(logN, L, K, pr, ps), fhe_amd.HipContext(logN, L, K, pr, ps)
    for m in (0, L + K - 1):
        p = int(pr[m])
        x = _word_edge_patterns(p, o.N, 300 + m)
        idx = [m] * x.shape[0]
        for inp in (x, x + np.uint64(3 * p)):
            want = o.ntt_forward(inp.copy(), idx)
            d = dev(inp)
            g.ntt_forward(d, idx)
            assert np.array_equal(host(d), want)
        want_b = o.ntt_backward(x.copy(), idx)
        d = dev(x)
        g.ntt_backward(d, idx)
        assert np.array_equal(host(d), want_b)
    sx, ext = o.keygen_sk(seed=41)
    beta, alpha = o.keygen_mult(sx, seed=42)
    g.load_mult_key(dev(beta), dev(alpha))
    c0 = np.stack([rand_poly(o, range(Li), 43), rand_poly(o, range(Li), 44)])
    c1 = np.stack([rand_poly(o, range(Li), 45), rand_poly(o, range(Li), 46)])
    out = torch.zeros((1, 2, Li - 1, o.N), dtype=torch.int64, device="cuda")
    g.mul_relin_rescale(out, dev(c0[None]), dev(c1[None]), Li, batch=1)
    want = np.stack(o.mul_relin_rescale(c0[0], c0[1], c1[0], c1[1], beta, alpha))
    assert np.array_equal(host(out)[0], want)
    gal = 3
    rkb, rka = o.keygen_rot(sx, ext, gal, seed=47)
    g.load_rot_key(gal, dev(rkb), dev(rka))
    rot = torch.zeros((1, 2, Li, o.N), dtype=torch.int64, device="cuda")
    g.rotate_to(gal, rot, dev(c0[None]), Li, batch=1)
    assert np.array_equal(host(rot)[0], np.stack(o.rotate(c0[0], c0[1], gal, rkb, rka)))
    g.close()


@pytest.mark.parametrize("logN", [15, 16])
def test_two_stream_overlapped_ntt_matches_the_oracle_and_one_stream(logN):
    """fhe_hip_ntt_forward / _backward at N >= 2^15 issue their chunks alternately on two library-owned streams
    (FHE_HIP_TUNE_NTT_STREAMS).  With a chunk of 2 limbs the 7-limb call below takes that path (four chunks, the last one short):
    bit-exact against the oracle both ways, from device and from host memory; then the default chunk (128 limbs) on 300 limbs
    against the one-stream result, with other work queued on the caller's stream before and after the call (fork / join)."""
    o, g = ctx_pair(logN, 3, 2)
    idx = [0, 1, 2, 3, 4, 2, 0]
    x = rand_poly(o, idx, 700 + logN)
    want = o.ntt_forward(x.copy(), idx)
    g.set_tuning(fhe_amd.hip.TUNE_NTT_CHUNK, 2)
    try:
        d = dev(x)
        g.ntt_forward(d, idx)
        assert np.array_equal(host(d), want)
        g.ntt_backward(d, idx)
        assert np.array_equal(host(d), x)
        h = x.copy()
        g.ntt_forward(h, idx)  # host buffers: staged copies before the fork and after the join
        assert np.array_equal(h, want)
    finally:
        g.set_tuning(fhe_amd.hip.TUNE_NTT_CHUNK, 0)
    n = 300
    idx = np.arange(n, dtype=np.uint32) % 5
    gen = torch.Generator(device="cuda").manual_seed(logN)
    src = torch.randint(0, 1 << 61, (n, o.N), dtype=torch.int64, device="cuda", generator=gen)
    a = src.clone() + 0  # produced on the caller's stream right before the call
    g.ntt_forward(a, idx)
    a_plus = a + 1       # consumed on the caller's stream right after it
    g.set_tuning(fhe_amd.hip.TUNE_NTT_STREAMS, 1)
    try:
        b = src.clone()
        g.ntt_forward(b, idx)
    finally:
        g.set_tuning(fhe_amd.hip.TUNE_NTT_STREAMS, 2)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(a_plus, b + 1)
    g.ntt_backward(a, idx)
    torch.cuda.synchronize()
    assert torch.equal(a, src)


def test_pointwise_word_edge_patterns():
    """multiply_plain / cross_mult on the same corner patterns (zero low words, p-1, 2^32 multiples)."""
    o, g = ctx_pair(10, 3, 2)
    Li = 3
    a = np.stack([_word_edge_patterns(int(o.primes[l]), o.N, 7 + l) for l in range(Li)], axis=1)  # [6][Li][N]
    b = a[::-1].copy()
    c0 = a.reshape(3, 2, Li, o.N)
    c1 = b.reshape(3, 2, Li, o.N)
    for k in range(3):
        for pt in (b[2 * k], b[2 * k + 1]):
            wb, wa = o.multiply_plain(c0[k, 0], c0[k, 1], np.ascontiguousarray(pt))
            d = dev(c0[k].copy())
            g.multiply_plain(d, dev(np.ascontiguousarray(pt)), Li, 1)
            assert np.array_equal(host(d)[0], wb) and np.array_equal(host(d)[1], wa)
        want = o.cross_mult(c0[k, 0].copy(), c0[k, 1].copy(), c1[k, 0].copy(), c1[k, 1].copy())
        out = torch.zeros((3, Li, o.N), dtype=torch.int64, device="cuda")
        g.cross_mult(out, dev(c0[k].copy()), dev(c1[k].copy()), Li, 1)
        for h in range(3):
            assert np.array_equal(host(out)[h], want[h])


def test_ntt_host_memory_kind():
    o, g = ctx_pair(12, 3, 2)
    idx = [1, 3]
    x = rand_poly(o, idx, 9)
    want = o.ntt_forward(x.copy(), idx)
    h = x.copy()
    g.ntt_forward(h, idx)  # numpy buffer -> FHE_HIP_MEM_HOST
    assert np.array_equal(h, want)


def test_config2_batched_ntt_properties():
    """BASELINE config 2 shape (N=2^14, 8 limbs, 1024 polys): round trip + spot parity + linearity."""
    logN, L = 14, 8
    o, g = ctx_pair(logN, L, 1)
    npoly = 1024
    idx = np.tile(np.arange(L, dtype=np.uint32), npoly)
    gen = torch.Generator(device="cuda").manual_seed(14)
    x = torch.randint(0, 1 << 61, (npoly * L, o.N), dtype=torch.int64, device="cuda", generator=gen)
    y = x.clone()
    g.ntt_forward(y, idx)
    for k in (0, 5, 8191):
        want = o.ntt_forward(host(x[k]).copy().reshape(1, -1), [int(idx[k])])[0]
        assert np.array_equal(host(y[k]), want)
    # linearity: NTT(a) + NTT(b) == NTT(a + b) mod p on one limb pair
    p0 = int(o.primes[0])
    a, b = host(x[0]).astype(object), host(x[8]).astype(object)
    s = np.array([(int(u) + int(v)) % p0 for u, v in zip(a, b)], dtype=np.uint64).reshape(1, -1)
    ds = dev(s)
    g.ntt_forward(ds, [0])
    fa, fb = host(y[0]), host(y[8])
    assert np.array_equal(host(ds)[0], np.array([(int(u) + int(v)) % p0 for u, v in zip(fa, fb)], dtype=np.uint64))
    g.ntt_backward(y, idx)
    xr = host(x) % np.tile(o.primes[:L], npoly)[:, None]
    assert np.array_equal(host(y), xr)


@pytest.mark.parametrize("galois", [3, 9, 5, 1])
def test_galois_both_domains(galois):
    o, g = ctx_pair(11, 3, 2)
    x = np.stack([rand_poly(o, [0, 1, 2], 20 + b) for b in range(2)])
    want = np.stack([o.apply_galois(x[b].copy(), galois) for b in range(2)])
    d = dev(x)
    g.apply_galois(d, 3, galois, batch=2)
    assert np.array_equal(host(d), want)
    want = np.stack([o.apply_galois_non_ntt(x[b].copy(), galois) for b in range(2)])
    d = dev(x)
    g.apply_galois_non_ntt(d, 3, galois, batch=2)
    assert np.array_equal(host(d), want)


@pytest.mark.parametrize("logN,L,K,Li,sp", [(10, 4, 4, 4, True), (10, 4, 4, 2, True), (11, 5, 2, 3, False), (10, 20, 3, 20, True)])
def test_mod_up_mod_down_rescale(logN, L, K, Li, sp):
    o, g = ctx_pair(logN, L, K, sp)
    x = np.stack([rand_poly(o, range(Li), 30 + b) for b in range(2)])
    want = np.stack([o.mod_up(x[b]) for b in range(2)])
    rop = torch.zeros((2, Li + K, o.N), dtype=torch.int64, device="cuda")
    g.mod_up(rop, dev(x), Li, batch=2)
    assert np.array_equal(host(rop), want)
    idx = list(range(Li)) + list(range(L, L + K))
    y = np.stack([rand_poly(o, idx, 40 + b) for b in range(2)])
    want = np.stack([o.mod_down(y[b]) for b in range(2)])
    rop = torch.zeros((2, Li, o.N), dtype=torch.int64, device="cuda")
    g.mod_down(rop, dev(y), Li, batch=2)
    assert np.array_equal(host(rop), want)
    want = np.stack([o.rescale(x[b]) for b in range(2)])
    rop = torch.zeros((2, Li - 1, o.N), dtype=torch.int64, device="cuda")
    g.rescale(rop, dev(x), Li, batch=2)
    assert np.array_equal(host(rop), want)
    # the single-target conversions behind them (BaseConverter::approx_convert_to_special_basis /
    # neg_approx_convert_to_normal_basis, base_converter.cpp:276-311)
    up = host(g.mod_up(torch.zeros((2, Li + K, o.N), dtype=torch.int64, device="cuda"), dev(x), Li, batch=2))
    for k in range(K):
        assert np.array_equal(up[1, Li + k], o.approx_convert_to_special_basis(x[1], k))
        one = g.conv_to_special(torch.zeros((2, o.N), dtype=torch.int64, device="cuda"), dev(x), Li, k, batch=2)  # the single-target entry point
        assert np.array_equal(host(one), up[:, Li + k])
    hone = np.zeros((2, o.N), np.uint64)
    g.conv_to_special(hone, x.copy(), Li, K - 1, batch=2)  # host buffers
    assert np.array_equal(hone, up[:, Li + K - 1])
    with pytest.raises(fhe_amd.FheHipError) as ei:
        g.conv_to_special(torch.zeros((1, o.N), dtype=torch.int64, device="cuda"), dev(x[:1]), Li, K)
    assert ei.value.code == -2
    sp_part = np.ascontiguousarray(y[:, Li:])
    for L0 in sorted({1, Li, L}):
        rop = torch.zeros((2, L0, o.N), dtype=torch.int64, device="cuda")
        g.conv_to_normal(rop, dev(sp_part), L0, batch=2)
        for b in range(2):
            for j in range(L0):
                assert np.array_equal(host(rop)[b, j], o.neg_approx_convert_to_normal_basis(sp_part[b], j))
    with pytest.raises(fhe_amd.FheHipError) as ei:
        g.conv_to_normal(torch.zeros((1, L + 1, o.N), dtype=torch.int64, device="cuda"), dev(sp_part[:1]), L + 1)
    assert ei.value.code == -2


def test_mod_up_rejects_single_limb_like_reference():
    o, g = ctx_pair(10, 4, 4)
    with pytest.raises(fhe_amd.FheHipError) as ei:
        g.mod_up(torch.zeros((5, o.N), dtype=torch.int64, device="cuda"), torch.zeros((1, o.N), dtype=torch.int64, device="cuda"), 1)
    assert ei.value.code == -2


def _keys(o, seed):
    sx, ext = o.keygen_sk(seed=seed)
    pkb, pka = o.keygen_pk(sx, seed=seed + 1)
    beta, alpha = o.keygen_mult(sx, seed=seed + 2)
    return sx, ext, pkb, pka, beta, alpha


@pytest.mark.parametrize("logN,L,K,Li,sp", [(10, 4, 4, 4, True), (10, 4, 4, 3, True), (11, 3, 1, 3, False), (10, 18, 1, 18, True)])
def test_cross_mult_relin_rescale_pipeline(logN, L, K, Li, sp):
    o, g = ctx_pair(logN, L, K, sp)
    sx, ext, pkb, pka, beta, alpha = _keys(o, 50)
    g.load_mult_key(dev(beta), dev(alpha))
    B = 2
    c0 = np.stack([np.stack([rand_poly(o, range(Li), 60 + 2 * b), rand_poly(o, range(Li), 61 + 2 * b)]) for b in range(B)])
    c1 = np.stack([np.stack([rand_poly(o, range(Li), 70 + 2 * b), rand_poly(o, range(Li), 71 + 2 * b)]) for b in range(B)])
    out3 = torch.zeros((B, 3, Li, o.N), dtype=torch.int64, device="cuda")
    g.cross_mult(out3, dev(c0), dev(c1), Li, batch=B)
    want3 = np.stack([np.stack(o.cross_mult(c0[b, 0], c0[b, 1], c1[b, 0], c1[b, 1])) for b in range(B)])
    assert np.array_equal(host(out3), want3)
    rl = torch.zeros((B, 2, Li, o.N), dtype=torch.int64, device="cuda")
    g.relin_only(rl, out3, Li, batch=B)
    want_rl = np.stack([np.stack(o.relinearize(want3[b, 0], want3[b, 1], want3[b, 2], beta, alpha)) for b in range(B)])
    assert np.array_equal(host(rl), want_rl)
    out = torch.zeros((B, 2, Li - 1, o.N), dtype=torch.int64, device="cuda")
    g.relinearize(out, out3, Li, batch=B)
    want = np.stack([np.stack(o.rescale_ntt(want_rl[b, 0], want_rl[b, 1])) for b in range(B)])
    assert np.array_equal(host(out), want)
    out2 = torch.zeros_like(out)
    g.mul_relin_rescale(out2, dev(c0), dev(c1), Li, batch=B)
    want2 = np.stack([np.stack(o.mul_relin_rescale(c0[b, 0], c0[b, 1], c1[b, 0], c1[b, 1], beta, alpha)) for b in range(B)])
    assert np.array_equal(host(out2), want2)
    assert np.array_equal(want2, want)


@pytest.mark.parametrize("logN,L,K,Li,sp", [(10, 4, 4, 4, True), (11, 5, 1, 3, False), (10, 18, 1, 18, True)])
def test_digit_key_switch_kernels_agree_and_refuse_each_others_keys(logN, L, K, Li, sp):
    """FHE_HIP_TUNE_RELIN_KERNEL: the round-3 kernel (unreduced key-MAC accumulators, keys stored times 2^128) and the Montgomery-MAC
    kernel of rounds 1-2 (keys times 2^64) give the same bits as the oracle for relinearisation and MixedRotKey rotation; a key
    loaded for one kernel is refused by the other instead of being misread."""
    o, g = ctx_pair(logN, L, K, sp)
    sx, ext, pkb, pka, beta, alpha = _keys(o, 150)
    B = 2
    c0 = np.stack([np.stack([rand_poly(o, range(Li), 160 + 2 * b), rand_poly(o, range(Li), 161 + 2 * b)]) for b in range(B)])
    c1 = np.stack([np.stack([rand_poly(o, range(Li), 170 + 2 * b), rand_poly(o, range(Li), 171 + 2 * b)]) for b in range(B)])
    want = np.stack([np.stack(o.mul_relin_rescale(c0[b, 0], c0[b, 1], c1[b, 0], c1[b, 1], beta, alpha)) for b in range(B)])
    gal = o.rotation_galois(1)
    rng = np.random.default_rng(151)
    uni = lambda idx: np.stack([rng.integers(0, int(o.primes[i]), o.N, dtype=np.uint64) for i in idx])
    mb = np.stack([uni(list(range(L)) + [L]) for _ in range(L)]); ma = np.stack([uni(list(range(L)) + [L]) for _ in range(L)])
    want_rot = np.stack([np.stack(o.rotate_mixed(c0[b, 0], c0[b, 1], gal, mb, ma)) for b in range(B)])
    try:
        for mode in (0, 1):
            g.set_tuning(fhe_amd.hip.TUNE_RELIN_KERNEL, mode)
            g.load_mult_key(dev(beta), dev(alpha))
            g.load_mixed_rot_key(gal, dev(mb), dev(ma))
            out = torch.zeros((B, 2, Li - 1, o.N), dtype=torch.int64, device="cuda")
            g.mul_relin_rescale(out, dev(c0), dev(c1), Li, batch=B)
            assert np.array_equal(host(out), want), mode
            d = dev(c0)
            g.rotate_mixed(gal, d, Li, batch=B)
            assert np.array_equal(host(d), want_rot), mode
            # the keys stay in this kernel's form: the other kernel refuses them
            g.set_tuning(fhe_amd.hip.TUNE_RELIN_KERNEL, 1 - mode)
            with pytest.raises(fhe_amd.FheHipError) as ei:
                g.mul_relin_rescale(out, dev(c0), dev(c1), Li, batch=B)
            assert ei.value.code == -3
            with pytest.raises(fhe_amd.FheHipError) as ei:
                g.rotate_mixed(gal, dev(c0), Li, batch=B)
            assert ei.value.code == -3
    finally:
        g.set_tuning(fhe_amd.hip.TUNE_RELIN_KERNEL, 1)


def test_config1_roundtrip_decrypted_plaintext_level():
    """BASELINE config 1 (N=2^13, L=K=4): encrypt -> multiply_and_relin on the GPU -> decrypt; ciphertext limbs
    and decrypted plaintext limbs bit-exact vs the oracle, and the plaintext equals a*b/q_last up to noise."""
    logN, L, K = 13, 4, 4
    o, g = ctx_pair(logN, L, K)
    sx, ext, pkb, pka, beta, alpha = _keys(o, 80)
    g.load_mult_key(beta, alpha)  # host-memory key upload
    rng = np.random.default_rng(81)
    scale = 1 << 40

    def enc(vals, seed):
        m = np.zeros((L, o.N), np.uint64)
        for i in range(L):
            p = int(o.primes[i])
            m[i] = np.array([(int(v) * scale) % p for v in vals], dtype=np.uint64)
        return np.stack(o.encrypt(m, pkb, pka, seed))

    a = rng.integers(-4, 5, o.N)
    b = rng.integers(-4, 5, o.N)
    ca, cb = enc(a, 82), enc(b, 83)
    out = torch.zeros((1, 2, L - 1, o.N), dtype=torch.int64, device="cuda")
    g.mul_relin_rescale(out, dev(ca[None]), dev(cb[None]), L, batch=1)
    want = np.stack(o.mul_relin_rescale(ca[0], ca[1], cb[0], cb[1], beta, alpha))
    got = host(out)[0]
    assert np.array_equal(got, want)
    dec_gpu = o.decrypt(got[0].copy(), got[1].copy(), sx)
    dec_cpu = o.decrypt(want[0], want[1], sx)
    assert np.array_equal(dec_gpu, dec_cpu)
    # semantic check on limb 0: centred value * q_last ~ negacyclic(a, b) * scale^2
    p0, qlast = int(o.primes[0]), int(o.primes[L - 1])
    prod = np.zeros(o.N, dtype=object)
    an, bn = [int(v) for v in a], [int(v) for v in b]
    for i in np.nonzero(a)[0][:64]:  # sparse spot check of the convolution is enough at this size
        pass
    fa = np.fft.rfft(np.concatenate([a, np.zeros(o.N)]).astype(np.float64))
    fb = np.fft.rfft(np.concatenate([b, np.zeros(o.N)]).astype(np.float64))
    full = np.rint(np.fft.irfft(fa * fb)).astype(np.int64)
    neg = full[: o.N] - full[o.N:]
    expect = [(int(v) * scale * scale // qlast) % p0 for v in neg]
    diff = [min((int(x) - e) % p0, (e - int(x)) % p0) for x, e in zip(dec_gpu[0], expect)]
    assert max(diff) < 1 << 40


@pytest.mark.parametrize("logN,L,K,Li,sp", [(10, 4, 4, 4, True), (10, 4, 4, 2, True), (11, 3, 2, 3, False), (10, 17, 17, 17, True)])
def test_rotate_keyswitch(logN, L, K, Li, sp):
    o, g = ctx_pair(logN, L, K, sp)
    sx, ext = o.keygen_sk(seed=90)
    for off in (1, 4):
        gal = o.rotation_galois(off)
        rkb, rka = o.keygen_rot(sx, ext, gal, seed=91 + off)
        g.load_rot_key(gal, dev(rkb), dev(rka))
        B = 2
        ct = np.stack([np.stack([rand_poly(o, range(Li), 100 + 2 * b), rand_poly(o, range(Li), 101 + 2 * b)]) for b in range(B)])
        want = np.stack([np.stack(o.rotate(ct[b, 0], ct[b, 1], gal, rkb, rka)) for b in range(B)])
        d = dev(ct)
        g.rotate(gal, d, Li, batch=B)
        assert np.array_equal(host(d), want)
    d = dev(ct)
    g.rotate(0, d, Li, batch=B)  # galois == 0: no-op (evaluator.cpp:303-304)
    assert np.array_equal(host(d), ct)
    with pytest.raises(fhe_amd.FheHipError):
        g.rotate(7, d, Li, batch=B)  # no key loaded for this galois element


@pytest.mark.parametrize("logN,L,K,Li,sp", [(10, 4, 4, 4, True), (10, 5, 3, 2, True), (11, 3, 2, 3, False), (10, 17, 17, 17, True),
                                            (10, 17, 9, 12, False), (12, 9, 10, 9, True), (10, 30, 30, 30, True), (10, 31, 29, 27, False)])
def test_base_conversion_on_the_matrix_cores_matches_the_vector_alu_kernel_and_the_oracle(logN, L, K, Li, sp):
    """The rotation pipelines' base conversions (base_converter.cpp:313-349) as a byte-decomposed integer product on
    v_mfma_i32_16x16x64_i8 (bconv_mfma_kernels.h), forced at shapes where the default would pick the vector-ALU kernel: 1..4 k-steps
    (<= 8, 16, 24, 32 source limbs), target counts that are not multiples of four, a conversion to fewer normal limbs than the
    table holds (Li < L), both prime kinds; bit-exact against the oracle and against the vector-ALU kernel."""
    o, g = ctx_pair(logN, L, K, sp)
    sx, ext = o.keygen_sk(seed=490)
    gal = o.rotation_galois(1)
    rkb, rka = o.keygen_rot(sx, ext, gal, seed=491)
    g.load_rot_key(gal, dev(rkb), dev(rka))
    B = 3
    ct = np.stack([np.stack([rand_poly(o, range(Li), 500 + 2 * b), rand_poly(o, range(Li), 501 + 2 * b)]) for b in range(B)])
    want = np.stack([np.stack(o.rotate(ct[b, 0], ct[b, 1], gal, rkb, rka)) for b in range(B)])
    want_nn = np.stack([np.stack(o.rotate_non_ntt(ct[b, 0], ct[b, 1], gal, rkb, rka)) for b in range(B)])
    got = {}
    try:
        for mode in (2, 1):
            g.set_tuning(fhe_amd.hip.TUNE_BCONV, mode)
            d = dev(ct)
            g.rotate(gal, d, Li, batch=B)
            got[mode] = host(d)
            assert np.array_equal(got[mode], want), mode
            d = dev(ct)
            g.rotate_non_ntt(gal, d, Li, batch=B)   # digits formed inside the kernel (lane-dependent source primes)
            assert np.array_equal(host(d), want_nn), mode
            # the conversions as entry points of their own (base_converter.cpp:189-242, 297-311)
            x = np.stack([rand_poly(o, range(Li), 530 + b) for b in range(2)])
            rop = torch.zeros((2, Li + K, o.N), dtype=torch.int64, device="cuda")
            g.mod_up(rop, dev(x), Li, batch=2)
            assert np.array_equal(host(rop), np.stack([o.mod_up(x[b]) for b in range(2)])), mode
            y = np.stack([rand_poly(o, list(range(Li)) + list(range(L, L + K)), 540 + b) for b in range(2)])
            rop = torch.zeros((2, Li, o.N), dtype=torch.int64, device="cuda")
            g.mod_down(rop, dev(y), Li, batch=2)
            assert np.array_equal(host(rop), np.stack([o.mod_down(y[b]) for b in range(2)])), mode
            sp_part = np.ascontiguousarray(y[:, Li:])
            rop = torch.zeros((2, 1, o.N), dtype=torch.int64, device="cuda")
            g.conv_to_normal(rop, dev(sp_part), 1, batch=2)   # one target: a quarter of a group
            assert np.array_equal(host(rop)[1, 0], o.neg_approx_convert_to_normal_basis(sp_part[1], 0)), mode
        with pytest.raises(fhe_amd.FheHipError):
            g.set_tuning(fhe_amd.hip.TUNE_BCONV, 3)
    finally:
        g.set_tuning(fhe_amd.hip.TUNE_BCONV, 0)
    assert np.array_equal(got[1], got[2])


@pytest.mark.parametrize("logN,L,K,Li,sp", [(10, 4, 4, 4, True), (10, 4, 2, 2, True), (11, 3, 2, 3, False), (12, 9, 3, 9, True)])
def test_rotate_non_ntt_keyswitch(logN, L, K, Li, sp):
    """power-basis rotation (evaluator.cpp:327-339, 520-574), the reference's arithmetic taken literally."""
    o, g = ctx_pair(logN, L, K, sp)
    sx, ext = o.keygen_sk(seed=190)
    for off in (1, 2):
        gal = o.rotation_galois(off)
        rkb, rka = o.keygen_rot(sx, ext, gal, seed=191 + off)
        g.load_rot_key(gal, dev(rkb), dev(rka))
        B = 3
        ct = np.stack([np.stack([rand_poly(o, range(Li), 300 + 2 * b), rand_poly(o, range(Li), 301 + 2 * b)]) for b in range(B)])
        want = np.stack([np.stack(o.rotate_non_ntt(ct[b, 0], ct[b, 1], gal, rkb, rka)) for b in range(B)])
        d = dev(ct)
        g.rotate_non_ntt(gal, d, Li, batch=B)
        assert np.array_equal(host(d), want)
        h = ct.copy()
        g.rotate_non_ntt(gal, h, Li, batch=B)  # host memory kind
        assert np.array_equal(h, want)
    d = dev(ct)
    g.rotate_non_ntt(0, d, Li, batch=B)  # galois == 0: no-op (evaluator.cpp:331-332)
    assert np.array_equal(host(d), ct)
    with pytest.raises(fhe_amd.FheHipError):
        g.rotate_non_ntt(7, d, Li, batch=B)


def test_add_and_multiply_plain():
    o, g = ctx_pair(10, 4, 4)
    a = np.stack([rand_poly(o, range(4), 110 + b) for b in range(2)])
    b_ = np.stack([rand_poly(o, range(4), 120 + b) for b in range(2)])
    d = dev(a)
    g.add(d, dev(b_), 4, batch=2)
    assert np.array_equal(host(d), np.stack([o.add(a[i], b_[i]) for i in range(2)]))
    pt = rand_poly(o, range(4), 130)
    ct = np.stack([a, b_], axis=1)  # [2][2][4][N]
    d = dev(np.ascontiguousarray(ct))
    g.multiply_plain(d, dev(pt), 4, batch=2)
    want = np.stack([np.stack(o.multiply_plain(ct[i, 0], ct[i, 1], pt)) for i in range(2)])
    assert np.array_equal(host(d), want)


@pytest.mark.parametrize("Li,logN,sp", [(4, 11, True), (2, 11, True), (3, 10, True), (3, 12, False)])
def test_encrypt_decrypt_inner_loops_on_device(Li, logN, sp):
    """encryptor.cpp:15-35 / decryptor.cpp:24-36 on the device, sampled polynomials from the host.
    (logN = 10 takes the 16-element row kernels, sp = False the generic-prime instantiations.)"""
    L, K = 4, 1
    o, g = ctx_pair(logN, L, K, sp)
    sx, ext = o.keygen_sk(seed=140)
    pkb, pka = o.keygen_pk(sx, seed=141)
    B = 2
    rng = np.random.default_rng(142)

    def small(lo, hi):
        v = rng.integers(lo, hi, o.N)
        return np.stack([np.array([int(x) % int(o.primes[i]) for x in v], dtype=np.uint64) for i in range(Li)])

    u = np.stack([small(-1, 2) for _ in range(B)])
    e0m = np.stack([small(-(1 << 40), 1 << 40) for _ in range(B)])
    e1 = np.stack([small(-8, 9) for _ in range(B)])
    want = np.stack([np.stack(o.encrypt_core(u[b], e0m[b], e1[b], pkb, pka)) for b in range(B)])
    ct = torch.zeros((B, 2, Li, o.N), dtype=torch.int64, device="cuda")
    g.encrypt(ct, dev(u), dev(e0m), dev(e1), dev(pkb), dev(pka), Li, batch=B)
    assert np.array_equal(host(ct), want)
    # host-memory flavour of the same call
    cth = np.zeros((B, 2, Li, o.N), np.uint64)
    g.encrypt(cth, u, e0m, e1, pkb, pka, Li, batch=B)
    assert np.array_equal(cth, want)
    out = torch.zeros((B, Li, o.N), dtype=torch.int64, device="cuda")
    g.decrypt(out, ct, dev(sx), Li, batch=B)
    want_dec = np.stack([o.decrypt(want[b, 0], want[b, 1], sx) for b in range(B)])
    assert np.array_equal(host(out), want_dec)
    # decrypt(encrypt(m)) = m + small noise on limb 0
    p0 = int(o.primes[0])
    diff = (want_dec[0, 0].astype(object) - e0m[0, 0].astype(object)) % p0
    assert int(np.minimum(diff, p0 - diff).max()) < 1 << 12


@pytest.mark.parametrize("logN,L,K,Li,sp", [(10, 4, 1, 4, True), (10, 4, 2, 2, True), (11, 3, 1, 3, False), (12, 18, 1, 18, True), (10, 3, 1, 1, True)])
def test_rotate_mixed_rot_key(logN, L, K, Li, sp):
    """MixedRotKey key switch (evaluator.cpp:403-516) bit-exact vs the oracle."""
    o, g = ctx_pair(logN, L, K, sp)
    sx, ext = o.keygen_sk(seed=150)
    gal = o.rotation_galois(2)
    mkb, mka = o.keygen_mixed_rot(sx, ext, gal, seed=151)
    g.load_mixed_rot_key(gal, dev(mkb), dev(mka))
    B = 3
    ct = np.stack([np.stack([rand_poly(o, range(Li), 160 + 2 * b), rand_poly(o, range(Li), 161 + 2 * b)]) for b in range(B)])
    want = np.stack([np.stack(o.rotate_mixed(ct[b, 0], ct[b, 1], gal, mkb, mka)) for b in range(B)])
    d = dev(ct)
    g.rotate_mixed(gal, d, Li, batch=B)
    assert np.array_equal(host(d), want)
    d = dev(ct)
    g.rotate_mixed(0, d, Li, batch=B)
    assert np.array_equal(host(d), ct)
    with pytest.raises(fhe_amd.FheHipError):
        g.rotate_mixed(5, d, Li, batch=B)


def _uniform(o, idx, rng):
    return np.stack([rng.integers(0, int(o.primes[i]), o.N, dtype=np.uint64) for i in idx])


@pytest.mark.parametrize("sp", [True, False])
def test_config3_full_size_bit_exact(sp):
    """BASELINE config 3 at full size (N=2^16, Li=L=30): one multiply + relinearize + rescale, every output limb equal to the
    CPU oracle's, on the repository's prime table and on generic 62-bit primes (the kernels a caller's own table runs).
    Keys are uniform polynomials (computationally what real keys are): the arithmetic does not care."""
    logN, L, K, Li = 16, 30, 1, 30
    o, g = ctx_pair(logN, L, K, sp)
    rng = np.random.default_rng(33)
    beta = np.stack([_uniform(o, range(L), rng) for _ in range(L)])
    alpha = np.stack([_uniform(o, range(L), rng) for _ in range(L)])
    c0 = np.stack([_uniform(o, range(Li), rng), _uniform(o, range(Li), rng)])
    c1 = np.stack([_uniform(o, range(Li), rng), _uniform(o, range(Li), rng)])
    wb, wa = o.mul_relin_rescale(c0[0], c0[1], c1[0], c1[1], beta, alpha)
    g.load_mult_key(dev(beta), dev(alpha))
    out = torch.zeros((2, 2, Li - 1, o.N), dtype=torch.int64, device="cuda")
    both0, both1 = np.stack([c0, c1]), np.stack([c1, c0])  # batch of two: (c0*c1) and (c1*c0) must agree as well
    g.mul_relin_rescale(out, dev(both0), dev(both1), Li, batch=2)
    h = host(out)
    assert np.array_equal(h[0, 0], wb) and np.array_equal(h[0, 1], wa)
    assert np.array_equal(h[1], h[0])
    del beta, alpha
    g.close(); _ctx_cache.pop((logN, L, K, sp), None)


@pytest.mark.parametrize("sp", [True, False])
def test_config4_full_size_bit_exact(sp):
    """BASELINE config 4 at full size (N=2^16, Li=L=K=30): one rotation key switch equal to the CPU oracle's, limb for limb, on both
    prime families; once as the library picks its base-conversion kernels for a batch of two, once with both conversions on the
    matrix cores (what the batch-64 bench shape runs)."""
    logN, L, K, Li = 16, 30, 30, 30
    o, g = ctx_pair(logN, L, K, sp)
    rng = np.random.default_rng(44)
    gal = o.rotation_galois(1)
    rkb, rka = _uniform(o, range(L + K), rng), _uniform(o, range(L + K), rng)
    ct = np.stack([_uniform(o, range(Li), rng), _uniform(o, range(Li), rng)])
    wb, wa = o.rotate(ct[0], ct[1], gal, rkb, rka)
    g.load_rot_key(gal, dev(rkb), dev(rka))
    d = dev(np.stack([ct, ct]))
    g.rotate(gal, d, Li, batch=2)
    h = host(d)
    assert np.array_equal(h[0, 0], wb) and np.array_equal(h[0, 1], wa) and np.array_equal(h[1], h[0])
    g.set_tuning(fhe_amd.hip.TUNE_BCONV, 2)
    d = dev(np.stack([ct, ct]))
    g.rotate(gal, d, Li, batch=2)
    h = host(d)
    assert np.array_equal(h[0, 0], wb) and np.array_equal(h[0, 1], wa) and np.array_equal(h[1], h[0])
    g.close(); _ctx_cache.pop((logN, L, K, sp), None)


@pytest.mark.parametrize("logN,L,K,Li,B", [(13, 6, 2, 5, 3), (14, 4, 4, 2, 2), (15, 3, 1, 3, 1), (13, 9, 3, 9, 2), (15, 16, 2, 16, 2)])  # last: BASELINE config 5 shape
def test_shape_sweep_relin_and_rotate(logN, L, K, Li, B):
    """Ragged shapes at mid sizes: Li < L, odd batches, K != L; multiply+relin+rescale and both rotation variants vs the oracle."""
    o, g = ctx_pair(logN, L, K)
    rng = np.random.default_rng(logN * 100 + Li)
    beta = np.stack([_uniform(o, range(L), rng) for _ in range(L)])
    alpha = np.stack([_uniform(o, range(L), rng) for _ in range(L)])
    g.load_mult_key(dev(beta), dev(alpha))
    c0 = np.stack([np.stack([_uniform(o, range(Li), rng), _uniform(o, range(Li), rng)]) for _ in range(B)])
    c1 = np.stack([np.stack([_uniform(o, range(Li), rng), _uniform(o, range(Li), rng)]) for _ in range(B)])
    out = torch.zeros((B, 2, Li - 1, o.N), dtype=torch.int64, device="cuda")
    g.mul_relin_rescale(out, dev(c0), dev(c1), Li, batch=B)
    h = host(out)
    for b in range(B):
        wb, wa = o.mul_relin_rescale(c0[b, 0], c0[b, 1], c1[b, 0], c1[b, 1], beta, alpha)
        assert np.array_equal(h[b, 0], wb) and np.array_equal(h[b, 1], wa)
    gal = o.rotation_galois(2)
    rkb, rka = _uniform(o, range(L + K), rng), _uniform(o, range(L + K), rng)
    g.load_rot_key(gal, dev(rkb), dev(rka))
    d = dev(c0)
    g.rotate(gal, d, Li, batch=B)
    h = host(d)
    for b in range(B):
        wb, wa = o.rotate(c0[b, 0], c0[b, 1], gal, rkb, rka)
        assert np.array_equal(h[b, 0], wb) and np.array_equal(h[b, 1], wa)
    mb = np.stack([_uniform(o, range(L + 1), rng) for _ in range(L)])
    ma = np.stack([_uniform(o, range(L + 1), rng) for _ in range(L)])
    # key limb L of a MixedRotKey lives under the special prime P[L]
    for i in range(L):
        mb[i, L] %= np.uint64(int(o.primes[L])); ma[i, L] %= np.uint64(int(o.primes[L]))
    g.load_mixed_rot_key(gal, dev(mb), dev(ma))
    d = dev(c1)
    g.rotate_mixed(gal, d, Li, batch=B)
    h = host(d)
    for b in range(B):
        wb, wa = o.rotate_mixed(c1[b, 0], c1[b, 1], gal, mb, ma)
        assert np.array_equal(h[b, 0], wb) and np.array_equal(h[b, 1], wa)


@pytest.mark.parametrize("logN,L,Li,group", [(12, 5, 5, 1), (12, 5, 4, 2), (13, 4, 4, 3), (10, 3, 3, 7)])
def test_relinearisation_in_groups_of_output_limbs(logN, L, Li, group):
    """FHE_HIP_TUNE_RELIN_GROUP: the column-pass output is produced and consumed `group` output limbs at a time; results
    must not depend on it (fhe_hip_set_tuning, include/fhe_hip.h)."""
    o, g = ctx_pair(logN, L, 1)
    sx, ext = o.keygen_sk(seed=900 + logN)
    beta, alpha = o.keygen_mult(sx, seed=901)
    g.load_mult_key(dev(beta), dev(alpha))
    B = 2
    c0 = np.stack([np.stack([rand_poly(o, range(Li), 910 + 4 * b), rand_poly(o, range(Li), 911 + 4 * b)]) for b in range(B)])
    c1 = np.stack([np.stack([rand_poly(o, range(Li), 912 + 4 * b), rand_poly(o, range(Li), 913 + 4 * b)]) for b in range(B)])
    want = np.stack([np.stack(o.mul_relin_rescale(c0[b, 0], c0[b, 1], c1[b, 0], c1[b, 1], beta, alpha)) for b in range(B)])
    try:
        g.set_tuning(fhe_amd.hip.TUNE_RELIN_GROUP, group)
        out = torch.zeros((B, 2, Li - 1, o.N), dtype=torch.int64, device="cuda")
        g.mul_relin_rescale(out, dev(c0), dev(c1), Li, batch=B)
        assert np.array_equal(host(out), want)
    finally:
        g.set_tuning(fhe_amd.hip.TUNE_RELIN_GROUP, 0)
    with pytest.raises(fhe_amd.FheHipError):
        g.set_tuning(99, 1)


@pytest.mark.parametrize("logN", [12, 13, 14])
@pytest.mark.parametrize("generic", [False, True])
def test_one_pass_ntt_matches_the_pass_pair_and_the_oracle(logN, generic):
    """N = 2^12 .. 2^14: plain NTTs run as one kernel with the limb in LDS (ntt_onepass_*); FHE_HIP_TUNE_ONEPASS = 0 selects the
    column/row pass pair.  Both must give the oracle's words: forward from lazy inputs (< 4p), in place and broadcast
    (one source limb under several primes), inverse, batch > 1, and a limb list with repeats."""
    o, g = ctx_pair(logN, 3, 1, sp=not generic)
    idx = [3, 0, 2, 2, 1]
    x = rand_poly(o, idx, 3000 + logN)
    want_f = o.ntt_forward(x.copy(), idx)
    lazy = x + np.array([3 * int(o.primes[i]) for i in idx], dtype=np.uint64)[:, None]   # same residues, values in [3p, 4p)
    xs = np.stack([rand_poly(o, range(3), 3100 + logN + b) for b in range(3)])
    want_r = np.stack([o.rescale(xs[b]) for b in range(3)])
    try:
        for mode in (1, 0):
            g.set_tuning(fhe_amd.hip.TUNE_ONEPASS, mode)
            d = dev(lazy); g.ntt_forward(d, idx)
            assert np.array_equal(host(d), want_f), mode
            g.ntt_backward(d, idx)
            assert np.array_equal(host(d), x), mode
            rop = torch.zeros((3, 2, o.N), dtype=torch.int64, device="cuda")
            g.rescale(rop, dev(xs), 3, batch=3)     # inverse of one limb + broadcast forward with an epilogue (pass pair) per entry
            assert np.array_equal(host(rop), want_r), mode
    finally:
        g.set_tuning(fhe_amd.hip.TUNE_ONEPASS, 1)


# ---- full-size results pinned by mathematics, not only by the oracle ----
@pytest.mark.parametrize("logN", [14, 15, 16])
def test_full_size_forward_ntt_against_horner_in_python_integers(logN):
    from horner import ntt_value_at, sample_positions
    o, g = ctx_pair(logN, 2, 1)
    rng = np.random.default_rng(2000 + logN)
    idx = [0, 2, 1]
    x = rand_poly(o, idx, 2100 + logN)
    d = dev(x)
    g.ntt_forward(d, idx)
    got = host(d)
    for row, m in enumerate(idx):
        p, psi = int(o.primes[m]), int(o.psi[m])
        coeffs = [int(v) for v in x[row]]
        for pos in sample_positions(logN, rng, 6):
            assert int(got[row][pos]) == ntt_value_at(coeffs, pos, logN, psi, p), (logN, m, pos)


def test_full_size_relinearisation_against_python_integers():
    """N = 2^16: sampled output coefficients of fhe_hip_relin_only recomputed from Horner evaluations of the digit
    polynomials and the key entries (no oracle in the loop except for the digits, which are pinned by evaluation too)."""
    from horner import ntt_value_at
    logN, L, K, Li = 16, 3, 1, 3
    o, g = ctx_pair(logN, L, K)
    rng = np.random.default_rng(78)
    poly = lambda: np.stack([rng.integers(0, int(o.primes[i]), o.N, dtype=np.uint64) for i in range(Li)])  # noqa: E731
    ct3 = np.stack([poly(), poly(), poly()])
    beta, alpha = np.stack([poly() for _ in range(L)]), np.stack([poly() for _ in range(L)])
    g.load_mult_key(dev(beta), dev(alpha))
    out = torch.zeros((1, 2, Li, o.N), dtype=torch.int64, device="cuda")
    g.relin_only(out, dev(ct3[None]), Li, batch=1)
    got = host(out)[0]
    dd = dev(ct3[2].copy())
    g.ntt_backward(dd, list(range(Li)))
    digits = host(dd)
    dig = [[int(v) for v in digits[j]] for j in range(Li)]
    for j in range(Li):
        assert ntt_value_at(dig[j], 777, logN, int(o.psi[j]), int(o.primes[j])) == int(ct3[2][j][777])
    for i, d in ((0, 5), (2, 65535), (1, 32768)):
        pi, si = int(o.primes[i]), int(o.psi[i])
        vals = [ntt_value_at(dig[j], d, logN, si, pi) for j in range(Li)]
        assert int(got[0][i][d]) == (sum(vals[j] * int(beta[j][i][d]) for j in range(Li)) + int(ct3[0][i][d])) % pi
        assert int(got[1][i][d]) == (sum(vals[j] * int(alpha[j][i][d]) for j in range(Li)) + int(ct3[1][i][d])) % pi


@pytest.mark.parametrize("logN,L,K,Li", [(12, 4, 4, 4), (10, 3, 2, 2), (13, 5, 3, 4)])
def test_rotate_out_of_place_matches_in_place_and_oracle(logN, L, K, Li):
    """fhe_hip_rotate_to (nine-launch pipeline: galois applied inside the loads, key product of the normal limbs and + galois(bx)
    in the mod-down tail) against the oracle and against fhe_hip_rotate, which forms the same result beside the input and copies."""
    o, g = ctx_pair(logN, L, K)
    sx, ext = o.keygen_sk(seed=300 + logN)
    B = 3
    ct = np.stack([np.stack([rand_poly(o, range(Li), 310 + 2 * b), rand_poly(o, range(Li), 311 + 2 * b)]) for b in range(B)])
    for offset in (1, 5):
        gal = o.rotation_galois(offset)
        rkb, rka = o.keygen_rot(sx, ext, gal, seed=320 + offset)
        g.load_rot_key(gal, dev(rkb), dev(rka))
        want = np.stack([np.stack(o.rotate(ct[b, 0], ct[b, 1], gal, rkb, rka)) for b in range(B)])
        src = dev(ct)
        out = torch.full((B, 2, Li, o.N), -1, dtype=torch.int64, device="cuda")
        g.rotate_to(gal, out, src, Li, batch=B)
        assert np.array_equal(host(out), want) and np.array_equal(host(src), ct)  # input untouched
        inplace = dev(ct)
        g.rotate(gal, inplace, Li, batch=B)
        assert np.array_equal(host(inplace), want)
        hout = np.zeros_like(ct)
        g.rotate_to(gal, hout, ct.copy(), Li, batch=B)  # host buffers
        assert np.array_equal(hout, want)
    # galois == 1: a key switch with the identity automorphism (no index table); galois == 0: copy
    rkb, rka = o.keygen_rot(sx, ext, 1, seed=399)
    g.load_rot_key(1, dev(rkb), dev(rka))
    want = np.stack([np.stack(o.rotate(ct[b, 0], ct[b, 1], 1, rkb, rka)) for b in range(B)])
    out = torch.zeros((B, 2, Li, o.N), dtype=torch.int64, device="cuda")
    g.rotate_to(1, out, dev(ct), Li, batch=B)
    assert np.array_equal(host(out), want)
    g.rotate_to(0, out, dev(ct), Li, batch=B)
    assert np.array_equal(host(out), ct)
    src = dev(ct)
    with pytest.raises(fhe_amd.FheHipError) as e:
        g.rotate_to(3, src, src, Li, batch=B)
    assert e.value.code == -1
    # partially overlapping ranges are refused as well (out = in + one half: the last kernel would gather what it overwrites)
    big = torch.zeros((B * 2 * Li + Li, o.N), dtype=torch.int64, device="cuda")
    with pytest.raises(fhe_amd.FheHipError) as e:
        g.rotate_to(3, big[Li:], big[:B * 2 * Li], Li, batch=B)
    assert e.value.code == -1 and "overlap" in str(e.value)
    with pytest.raises(fhe_amd.FheHipError) as e:  # the limb count is validated before the galois == 0 copy
        g.rotate_to(0, out, dev(ct), o.L + 1, batch=0)
    assert e.value.code == -2


def test_full_size_rotation_and_multiplication_decrypt_to_the_right_plaintext():
    """N = 2^16, pinned by what the operations MEAN rather than by a second implementation: an encryption of m rotated on the GPU
    decrypts (on the GPU) to m(X^g) up to key-switch noise, and the product of two encryptions decrypts to the negacyclic
    product / q_last up to relinearisation + rescale noise.  Plaintext coefficients are lifted by CRT in Python integers at
    sampled positions; keys and fresh encryptions come from the definitions in the CPU restatement (keygen / encrypt only)."""
    logN, L, K = 16, 4, 4
    o, g = ctx_pair(logN, L, K)
    N = o.N
    q = [int(v) for v in o.primes[:L]]
    sx, ext = o.keygen_sk(seed=41)
    pkb, pka = o.keygen_pk(sx, seed=42)
    gal = o.rotation_galois(1)
    rkb, rka = o.keygen_rot(sx, ext, gal, seed=43)
    beta, alpha = o.keygen_mult(sx, seed=44)
    g.load_rot_key(gal, dev(rkb), dev(rka))
    g.load_mult_key(dev(beta), dev(alpha))
    rng = np.random.default_rng(9)
    sa, sb = rng.integers(-8, 9, N), rng.integers(-8, 9, N)

    def encode(small, shift):
        return np.stack([np.array([(int(v) << shift) % p for v in small], dtype=np.uint64) for p in q])

    def lifted(limbs, mods, d):
        Q = 1
        for m in mods:
            Q *= m
        x = sum(int(limbs[j, d]) * pow(Q // m, -1, m) % m * (Q // m) for j, m in enumerate(mods)) % Q
        return x - Q if x > Q // 2 else x

    def gpu_decrypt(ct, Li):
        out = torch.zeros((1, Li, N), dtype=torch.int64, device="cuda")
        g.decrypt(out, ct, dev(sx), Li, batch=1)
        return host(out)[0]

    pos = [0, 1, N - 1, N // 2] + [int(v) for v in rng.integers(0, N, 12)]
    # rotation: coefficient d of m lands on d*g mod 2N, negated when it wraps past N
    ca = np.stack(o.encrypt(encode(sa, 50), pkb, pka, seed=45))
    rot = torch.zeros((1, 2, L, N), dtype=torch.int64, device="cuda")
    g.rotate_to(gal, rot, dev(ca[None]), L, batch=1)
    dec = gpu_decrypt(rot, L)
    for d in pos:
        e = d * gal % (2 * N)
        want, at = ((int(sa[d]) << 50), e) if e < N else (-(int(sa[d]) << 50), e - N)
        assert abs(lifted(dec, q, at) - want) < 1 << 24, d
    # the same rotation through the MixedRotKey key switch (digit decomposition + one special prime)
    mb, ma = o.keygen_mixed_rot(sx, ext, gal, seed=48)
    g.load_mixed_rot_key(gal, dev(mb), dev(ma))
    rot = dev(ca[None].copy())
    g.rotate_mixed(gal, rot, L, batch=1)
    dec = gpu_decrypt(rot, L)
    for d in pos:
        e = d * gal % (2 * N)
        want, at = ((int(sa[d]) << 50), e) if e < N else (-(int(sa[d]) << 50), e - N)
        assert abs(lifted(dec, q, at) - want) < 1 << 24, d
    # multiplication: (a * b mod X^N + 1) / q_last
    c0 = np.stack(o.encrypt(encode(sa, 40), pkb, pka, seed=47))
    c1 = np.stack(o.encrypt(encode(sb, 40), pkb, pka, seed=46))
    out = torch.zeros((1, 2, L - 1, N), dtype=torch.int64, device="cuda")
    g.mul_relin_rescale(out, dev(c0[None]), dev(c1[None]), L, batch=1)
    dec = gpu_decrypt(out, L - 1)
    idx = np.arange(N)
    for e in pos:
        sign = np.where(idx <= e, 1, -1)
        coef = int(np.sum(sa.astype(np.int64) * sb[(e - idx) % N].astype(np.int64) * sign)) << 80
        assert abs(lifted(dec, q[:L - 1], e) * q[L - 1] - coef) < 1 << 80, e
