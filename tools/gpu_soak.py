"""Randomised soak of the C-ABI against the CPU oracle (many shapes): tools/gpu_soak.py [seconds] [seed] [logN_lo] [logN_hi] [generic] [Lmax] [bconv].
`bconv` = FHE_HIP_TUNE_BCONV for every context (2: base conversions on the matrix cores at every shape that allows it; default 0 picks by size).
`generic` = 1 alternates the repository's prime table with primes that are not 1 mod 2^32 (generic Montgomery kernels).
Not part of the test suite (run time); every mismatch prints the configuration and exits non-zero."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import fhe_amd
from oracle import oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
host = lambda t: t.cpu().numpy().view(np.uint64)
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 10
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 13
generic = len(sys.argv) > 5 and sys.argv[5] == "1"
Lmax = int(sys.argv[6]) if len(sys.argv) > 6 else 8
bconv = int(sys.argv[7]) if len(sys.argv) > 7 else 0


def generic_primes(logN, count):
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


t_end = time.time() + budget
n_cfg = n_chk = 0
while time.time() < t_end:
    logN = int(rng.integers(lo, hi + 1))
    L = int(rng.integers(2, Lmax + 1)); K = int(rng.integers(1, L + 2)); Li = int(rng.integers(2, L + 1)); B = int(rng.integers(1, 4))
    pr, ps = generic_primes(logN, L + K) if (generic and n_cfg % 2) else orc.gen_primes(logN, L + K)
    o = orc.Oracle(logN, L, K, pr, ps)
    g = fhe_amd.HipContext(logN, L, K, pr, ps)
    if bconv:
        g.set_tuning(fhe_amd.hip.TUNE_BCONV, bconv)
    N = o.N
    cfg = dict(logN=logN, L=L, K=K, Li=Li, B=B, generic=bool(generic and n_cfg % 2))
    def uni(idx):
        return np.stack([rng.integers(0, int(o.primes[i]), N, dtype=np.uint64) for i in idx])
    def check(name, got, want):
        global n_chk
        n_chk += 1
        if not np.array_equal(got, want):
            print("MISMATCH", name, cfg, flush=True); sys.exit(1)
    # NTT round trip and forward against the oracle, ragged modulus list
    idx = rng.integers(0, L + K, size=int(rng.integers(1, 7))).astype(np.uint32)
    x = uni(idx)
    d = dev(x); g.ntt_forward(d, idx); check("ntt_fwd", host(d), o.ntt_forward(x.copy(), idx))
    g.ntt_backward(d, idx); check("ntt_roundtrip", host(d), x)
    # multiply + relin + rescale
    beta = np.stack([uni(range(L)) for _ in range(L)]); alpha = np.stack([uni(range(L)) for _ in range(L)])
    g.load_mult_key(dev(beta), dev(alpha))
    c0 = np.stack([np.stack([uni(range(Li)), uni(range(Li))]) for _ in range(B)])
    c1 = np.stack([np.stack([uni(range(Li)), uni(range(Li))]) for _ in range(B)])
    out = torch.zeros((B, 2, Li - 1, N), dtype=torch.int64, device="cuda")
    g.mul_relin_rescale(out, dev(c0), dev(c1), Li, batch=B)
    for b in range(B):
        wb, wa = o.mul_relin_rescale(c0[b, 0], c0[b, 1], c1[b, 0], c1[b, 1], beta, alpha)
        check("mul_relin_rescale", host(out)[b], np.stack([wb, wa]))
    if B * (9 * Li - 2) * N * 8 <= fhe_amd.hip.NOALLOC_SCRATCH_BYTES:
        out2 = torch.zeros_like(out); g.mul_relin_rescale_noalloc(out2, dev(c0), dev(c1), Li, batch=B)
        check("mul_relin_rescale_noalloc", host(out2), host(out))
    # rotations with a random galois element
    gal = int(rng.integers(1, N)) * 2 + 1
    rkb, rka = uni(range(L + K)), uni(range(L + K))
    g.load_rot_key(gal, dev(rkb), dev(rka))
    d = dev(c0); g.rotate(gal, d, Li, batch=B)
    for b in range(B):
        wb, wa = o.rotate(c0[b, 0], c0[b, 1], gal, rkb, rka)
        check("rotate", host(d)[b], np.stack([wb, wa]))
    d2 = torch.zeros_like(d); g.rotate_to(gal, d2, dev(c0), Li, batch=B); check("rotate_to", host(d2), host(d))
    d = dev(c1); g.rotate_non_ntt(gal, d, Li, batch=B)
    for b in range(B):
        wb, wa = o.rotate_non_ntt(c1[b, 0], c1[b, 1], gal, rkb, rka)
        check("rotate_non_ntt", host(d)[b], np.stack([wb, wa]))
    mb = np.stack([uni(list(range(L)) + [L]) for _ in range(L)]); ma = np.stack([uni(list(range(L)) + [L]) for _ in range(L)])
    g.load_mixed_rot_key(gal, dev(mb), dev(ma))
    d = dev(c0); g.rotate_mixed(gal, d, Li, batch=B)
    for b in range(B):
        wb, wa = o.rotate_mixed(c0[b, 0], c0[b, 1], gal, mb, ma)
        check("rotate_mixed", host(d)[b], np.stack([wb, wa]))
    # base conversions, galois, encrypt / decrypt
    xs = np.stack([uni(range(Li)) for _ in range(B)])
    rop = torch.zeros((B, Li + K, N), dtype=torch.int64, device="cuda"); g.mod_up(rop, dev(xs), Li, batch=B)
    check("mod_up", host(rop), np.stack([o.mod_up(xs[b]) for b in range(B)]))
    ys = np.stack([uni(list(range(Li)) + list(range(L, L + K))) for _ in range(B)])
    rop = torch.zeros((B, Li, N), dtype=torch.int64, device="cuda"); g.mod_down(rop, dev(ys), Li, batch=B)
    check("mod_down", host(rop), np.stack([o.mod_down(ys[b]) for b in range(B)]))
    rop = torch.zeros((B, Li - 1, N), dtype=torch.int64, device="cuda"); g.rescale(rop, dev(xs), Li, batch=B)
    check("rescale", host(rop), np.stack([o.rescale(xs[b]) for b in range(B)]))
    d = dev(xs); g.apply_galois(d, Li, gal, batch=B); check("galois", host(d), np.stack([o.apply_galois(xs[b].copy(), gal) for b in range(B)]))
    d = dev(xs); g.apply_galois_non_ntt(d, Li, gal, batch=B); check("galois_nn", host(d), np.stack([o.apply_galois_non_ntt(xs[b].copy(), gal) for b in range(B)]))
    sx = uni(range(L)); pkb, pka = uni(range(L)), uni(range(L))
    u, e0m, e1 = xs, np.stack([uni(range(Li)) for _ in range(B)]), np.stack([uni(range(Li)) for _ in range(B)])
    ct = torch.zeros((B, 2, Li, N), dtype=torch.int64, device="cuda"); g.encrypt(ct, dev(u), dev(e0m), dev(e1), dev(pkb), dev(pka), Li, batch=B)
    want = np.stack([np.stack(o.encrypt_core(u[b], e0m[b], e1[b], pkb, pka)) for b in range(B)])
    check("encrypt", host(ct), want)
    dd = torch.zeros((B, Li, N), dtype=torch.int64, device="cuda"); g.decrypt(dd, ct, dev(sx), Li, batch=B)
    check("decrypt", host(dd), np.stack([o.decrypt(want[b, 0], want[b, 1], sx) for b in range(B)]))
    g.close()
    n_cfg += 1
    if n_cfg % 10 == 0:
        print("configs", n_cfg, "checks", n_chk, flush=True)
print("soak OK: %d configurations, %d checks" % (n_cfg, n_chk))
