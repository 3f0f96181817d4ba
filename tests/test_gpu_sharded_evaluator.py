"""GPU: fhe_amd.batching.ShardedEvaluator on one rank (the test box has one GPU; the world_size-2 exchange is covered on
CPU by tests/test_batching_gloo.py): chunked and unchunked runs must equal the CPU oracle bit for bit."""
import numpy as np
import pytest

import fhe_amd
from fhe_amd.batching import ShardedEvaluator
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("chunk", [None, 2])
def test_sharded_evaluator_single_rank(chunk):
    logN, L, K = 10, 3, 2
    pr, ps = orc.gen_primes(logN, L + K)
    o = orc.Oracle(logN, L, K, pr, ps)
    g = fhe_amd.HipContext(logN, L, K, pr, ps)
    N, Li, n = o.N, L, 5
    rng = np.random.default_rng(3)
    rnd = lambda *shape: rng.integers(0, int(min(o.primes)), size=shape, dtype=np.uint64)  # noqa: E731
    dev = lambda a: torch.from_numpy(a.view(np.int64)).cuda()  # noqa: E731
    host = lambda t: t.cpu().numpy().view(np.uint64)  # noqa: E731
    beta, alpha = rnd(L, L, N), rnd(L, L, N)
    gal = o.rotation_galois(1)
    rkb, rka = rnd(L + K, N), rnd(L + K, N)
    ev = ShardedEvaluator(g)
    ev.load_mult_key(dev(beta), dev(alpha))
    g.load_rot_key(gal, dev(rkb), dev(rka))
    c0, c1 = rnd(n, 2, Li, N), rnd(n, 2, Li, N)
    out = host(ev.mul_relin_rescale(dev(c0), dev(c1), n, Li, chunk=chunk))
    assert out.shape == (n, 2, Li - 1, N)
    for b in range(n):
        wb, wa = o.mul_relin_rescale(c0[b, 0], c0[b, 1], c1[b, 0], c1[b, 1], beta, alpha)
        assert np.array_equal(out[b, 0], wb) and np.array_equal(out[b, 1], wa)
    rot = host(ev.rotate(gal, dev(c0), n, Li, chunk=chunk))
    for b in range(n):
        wb, wa = o.rotate(c0[b, 0], c0[b, 1], gal, rkb, rka)
        assert np.array_equal(rot[b, 0], wb) and np.array_equal(rot[b, 1], wa)
    assert host(ev.mul_relin_rescale(dev(c0[:0]), dev(c1[:0]), 0, Li)).shape[0] == 0
