"""GPU: the entry point that allocates nothing (fhe_hip_mul_relin_rescale_noalloc, include/fhe_hip.h) against the allocating
entry point it is a variant of (fhe_hip_mul_relin_rescale), through the C-ABI, bit-exact, on two calls in a row, the second
with a smaller input (the static scratch array keeps the first call's contents)."""
import numpy as np
import pytest

import fhe_amd
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _rand(shape, gen):
    return torch.randint(0, 1 << 61, shape, dtype=torch.int64, device="cuda", generator=gen)


def test_noalloc_entry_point_matches_the_allocating_one_twice():
    logN, L, K = 13, 6, 1
    pr, ps = orc.gen_primes(logN, L + K)
    g = fhe_amd.HipContext(logN, L, K, pr, ps)
    N = 1 << logN
    gen = torch.Generator(device="cuda").manual_seed(77)
    g.load_mult_key(_rand((L, L, N), gen), _rand((L, L, N), gen))
    for Li, B in ((6, 5), (4, 2)):  # second call: fewer limbs and a smaller batch
        c0, c1 = _rand((B, 2, Li, N), gen), _rand((B, 2, Li, N), gen)
        want = torch.zeros((B, 2, Li - 1, N), dtype=torch.int64, device="cuda")
        g.mul_relin_rescale(want, c0, c1, Li, batch=B)
        got = torch.full((B, 2, Li - 1, N), -1, dtype=torch.int64, device="cuda")
        g.mul_relin_rescale_noalloc(got, c0, c1, Li, batch=B)
        g.sync()
        assert torch.equal(got, want)
    # the allocating entry point still works after its workspaces were set aside and restored
    g.mul_relin_rescale(want, c0, c1, Li, batch=B)
    g.sync()
    assert torch.equal(got, want)
    g.close()


def test_noalloc_against_the_oracle_and_in_forced_groups():
    """Small shape against the CPU oracle; then a shape whose column-pass output cannot fit the scratch at once, so the
    relinearisation has to run in groups of output limbs."""
    logN, L, K, Li = 12, 4, 1, 4
    pr, ps = orc.gen_primes(logN, L + K)
    o, g = orc.Oracle(logN, L, K, pr, ps), fhe_amd.HipContext(logN, L, K, pr, ps)
    sx, ext = o.keygen_sk(seed=5)
    beta, alpha = o.keygen_mult(sx, seed=6)
    dev = lambda a: torch.from_numpy(a.view(np.int64)).cuda()
    g.load_mult_key(dev(beta), dev(alpha))
    rng = np.random.default_rng(9)
    poly = lambda: np.stack([rng.integers(0, int(pr[i]), o.N, dtype=np.uint64) for i in range(Li)])
    c0, c1 = np.stack([poly(), poly()]), np.stack([poly(), poly()])
    wb, wa = o.mul_relin_rescale(c0[0], c0[1], c1[0], c1[1], beta, alpha)
    out = torch.zeros((1, 2, Li - 1, o.N), dtype=torch.int64, device="cuda")
    g.mul_relin_rescale_noalloc(out, dev(c0[None]), dev(c1[None]), Li, batch=1)
    h = out.cpu().numpy().view(np.uint64)
    assert np.array_equal(h[0, 0], wb) and np.array_equal(h[0, 1], wa)
    g.close()
    # N = 2^16, Li = 12, batch 3: 3 * (9*12 - 2) limbs of 512 KiB = 159 MiB minimum; all output limbs at once would need
    # 3 * (8*12 - 2 + 144) limbs = 357 MiB: fits the 256 MiB only with the relinearisation in groups of output limbs
    logN, L, K, Li, B = 16, 12, 1, 12, 3
    pr, ps = orc.gen_primes(logN, L + K)
    g = fhe_amd.HipContext(logN, L, K, pr, ps)
    N = 1 << logN
    gen = torch.Generator(device="cuda").manual_seed(78)
    g.load_mult_key(_rand((L, L, N), gen), _rand((L, L, N), gen))
    c0, c1 = _rand((B, 2, Li, N), gen), _rand((B, 2, Li, N), gen)
    want = torch.zeros((B, 2, Li - 1, N), dtype=torch.int64, device="cuda")
    g.mul_relin_rescale(want, c0, c1, Li, batch=B)
    got = torch.zeros_like(want)
    g.mul_relin_rescale_noalloc(got, c0, c1, Li, batch=B)
    g.sync()
    assert torch.equal(got, want)
    g.close()


def test_noalloc_rejects_what_does_not_fit_and_host_buffers():
    logN, L, K, Li = 16, 12, 1, 12
    pr, ps = orc.gen_primes(logN, L + K)
    g = fhe_amd.HipContext(logN, L, K, pr, ps)
    N = 1 << logN
    gen = torch.Generator(device="cuda").manual_seed(79)
    g.load_mult_key(_rand((L, L, N), gen), _rand((L, L, N), gen))
    B = 8  # 8 * 106 limbs * 512 KiB = 424 MiB > 256 MiB
    c0 = _rand((B, 2, Li, N), gen)
    out = torch.full((B, 2, Li - 1, N), 7, dtype=torch.int64, device="cuda")
    with pytest.raises(fhe_amd.FheHipError) as e:
        g.mul_relin_rescale_noalloc(out, c0, c0, Li, batch=B)
    assert e.value.code == -2 and "FHE_HIP_NOALLOC_SCRATCH_BYTES" in str(e.value)
    assert bool((out == 7).all())  # refused before anything ran
    with pytest.raises(TypeError):
        g.mul_relin_rescale_noalloc(np.zeros((1, 2, Li - 1, N), np.uint64), np.zeros((1, 2, Li, N), np.uint64),
                                    np.zeros((1, 2, Li, N), np.uint64), Li, batch=1)
    g.close()


def test_noalloc_exact_fit_of_the_documented_budget():
    """batch * (9 Li - 2) * N * 8 == FHE_HIP_NOALLOC_SCRATCH_BYTES exactly (N = 2^16, Li = 2, batch 32: 512 limbs of 512 KiB) is
    admitted and correct: the entry point's precheck and the relinearisation's own sizing use one formula.  One op more is refused
    before anything runs."""
    logN, L, K, Li, B = 16, 2, 1, 2, 32
    pr, ps = orc.gen_primes(logN, L + K)
    g = fhe_amd.HipContext(logN, L, K, pr, ps)
    N = 1 << logN
    assert B * (9 * Li - 2) * N * 8 == 256 << 20
    gen = torch.Generator(device="cuda").manual_seed(80)
    g.load_mult_key(_rand((L, L, N), gen), _rand((L, L, N), gen))
    c0, c1 = _rand((B + 1, 2, Li, N), gen), _rand((B + 1, 2, Li, N), gen)
    want = torch.zeros((B, 2, Li - 1, N), dtype=torch.int64, device="cuda")
    g.mul_relin_rescale(want, c0[:B].contiguous(), c1[:B].contiguous(), Li, batch=B)
    got = torch.full_like(want, -1)
    g.mul_relin_rescale_noalloc(got, c0[:B].contiguous(), c1[:B].contiguous(), Li, batch=B)
    g.sync()
    assert torch.equal(got, want)
    out = torch.full((B + 1, 2, Li - 1, N), 7, dtype=torch.int64, device="cuda")
    with pytest.raises(fhe_amd.FheHipError) as e:
        g.mul_relin_rescale_noalloc(out, c0, c1, Li, batch=B + 1)
    assert e.value.code == -2
    assert bool((out == 7).all())
    # the allocating entry point is untouched by the refusal (its workspaces were put back)
    g.mul_relin_rescale(got, c0[:B].contiguous(), c1[:B].contiguous(), Li, batch=B)
    g.sync()
    assert torch.equal(got, want)
    g.close()


def test_noalloc_entry_point_can_be_captured_in_a_hip_graph_and_replayed():
    """What the entry point is for: a caller under stream capture.  The call is captured on a side stream (nothing executes, no
    allocation, event or synchronisation is attempted inside it), and replays give the allocating entry point's bits, also
    after the input buffers' contents changed."""
    logN, L, K, Li, B = 13, 4, 1, 4, 2
    pr, ps = orc.gen_primes(logN, L + K)
    g = fhe_amd.HipContext(logN, L, K, pr, ps)
    N = 1 << logN
    gen = torch.Generator(device="cuda").manual_seed(78)
    g.load_mult_key(_rand((L, L, N), gen), _rand((L, L, N), gen))
    c0, c1 = _rand((B, 2, Li, N), gen), _rand((B, 2, Li, N), gen)
    want = torch.zeros((B, 2, Li - 1, N), dtype=torch.int64, device="cuda")
    out = torch.zeros_like(want)
    g.mul_relin_rescale(want, c0, c1, Li, batch=B)
    g.mul_relin_rescale_noalloc(out, c0, c1, Li, batch=B)  # first call outside the capture
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    out.zero_()
    side, graph = torch.cuda.Stream(), torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(side):
            g.set_stream(side.cuda_stream)
            with torch.cuda.graph(graph, stream=side):
                g.mul_relin_rescale_noalloc(out, c0, c1, Li, batch=B)
    finally:
        g.set_stream(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert not bool(out.any()), "a captured call must not run"
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    c0.copy_(_rand((B, 2, Li, N), gen))
    g.mul_relin_rescale(want, c0, c1, Li, batch=B)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    del graph
    g.close()
