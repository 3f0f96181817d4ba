"""HIP-graph capture of the entry point that allocates nothing: tools/gpu_graph_probe.py
Captures fhe_hip_mul_relin_rescale_noalloc at the C1 size (N=2^13, L=4) on a side stream with torch.cuda.CUDAGraph, replays it,
checks the result against the eager call and times eager launches against graph replays (the small configuration is launch-bound)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import fhe_amd

logN, L, K, Li, B = 13, 4, 4, 4, 1
from bench import gen_primes_host
pr, ps = gen_primes_host(logN, L + K)
g = fhe_amd.HipContext(logN, L, K, pr, ps)
N = 1 << logN
gen = torch.Generator(device="cuda").manual_seed(5)
rnd = lambda *s: torch.randint(0, int(min(pr)), s, dtype=torch.int64, device="cuda", generator=gen)
g.load_mult_key(rnd(L, L, N), rnd(L, L, N))
c0, c1 = rnd(B, 2, Li, N), rnd(B, 2, Li, N)
want = torch.zeros((B, 2, Li - 1, N), dtype=torch.int64, device="cuda")
g.mul_relin_rescale(want, c0, c1, Li, batch=B)
out = torch.zeros_like(want)
g.mul_relin_rescale_noalloc(out, c0, c1, Li, batch=B)   # warm: galois tables, first-call set-up
torch.cuda.synchronize()
assert torch.equal(out, want)
side = torch.cuda.Stream()
graph = torch.cuda.CUDAGraph()
out.zero_()
with torch.cuda.stream(side):
    g.set_stream(side.cuda_stream)
    with torch.cuda.graph(graph, stream=side):
        g.mul_relin_rescale_noalloc(out, c0, c1, Li, batch=B)
torch.cuda.synchronize()
assert not torch.equal(out, want), "capture must not execute"
graph.replay(); torch.cuda.synchronize()
assert torch.equal(out, want), "graph replay differs from the eager result"
c0.copy_(rnd(B, 2, Li, N)); g.set_stream(torch.cuda.current_stream().cuda_stream)
g.mul_relin_rescale(want, c0, c1, Li, batch=B); torch.cuda.synchronize()
graph.replay(); torch.cuda.synchronize()
assert torch.equal(out, want), "replay on new input contents differs"
def timeit(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
eager = timeit(lambda: g.mul_relin_rescale_noalloc(out, c0, c1, Li, batch=B))
replay = timeit(graph.replay)
print("graph capture OK: eager %.1f us/op, graph replay %.1f us/op (N=2^%d, Li=%d, batch %d)" % (eager, replay, logN, Li, B))
