"""Experiment: does running two half-batches of c3 on two streams (two contexts) beat one full batch on one stream?
(kernels of one half that wait on memory could overlap the other half's VALU-bound kernels)"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import fhe_amd
from bench import gen_primes_host

logN, L, K, Li, B = 16, 30, 1, 30, 16
N = 1 << logN
primes, psi = gen_primes_host(logN, L + K)
gen = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *s: torch.randint(0, 1 << 61, s, dtype=torch.int64, device="cuda", generator=gen)
beta, alpha = rnd(L, L, N), rnd(L, L, N)
c0, c1 = rnd(B, 2, Li, N), rnd(B, 2, Li, N)
out = torch.empty((B, 2, Li - 1, N), dtype=torch.int64, device="cuda")

def run(nstreams, offset_frac, steps=60):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    ctxs = []
    for s in streams:
        c = fhe_amd.HipContext(logN, L, K, primes, psi)
        c.set_stream(s.cuda_stream)
        c.load_mult_key(beta, alpha)
        ctxs.append(c)
    per = B // nstreams
    def step():
        for i, c in enumerate(ctxs):
            sl = slice(i * per, (i + 1) * per)
            c.mul_relin_rescale(out[sl], c0[sl], c1[sl], Li, batch=per)
    torch.cuda.synchronize()
    t_end = time.perf_counter() + 0.6
    while time.perf_counter() < t_end:
        step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for c in ctxs:
        c.close()
    return B * steps / dt

for ns in (1, 2, 4, 1, 2):
    print("streams", ns, "ops/s", round(run(ns, 0), 1), flush=True)
