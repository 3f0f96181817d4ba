"""fhe_hip_batch_run from pageable and from pinned host memory (C5-shaped ops, one context): tools/gpu_batch_run_pinned.py"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import fhe_amd
from fhe_amd import hip
from bench import gen_primes_host

logN, Lc, Kc, Li, batch = 15, 16, 1, 16, 192
pr, ps = gen_primes_host(logN, Lc + Kc)
g = fhe_amd.HipContext(logN, Lc, Kc, pr, ps)
N = 1 << logN
gen = torch.Generator(device="cuda").manual_seed(5)
rnd = lambda *s: torch.randint(0, 1 << 61, s, dtype=torch.int64, device="cuda", generator=gen)
g.load_mult_key(rnd(Lc, Lc, N), rnd(Lc, Lc, N))
src0, src1 = rnd(batch, 2, Li, N).cpu(), rnd(batch, 2, Li, N).cpu()
res = {}
for kind in ("pageable", "pinned"):
    pin = kind == "pinned"
    c0 = torch.empty(src0.shape, dtype=torch.int64, pin_memory=pin); c0.copy_(src0)
    c1 = torch.empty(src1.shape, dtype=torch.int64, pin_memory=pin); c1.copy_(src1)
    out = torch.zeros((batch, 2, Li - 1, N), dtype=torch.int64, pin_memory=pin)
    a0, a1, ao = (t.numpy().view(np.uint64) for t in (c0, c1, out))
    hip.batch_run([g], hip.OP_MUL_RELIN_RESCALE, ao, a0, a1, Li=Li, batch=batch)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); hip.batch_run([g], hip.OP_MUL_RELIN_RESCALE, ao, a0, a1, Li=Li, batch=batch); ts.append(time.perf_counter() - t0)
    res[kind] = ao.copy()
    print("%-9s %.1f ms  %.0f op/s  (%.1f GB/s up, %.1f GB/s down)" % (kind, min(ts) * 1e3, batch / min(ts), batch * 2 * 2 * Li * N * 8 / min(ts) / 1e9,
                                                                       batch * 2 * (Li - 1) * N * 8 / min(ts) / 1e9), flush=True)
assert np.array_equal(res["pageable"], res["pinned"])
print("same results")
