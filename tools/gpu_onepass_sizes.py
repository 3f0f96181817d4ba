"""One-pass NTT against the column/row pass pair at N = 2^12, 2^13, 2^14 (256 MiB of limbs per call, forward + inverse in place):
tools/gpu_onepass_sizes.py"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import fhe_amd
from bench import gen_primes_host

for logN in (12, 13, 14):
    M = 8
    pr, ps = gen_primes_host(logN, M)
    g = fhe_amd.HipContext(logN, M, 0, pr, ps)
    N = 1 << logN
    limbs = (256 << 20) // (8 * N)
    x = torch.randint(0, int(min(pr)), (limbs, N), dtype=torch.int64, device="cuda")
    idx = np.arange(limbs, dtype=np.uint32) % M
    out = []
    for mode in (1, 0):
        g.set_tuning(fhe_amd.hip.TUNE_ONEPASS, mode)
        def step():
            g.ntt_forward(x, idx); g.ntt_backward(x, idx)
        t_end = time.perf_counter() + 0.4
        while time.perf_counter() < t_end:
            step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize()
        out.append(2 * limbs * 20 / (time.perf_counter() - t0) / 1e6)
    print("N=2^%d  %6d limbs  one-pass %.2f M limb-NTT/s  pass pair %.2f M  (%+.0f %%)  = %.0f / %.0f GB/s algorithmic" % (
        logN, limbs, out[0], out[1], 100 * (out[0] / out[1] - 1), out[0] * 16 * N / 1e3, out[1] * 16 * N / 1e3), flush=True)
    g.close()
