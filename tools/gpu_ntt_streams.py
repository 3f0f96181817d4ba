"""Stand-alone N=2^16 NTT over 2048 limbs issued as chunks on two streams (two contexts), so that the column pass of one chunk overlaps
the row pass of another and a chunk's intermediate can stay in the Infinity Cache: tools/gpu_ntt_streams.py
Prints limb-NTT/s for one call over all limbs, and for chunked issue on one and on two streams."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import fhe_amd
from bench import gen_primes_host

logN, M, LIMBS = 16, 8, 2048
pr, ps = gen_primes_host(logN, M)
N = 1 << logN
NS = 4
ctxs = [fhe_amd.HipContext(logN, M, 0, pr, ps) for _ in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]
x = torch.randint(0, int(min(pr)), (LIMBS, N), dtype=torch.int64, device="cuda")
idx_all = np.arange(LIMBS, dtype=np.uint32) % M

def run_single():
    ctxs[0].ntt_forward(x, idx_all)

def run_chunks(chunk, nstreams):
    evs = []
    for k, lo in enumerate(range(0, LIMBS, chunk)):
        s = k % nstreams
        with torch.cuda.stream(streams[s]):
            ctxs[s].set_stream(streams[s].cuda_stream)
            ctxs[s].ntt_forward(x[lo:lo + chunk], idx_all[lo:lo + chunk])
    for s in range(nstreams):
        torch.cuda.current_stream().wait_stream(streams[s])

def timeit(f, reps=30):
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        f(); torch.cuda.synchronize()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps

ctxs[0].set_stream(torch.cuda.current_stream().cuda_stream)
t = timeit(run_single); print("one call            %.3f ms  %.3f M limb-NTT/s" % (t * 1e3, LIMBS / t / 1e6), flush=True)
for chunk in (32, 64, 96, 128, 192, 256):
    for ns in (2, 3, 4):
        t = timeit(lambda: run_chunks(chunk, ns))
        print("chunk %4d streams %d  %.3f ms  %.3f M limb-NTT/s" % (chunk, ns, t * 1e3, LIMBS / t / 1e6), flush=True)
