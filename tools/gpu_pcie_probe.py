import time, numpy as np, torch
torch.cuda.init()
rt = torch.cuda.cudart()
n = 1 << 30
a = np.empty(n, np.uint8); a[:] = 1
d = torch.empty(n, dtype=torch.uint8, device="cuda")
t = torch.from_numpy(a)
def tm(f, reps=3):
    torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
print("pageable H2D 1GiB: %.1f GB/s" % (n / tm(lambda: d.copy_(t)) / 1e9))
print("pageable D2H 1GiB: %.1f GB/s" % (n / tm(lambda: t.copy_(d)) / 1e9))
t0 = time.perf_counter(); rc = rt.cudaHostRegister(a.ctypes.data, n, 0); t1 = time.perf_counter()
print("hostRegister rc", rc, "%.3f s = %.1f GB/s" % (t1 - t0, n / (t1 - t0) / 1e9))
print("registered H2D: %.1f GB/s" % (n / tm(lambda: d.copy_(t, non_blocking=True)) / 1e9))
print("registered D2H: %.1f GB/s" % (n / tm(lambda: t.copy_(d, non_blocking=True)) / 1e9))
t0 = time.perf_counter(); rt.cudaHostUnregister(a.ctypes.data); print("unregister %.3f s" % (time.perf_counter() - t0))
p = torch.empty(n, dtype=torch.uint8).pin_memory()
print("pinned H2D: %.1f GB/s" % (n / tm(lambda: d.copy_(p, non_blocking=True)) / 1e9))
print("pinned D2H: %.1f GB/s" % (n / tm(lambda: p.copy_(d, non_blocking=True)) / 1e9))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
d2 = torch.empty(n, dtype=torch.uint8, device="cuda"); p2 = torch.empty(n, dtype=torch.uint8).pin_memory()
def duplex():
    with torch.cuda.stream(s1): d.copy_(p, non_blocking=True)
    with torch.cuda.stream(s2): p2.copy_(d2, non_blocking=True)
print("duplex pinned H2D+D2H: %.1f GB/s each way" % (n / tm(duplex) / 1e9))
t0 = time.perf_counter(); b = a.copy(); print("host memcpy 1 thread: %.1f GB/s" % (n / (time.perf_counter() - t0) / 1e9))
