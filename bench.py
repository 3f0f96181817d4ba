#!/usr/bin/env python3
"""bench.py — throughput of the fHE RNS-CKKS hot path on MI355X (contract: see task brief / DESIGN.md §Measurement).

A "step" is one pass of the hot path over one batch of synthetic, HBM-resident input:
  --workload c3  (default; BASELINE.json configs[2], the config the metric is quoted on):
                 N=2^16, Li=L=30 ciphertext multiply + relinearize + rescale, `--batch` ops per step
  --workload ntt batched forward limb-NTT, N=2^16 (north_star headline kernel), `--batch` limbs per step
  --workload c2  N=2^14, 8 limbs x 1024 polys forward+inverse NTT
  --workload c4  N=2^16, Li=K=30 rotation key switch
  --workload c5  N=2^15, Li=16 multiply+relin+rescale
  --workload rotm | rotnn | dec | enc   the SURVEY 8(f) rows: MixedRotKey rotation, power-basis rotation, decrypt and encrypt
                 inner loops (N=2^16, Li=30); their `roofline` is the whole pipeline against the op's algorithmic bytes
Before the W warmup steps the steps run untimed for --prewarm seconds (clock ramp).
One rank per GPU (torch.distributed / RCCL is used only for the barrier and the max-over-ranks of the time:
independent ciphertexts shard with no data-path collective -> "scaling": "weak").
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    #        logN  L   K   Li  default batch
    "c3": (16, 30, 1, 30, 16),
    "c5": (15, 16, 1, 16, 64),
    "c4": (16, 30, 30, 30, 64),
    "ntt": (16, 8, 0, 8, 2048),
    "c2": (14, 8, 0, 8, 8192),
    # SURVEY 8(f) "next" rows, measured to the same bar (whole-pipeline roofline):
    "rotm": (16, 30, 1, 30, 16),    # rank 1: MixedRotKey rotation
    "rotnn": (16, 30, 30, 30, 16),  # rank 2: power-basis rotation (rotate_slots_non_ntt)
    "dec": (16, 30, 1, 30, 64),     # rank 3: decrypt inner loops
    "enc": (16, 30, 1, 30, 64),     # rank 3: encrypt inner loops
}
GENERIC = ("rotm", "rotnn", "dec", "enc")


def gen_primes_host(logN, count):
    """The repo's prime table (same rule as oracle/fhe_oracle.c:orc_gen_primes) without touching the oracle."""
    def is_prime(n):
        if n < 2:
            return False
        small = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)
        for q in small:
            if n % q == 0:
                return n == q
        d, r = n - 1, 0
        while d % 2 == 0:
            d //= 2
            r += 1
        for a in small:
            x = pow(a, d, n)
            if x in (1, n - 1):
                continue
            for _ in range(r - 1):
                x = x * x % n
                if x == n - 1:
                    break
            else:
                return False
        return True

    pr, ps = [], []
    k = (1 << 30) - 1
    while len(pr) < count:
        p = (k << 32) + 1
        k -= 1
        if not is_prime(p):
            continue
        x = 2
        while pow(x, (p - 1) // 2, p) != p - 1:
            x += 1
        pr.append(p)
        ps.append(pow(x, (p - 1) >> (logN + 1), p))
    return np.array(pr, np.uint64), np.array(ps, np.uint64)


def algorithmic_bytes(workload, N, Li, K):
    """SURVEY.md §8(d) per-unit algorithmic bytes (8-byte limbs)."""
    if workload in ("c3", "c5"):
        return 8 * N * (2 * Li * Li + 6 * Li - 2)
    if workload in ("c4", "rotnn"):
        return 8 * N * (6 * Li + 2 * K)
    if workload == "rotm":  # 2*Li in, 2*Li out, 2*Li*(Li+1) key limbs
        return 8 * N * (4 * Li + 2 * Li * (Li + 1))
    if workload == "dec":   # ct (2*Li) + sx (Li) in, Li out
        return 8 * N * 4 * Li
    if workload == "enc":   # u, e0+m, e1 (3*Li) + pk (2*Li) in, 2*Li out
        return 8 * N * 7 * Li
    return 16 * N  # one limb-NTT


def cpu_baseline(workload, logN, L, K, Li, inputs, budget_s=20.0):
    """Times the CPU oracle (kind "port": restatement of the reference's loop structure) on rank 0's host cores,
    single thread like the reference, on a bounded sample of the same workload."""
    from oracle import oracle as orc
    pr, ps = inputs["primes"], inputs["psi"]
    o = orc.Oracle(logN, L, K, pr, ps)
    N = 1 << logN
    t_used, units = 0.0, 0
    if workload in ("c3", "c5"):
        c0, c1, beta, alpha = inputs["c0"], inputs["c1"], inputs["beta"], inputs["alpha"]
        while t_used < budget_s and units < c0.shape[0]:
            t0 = time.perf_counter()
            o.mul_relin_rescale(c0[units, 0], c0[units, 1], c1[units, 0], c1[units, 1], beta, alpha)
            t_used += time.perf_counter() - t0
            units += 1
        unit = "ct-mult+relin+rescale/s"
        sample = "%d op(s) of the bench batch, N=2^%d Li=%d" % (units, logN, Li)
    elif workload in GENERIC:
        one_unit = inputs["cpu_one"](o)
        while t_used < budget_s and units < 4:
            t0 = time.perf_counter()
            one_unit(units)
            t_used += time.perf_counter() - t0
            units += 1
        unit = inputs["cpu_unit"]
        sample = "%d op(s) of the bench batch, N=2^%d Li=%d" % (units, logN, Li)
    elif workload == "c4":
        ct, rkb, rka, gal = inputs["ct"], inputs["rkb"], inputs["rka"], inputs["galois"]
        while t_used < budget_s and units < ct.shape[0]:
            t0 = time.perf_counter()
            o.rotate(ct[units, 0], ct[units, 1], gal, rkb, rka)
            t_used += time.perf_counter() - t0
            units += 1
        unit = "rotation-keyswitch/s"
        sample = "%d op(s) of the bench batch, N=2^%d Li=K=%d" % (units, logN, Li)
    else:
        x, idx = inputs["x"], inputs["idx"]
        n = min(x.shape[0], 256)
        t0 = time.perf_counter()
        o.ntt_forward(x[:n].copy(), idx[:n])
        t_used = time.perf_counter() - t0
        units = n
        unit = "limb-NTT/s"
        sample = "%d forward limb-NTTs of the bench batch, N=2^%d" % (n, logN)
    res = {"value": units / t_used, "unit": unit, "cores": 1, "kind": "port", "sample": sample}
    # second figure: one independent unit per host thread (the C checker releases the GIL), bounded like the first
    try:
        from concurrent.futures import ThreadPoolExecutor
        T = max(1, min(16, os.cpu_count() or 1))
        per_unit = t_used / max(units, 1)
        if workload in ("c3", "c5"):
            one = lambda i: o.mul_relin_rescale(c0[0, 0], c0[0, 1], c1[0, 0], c1[0, 1], beta, alpha)  # noqa: E731
            reps = max(1, int(min(budget_s / 2, 10.0) / per_unit))
        elif workload in GENERIC:
            one = one_unit
            reps = max(1, int(min(budget_s / 2, 10.0) / per_unit))
        elif workload == "c4":
            one = lambda i: o.rotate(ct[0, 0], ct[0, 1], gal, rkb, rka)  # noqa: E731
            reps = max(1, int(min(budget_s / 2, 10.0) / per_unit))
        else:
            nn = min(x.shape[0], 64)
            one = lambda i: o.ntt_forward(x[:nn].copy(), idx[:nn])  # noqa: E731
            per_unit = per_unit * nn
            reps = max(1, int(5.0 / per_unit))
        def worker(i):
            for _ in range(reps):
                one(i)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(T) as ex:
            list(ex.map(worker, range(T)))
        dt = time.perf_counter() - t0
        per = 1 if workload in ("c3", "c4", "c5") + GENERIC else nn
        res["all_cores"] = {"value": T * reps * per / dt, "unit": unit, "cores": T,
                            "sample": "%d thread(s) x %d repetition(s) of one unit each" % (T, reps)}
    except Exception as e:  # the single-thread figure is the contract; this one is extra
        res["all_cores"] = {"error": str(e)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="units per step per GPU (0 = workload default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--prewarm", type=float, default=0.5,
                    help="seconds of untimed steps before the W warmup steps: the GPU needs ~0.3 s of load to reach its "
                         "sustained clocks (a 5 ms NTT run measures 1.96 M limb-NTT/s, a 0.4 s one 2.40 M)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import fhe_amd
    logN, L, K, Li, dflt_batch = WORKLOADS[args.workload]
    B = args.batch or dflt_batch
    N = 1 << logN
    primes, psi = gen_primes_host(logN, L + K)
    ctx = fhe_amd.HipContext(logN, L, K, primes, psi, device=local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    # synthetic inputs, resident in HBM: uniform residues below 2^61 (< every prime of the table)
    gen = torch.Generator(device="cuda").manual_seed(0x5EED0000 + rank)

    def rnd(*shape):
        return torch.randint(0, 1 << 61, shape, dtype=torch.int64, device="cuda", generator=gen)

    inputs = {"primes": primes, "psi": psi}
    if args.workload in ("c3", "c5"):
        c0, c1 = rnd(B, 2, Li, N), rnd(B, 2, Li, N)
        beta, alpha = rnd(L, L, N), rnd(L, L, N)
        out = torch.empty((B, 2, Li - 1, N), dtype=torch.int64, device="cuda")
        ctx.load_mult_key(beta, alpha)
        step = lambda: ctx.mul_relin_rescale(out, c0, c1, Li, batch=B)  # noqa: E731
        metric, unit = "ct-mult+relin+rescale/sec at N=2^%d, Li=%d" % (logN, Li), "ops/s"
    elif args.workload == "c4":
        ct = rnd(B, 2, Li, N)
        rkb, rka = rnd(L + K, N), rnd(L + K, N)
        galois = 3
        ctx.load_rot_key(galois, rkb, rka)
        work = ct.clone()
        step = lambda: ctx.rotate(galois, work, Li, batch=B)  # noqa: E731
        metric, unit = "rotation-keyswitch/sec at N=2^%d, Li=K=%d" % (logN, Li), "ops/s"
    elif args.workload in GENERIC:
        host = lambda t: t.cpu().numpy().view(np.uint64)  # noqa: E731
        if args.workload in ("rotm", "rotnn"):
            ct = rnd(B, 2, Li, N)
            work = ct.clone()
            galois = 3
            if args.workload == "rotm":
                kb, ka = rnd(L, L + 1, N), rnd(L, L + 1, N)
                ctx.load_mixed_rot_key(galois, kb, ka)
                step = lambda: ctx.rotate_mixed(galois, work, Li, batch=B)  # noqa: E731
                metric = "MixedRotKey rotation-keyswitch/sec at N=2^%d, Li=%d" % (logN, Li)
                inputs["cpu_one"] = lambda o: (lambda i, c=host(ct[:1]), b=host(kb), a=host(ka): o.rotate_mixed(c[0, 0], c[0, 1], galois, b, a))
            else:
                kb, ka = rnd(L + K, N), rnd(L + K, N)
                ctx.load_rot_key(galois, kb, ka)
                step = lambda: ctx.rotate_non_ntt(galois, work, Li, batch=B)  # noqa: E731
                metric = "power-basis rotation-keyswitch/sec at N=2^%d, Li=K=%d" % (logN, Li)
                inputs["cpu_one"] = lambda o: (lambda i, c=host(ct[:1]), b=host(kb), a=host(ka): o.rotate_non_ntt(c[0, 0], c[0, 1], galois, b, a))
            unit = "ops/s"
        elif args.workload == "dec":
            ct, sx = rnd(B, 2, Li, N), rnd(L, N)
            out = torch.empty((B, Li, N), dtype=torch.int64, device="cuda")
            step = lambda: ctx.decrypt(out, ct, sx, Li, batch=B)  # noqa: E731
            metric, unit = "decrypt (bx + sx*ax, INTT)/sec at N=2^%d, Li=%d" % (logN, Li), "ops/s"
            inputs["cpu_one"] = lambda o: (lambda i, c=host(ct[:1]), s_=host(sx): o.decrypt(c[0, 0], c[0, 1], s_))
        else:
            u, e0m, e1 = rnd(B, Li, N), rnd(B, Li, N), rnd(B, Li, N)
            pkb, pka = rnd(L, N), rnd(L, N)
            out = torch.empty((B, 2, Li, N), dtype=torch.int64, device="cuda")
            step = lambda: ctx.encrypt(out, u, e0m, e1, pkb, pka, Li, batch=B)  # noqa: E731
            metric, unit = "encrypt (NTT, u*pk + e)/sec at N=2^%d, Li=%d" % (logN, Li), "ops/s"
            inputs["cpu_one"] = lambda o: (lambda i, a_=host(u[:1]), b_=host(e0m[:1]), c_=host(e1[:1]), p_=host(pkb), q_=host(pka):
                                           o.encrypt_core(a_[0], b_[0], c_[0], p_, q_))
        inputs["cpu_unit"] = unit
    else:
        x = rnd(B, N)
        idx = np.tile(np.arange(L, dtype=np.uint32), (B + L - 1) // L)[:B]
        if args.workload == "c2":
            def step():
                ctx.ntt_forward(x, idx)
                ctx.ntt_backward(x, idx)
            per_step_units = 2 * B
        else:
            step = lambda: ctx.ntt_forward(x, idx)  # noqa: E731
            per_step_units = B
        metric, unit = "limb-NTTs/sec at N=2^%d" % logN, "limb-NTT/s"
    units_per_step = B if args.workload in ("c3", "c4", "c5") + GENERIC else per_step_units

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t_end = time.perf_counter() + max(0.0, args.prewarm)
    while time.perf_counter() < t_end:  # untimed: clock ramp (see --prewarm)
        for _ in range(4):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    ctx.profile_enable(not os.environ.get("FHE_BENCH_NOPROF"))  # development switch: cost of the per-launch events
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    value = units_per_step * args.steps * world / elapsed
    alg = algorithmic_bytes(args.workload, N, Li, K)

    # roofline of the dominant kernel: algorithmic bytes per launch / average launch duration, durations from HIP
    # events recorded on the launch stream inside the timed region (fhe_hip_profile_*).
    #  - c3 / c5: relin_rows_mac_kernel (row pass of the Li^2 digit NTTs fused with the key MAC): per ciphertext it
    #    reads 2*Li^2 key limbs + the Li limbs of ct[2] + the 2*Li addend limbs and writes 2*Li limbs.
    #  - ntt / c2: the forward cols+rows kernel pair = one limb-NTT = 16*N bytes.
    #  - c4: the two base-conversion kernels, Li (or K) limbs in, K (or Li) limbs out per conversion.
    roofline = None
    def kms(name):
        return prof.get(name, (0, 0.0))
    if args.workload in ("c3", "c5"):
        launches, ms = kms("relin_rows_mac")
        kname = "relin_rows_mac_kernel"
        alg_total = units_per_step * args.steps * 8 * N * (2 * Li * Li + 5 * Li)
        # SURVEY 8(d) also asks for the key-amortised figure: the 2*Li^2 key limbs counted once per launch, not per ciphertext
        alg_amortised = args.steps * 8 * N * (2 * Li * Li + units_per_step * (Li * Li + 5 * Li))
    elif args.workload in GENERIC:  # the "next" rows: the whole pipeline of the op against its algorithmic bytes
        launches = args.steps
        ms = sum(v[1] for v in prof.values())
        kname = "whole pipeline: " + " + ".join(sorted(prof))
        alg_total = units_per_step * args.steps * alg
    elif args.workload == "c4":
        launches = kms("bconv_modup")[0] + kms("bconv_moddown")[0]
        ms = kms("bconv_modup")[1] + kms("bconv_moddown")[1]
        kname = "bconv_kernel (mod-up + 2x mod-down)"
        alg_total = units_per_step * args.steps * 8 * N * 3 * (Li + K)
    else:
        launches = kms("ntt_fwd_cols")[0]
        ms = kms("ntt_fwd_cols")[1] + kms("ntt_fwd_rows")[1]
        kname = "ntt_fwd (ntt_cols_kernel + ntt_rows_kernel pair)"
        alg_total = B * args.steps * 16 * N
    if launches and ms > 0:
        achieved = alg_total / (ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        if os.path.exists(tpath) and not args.batch:
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "alg_bytes_per_launch": alg_total / launches, "avg_launch_ms": ms / launches, "launches": launches}
        if args.workload in ("c3", "c5"):
            roofline["key_amortised"] = {"alg_bytes_per_launch": alg_amortised / launches,
                                         "achieved": alg_amortised / (ms * 1e-3) / 1e9,
                                         "frac": alg_amortised / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                         "note": "keys counted once per launch (shared by the batch through L2), column-pass output t1 read once per ciphertext"}

    result = {
        "metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": "%s: N=2^%d Li=%d K=%d, %d units/step/GPU" % (args.workload, logN, Li, K, units_per_step),
                   "parallelism": "independent ciphertexts sharded over %d GPU(s), no data-path collective" % world},
        "roofline_whole_op": {"alg_bytes_per_unit": alg, "achieved_GBs": value / world * alg / 1e9,
                              "frac_of_hbm_peak": value / world * alg / 1e9 / HBM_PEAK_GBS},
        "kernel_ms": {k: {"launches": v[0], "total_ms": round(v[1], 3)} for k, v in sorted(prof.items())},
    }
    if roofline:
        result["roofline"] = roofline

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if args.workload in GENERIC:
            pass  # inputs["cpu_one"] was set with the workload
        elif args.workload in ("c3", "c5"):
            nb = min(B, 2)
            inputs.update(c0=c0[:nb].cpu().numpy().view(np.uint64), c1=c1[:nb].cpu().numpy().view(np.uint64),
                          beta=beta.cpu().numpy().view(np.uint64), alpha=alpha.cpu().numpy().view(np.uint64))
        elif args.workload == "c4":
            nb = min(B, 2)
            inputs.update(ct=ct[:nb].cpu().numpy().view(np.uint64), rkb=rkb.cpu().numpy().view(np.uint64),
                          rka=rka.cpu().numpy().view(np.uint64), galois=galois)
        else:
            inputs.update(x=rnd(min(B, 256), N).cpu().numpy().view(np.uint64), idx=idx)
        result["cpu_baseline"] = cpu_baseline(args.workload, logN, L, K, Li, inputs, args.cpu_budget)

    if rank == 0:
        print(json.dumps(result))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
