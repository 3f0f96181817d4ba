#!/usr/bin/env python3
"""bench.py — throughput of the fHE RNS-CKKS hot path on MI355X (contract: see task brief / DESIGN.md §Measurement).

A "step" is one pass of the hot path over one batch of synthetic, HBM-resident input:
  --workload c3  (default; BASELINE.json configs[2], the config the metric is quoted on):
                 N=2^16, Li=L=30 ciphertext multiply + relinearize + rescale, `--batch` ops per step
  --workload ntt batched forward limb-NTT, N=2^16 (north_star headline kernel), `--batch` limbs per step
  --workload c2  N=2^14, 8 limbs x 1024 polys forward+inverse NTT
  --workload c4  N=2^16, Li=K=30 rotation key switch
  --workload c5  N=2^15, Li=16 multiply+relin+rescale
  --workload rotm | rotnn | dec | enc | encode | decode   the SURVEY 8(f) rows: MixedRotKey rotation, power-basis rotation, decrypt
                 and encrypt inner loops, the Encoder (N=2^16, Li=30); their `roofline` is the whole pipeline against the op's
                 algorithmic bytes
Before the W warmup steps the steps run untimed for --prewarm seconds (clock ramp).  The K timed steps
run with the library's per-launch events OFF; the per-kernel durations (`kernel_ms`, `roofline`) come from a second, untimed
pass of the same steps with the events on.  The default (c3) run also carries `ntt_n16`: 2048 forward limb-NTTs at N=2^16,
so that BASELINE.json's other headline figure (limb-NTTs/sec) is on the driver's line too.

One rank per GPU.  `--gpus N` (N > 1) without a torch.distributed launcher starts N child processes of this script itself,
before anything touches the GPU; under `python -m torch.distributed.run` it reads RANK / LOCAL_RANK / WORLD_SIZE and must
agree with --gpus.  Independent ciphertexts shard with no data-path collective ("scaling": "weak"); RCCL carries the barrier,
the max-over-ranks of the time, the key broadcast and, after the timed region, one operation per rank to rank 0, which
recomputes it on its own GPU and requires byte equality (`rccl_parity`).  `--e2e` adds the end-to-end figure of SURVEY 8(e):
the same operations with inputs and results resident on rank 0 only, through fhe_amd.batching.ShardedEvaluator
(chunked star scatter / gather over RCCL).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SIMD = 256 * 4       # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs
VALU_CHEAP_CYCLES = 2  # plain 32-bit VOP1/VOP2 (measured 2.3-2.7 per wave64: profiles/r02_first/instrate2.txt)
VALU_ISSUE_CYCLES = 4  # cycles one wave64 multiply-add / 64-bit / carry instruction holds a SIMD's issue port

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
    "encode": (16, 30, 1, 30, 64),  # rank 4: Encoder::encode on the device (slots -> Li-limb plaintext polynomial)
    "decode": (16, 30, 1, 30, 64),  # rank 4: Encoder::decode on the device
}
GENERIC = ("rotm", "rotnn", "dec", "enc", "encode", "decode")
MULT = ("c3", "c5")


def _is_prime_host(n):
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


def gen_primes_host(logN, count):
    """The repo's prime table (same rule as oracle/fhe_oracle.c:orc_gen_primes) without touching the oracle."""
    is_prime = _is_prime_host
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


def gen_generic_primes_host(logN, count):
    """`--primes generic`: descending 62-bit primes with 2N | p - 1 that are NOT of the k*2^32 + 1 form, i.e. a table like the
    one a drop-in caller passes (yell's params::P); the library then runs its generic-prime kernel instantiations."""
    base = _is_prime_host
    pr, ps = [], []
    step = 1 << (logN + 1)
    p = (1 << 62) - step + 1
    while len(pr) < count:
        if p % (1 << 32) != 1 and base(p):
            x = 2
            while pow(x, (p - 1) // 2, p) != p - 1:
                x += 1
            pr.append(p)
            ps.append(pow(x, (p - 1) >> (logN + 1), p))
        p -= step
    return np.array(pr, np.uint64), np.array(ps, np.uint64)


def algorithmic_bytes(workload, N, Li, K):
    """SURVEY.md §8(d) per-unit algorithmic bytes (8-byte limbs)."""
    if workload in MULT:
        return 8 * N * (2 * Li * Li + 6 * Li - 2)
    if workload in ("c4", "rotnn"):
        return 8 * N * (6 * Li + 2 * K)
    if workload == "rotm":  # 2*Li in, 2*Li out, 2*Li*(Li+1) key limbs
        return 8 * N * (4 * Li + 2 * Li * (Li + 1))
    if workload == "dec":   # ct (2*Li) + sx (Li) in, Li out
        return 8 * N * 4 * Li
    if workload in ("encode", "decode"):  # N/2 complex slots (16 B each) on one side, Li limbs on the other
        return 8 * N + 8 * N * Li
    if workload == "enc":   # u, e0+m, e1 (3*Li) + pk (2*Li) in, 2*Li out
        return 8 * N * 7 * Li
    return 16 * N  # one limb-NTT


def dominant_kernel_bytes(workload, N, Li, K, units_per_launch):
    """Algorithmic HBM bytes ONE launch of the workload's dominant kernel has to move, and the same
    figure with the key limbs charged to every ciphertext of the batch (`per_op_keys`, SURVEY 8(d)'s per-op convention).
    c3 / c5, relin_rows8p_mac_kernel: the 2*Li^2 key limbs ONCE per launch (the ciphertexts of a batch share them through
    L2, and the counters confirm they are fetched once), per ciphertext the Li^2 digit limbs (column-pass output + diagonal),
    2*Li addend limbs in and 2*Li result limbs out."""
    if workload in MULT:
        shared = 8 * N * (2 * Li * Li + units_per_launch * (Li * Li + 4 * Li))
        per_op = 8 * N * units_per_launch * (2 * Li * Li + Li * Li + 4 * Li)
        return shared, per_op
    if workload == "c4":
        # ntt_rows8_fwd_kernel<SP, 4>, the rotation's last launch (both halves): per ciphertext and normal limb the two column-pass
        # limbs in, galois(ax) and galois(bx) of the input, two result limbs out; the 2*Li key limbs once per launch
        b = 8 * N * Li * (6 * units_per_launch + 2)
        return b, b
    b = units_per_launch * 16 * N  # ntt / c2: the cols + rows pair is one limb-NTT = read N + write N words
    return b, b


def cpu_baseline(workload, logN, L, K, Li, inputs, budget_s=20.0):
    """Times the CPU oracle (kind "port": restatement of the reference's loop structure) on rank 0's host cores,
    single thread like the reference, on a bounded sample of the same workload."""
    from oracle import oracle as orc
    pr, ps = inputs["primes"], inputs["psi"]
    o = orc.Oracle(logN, L, K, pr, ps)
    t_used, units = 0.0, 0
    if workload in MULT:
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
    if workload in ("encode", "decode"):
        res["all_cores"] = {"note": "not measured: the encoder's checker is numpy/Python (oracle/encoder.py) and holds the interpreter lock"}
        return res
    try:
        from concurrent.futures import ThreadPoolExecutor
        T = max(1, min(16, os.cpu_count() or 1))
        per_unit = t_used / max(units, 1)
        if workload in MULT:
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


def verify_over_collective(sb, compute, c0_one, c1_one, out_one, in_shape, out_shape):
    """After the timed region of a multi-rank run: every rank contributes ONE operation (its inputs and the result its GPU
    produced) to rank 0 through ShardedBatch.gather (RCCL on GPUs); rank 0 recomputes all of them on its own device and
    requires byte equality.  Returns (parity, ranks_seen) on rank 0, (None, None) elsewhere.  compute(c0, c1, n) -> out."""
    import torch
    n = sb.world
    g0 = sb.gather(c0_one, n, in_shape)
    g1 = sb.gather(c1_one, n, in_shape)
    go = sb.gather(out_one, n, out_shape)
    if sb.rank != sb.root:
        return None, None
    want = compute(g0, g1, n)
    rows_ok = [bool(torch.equal(want[r], go[r])) for r in range(n)]
    # a rank whose transfer never arrived leaves its slot of the gather buffer unequal to the recomputation
    return all(rows_ok), sum(rows_ok)


def launch_ranks(n):
    """`--gpus N` without a launcher: N fresh children of this script, one per GPU, started before this process has made any
    GPU call (it makes none afterwards either); rank 0's stdout is this process's."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [p.wait() for p in procs]
    return next((rc for rc in rcs if rc), 0)


KERNEL_SOURCES = ("modarith.h", "ntt_kernels.h", "pointwise_kernels.h", "bconv_mfma_kernels.h", "fhe_hip.hip")


def kernel_source_sha16():
    """Identity of the kernel sources of the library being timed (the committed counter files carry the hash of the sources
    they were measured on: tools/stamp_profiles.py)."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "fhe_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def read_profile_json(kind, workload):
    """profiles/<kind>_<workload>.json, or None when absent or measured on other kernel sources than the ones being timed."""
    path = os.path.join(ROOT, "profiles", "%s_%s.json" % (kind, workload))
    try:
        j = json.load(open(path))
    except Exception:
        return None
    return j if j.get("kernel_source_sha16") == kernel_source_sha16() else None


class Setup:
    """One workload on one context: synthetic HBM-resident inputs and the step() that runs the hot path over them."""

    def __init__(self, workload, B, local_rank, rank, sb=None, relin_group=0, ntt_chunk=0, onepass=1, primes="sp", ntt_streams=2):
        import torch
        import fhe_amd
        self.workload = workload
        self.logN, self.L, self.K, self.Li, dflt = WORKLOADS[workload]
        self.B = B or dflt
        logN, L, K, Li, B = self.logN, self.L, self.K, self.Li, self.B
        self.N = N = 1 << logN
        self.primes_kind = primes
        self.primes, self.psi = (gen_generic_primes_host if primes == "generic" else gen_primes_host)(logN, L + K)
        self.ctx = ctx = fhe_amd.HipContext(logN, L, K, self.primes, self.psi, device=local_rank)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        if relin_group:
            ctx.set_tuning(fhe_amd.hip.TUNE_RELIN_GROUP, relin_group)
        if ntt_chunk:
            ctx.set_tuning(fhe_amd.hip.TUNE_NTT_CHUNK, ntt_chunk)
        if not onepass:
            ctx.set_tuning(fhe_amd.hip.TUNE_ONEPASS, 0)
        if os.environ.get("FHE_BENCH_BCONV"):          # A/B switch of the base-conversion kernel (tools only)
            ctx.set_tuning(fhe_amd.hip.TUNE_BCONV, int(os.environ["FHE_BENCH_BCONV"]))
        if os.environ.get("FHE_BENCH_RELIN_KERNEL"):   # A/B switch of the digit key-switch kernel (tools only)
            ctx.set_tuning(fhe_amd.hip.TUNE_RELIN_KERNEL, int(os.environ["FHE_BENCH_RELIN_KERNEL"]))
        if ntt_streams != 2:
            ctx.set_tuning(fhe_amd.hip.TUNE_NTT_STREAMS, ntt_streams)
        gen = torch.Generator(device="cuda").manual_seed(0x5EED0000 + rank)

        def rnd(*shape):  # uniform residues below 2^61 (< every prime of the table)
            return torch.randint(0, 1 << 61, shape, dtype=torch.int64, device="cuda", generator=gen)

        def key(*shape):  # key material is replicated: rank 0 draws it, RCCL broadcasts it once
            t = rnd(*shape)
            if sb is not None:
                sb.broadcast(t)
            return t
        self.rnd = rnd
        self.inputs = {"primes": self.primes, "psi": self.psi}
        self.per_step_units = B
        host = lambda t: t.cpu().numpy().view(np.uint64)  # noqa: E731
        if workload in MULT:
            self.c0, self.c1 = rnd(B, 2, Li, N), rnd(B, 2, Li, N)
            self.beta, self.alpha = key(L, L, N), key(L, L, N)
            self.out = torch.empty((B, 2, Li - 1, N), dtype=torch.int64, device="cuda")
            ctx.load_mult_key(self.beta, self.alpha)
            self.step = lambda: ctx.mul_relin_rescale(self.out, self.c0, self.c1, Li, batch=B)
            self.metric, self.unit = "ct-mult+relin+rescale/sec at N=2^%d, Li=%d" % (logN, Li), "ops/s"
        elif workload == "c4":
            self.ct = rnd(B, 2, Li, N)
            self.rkb, self.rka = key(L + K, N), key(L + K, N)
            self.galois = 3
            ctx.load_rot_key(self.galois, self.rkb, self.rka)
            self.work = torch.empty_like(self.ct)
            self.step = lambda: ctx.rotate_to(self.galois, self.work, self.ct, Li, batch=B)  # out of place: B ciphertexts in, B out
            self.metric, self.unit = "rotation-keyswitch/sec at N=2^%d, Li=K=%d" % (logN, Li), "ops/s"
        elif workload in GENERIC:
            galois = 3
            if workload in ("rotm", "rotnn"):
                ct = rnd(B, 2, Li, N)
                work = ct.clone()
                if workload == "rotm":
                    kb, ka = key(L, L + 1, N), key(L, L + 1, N)
                    ctx.load_mixed_rot_key(galois, kb, ka)
                    self.step = lambda: ctx.rotate_mixed(galois, work, Li, batch=B)
                    self.metric = "MixedRotKey rotation-keyswitch/sec at N=2^%d, Li=%d" % (logN, Li)
                    self.inputs["cpu_one"] = lambda o: (lambda i, c=host(ct[:1]), b=host(kb), a=host(ka): o.rotate_mixed(c[0, 0], c[0, 1], galois, b, a))
                else:
                    kb, ka = key(L + K, N), key(L + K, N)
                    ctx.load_rot_key(galois, kb, ka)
                    self.step = lambda: ctx.rotate_non_ntt(galois, work, Li, batch=B)
                    self.metric = "power-basis rotation-keyswitch/sec at N=2^%d, Li=K=%d" % (logN, Li)
                    self.inputs["cpu_one"] = lambda o: (lambda i, c=host(ct[:1]), b=host(kb), a=host(ka): o.rotate_non_ntt(c[0, 0], c[0, 1], galois, b, a))
            elif workload in ("encode", "decode"):
                scale = 2.0 ** 50
                vals = (torch.rand((B, N // 2), dtype=torch.float64, device="cuda", generator=gen) * 8 - 4).to(torch.complex128)
                vals = vals + 1j * (torch.rand((B, N // 2), dtype=torch.float64, device="cuda", generator=gen) * 8 - 4)
                msg = torch.zeros((B, Li, N), dtype=torch.int64, device="cuda")
                ctx.encode(msg, vals, Li, scale, batch=B)
                pr_li = self.primes[:Li]

                def enc_oracle(o):  # the numpy/Python restatement of the reference's Encoder (oracle/encoder.py)
                    from oracle.encoder import EncoderOracle
                    e, v0, m0 = EncoderOracle(logN), vals[0].cpu().numpy(), host(msg[:1])[0]
                    return (lambda i: e.encode(v0, scale, pr_li)) if workload == "encode" else (lambda i: e.decode(m0, pr_li, scale))
                if workload == "encode":
                    self.step = lambda: ctx.encode(msg, vals, Li, scale, batch=B)
                    self.metric = "Encoder::encode/sec at N=2^%d, %d limbs (complex double FFT + rounding, on the device)" % (logN, Li)
                else:
                    slots = torch.zeros((B, N // 2), dtype=torch.complex128, device="cuda")
                    self.step = lambda: ctx.decode(slots, msg, Li, scale, batch=B)
                    self.metric = "Encoder::decode/sec at N=2^%d, %d limbs (CRT lift + complex double FFT, on the device)" % (logN, Li)
                self.inputs["cpu_one"] = enc_oracle
            elif workload == "dec":
                ct, sx = rnd(B, 2, Li, N), key(L, N)
                out = torch.empty((B, Li, N), dtype=torch.int64, device="cuda")
                self.step = lambda: ctx.decrypt(out, ct, sx, Li, batch=B)
                self.metric = "decrypt (bx + sx*ax, INTT)/sec at N=2^%d, Li=%d" % (logN, Li)
                self.inputs["cpu_one"] = lambda o: (lambda i, c=host(ct[:1]), s_=host(sx): o.decrypt(c[0, 0], c[0, 1], s_))
            else:
                u, e0m, e1 = rnd(B, Li, N), rnd(B, Li, N), rnd(B, Li, N)
                pkb, pka = key(L, N), key(L, N)
                out = torch.empty((B, 2, Li, N), dtype=torch.int64, device="cuda")
                self.step = lambda: ctx.encrypt(out, u, e0m, e1, pkb, pka, Li, batch=B)
                self.metric = "encrypt (NTT, u*pk + e)/sec at N=2^%d, Li=%d" % (logN, Li)
                self.inputs["cpu_one"] = lambda o: (lambda i, a_=host(u[:1]), b_=host(e0m[:1]), c_=host(e1[:1]), p_=host(pkb), q_=host(pka):
                                                    o.encrypt_core(a_[0], b_[0], c_[0], p_, q_))
            self.unit = "ops/s"
            self.inputs["cpu_unit"] = self.unit
        else:
            self.x = rnd(B, N)
            self.idx = np.tile(np.arange(L, dtype=np.uint32), (B + L - 1) // L)[:B]
            if workload == "c2":
                def step():
                    ctx.ntt_forward(self.x, self.idx)
                    ctx.ntt_backward(self.x, self.idx)
                self.step = step
                self.per_step_units = 2 * B
            else:
                self.step = lambda: ctx.ntt_forward(self.x, self.idx)
            self.metric, self.unit = "limb-NTTs/sec at N=2^%d" % logN, "limb-NTT/s"

    def timed(self, steps, warmup, prewarm, barrier):
        """(elapsed seconds of `steps` steps with the library's events off, per-kernel profile of a second untimed pass)."""
        import torch
        ctx = self.ctx
        ctx.profile_enable(False)
        t_end = time.perf_counter() + max(0.0, prewarm)
        while time.perf_counter() < t_end:  # untimed: clock ramp (see --prewarm)
            for _ in range(4):
                self.step()
            torch.cuda.synchronize()
        for _ in range(warmup):
            self.step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        barrier()
        elapsed = time.perf_counter() - t0
        # per-kernel durations: same steps again, untimed, with HIP events recorded on the launch stream around every launch.
        # Two such passes; the one whose kernel times add up closest to the timed region's step time is reported (a pass that
        # catches a clock excursion would otherwise put a duration on the roofline line that the timed region never saw).
        ctx.profile_enable(True)
        best = None
        for _ in range(2):
            ctx.profile_reset()
            for _ in range(steps):
                self.step()
            torch.cuda.synchronize()
            prof = ctx.profile_get()
            gap = abs(sum(v[1] for v in prof.values()) * 1e-3 - elapsed)
            if best is None or gap < best[0]:
                best = (gap, prof)
        ctx.profile_enable(False)
        return elapsed, best[1]

    def roofline(self, prof, steps):
        """`roofline` (HBM) and `roofline_valu` blocks for the workload's dominant kernel."""
        w, N, Li, K, B = self.workload, self.N, self.Li, self.K, self.B
        kms = lambda name: prof.get(name, (0, 0.0))  # noqa: E731
        if w in MULT:
            launches, ms = kms("relin_rows_mac")
            kname = "relin_rows4_mac_kernel" if os.environ.get("FHE_BENCH_RELIN_KERNEL", "1") != "0" else "relin_rows8p_mac_kernel"
            units_per_launch = B
        elif w in GENERIC:  # the "next" rows: the whole pipeline of the op against its algorithmic bytes
            launches, ms = steps, sum(v[1] for v in prof.values())
            kname = "whole pipeline: " + " + ".join(sorted(prof))
            units_per_launch = B
        elif w == "c4":
            # longest launch of the step since the base conversions moved to the matrix cores: the last forward row pass with the
            # normal limbs' key product, the mod-down tail and + galois(bx) fused (both halves in one launch)
            launches, ms = kms("ntt_fwd_rows_rotdown")
            kname = "ntt_rows8_fwd_kernel<SP, 4> (row pass + key product of the normal limbs + mod-down tail, both halves in one launch)"
            units_per_launch = B
        elif kms("ntt_fwd_overlapped")[0]:
            # N >= 2^15: the library issues the call's chunks on two internal streams (column pass of one chunk under the row pass of
            # another), so per-kernel durations overlap; the unit timed is the whole call, first launch -> last completion
            launches, ms = kms("ntt_fwd_overlapped")
            kname = "ntt_fwd: ntt_cols_kernel + ntt_rows8_fwd_kernel over the whole call (128-limb chunks overlapped on two streams; first launch -> last completion)"
            units_per_launch = B
        elif kms("ntt_fwd_onepass")[0]:  # N <= 2^14: the limb stays in one workgroup's LDS, one kernel per transform
            launches, ms = kms("ntt_fwd_onepass")
            kname = "ntt_onepass_fwd_kernel"
            units_per_launch = B
        else:
            launches = kms("ntt_fwd_cols")[0]
            ms = kms("ntt_fwd_cols")[1] + kms("ntt_fwd_rows")[1]
            kname = "ntt_fwd (ntt_cols_kernel + ntt_rows8_fwd_kernel pair)"
            units_per_launch = B
        if not launches or ms <= 0:
            return None, None
        avg_ms = ms / launches
        if w in GENERIC:
            alg = per_op = units_per_launch * algorithmic_bytes(w, N, Li, K)
        else:
            alg, per_op = dominant_kernel_bytes(w, N, Li, K, units_per_launch)
        achieved = alg / (avg_ms * 1e-3) / 1e9
        default_batch = self.B == WORKLOADS[w][4] and self.primes_kind == "sp"  # the committed counter passes ran this shape on the SP table
        tj = read_profile_json("traffic", w) if default_batch else None
        traffic = tj.get("hbm_bytes_per_launch") if tj else None
        bytes_key = "min_bytes_of_this_design_per_launch" if w in MULT else "alg_bytes_per_launch"
        roof = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                bytes_key: alg, "avg_launch_ms": avg_ms, "launches": launches}
        if w == "c4" and kms("bconv_moddown")[0]:
            # the mod-down base conversion (bconv_mfma_kernel when the shape allows): 2 x (K special limbs in, Li normal limbs out) per ciphertext
            bl, bms = kms("bconv_moddown")
            bb = B * 2 * 8 * N * (Li + K)
            roof["base_conversion"] = {"kernel": "bconv_moddown (both halves in one launch)", "alg_bytes_per_launch": bb, "avg_launch_ms": bms / bl,
                                       "achieved": bb / (bms / bl * 1e-3) / 1e9, "frac": bb / (bms / bl * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if w in MULT:
            roof["bytes_note"] = ("bytes this launch of the two-kernel relinearisation has to move once: 2*Li^2 key limbs, per ciphertext the Li^2 "
                                  "column-pass limbs (an intermediate of this design, not algorithmic bytes of the operation), 2*Li addends, 2*Li results")
        if default_batch and not tj:
            roof["traffic_note"] = "null: profiles/traffic_%s.json is absent or was measured on other kernel sources (kernel_source_sha16)" % w
        if traffic:
            roof["traffic_frac"] = traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if tj and tj.get("hbm_bytes_per_step"):
            # what the counters saw over ALL kernels of a step against the operation's algorithmic bytes (keys amortised
            # over the batch): the traffic the design adds (column-pass output written and re-read, intermediate ciphertexts)
            alg_step = B * 8 * N * (6 * Li - 2 + 2 * Li * Li / B) if w in MULT else self.per_step_units * algorithmic_bytes(w, N, Li, K)
            roof["traffic_per_step"] = tj["hbm_bytes_per_step"]
            roof["alg_bytes_per_step_key_amortised" if w in MULT else "alg_bytes_per_step"] = alg_step
            roof["traffic_over_algorithmic"] = tj["hbm_bytes_per_step"] / alg_step
        if per_op != alg:
            roof["per_op_keys"] = {"alg_bytes_per_launch": per_op, "achieved": per_op / (avg_ms * 1e-3) / 1e9,
                                   "frac": per_op / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "note": "SURVEY 8(d) per-op convention: the 2*Li^2 key limbs charged to every ciphertext of the batch; "
                                           "the counters show they are fetched once per launch, so this is not HBM traffic"}
        # VALU issue, entirely from ONE committed PMC pass of the dominant kernel (profiles/valu_<workload>.json: SQ_INSTS_VALU and
        # GRBM_GUI_ACTIVE of the same dispatches; no figure of this run enters): issue cycles the instruction stream needs at the
        # least each class can cost (2 for plain 32-bit VOP1/VOP2, 4 for everything else: profiles/r02_first/instrate2.txt measures
        # 2.3-2.7 and 4.1-4.7) over the SIMD cycles the launch had.  A lower bound on busy cycles over available cycles: <= 1.
        valu = None
        vj = read_profile_json("valu", w) if default_batch else None
        if vj and vj.get("insts_valu_per_launch") and vj.get("gui_active_per_xcd_per_launch") and vj.get("cheap_share") is not None:
            cs = vj["cheap_share"]
            need = vj["insts_valu_per_launch"] * (VALU_ISSUE_CYCLES * (1 - cs) + VALU_CHEAP_CYCLES * cs)
            have = N_SIMD * vj["gui_active_per_xcd_per_launch"]
            valu = {"bound": "valu-issue", "kernel": kname, "insts_valu_per_launch": vj["insts_valu_per_launch"],
                    "cheap_share": cs, "issue_cycles": {"plain_32bit_vop": VALU_CHEAP_CYCLES, "other": VALU_ISSUE_CYCLES}, "simds": N_SIMD,
                    "simd_cycles_available": have, "issue_cycles_needed": need, "frac": need / have,
                    "sclk_mhz_pmc_pass": vj["gui_active_per_xcd_per_launch"] / (vj["launch_ms_in_pmc_pass"] * 1e3) if vj.get("launch_ms_in_pmc_pass") else None,
                    "note": "SQ_INSTS_VALU x class-weighted minimum issue cycles / (1024 SIMDs x GRBM_GUI_ACTIVE per XCD), all from the committed "
                            "PMC pass of this kernel on these kernel sources", "source": vj.get("source")}
            if valu["frac"] > roof["frac"]:
                roof["bound"] = "valu-issue (see roofline_valu): the kernel sits nearer its VALU issue limit than the HBM roofline"
        return roof, valu


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
    ap.add_argument("--no-ntt-n16", action="store_true", help="skip the ntt_n16 sub-record of the default (c3) run")
    ap.add_argument("--relin-group", type=int, default=0, help="FHE_HIP_TUNE_RELIN_GROUP (0 = library default)")
    ap.add_argument("--ntt-chunk", type=int, default=0, help="FHE_HIP_TUNE_NTT_CHUNK (0 = library default)")
    ap.add_argument("--ntt-streams", type=int, default=2, help="FHE_HIP_TUNE_NTT_STREAMS: 1 = stand-alone NTTs at N >= 2^15 on one stream (A/B)")
    ap.add_argument("--onepass", type=int, default=1, help="FHE_HIP_TUNE_ONEPASS: 0 = column/row pass pair also for N <= 2^14")
    ap.add_argument("--primes", default="sp", choices=("sp", "generic"),
                    help="sp: the repo's k*2^32+1 table (multiplication-free Montgomery quotient); generic: 62-bit primes of no special "
                         "form, the kernels a caller's own table (yell's params::P) runs")
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="run the ranks of --gpus N on the GPUs that exist (rank %% device count) over gloo: exercises the multi-rank "
                         "logic (barriers, key broadcast, parity check, --e2e) on a one-GPU box; the line is marked and is not a measurement")
    ap.add_argument("--e2e", action="store_true",
                    help="c3 / c5 only: also time the same operations end to end with inputs and results resident on rank 0 only "
                         "(ShardedEvaluator: chunked scatter / compute / gather), reported beside the compute-only value")
    ap.add_argument("--e2e-chunk", type=int, default=8, help="ops per rank per scatter/compute/gather chunk in --e2e")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))  # this process never touches the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d disagrees with WORLD_SIZE=%d of the launcher" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.rehearse_shared_gpu:  # the multi-rank logic on fewer GPUs than ranks (gloo, host-staged transfers): NOT a measurement
        local_rank = local_rank % torch.cuda.device_count()
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d has no GPU: %d device(s) visible, --gpus %d" % (rank, torch.cuda.device_count(), args.gpus))
    torch.cuda.set_device(local_rank)
    sb = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        from fhe_amd.batching import ShardedBatch
        sb = ShardedBatch(root=0, device=torch.device("cuda", local_rank))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if args.rehearse_shared_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    su = Setup(args.workload, args.batch, local_rank, rank, sb, args.relin_group, args.ntt_chunk, args.onepass, args.primes, args.ntt_streams)
    logN, L, K, Li, N, B = su.logN, su.L, su.K, su.Li, su.N, su.B
    elapsed, prof = su.timed(args.steps, args.warmup, args.prewarm, barrier)
    elapsed = max_over_ranks(elapsed)
    value = su.per_step_units * args.steps * world / elapsed
    alg = algorithmic_bytes(args.workload, N, Li, K)
    roofline, roofline_valu = su.roofline(prof, args.steps)

    result = {
        "metric": su.metric, "value": value, "unit": su.unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64 (complex) + u64" if args.workload in ("encode", "decode") else "u64", "data": "synthetic",
        "config": {"workload": "%s: N=2^%d Li=%d K=%d, %d units/step/GPU" % (args.workload, logN, Li, K, su.per_step_units),
                   "primes": "k*2^32+1 table (SP kernels)" if args.primes == "sp" else "generic 62-bit primes (generic-prime kernels)",
                   "parallelism": "independent ciphertexts sharded over %d GPU(s), no data-path collective" % world},
        "roofline_whole_op": {"alg_bytes_per_unit": alg, "achieved_GBs": value / world * alg / 1e9,
                              "frac_of_hbm_peak": value / world * alg / 1e9 / HBM_PEAK_GBS},
        "kernel_ms": {k: {"launches": v[0], "total_ms": round(v[1], 3)} for k, v in sorted(prof.items())},
        "kernel_ms_note": "untimed pass of the same steps with per-launch HIP events (of two passes, the one whose sum is closest to the "
                          "timed region); the timed region runs with them off",
        "kernel_ms_sum_per_step": sum(v[1] for v in prof.values()) / args.steps,
    }
    if args.workload in MULT:  # "also report the key-amortised figure 8*N*(6Li - 2 + 2Li^2/B) for batch B"
        am = 8 * N * (6 * Li - 2 + 2 * Li * Li / B)
        result["roofline_whole_op"]["key_amortised"] = {"alg_bytes_per_unit": am, "achieved_GBs": value / world * am / 1e9,
                                                        "frac_of_hbm_peak": value / world * am / 1e9 / HBM_PEAK_GBS}
    if roofline:
        result["roofline"] = roofline
    if roofline_valu:
        result["roofline_valu"] = roofline_valu

    # ---- multi-rank only: one operation per rank to rank 0 over RCCL, recomputed there ---------------------------------
    if world > 1 and args.workload in MULT:
        def recompute(g0, g1, n):
            o = torch.empty((n, 2, Li - 1, N), dtype=torch.int64, device="cuda")
            su.ctx.mul_relin_rescale(o, g0.contiguous(), g1.contiguous(), Li, batch=n)
            return o
        su.step()
        torch.cuda.synchronize()
        try:
            ok, seen = verify_over_collective(sb, recompute, su.c0[:1], su.c1[:1], su.out[:1], (2, Li, N), (2, Li - 1, N))
            if rank == 0:
                result["rccl_parity"], result["ranks_seen"] = ok, seen
        except Exception as e:  # the measured line must survive a failing check: report it instead of losing the run
            result["rccl_parity"], result["ranks_seen"], result["rccl_parity_error"] = None, None, repr(e)[:300]

    # ---- end to end: data starts and ends on rank 0 ----------------------------------------------------------
    if args.e2e and args.workload in MULT:
        from fhe_amd.batching import ShardedEvaluator
        ev = ShardedEvaluator(su.ctx)
        n_items = B * world
        r0, r1 = (su.rnd(n_items, 2, Li, N), su.rnd(n_items, 2, Li, N)) if rank == 0 else (None, None)
        run = lambda: ev.mul_relin_rescale(r0, r1, n_items, Li, chunk=args.e2e_chunk)  # noqa: E731
        got = run()
        barrier()
        t0 = time.perf_counter()
        for _ in range(max(1, args.steps // 2)):
            got = run()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        e2e = {"value": n_items * max(1, args.steps // 2) / dt, "unit": su.unit, "ops_per_call": n_items, "chunk_per_rank": args.e2e_chunk,
               "compute_only_value": value,
               "note": "inputs and results resident on rank 0's GPU only; chunked star scatter -> compute -> gather (fhe_amd/batching.py)"}
        if rank == 0:
            # bytes through the root per op: 2 inputs out, 1 result back, for the (world-1)/world share that leaves the root
            e2e["root_link_GBs"] = e2e["value"] * (world - 1) / world * 8 * N * (4 * Li + 2 * (Li - 1)) / 1e9
            check = torch.empty((1, 2, Li - 1, N), dtype=torch.int64, device="cuda")
            su.ctx.mul_relin_rescale(check, r0[n_items - 1:].contiguous(), r1[n_items - 1:].contiguous(), Li, batch=1)
            e2e["last_op_matches_local_recompute"] = bool(torch.equal(check[0], got[n_items - 1]))
            result["e2e"] = e2e
        del r0, r1, got

    # bounded host-side sample for the CPU baseline, taken before the inputs are released
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    inputs = su.inputs
    if want_cpu:
        if args.workload in MULT:
            nb = min(B, 2)
            inputs.update(c0=su.c0[:nb].cpu().numpy().view(np.uint64), c1=su.c1[:nb].cpu().numpy().view(np.uint64),
                          beta=su.beta.cpu().numpy().view(np.uint64), alpha=su.alpha.cpu().numpy().view(np.uint64))
        elif args.workload == "c4":
            nb = min(B, 2)
            inputs.update(ct=su.ct[:nb].cpu().numpy().view(np.uint64), rkb=su.rkb.cpu().numpy().view(np.uint64),
                          rka=su.rka.cpu().numpy().view(np.uint64), galois=su.galois)
        elif args.workload not in GENERIC:
            inputs.update(x=su.rnd(min(B, 256), N).cpu().numpy().view(np.uint64), idx=su.idx)

    # ---- the other headline figure on the same line: batched forward limb-NTT at N = 2^16 ---------------------------------
    if args.workload == "c3" and not args.no_ntt_n16 and not args.batch:
        del su.c0, su.c1, su.out
        nt = Setup("ntt", 0, local_rank, rank, primes=args.primes, ntt_streams=args.ntt_streams, ntt_chunk=args.ntt_chunk)
        el, npf = nt.timed(2 * args.steps, args.warmup, args.prewarm, barrier)
        el = max_over_ranks(el)
        nroof, nvalu = nt.roofline(npf, 2 * args.steps)
        result["ntt_n16"] = {"metric": nt.metric, "value": nt.per_step_units * 2 * args.steps * world / el, "unit": nt.unit,
                             "limbs_per_step_per_gpu": nt.B, "steps": 2 * args.steps, "ms_per_step": el / (2 * args.steps) * 1e3,
                             "roofline": nroof, "roofline_valu": nvalu}
        nt.ctx.close()

    if want_cpu:
        result["cpu_baseline"] = cpu_baseline(args.workload, logN, L, K, Li, inputs, args.cpu_budget)

    if args.rehearse_shared_gpu:
        result["rehearsal"] = "ranks share GPUs, gloo backend with host-staged transfers: not a measurement"
    if rank == 0:
        print(json.dumps(result))
    su.ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
