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


def _run_py(code, timeout=600, extra_env=None):
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    env.update(extra_env or {})
    return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=timeout, cwd=root)


def test_rccl_backend_single_rank_collectives_and_sharded_batch():
    """The "nccl" backend (= RCCL) brought up on the one GPU a test box has: process group of one rank, the collectives
    bench.py and ShardedBatch use (broadcast of key material, all_reduce MAX of the timing, barrier, all_gather) run through
    RCCL on the device, and ShardedEvaluator on that group gives the oracle's bits.  (Peer-to-peer transfers need a second GPU:
    those run under gloo in tests/test_batching_gloo.py and in the shared-GPU rehearsal below.)"""
    import socket, textwrap
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    code = textwrap.dedent('''
        import numpy as np, torch, torch.distributed as dist, fhe_amd
        from fhe_amd.batching import ShardedBatch, ShardedEvaluator
        from oracle import oracle as orc
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        assert dist.get_backend() == "nccl"
        t = torch.arange(1 << 16, dtype=torch.int64, device="cuda")
        dist.broadcast(t, 0); dist.barrier()
        m = torch.tensor([3.5], dtype=torch.float64, device="cuda"); dist.all_reduce(m, op=dist.ReduceOp.MAX)
        out = [torch.empty_like(t)]; dist.all_gather(out, t)
        torch.cuda.synchronize()
        assert float(m.item()) == 3.5 and torch.equal(out[0], t)
        logN, L, K, Li, B = 12, 3, 1, 3, 3
        pr, ps = orc.gen_primes(logN, L + K)
        o, g = orc.Oracle(logN, L, K, pr, ps), fhe_amd.HipContext(logN, L, K, pr, ps, device=0)
        rng = np.random.default_rng(7)
        poly = lambda: np.stack([rng.integers(0, int(pr[i]), o.N, dtype=np.uint64) for i in range(Li)])
        dev = lambda a: torch.from_numpy(a.view(np.int64)).cuda()
        sx, ext = o.keygen_sk(seed=1); beta, alpha = o.keygen_mult(sx, seed=2)
        ev = ShardedEvaluator(g)
        assert ev.batch.world == 1 and not ev.batch.stage
        ev.load_mult_key(dev(beta), dev(alpha))
        c0 = np.stack([np.stack([poly(), poly()]) for _ in range(B)]); c1 = np.stack([np.stack([poly(), poly()]) for _ in range(B)])
        got = ev.mul_relin_rescale(dev(c0), dev(c1), B, Li, chunk=2).cpu().numpy().view(np.uint64)
        for b in range(B):
            wb, wa = o.mul_relin_rescale(c0[b, 0], c0[b, 1], c1[b, 0], c1[b, 1], beta, alpha)
            assert np.array_equal(got[b, 0], wb) and np.array_equal(got[b, 1], wa)
        g.close(); dist.destroy_process_group()
        print("rccl single rank OK")
    ''')
    r = _run_py(code, extra_env={"MASTER_PORT": str(port)})
    assert r.returncode == 0 and "rccl single rank OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_bench_multi_rank_logic_rehearsed_on_one_gpu():
    """bench.py --gpus 2 with both ranks on the one GPU (gloo, host-staged transfers): the launcher, the barriers and the
    MAX over ranks, the key broadcast, the parity check over the collective (`rccl_parity`, `ranks_seen`) and --e2e all run;
    the line says it is a rehearsal.  What it cannot cover is RCCL's own peer-to-peer transport."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-shared-gpu", "--workload", "c5", "--batch", "4",
                        "--steps", "2", "--warmup", "1", "--prewarm", "0", "--e2e", "--e2e-chunk", "2", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["rccl_parity"] is True and j["ranks_seen"] == 2 and "rehearsal" in j
    assert j["e2e"]["last_op_matches_local_recompute"] is True and j["e2e"]["ops_per_call"] == 8
    assert j["value"] > 0 and j["scaling"] == "weak"
