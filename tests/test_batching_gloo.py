"""N>1 path on CPU: world_size-2 gloo run of the batching layer.  The per-rank compute is the CPU oracle (test code may
call it), so the check is the property the multi-GPU path must have: the concatenated sharded output equals the
single-rank output byte for byte, for even, uneven, tiny and chunked batches."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fhe_amd.batching import ShardedBatch, shard_bounds


def test_shard_bounds_cover_the_batch():
    for n in (0, 1, 2, 7, 8, 8192):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_items, chunk, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    logN, L, K, Li = 10, 3, 1, 3
    o = orc.Oracle(logN, L, K)
    N = o.N
    sx, _ = o.keygen_sk(seed=5)
    beta, alpha = o.keygen_mult(sx, seed=6)  # every rank derives the same keys here; GPUs get them by broadcast
    sb = ShardedBatch(root=0)
    c0 = c1 = None
    if rank == 0:
        rng = np.random.default_rng(7)
        mk = lambda: np.stack([np.stack([np.stack([rng.integers(0, int(o.primes[i]), N, dtype=np.uint64) for i in range(Li)])  # noqa: E731
                                         for _ in range(2)]) for _ in range(n_items)]) if n_items else np.zeros((0, 2, Li, N), np.uint64)
        c0n, c1n = mk(), mk()
        c0, c1 = torch.from_numpy(c0n.view(np.int64)), torch.from_numpy(c1n.view(np.int64))

    def compute(local, n):
        out = np.zeros((n, 2, Li - 1, N), np.uint64)
        a, b = local[0].numpy().view(np.uint64), local[1].numpy().view(np.uint64)
        for k in range(n):
            ob, oa = o.mul_relin_rescale(np.ascontiguousarray(a[k, 0]), np.ascontiguousarray(a[k, 1]),
                                         np.ascontiguousarray(b[k, 0]), np.ascontiguousarray(b[k, 1]), beta, alpha)
            out[k, 0], out[k, 1] = ob, oa
        return torch.from_numpy(out.view(np.int64))

    got = sb.run(compute, [c0, c1], n_items, [(2, Li, N)] * 2, (2, Li - 1, N), chunk)
    kb = torch.arange(16, dtype=torch.int64) if rank == 0 else torch.zeros(16, dtype=torch.int64)
    sb.broadcast(kb)
    assert kb.tolist() == list(range(16))
    if rank == 0:
        want = compute([c0, c1], n_items)
        ok = got.shape == want.shape and bool((got == want).all())
        open(result_path, "w").write("ok" if ok else "mismatch")
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items,chunk", [(4, None), (5, None), (1, None), (0, None), (7, 2)])
def test_sharded_output_equals_single_rank_output(tmp_path, n_items, chunk):
    result = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), n_items, chunk, result), nprocs=2, join=True)
    assert open(result).read() == "ok"


def _verify_worker(rank, world, port, corrupt, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from oracle import oracle as orc
    logN, L, K, Li = 10, 3, 1, 3
    o = orc.Oracle(logN, L, K)
    N = o.N
    sx, _ = o.keygen_sk(seed=5)
    beta, alpha = o.keygen_mult(sx, seed=6)

    def compute(c0, c1, n):  # stands in for the rank's GPU: same keys on every rank
        out = np.zeros((n, 2, Li - 1, N), np.uint64)
        a, b = c0.numpy().view(np.uint64), c1.numpy().view(np.uint64)
        for k in range(n):
            out[k, 0], out[k, 1] = o.mul_relin_rescale(np.ascontiguousarray(a[k, 0]), np.ascontiguousarray(a[k, 1]),
                                                       np.ascontiguousarray(b[k, 0]), np.ascontiguousarray(b[k, 1]), beta, alpha)
        return torch.from_numpy(out.view(np.int64))

    rng = np.random.default_rng(100 + rank)  # every rank has its own operation, as in bench.py
    mk = lambda: torch.from_numpy(np.stack([np.stack([rng.integers(0, int(o.primes[i]), N, dtype=np.uint64) for i in range(Li)])  # noqa: E731
                                            for _ in range(2)])[None].view(np.int64))
    c0, c1 = mk(), mk()
    out = compute(c0, c1, 1)
    if corrupt and rank == 1:
        out[0, 1, 0, 7] ^= 1  # one flipped bit in what rank 1 "computed"
    ok, seen = bench.verify_over_collective(ShardedBatch(root=0), compute, c0, c1, out, (2, Li, N), (2, Li - 1, N))
    if rank == 0:
        open(result_path, "w").write("%s %s" % (ok, seen))
    else:
        assert ok is None and seen is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("corrupt,expect", [(False, "True 2"), (True, "False 1")])
def test_post_run_parity_check_over_the_collective(tmp_path, corrupt, expect):
    """bench.py --gpus N: one operation per rank travels to rank 0 and is recomputed there (`rccl_parity`, `ranks_seen`)."""
    result = str(tmp_path / "verify.txt")
    mp.spawn(_verify_worker, args=(2, _free_port(), corrupt, result), nprocs=2, join=True)
    assert open(result).read() == expect
