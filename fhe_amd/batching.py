"""Batching layer (north_star subsystem 4): shard independent ciphertext operations over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on GPUs, "gloo" in the CPU tests).  Nothing in
fHE's path couples different ciphertexts, so the data path needs NO collective: ranks work on disjoint,
contiguous slices of the batch with keys and tables replicated.  RCCL is used only either side of the path, for callers
whose data starts on one rank: a star scatter of inputs from the root (grouped point-to-point sends, so all xGMI links of
the root carry traffic concurrently), a broadcast of key material once, and the mirror-image gather of results.
The reference has no counterpart (it is single-process, single-thread).
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, world, rank):
    """Contiguous split [lo, hi) of n_items over `world` ranks; the first n_items % world ranks get one more."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedBatch:
    """Scatter / compute / gather of a batch whose leading dimension indexes independent ciphertexts."""

    def __init__(self, group=None, root=0, device=None):
        self.group = group
        self.root = root
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.device = device
        # gloo moves host memory only: device tensors are staged through host copies there (CPU tests, and rehearsals of the
        # multi-rank logic on a box without RCCL peers); with nccl (= RCCL) the device buffers go on the wire as they are
        self.stage = dist.is_initialized() and dist.get_backend(group) == "gloo"

    def _wire(self, t):
        """The tensor a point-to-point operation is posted with."""
        return t.cpu() if (self.stage and t.is_cuda) else t

    def bounds(self, n_items, rank=None):
        return shard_bounds(n_items, self.world, self.rank if rank is None else rank)

    def _p2p_start(self, ops):
        return dist.batch_isend_irecv(ops) if ops else []

    @staticmethod
    def _p2p_wait(reqs):
        for req in reqs:
            if callable(req):
                req()       # staged receive: copy the host buffer into its device destination
            else:
                req.wait()  # RCCL: the current stream waits, the host does not

    def _recv_op(self, dst, peer, post):
        """irecv into dst (staged through the host under gloo); `post` collects the copies to run after the wait."""
        if self.stage and dst.is_cuda:
            host = torch.empty(dst.shape, dtype=dst.dtype)
            post.append(lambda: dst.copy_(host))
            return dist.P2POp(dist.irecv, host, peer, self.group)
        return dist.P2POp(dist.irecv, dst, peer, self.group)

    def _p2p(self, ops):
        self._p2p_wait(self._p2p_start(ops))

    def scatter_start(self, full, n_items, item_shape, dtype=torch.int64):
        """Begin a scatter: returns (local slice, pending requests, buffers to keep alive).  The slice may be used after
        `_p2p_wait(requests)`."""
        lo, hi = self.bounds(n_items)
        if self.world == 1:
            return full[lo:hi], [], None
        if self.rank == self.root:
            ops, keep = [], []
            for r in range(self.world):
                if r == self.root:
                    continue
                rlo, rhi = self.bounds(n_items, r)
                if rhi > rlo:
                    keep.append(self._wire(full[rlo:rhi].contiguous()))
                    ops.append(dist.P2POp(dist.isend, keep[-1], r, self.group))
            return full[lo:hi], self._p2p_start(ops), keep
        local = torch.empty((hi - lo,) + tuple(item_shape), dtype=dtype, device=self.device)
        post = []
        reqs = self._p2p_start([self._recv_op(local, self.root, post)]) if hi > lo else []
        return local, list(reqs) + post, None

    def scatter(self, full, n_items, item_shape, dtype=torch.int64):
        """Root holds `full` = [n_items, *item_shape]; every rank returns its slice [hi-lo, *item_shape]."""
        local, reqs, _keep = self.scatter_start(full, n_items, item_shape, dtype)
        self._p2p_wait(reqs)
        return local

    def gather(self, local, n_items, item_shape, dtype=torch.int64):
        """Mirror of scatter: the root returns [n_items, *item_shape], other ranks None."""
        lo, hi = self.bounds(n_items)
        if self.world == 1:
            return local
        if self.rank == self.root:
            full = torch.empty((n_items,) + tuple(item_shape), dtype=dtype, device=self.device)
            full[lo:hi] = local
            ops, post = [], []
            for r in range(self.world):
                if r == self.root:
                    continue
                rlo, rhi = self.bounds(n_items, r)
                if rhi > rlo:
                    ops.append(self._recv_op(full[rlo:rhi], r, post))
            self._p2p_wait(list(self._p2p_start(ops)) + post)
            return full
        if hi > lo:
            self._p2p([dist.P2POp(dist.isend, self._wire(local.contiguous()), self.root, self.group)])
        return None

    def broadcast(self, tensor):
        """Key material: once, from the root to every rank."""
        if self.world > 1:
            if self.stage and tensor.is_cuda:
                host = tensor.cpu()
                dist.broadcast(host, self.root, group=self.group)
                tensor.copy_(host)
            else:
                dist.broadcast(tensor, self.root, group=self.group)
        return tensor

    def run(self, compute, inputs, n_items, in_shapes, out_shape, chunk=None):
        """inputs: list of root-resident tensors (None on other ranks); compute(local_inputs, n_local) -> local output.
        With `chunk`, the scatter / compute / gather proceeds in chunks of that many items per rank so that a batch
        larger than one GPU's memory can stream through."""
        outs = []
        step = max(1, n_items if not chunk else chunk * self.world)
        starts = list(range(0, max(n_items, 1), step))

        def begin(c0):
            n = min(step, n_items - c0)
            return n, [self.scatter_start(None if x is None else x[c0:c0 + n], n, shp) for x, shp in zip(inputs, in_shapes)]

        # the scatter of chunk k+1 is issued before chunk k is computed, so on GPUs it travels while chunk k computes
        # (its requests sit ahead of chunk k's gather on the communication stream)
        nxt = begin(starts[0])
        for k, c0 in enumerate(starts):
            n, pend = nxt
            for _local, reqs, _keep in pend:
                self._p2p_wait(reqs)
            local_in = [p[0] for p in pend]
            nxt = begin(starts[k + 1]) if k + 1 < len(starts) else None
            lo, hi = self.bounds(n)
            local_out = compute(local_in, hi - lo)
            outs.append(self.gather(local_out, n, out_shape))
        if self.rank != self.root:
            return None
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)


class ShardedEvaluator:
    """Evaluator::multiply_and_relin / rotate_slots over a sharded batch: each rank owns one HipContext."""

    def __init__(self, ctx, group=None, root=0):
        self.ctx = ctx
        self.N = ctx.N
        self.batch = ShardedBatch(group, root, device=torch.device("cuda", ctx.device))

    def load_mult_key(self, beta, alpha):
        L, N = self.ctx.L, self.N
        if beta is None:
            beta = torch.empty((L, L, N), dtype=torch.int64, device=self.batch.device)
            alpha = torch.empty_like(beta)
        self.batch.broadcast(beta)
        self.batch.broadcast(alpha)
        self.ctx.load_mult_key(beta, alpha)

    def mul_relin_rescale(self, c0, c1, n_items, Li, chunk=None):
        N = self.N

        def compute(local, n):
            out = torch.empty((n, 2, Li - 1, N), dtype=torch.int64, device=self.batch.device)
            if n:
                self.ctx.mul_relin_rescale(out, local[0].contiguous(), local[1].contiguous(), Li, batch=n)
            return out

        return self.batch.run(compute, [c0, c1], n_items, [(2, Li, N)] * 2, (2, Li - 1, N), chunk)

    def rotate(self, galois, ct, n_items, Li, chunk=None):
        N = self.N

        def compute(local, n):
            work = torch.empty((n, 2, Li, N), dtype=torch.int64, device=self.batch.device)
            if n:
                self.ctx.rotate_to(galois, work, local[0].contiguous(), Li, batch=n)
            return work

        return self.batch.run(compute, [ct], n_items, [(2, Li, N)], (2, Li, N), chunk)
