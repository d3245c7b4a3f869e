"""Multi-GPU driver: the source cloud shards across ranks, the target cloud is replicated.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  Per outer ICP
iteration (src/lib.rs:155-171):

  1. every rank transforms + nearest-neighbour-matches ITS contiguous range of the source
     cloud (no communication: source points are independent);
  2. one all-gather gives every rank all N matched pairs in the global point order: of the
     4-byte correspondence indices when the host hands every rank the whole source cloud
     (`src_full`; 24 MB at 1M points, replicated once) -- each rank then rebuilds the pairs
     locally with the same arithmetic -- or else of the 32-byte pairs themselves;
  3. every rank runs the identical, deterministic inner Gauss-Newton loop on all N pairs
     (exact medians are not all-reducible sums; replicating the loop costs < 1 % of step 1
     and needs no further collective), so all ranks hold bit-identical poses and the
     N-GPU result equals the 1-GPU result bit for bit.

The compute is delegated to a `stages` object so that the orchestration (ranges, gather,
replication) is testable on CPU with the gloo backend; production uses HipStages.
"""
import numpy as np

from .api import Transform


def shard_range(n, rank, world):
    """contiguous range [lo, hi) of rank `rank`: sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class HipStages:
    """The two device stages of the C ABI on torch CUDA tensors (include/icp_mi355x.h, section 4)."""

    def __init__(self, icp):
        import torch

        self.icp = icp
        self.torch = torch
        icp.set_stream(torch.cuda.current_stream().cuda_stream)

    def prepare(self, src_shard, T):
        self.icp.prepare_source_device(src_shard, T)

    def correspond(self, src_shard, T, a_out, b_out):
        self.icp.correspond_device(src_shard, T, a_out, b_out)

    def correspond_idx(self, src_shard, T, idx_out):
        self.icp.correspond_device(src_shard, T, None, None, idx_out)

    def materialize(self, src_full, T, idx_full, a_out, b_out):
        self.icp.materialize_pairs_device(src_full, T, idx_full, a_out, b_out)

    def estimate_transform(self, a_full, b_full):
        return self.icp.estimate_transform_device(a_full, b_full)


class ShardedIcp:
    def __init__(self, stages, n_total, rank=0, world=1, group=None, mul=None, src_full=None):
        self.stages = stages
        self.n = n_total
        self.src_full = src_full  # optional: the whole source cloud on every rank -> index gather
        self.rank, self.world, self.group = rank, world, group
        self.lo, self.hi = shard_range(n_total, rank, world)
        self.max_shard = shard_range(n_total, 0, world)[1]
        self._mul = mul or (lambda a, b: a * b)
        self._bufs = None

    def _buffers(self, like):
        import torch

        if self._bufs is None:
            kw = dict(dtype=torch.float64, device=like.device)
            W, ms = self.world, self.max_shard
            even = self.n % W == 0
            gathered = torch.empty((2, W * ms, 2), **kw)  # [a|b] x rank-major padded shards
            local = torch.empty((2, ms, 2), **kw)
            full = gathered if even else torch.empty((2, self.n, 2), **kw)
            self._bufs = (local, gathered, full, even)
        return self._bufs

    def _step_idx(self, src_shard, T):
        """index-gather variant: 4 B/point on the wire, pairs rebuilt locally."""
        import torch
        import torch.distributed as dist

        if getattr(self, "_ibufs", None) is None:
            W, ms = self.world, self.max_shard
            kw = dict(device=src_shard.device)
            self._ibufs = (torch.empty(ms, dtype=torch.int32, **kw), torch.empty(W * ms, dtype=torch.int32, **kw),
                           torch.empty(self.n, dtype=torch.int32, **kw),
                           torch.empty((2, self.n, 2), dtype=torch.float64, **kw))
        loc, gat, idx_full, ab = self._ibufs
        ns = self.hi - self.lo
        self.stages.correspond_idx(src_shard, T, loc[:ns])
        dist.all_gather_into_tensor(gat, loc, group=self.group)
        if self.n % self.world == 0:
            idx_full = gat
        else:
            ms = self.max_shard
            for r in range(self.world):
                lo, hi = shard_range(self.n, r, self.world)
                idx_full[lo:hi] = gat[r * ms: r * ms + (hi - lo)]
        self.stages.materialize(self.src_full, T, idx_full[: self.n], ab[0], ab[1])
        dT, inner = self.stages.estimate_transform(ab[0], ab[1])
        return self._mul(dT, T), inner

    def step(self, src_shard, T):
        """one outer iteration; returns (dT * T, inner_iters)."""
        import torch
        import torch.distributed as dist

        if self.world > 1 and self.src_full is not None and hasattr(self.stages, "materialize"):
            return self._step_idx(src_shard, T)
        local, gathered, full, even = self._buffers(src_shard)
        ns = self.hi - self.lo
        self.stages.correspond(src_shard, T, local[0, :ns], local[1, :ns])
        if self.world > 1:
            # a and b travel in one collective each so that the receive layout is rank-major
            dist.all_gather_into_tensor(gathered[0], local[0], group=self.group)
            dist.all_gather_into_tensor(gathered[1], local[1], group=self.group)
            if not even:
                ms = self.max_shard
                for r in range(self.world):
                    lo, hi = shard_range(self.n, r, self.world)
                    full[:, lo:hi] = gathered[:, r * ms: r * ms + (hi - lo)]
        else:
            full = local
        dT, inner = self.stages.estimate_transform(full[0, : self.n], full[1, : self.n])
        return self._mul(dT, T), inner

    def estimate(self, src_shard, initial_transform, max_iter):
        """Icp{2,3}d::estimate over the sharded source (src/lib.rs:105-130, 148-173)."""
        T = initial_transform
        inner = []
        if max_iter > 0 and hasattr(self.stages, "prepare"):
            self.stages.prepare(src_shard, T)  # once per estimate call, like Icp::estimate itself
        for _ in range(max_iter):
            T, k = self.step(src_shard, T)
            inner.append(k)
        return T, np.array(inner, dtype=np.uint32)
