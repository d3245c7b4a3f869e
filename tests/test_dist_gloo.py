"""The N>1 path on CPU: world_size-2 (and 3, uneven shards) gloo runs of the sharded driver.

The orchestration under test is icp_rust_amd.dist.ShardedIcp (ranges, all-gather of the
matched pairs in global point order, replicated inner loop, pose composition through the
C ABI's host pose algebra).  The two device stages are stood in for by the CPU oracle --
allowed here because this is a test; the product's HipStages needs a GPU.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import icp_rust_amd as I
import oracle_ffi as O
from icp_rust_amd import synth
from icp_rust_amd.dist import ShardedIcp, shard_range


class OracleStages:
    """correspond / estimate_transform with the oracle, on CPU tensors."""

    def __init__(self, dst):
        self.dst = dst
        self.tree = O.KdTree(dst)
        self.prepared = 0

    def prepare(self, src_shard, T):
        self.prepared += 1

    def correspond(self, src_shard, T, a_out, b_out):
        p = T.pose
        s = src_shard.numpy()
        st = s.copy()
        st[:, 0] = (p.r00 * s[:, 0] + p.r01 * s[:, 1]) + p.tx
        st[:, 1] = (p.r10 * s[:, 0] + p.r11 * s[:, 1]) + p.ty
        rc, idx = self.tree.search(st)
        assert rc == O.OK
        self._last_idx = idx
        a_out.copy_(torch.from_numpy(np.ascontiguousarray(st[:, :2])))
        b_out.copy_(torch.from_numpy(np.ascontiguousarray(self.dst[idx][:, :2])))

    def correspond_idx(self, src_shard, T, idx_out):
        a = torch.empty((src_shard.shape[0], 2), dtype=torch.float64)
        b = torch.empty_like(a)
        self.correspond(src_shard, T, a, b)
        idx_out.copy_(torch.from_numpy(self._last_idx.astype(np.int32)))

    def materialize(self, src_full, T, idx_full, a_out, b_out):
        p = T.pose
        s = src_full.numpy()
        a_out[:, 0] = torch.from_numpy((p.r00 * s[:, 0] + p.r01 * s[:, 1]) + p.tx)
        a_out[:, 1] = torch.from_numpy((p.r10 * s[:, 0] + p.r11 * s[:, 1]) + p.ty)
        b_out.copy_(torch.from_numpy(np.ascontiguousarray(self.dst[idx_full.numpy()][:, :2])))

    def estimate_transform(self, a_full, b_full):
        T, applied = O.estimate_transform(a_full.numpy(), b_full.numpy())
        return I.Transform.from_pose(I.Pose(*[float(x) for x in T.as_array()])), applied


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, m, max_iter, out, idx_gather=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(n, rank, world)
        src_shard, dst = synth.synthetic_pair(n, m, src_first=lo, src_count=hi - lo)
        stages = OracleStages(dst)
        full = torch.from_numpy(synth.synthetic_pair(n, 1)[0]) if idx_gather else None
        drv = ShardedIcp(stages, n, rank, world, src_full=full)
        T, inner = drv.estimate(torch.from_numpy(src_shard), I.Transform(), max_iter)
        assert stages.prepared == 1
        # every rank must hold the same pose, bit for bit
        t = torch.from_numpy(T.as_array().copy())
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        for g in gathered:
            assert torch.equal(g, gathered[0])
        if rank == 0:
            np.save(out, np.concatenate([T.as_array(), inner.astype(np.float64)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,idx_gather", [(2, 4000, False), (2, 4001, False), (3, 3001, False),
                                                (2, 4000, True), (3, 3001, True)])
def test_sharded_driver_equals_single_process(tmp_path, world, n, idx_gather):
    m, max_iter = 3000, 4
    out = str(tmp_path / "pose.npy")
    mp.spawn(_worker, args=(world, _free_port(), n, m, max_iter, out, idx_gather), nprocs=world, join=True)
    got = np.load(out)
    src, dst = synth.synthetic_pair(n, m)
    rc, T, _, inner = O.icp_estimate(3, dst, src, O.transform_identity(), max_iter, use_kdtree=True)
    assert rc == O.OK
    assert np.array_equal(got[:6], T.as_array())          # N ranks == 1 process, bit for bit
    assert np.array_equal(got[6:], inner.astype(np.float64))


def test_shard_ranges_partition_the_cloud():
    for n in (0, 1, 7, 8, 1_000_000, 1_000_003):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1


def test_synthetic_shards_concatenate_to_the_full_cloud():
    full, _ = synth.synthetic_pair(1000, 10)
    parts = [synth.synthetic_pair(1000, 10, src_first=lo, src_count=hi - lo)[0]
             for lo, hi in (shard_range(1000, k, 3) for k in range(3))]
    assert np.array_equal(np.concatenate(parts), full)
