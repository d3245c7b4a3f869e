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


# =========================================================================================
# Block-sharded driver (round 2): source sharded by reduction-tree block, three small exchanges per
# evaluation, bit-identical to one rank.  CPU stand-in for the device stages below; the real
# HipStages run the same orchestration in tests/test_gpu_shard.py (virtual ranks on one GPU).
# =========================================================================================
from icp_rust_amd import _lib  # noqa: E402
from icp_rust_amd.dist import BlockShardedIcp, LocalComm, TorchComm, block_shard, local_indices  # noqa: E402


def _opose(T):
    return O.Pose(*T.pose.as_tuple())


class OracleBlockStages:
    """The block-sharded stage interface on CPU tensors.  What is under test is the ORCHESTRATION
    (ownership, the three exchanges, the replicated fallback, lockstep decisions), so the stand-in keeps
    the protocol's shape but not the device's selection kernels: its "histogram" is one counter, its
    "candidates" are all of its residuals; statistics and sums come from the oracle (exact medians,
    tree-order block sums)."""

    hist_words = 16

    def __init__(self, dst, n_total, world, miss_every=0):
        self.dst, self.tree = dst, O.KdTree(dst)
        self.n_total, self.world = n_total, world
        self.nl_max = max(block_shard(n_total, r, world)[3] for r in range(world))
        self.seen_kind = set()
        self.evals = 0
        self.miss_every = miss_every
        self.prepared = 0

    def exch_bytes(self, world):
        """what a rank hands to the others between compact and finish: [count, its residuals][blocks, its block sums]"""
        return 8 * (1 + 2 * self.nl_max) + 8 * (O.TREE_SUMS * (256 // world + 2) + 4)

    def empty(self, nbytes, like):
        return torch.zeros(nbytes, dtype=torch.uint8)

    def empty_points(self, n, cols, like):
        return torch.zeros((n, cols), dtype=torch.float64)

    def empty_index(self, n, like):
        return torch.zeros(n, dtype=torch.int32)

    def prepare(self, src_local, T):
        self.prepared += 1

    def take(self, full, local, n_total, rank, world):
        local.copy_(full[torch.from_numpy(local_indices(n_total, rank, world))])

    def put(self, local, full, n_total, rank, world):
        full[torch.from_numpy(local_indices(n_total, rank, world))] = local

    def correspond(self, src_local, T, a_out, b_out, idx_out=None):
        p = T.pose
        s = src_local.numpy()
        st = s.copy()
        st[:, 0] = (p.r00 * s[:, 0] + p.r01 * s[:, 1]) + p.tx
        st[:, 1] = (p.r10 * s[:, 0] + p.r11 * s[:, 1]) + p.ty
        rc, idx = self.tree.search(st)
        assert rc == O.OK
        a_out.copy_(torch.from_numpy(np.ascontiguousarray(st[:, :2])))
        b_out.copy_(torch.from_numpy(np.ascontiguousarray(self.dst[idx][:, :2])))
        if idx_out is not None:
            idx_out.copy_(torch.from_numpy(idx.astype(np.int32)))

    def eval_hist(self, a, b, n_total, rank, world, T, kind, refined=False):
        if n_total < 2:
            return _lib.NONE, None
        if kind not in self.seen_kind:  # no prediction for this kind of evaluation yet
            return _lib.RETRY_REPLICATED, None
        self.refined = refined
        self.cur = (a.numpy(), b.numpy(), rank, world, T)
        p = T.pose
        an = self.cur[0]
        self.res = np.stack([((p.r00 * an[:, 0] + p.r01 * an[:, 1]) + p.tx) - self.cur[1][:, 0],
                             ((p.r10 * an[:, 0] + p.r11 * an[:, 1]) + p.ty) - self.cur[1][:, 1]], axis=1)
        h = torch.zeros(self.hist_words, dtype=torch.int32)
        h[0] = len(an)
        self.hist = h
        return _lib.OK, h

    def eval_compact(self, exch_out):
        assert int(self.hist[0]) == self.n_total  # the histogram now holds the sum over ranks
        a, b, rank, world, T = self.cur
        v = exch_out.view(torch.float64)
        v.zero_()
        v[0] = float(len(self.res))
        v[1:1 + 2 * len(self.res)] = torch.from_numpy(self.res.reshape(-1).copy())
        # the sums of this rank's blocks do not wait for the statistics (no 1 / sigma in them: applied after the fold)
        b0, b1, blocks, nl = block_shard(self.n_total, rank, world)
        parts = O.wgn_tree_partials(_opose(T), a, b, b1 - b0, 512)
        off = 1 + 2 * self.nl_max
        v[off] = float(b1 - b0)
        v[off + 1:off + 1 + parts.size] = torch.from_numpy(parts.reshape(-1).copy())
        return _lib.OK

    def eval_finish(self, exch_all):
        world = self.cur[3]
        allv = exch_all.view(torch.float64).numpy().reshape(world, -1)
        res = np.concatenate([allv[q, 1:1 + 2 * int(allv[q, 0])].reshape(-1, 2) for q in range(world)])
        assert len(res) == self.n_total
        rc, sd = O.calc_stddevs(res)  # every rank selects the same statistics from the same candidates
        assert rc == O.OK
        K = O.TREE_SUMS
        off = 1 + 2 * self.nl_max
        parts = np.concatenate([allv[q, off + 1:off + 1 + K * int(allv[q, off])].reshape(-1, K) for q in range(world)])
        sd = np.asarray(sd, dtype=np.float64)
        if not self.refined:
            self.evals += 1
        if self.miss_every and self.evals % self.miss_every == 0:  # "the predicted window missed"
            if not self.refined:
                return _lib.RETRY_SHARDED, None, 0.0      # its counts place a refined window: again, sharded
            if self.evals % (2 * self.miss_every) == 0:
                return _lib.RETRY_REPLICATED, None, 0.0   # every other time the refined attempt misses too
        rc, delta, err = O.wgn_tree_fold(parts, 512, sd)
        return (_lib.OK if rc == O.OK else _lib.NONE), delta, err

    def gn_step(self, a_full, b_full, T, kind):
        self.seen_kind.add(kind)
        blocks, threads = I.reduce_geometry(a_full.shape[0])
        rc, delta, err = O.weighted_gauss_newton_update_tree(_opose(T), a_full.numpy(), b_full.numpy(), blocks, threads)
        return (_lib.OK if rc == O.OK else _lib.NONE), delta, err


def test_block_shard_geometry_partitions_the_fold_order():
    for n in (2, 511, 512, 513, 4096, 70_001, 131_072, 131_073, 1_000_000):
        blocks, threads = I.reduce_geometry(n)
        G = blocks * threads
        for w in (1, 2, 3, 8):
            seen = []
            for r in range(w):
                b0, b1, bl, nl = block_shard(n, r, w)
                assert bl == blocks and b0 == blocks * r // w and b1 == blocks * (r + 1) // w
                idx = local_indices(n, r, w)
                assert len(idx) == nl
                # exactly the points whose fold thread lives in the rank's blocks, in fold order per thread
                assert np.all(((idx % G) // threads >= b0) & ((idx % G) // threads < b1))
                assert np.all(np.diff(idx) > 0)
                seen.append(idx)
            assert np.array_equal(np.sort(np.concatenate(seen)), np.arange(n))


def _reference(n, m, max_iter):
    src, dst = synth.synthetic_pair(n, m)
    blocks, threads = I.reduce_geometry(n)
    rc, T, idx, inner = O.icp_estimate(3, dst, src, O.transform_identity(), max_iter, use_kdtree=True, sum_mode=1,
                                       reduce_blocks=blocks, reduce_threads=threads)
    assert rc == O.OK
    return src, dst, T, idx, inner


@pytest.mark.parametrize("world,n,miss_every", [(1, 5000, 0), (2, 5000, 0), (3, 9001, 0), (8, 6000, 0), (4, 70_001, 3)])
def test_block_sharded_driver_with_virtual_ranks_equals_one_rank(world, n, miss_every):
    """all ranks in one process (LocalComm): the lockstep path the 1-GPU test of the N-rank code uses"""
    m, max_iter = 3000, 4
    src, dst, want_T, want_idx, want_inner = _reference(n, m, max_iter)
    stages = {r: OracleBlockStages(dst, n, world, miss_every) for r in range(world)}
    drv = BlockShardedIcp(stages, n, world, LocalComm(world))
    full = torch.from_numpy(src)
    T, inner = drv.estimate(drv.take_source(full), I.Transform(), max_iter)
    assert np.array_equal(T.as_array(), want_T.as_array())
    assert np.array_equal(inner, want_inner)
    assert drv.counters["sharded"] > 0 and drv.counters["replicated"] >= 2
    got_idx = np.zeros(n, dtype=np.uint32)
    for r, ix in drv.last_indices().items():
        got_idx[local_indices(n, r, world)] = ix.numpy().astype(np.uint32)
    assert np.array_equal(got_idx, want_idx)
    assert all(s.prepared == 1 for s in stages.values())


class GivingUpLoopStages(OracleBlockStages):
    """OracleBlockStages + a stand-in for the one-launch inner loop (include/icp_mi355x.h section 5b) that connects,
    accepts `good` launches without serving an evaluation (it hands evaluation `it` back at once, state untouched) and
    then GIVES UP: icp_shard_loop_wait -> ICP_HIP_ERROR, as when a peer did not arrive.  The driver must take nothing
    from such a launch, drop the rank's window predictions and go on with the stage calls -- on every rank alike."""

    def __init__(self, dst, n_total, world, good=1):
        super().__init__(dst, n_total, world)
        self.good, self.launched, self.resets, self.state = good, 0, 0, None

    def loop_inbox(self, kind=0):
        return 1000 + id(self) % 1000

    def loop_connect(self, rank, world, ptrs):
        assert len(ptrs) == world

    def loop_launch(self, a, b, n_total, launch_no, eval_base, it0, applied, Ti, prev_error, first_kind=0, second_kind=1):
        self.launched += 1
        self.state = (Ti, prev_error, applied, it0)
        return I._lib.OK

    def loop_wait(self):
        Ti, pe, ap, it = self.state
        if self.launched > self.good:
            return I._lib.HIP_ERROR, I.Transform(), 0.0, 0, 0, False, 0
        return I._lib.OK, Ti, pe, ap, it, False, 0  # (evaluation `it` handed back, nothing served)

    def reset_predictions(self):
        self.resets += 1


@pytest.mark.parametrize("world,n", [(2, 5000), (3, 9001)])
def test_a_loop_launch_that_gives_up_sends_every_rank_to_the_stage_calls(world, n):
    """a sharded launch that gave up used to raise on ONE rank while the others went on.  Now the kernel
    raises abort in every inbox, every rank's wait reports ICP_HIP_ERROR for that launch and the driver falls back on
    all of them: same pose as a run that never had the loop, predictions dropped once per rank, no launch afterwards."""
    m, max_iter = 3000, 4
    src, dst, want_T, _, want_inner = _reference(n, m, max_iter)
    stages = {r: GivingUpLoopStages(dst, n, world, good=2) for r in range(world)}
    drv = BlockShardedIcp(stages, n, world, LocalComm(world))
    assert drv.connect_loop() is not None
    T, inner = drv.estimate(drv.take_source(torch.from_numpy(src)), I.Transform(), max_iter)
    assert np.array_equal(T.as_array(), want_T.as_array()) and np.array_equal(inner, want_inner)
    assert drv.counters["loop_gave_up"] == 1 and drv._loop is None
    assert all(s.resets == 1 and s.launched == 3 for s in stages.values()), [(s.resets, s.launched) for s in stages.values()]


def _block_worker(rank, world, port, n, m, max_iter, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        src, dst = synth.synthetic_pair(n, m)
        stages = {rank: OracleBlockStages(dst, n, world, miss_every=4)}
        drv = BlockShardedIcp(stages, n, world, TorchComm(rank, world))
        local = drv.take_source(torch.from_numpy(src))
        assert local[rank].shape[0] == block_shard(n, rank, world)[3]
        T, inner = drv.estimate(local, I.Transform(), max_iter)
        t = torch.from_numpy(T.as_array().copy())
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        for g in gathered:
            assert torch.equal(g, gathered[0])
        if rank == 0:
            np.save(out, np.concatenate([T.as_array(), inner.astype(np.float64),
                                         [drv.counters["sharded"], drv.counters["replicated"]]]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 6000), (3, 9001)])
def test_block_sharded_driver_over_gloo_equals_one_rank(tmp_path, world, n):
    m, max_iter = 3000, 4
    out = str(tmp_path / "pose.npy")
    mp.spawn(_block_worker, args=(world, _free_port(), n, m, max_iter, out), nprocs=world, join=True)
    got = np.load(out)
    _, _, want_T, _, want_inner = _reference(n, m, max_iter)
    assert np.array_equal(got[:6], want_T.as_array())
    assert np.array_equal(got[6:6 + max_iter], want_inner.astype(np.float64))
    assert got[-2] > 0 and got[-1] > 0
