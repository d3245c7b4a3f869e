"""The N-rank path on ONE GPU: N "virtual ranks" (N handles on cuda:0) run the block-sharded driver
(icp_rust_amd/dist.py: BlockShardedIcp + HipStages, include/icp_mi355x.h section 5) in lockstep and must
reproduce the one-handle registration bit for bit -- pose, inner-iteration counts, correspondence
indices.  (8 real GPUs are the driver's to launch; tests/test_dist_gloo.py covers the same
orchestration over gloo with a CPU stand-in for the stages.)"""
import os

import numpy as np
import pytest

import icp_rust_amd as I
import oracle_ffi as O
from parity_util import oracle_in_device_order
from icp_rust_amd import synth
from icp_rust_amd.dist import BlockShardedIcp, HipStages, LocalComm, block_shard, local_indices

pytestmark = pytest.mark.gpu


def _run(world, n, m, max_iter, dim=3):
    import torch

    src, dst = synth.synthetic_pair(n, m)
    if dim == 2:
        src, dst = np.ascontiguousarray(src[:, :2]), np.ascontiguousarray(dst[:, :2])
    cls = I.Icp3d if dim == 3 else I.Icp2d
    d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
    one = cls(d_dst)
    T1, idx1, inner1 = one.estimate(d_src, I.Transform(), max_iter, return_info=True)
    handles = [cls(d_dst) for _ in range(world)]
    stages = {r: HipStages(handles[r]) for r in range(world)}
    drv = BlockShardedIcp(stages, n, world, LocalComm(world))
    # the one-GPU call folds its sums over the cell-sorted cloud (icp_last_fold_order); the sharded driver
    # shards that order: every rank sorts the full cloud the same way, then takes its blocks' points
    srt, perms = drv.sort_source(d_src, I.Transform())
    perm = perms[0].cpu().numpy().astype(np.int64)
    assert np.array_equal(perm, one.last_fold_order(n))
    for r in range(world):
        assert torch.equal(perms[r], perms[0]) and torch.equal(srt[r], srt[0])
    assert np.array_equal(srt[0].cpu().numpy(), src[perm])
    local = drv.take_source(srt)
    for r in range(world):
        assert np.array_equal(local[r].cpu().numpy(), src[perm][local_indices(n, r, world)])
    T, inner = drv.estimate(local, I.Transform(), max_iter)
    torch.cuda.synchronize()
    idx_s = np.zeros(n, dtype=np.uint32)
    for r, ix in drv.last_indices().items():
        idx_s[local_indices(n, r, world)] = ix.cpu().numpy().view(np.uint32)
    idx = np.zeros(n, dtype=np.uint32)
    idx[perm] = idx_s
    return (T1, idx1, inner1), (T, idx, inner), drv, (src, dst), one


@pytest.mark.parametrize("world,n,m", [(2, 150_000, 120_000), (3, 70_001, 50_000), (8, 150_000, 120_000), (4, 20_000, 9_000)])
def test_virtual_ranks_reproduce_one_rank_bit_for_bit(world, n, m):
    (T1, idx1, inner1), (T, idx, inner), drv, _, _ = _run(world, n, m, 6)
    assert np.array_equal(T.as_array(), T1.as_array())
    assert np.array_equal(inner, inner1)
    assert np.array_equal(idx, idx1)
    # the steady state really ran sharded (three small exchanges), not on gathered pairs
    assert drv.counters["sharded"] >= 4, drv.counters
    assert drv.counters["replicated"] >= 1  # the very first evaluation has no prediction of its statistics yet


def test_virtual_ranks_2d_and_against_the_oracle():
    (T1, idx1, inner1), (T, idx, inner), drv, (src, dst), one = _run(4, 70_000, 40_000, 4, dim=2)
    assert np.array_equal(T.as_array(), T1.as_array()) and np.array_equal(idx, idx1) and np.array_equal(inner, inner1)
    rc, oT, oidx, oinner = oracle_in_device_order(one, 2, dst, src, O.transform_identity(), 4)
    assert rc == O.OK
    assert np.array_equal(T.as_array(), oT.as_array()) and np.array_equal(idx, oidx) and np.array_equal(inner, oinner)


def test_eight_virtual_ranks_at_the_full_size():
    """BASELINE configs[3] (1M x 1M over 8 ranks), as far as one GPU can rehearse it"""
    (T1, idx1, inner1), (T, idx, inner), drv, _, _ = _run(8, 1_000_000, 1_000_000, 5)
    assert np.array_equal(T.as_array(), T1.as_array())
    assert np.array_equal(inner, inner1) and np.array_equal(idx, idx1)
    assert drv.counters["sharded"] >= 6


def test_ranks_that_answer_differently_raise_together_instead_of_hanging():
    """every rank picks its branch from its own hist stage's answer.  The answers travel as status
    counters behind the histograms (include/icp_mi355x.h section 5), every rank joins the same three exchanges
    whatever it answered, and a disagreement -- here: rank 1's handle replaced by a fresh one without any window
    prediction, so it answers RETRY_REPLICATED while rank 0 answers OK -- is the same exception on every rank."""
    import torch

    n, m, world = 150_000, 120_000, 2
    src, dst = synth.synthetic_pair(n, m)
    d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
    handles = [I.Icp3d(d_dst) for _ in range(world)]
    drv = BlockShardedIcp({r: HipStages(handles[r]) for r in range(world)}, n, world, LocalComm(world))
    T, inner, _ = drv.estimate_full(d_src, I.Transform(), 4)
    assert drv.counters["sharded"] >= 4
    srt, _ = drv.sort_source(d_src, I.Transform())
    local = drv.take_source(srt)
    fresh = I.Icp3d(d_dst)
    drv.ranks[1].stages = HipStages(fresh)
    drv.ranks[1].bufs = None
    with pytest.raises(RuntimeError, match="answered differently"):
        drv.step(local, T)
    # both handles are back in their rest state: an ordinary registration on each of them still equals the oracle's
    for h in (handles[0], fresh):
        Tq, idx, inn = h.estimate(d_src, I.Transform(), 3, return_info=True)
        rc, oT, oidx, oinn = oracle_in_device_order(h, 3, dst, src, O.transform_identity(), 3)
        assert rc == O.OK and np.array_equal(Tq.as_array(), oT.as_array()) and np.array_equal(idx, oidx)


def test_stage_calls_refuse_what_they_cannot_serve():
    import torch

    src, dst = synth.synthetic_pair(3000, 9000)  # 6 blocks: fewer than 8 ranks
    h = I.Icp3d(torch.from_numpy(dst).cuda())
    st = HipStages(h)
    a = torch.zeros((3000, 2), dtype=torch.float64, device="cuda")
    rc, _ = st.eval_hist(a, a, 3000, 0, 8, I.Transform(), 0)
    assert rc == I._lib.RETRY_REPLICATED
    assert block_shard(3000, 7, 8)[3] + block_shard(3000, 0, 8)[3] <= 3000
    assert I.lib().icp_shard_eval_compact_device(h._h, None) == I._lib.BAD_ARGUMENT


@pytest.mark.parametrize("world,n,m,dim", [(2, 150_000, 120_000, 3), (8, 150_000, 120_000, 3), (3, 70_001, 50_000, 2),
                                            (4, 3_000, 9_000, 3), (4, 1, 5000, 3), (2, 0, 5000, 3)])
def test_icp_create_multi_with_virtual_ranks_equals_one_handle(world, n, m, dim):
    """icp_create_multi (one process, in-library exchange) with every rank on cuda:0: pose, indices
    and inner-iteration counts of icp_estimate on one handle, bit for bit.  3 000 points: fewer tree
    blocks than ranks -> every evaluation falls back to gathered pairs (still the same bits)."""
    src, dst = synth.synthetic_pair(max(n, 1), m)
    src = src[:n]
    if dim == 2:
        src, dst = np.ascontiguousarray(src[:, :2]), np.ascontiguousarray(dst[:, :2])
    one = (I.Icp3d if dim == 3 else I.Icp2d)(dst)
    init = I.Transform([0.01, -0.02, 0.001])
    T1, idx1, inner1 = one.estimate(src, init, 5, return_info=True)
    multi = I.IcpMulti(dst, [0] * world, dim=dim)
    T, idx, inner = multi.estimate(src, init, 5, return_info=True)
    assert np.array_equal(T.as_array(), T1.as_array())
    assert np.array_equal(inner, inner1) and np.array_equal(idx, idx1)
    sharded, replicated = multi.counters()
    if n >= 20_000:
        assert sharded >= 4 and replicated >= 1
    # a second call on the same object (buffers, flags and generations carry over)
    T2 = multi.estimate(src, init, 5)
    assert np.array_equal(T2.as_array(), T1.as_array())
    multi.close()


@pytest.mark.parametrize("world,n", [(2, 2_000_000), (4, 4_000_000), (8, 8 * 1024 * 1024)])
def test_more_than_a_million_pairs_across_ranks_equal_one_handle(world, n):
    """The weak-scaling regime (the only one where more GPUs can pay).  Round 6: beyond 2^20 points the reduction tree grows
    with the cloud -- a block per 4 096 points -- so a rank of `world` owns up to 256 blocks and evaluates like one GPU on
    1M points; the steady state goes through the pipelined evaluation (csrc/pipe.hip), everything else through the stage
    calls (the one-launch loop keeps to 2^20 pairs).  The result must be ONE handle's, bit for bit (one handle steps such
    clouds from the host: /root/reference/src/lib.rs:59-84)."""
    m = 500_000
    src, dst = synth.synthetic_pair(n, m)
    init = I.Transform([0.01, -0.02, 0.001])
    one = I.Icp3d(dst)
    T1, idx1, inner1 = one.estimate(src, init, 3, return_info=True)
    one.close()
    multi = I.IcpMulti(dst, [0] * world)
    T, idx, inner = multi.estimate(src, init, 3, return_info=True)
    assert np.array_equal(T.as_array(), T1.as_array())
    assert np.array_equal(inner, inner1) and np.array_equal(idx, idx1)
    sharded, replicated = multi.counters()
    assert sharded >= 3, (sharded, replicated, multi.loop_counters(), multi.pipe_iterations())
    multi.close()


def test_grown_multi_equals_fresh_multi_equals_one_handle():
    """BASELINE configs[4] across ranks (EXTENSION, include/icp_mi355x.h section 6): every rank appends the registered
    scan to its replica of the target cloud; the grown object must answer like a fresh one on the concatenated cloud
    and like ONE handle on it -- pose, indices, inner counts, bit for bit (frame loop: /root/reference/examples/scan3d.rs:113-133)."""
    world = synth.box_cloud(synth.SEED + 11, 90_000, synth.ROOM_LO, synth.ROOM_HI)
    first, rest = world[:50_000], world[50_000:]
    rng = np.random.default_rng(3)
    scan = world[rng.choice(len(world), 70_000, replace=False)] + rng.normal(size=(70_000, 3)) * 0.01
    T0 = I.Transform([0.04, -0.03, 0.008])
    moved = I.Transform([0.01, 0.02, -0.003])
    grown = I.IcpMulti(first, [0] * 4)
    Tg0 = grown.estimate(scan, T0, 3)  # (a registration before the append: snapshots, predictions, matches exist)
    grown.append(rest, moved)
    assert grown.target_count == len(world)
    one = I.Icp3d(first)
    one.append(rest, moved)
    cat = one.read_targets(0, len(world))
    fresh = I.IcpMulti(cat, [0] * 4)
    a = grown.estimate(scan, T0, 4, return_info=True)
    b = fresh.estimate(scan, T0, 4, return_info=True)
    c = one.estimate(scan, T0, 4, return_info=True)
    for x in (b, c):
        assert np.array_equal(a[0].as_array(), x[0].as_array())
        assert np.array_equal(a[1], x[1]) and np.array_equal(a[2], x[2])
    grown.close()
    fresh.close()


def test_icp_create_multi_argument_checks():
    _, dst = synth.synthetic_pair(1, 5000)
    with pytest.raises(I.IcpError):
        I.IcpMulti(dst, [0, 99])
    with pytest.raises(I.IcpError):
        I.IcpMulti(dst, [])
    e = I.IcpMulti(np.zeros((0, 3)), [0, 0])
    with pytest.raises(I.IcpError) as ex:
        e.estimate(dst[:10], I.Transform(), 1)
    assert ex.value.status == I._lib.EMPTY_DST


def test_virtual_ranks_beyond_4m_points_use_refined_windows_not_gathered_pairs():
    """past 4M points the one-GPU path finds its windows in two passes; the sharded path instead refines
    a window that missed from that attempt's own (global, exact) counts and stays sharded"""
    (T1, idx1, inner1), (T, idx, inner), drv, _, _ = _run(2, 4_500_000, 400_000, 3)
    assert np.array_equal(T.as_array(), T1.as_array())
    assert np.array_equal(inner, inner1) and np.array_equal(idx, idx1)
    assert drv.counters["sharded"] >= 3, drv.counters
    assert drv.counters["replicated"] <= 2, drv.counters


def test_recycled_handles_start_their_sharded_evaluations_in_step():
    """regression (profiles/multi_fuzz.py, seeds 40 .. 47): a sharded evaluation whose window missed parked its
    verdict in the handle's selection scratch; recycled through the handle pool, that handle sent its next owner's
    first un-sharded evaluation to the radix pipeline (no statistics recorded), and as one rank of a later sharded
    run it then disagreed with its peers about whether a window could be predicted (`icp_multi_estimate` refused
    with ICP_HIP_ERROR; over torch.distributed the ranks would have entered different collectives)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "profiles", "multi_fuzz.py"), "40", "8"], capture_output=True,
                         text=True, timeout=600, cwd=root)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert "8 seeds" in out.stdout or "0 mismatches" in out.stdout


@pytest.mark.parametrize("world", [2, 8])
def test_long_inner_loops_across_virtual_ranks_equal_one_handle(world):
    """The converging millimetre pair (inner loops of tens of updates): every inner loop is one launch per rank (fused
    for ranks that share a device) with three waits per evaluation across the ranks' inboxes, windows that follow the
    statistics, repeats inside the launch -- pose, indices and inner counts of one handle, bit for bit."""
    n = m = 150_000
    src, dst = synth.converging_pair(n, m)[:2]
    one = I.Icp3d(dst)
    T1, idx1, inner1 = one.estimate(src, I.Transform(), 8, return_info=True)
    assert max(int(x) for x in inner1) >= 5
    multi = I.IcpMulti(dst, [0] * world)
    T, idx, inner = multi.estimate(src, I.Transform(), 8, return_info=True)
    assert np.array_equal(T.as_array(), T1.as_array())
    assert np.array_equal(inner, inner1) and np.array_equal(idx, idx1)
    launches, served, handbacks = multi.loop_counters()
    piped = multi.pipe_iterations()  # (round 6: iterations whose loop applies one update go through the pipelined evaluation)
    assert launches + piped >= 6 and served + 2 * piped >= int(np.sum(inner1)), (launches, served, handbacks, piped)
    T2 = multi.estimate(src, I.Transform(), 8)  # (generations, parities and predictions carry over)
    assert np.array_equal(T2.as_array(), T1.as_array())
    multi.close()


def test_an_inner_loop_launch_that_hands_back_across_virtual_ranks():
    """A third of the source points sit EXACTLY on their targets along x (equal residuals on the median: more candidates
    than a workgroup can stage): the launches of all ranks report the miss together, the stage calls (and behind them
    the gathered pairs) serve that evaluation on every rank, and the result is still one handle's."""
    rng = np.random.default_rng(3)
    m = 90_000
    dst = synth.box_cloud(synth.SEED + 61, m, synth.ROOM_LO, synth.ROOM_HI)
    n = 60_000
    sel = rng.choice(m, n, replace=False)
    src = dst[sel].copy()
    noise = rng.normal(size=(n, 3)) * 0.004
    noise[: n // 3, 0] = 0.0  # exact x for a third of the points
    src += noise
    one = I.Icp3d(dst, nn_mode=I.NN_GRID)
    T1, idx1, inner1 = one.estimate(src, I.Transform(), 4, return_info=True)
    multi = I.IcpMulti(dst, [0, 0, 0])
    T, idx, inner = multi.estimate(src, I.Transform(), 4, return_info=True)
    assert np.array_equal(T.as_array(), T1.as_array())
    assert np.array_equal(inner, inner1) and np.array_equal(idx, idx1)
    multi.close()
