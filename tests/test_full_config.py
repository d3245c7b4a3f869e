"""Parity at BASELINE.json's full sizes ("close parity at the full configs").

configs[2]: the whole Icp3d::estimate (src/lib.rs:148-173) on the synthetic 1M x 1M pair, 20 outer
iterations from the identity pose -- pose, indices and inner-iteration counts bit-equal to the
oracle evaluated in the device's reduction order, and within the north_star's 1e-5 of the
oracle in the reference's own (left fold) order.
configs[4] (first half, the growing map): size-independent properties on a >= 10M-point target
cloud -- every target is its own nearest neighbour, and a handle grown by an append is
indistinguishable from a fresh handle on the concatenated cloud.
"""
import numpy as np
import pytest

import icp_rust_amd as I
import oracle_ffi as O
from parity_util import check_fold_order
from icp_rust_amd import synth

pytestmark = pytest.mark.gpu


def test_estimate_1m_x_1m_is_bit_equal_to_the_oracle_in_device_order():
    import torch

    n = m = 1_000_000
    src, dst = synth.synthetic_pair(n, m)
    icp = I.Icp3d(torch.from_numpy(dst).cuda())
    T, idx, inner = icp.estimate(torch.from_numpy(src).cuda(), I.Transform(), 20, return_info=True)
    blocks, threads = I.reduce_geometry(n)
    assert (blocks, threads) == (256, 512)
    # every iteration of this pair applies one update: from the third on the search of iteration k + 2 was enqueued
    # behind the first evaluation of iteration k + 1 with a pose the DEVICE derived -- and the host, deriving the same
    # pose from the evaluation's sums, must have found the same bits every time (a miss would only cost time, but
    # on this pair it means the two solves disagree)
    hits, misses = I.run_ahead_counters(icp)
    assert hits >= 10 and misses == 0, (hits, misses)
    # the order the call folded its sums in: the cell-sorted snapshot of the source cloud (icp_last_fold_order)
    perm, cell = icp.last_fold_order(n, with_cells=True)
    check_fold_order(perm, cell)
    assert not np.array_equal(perm, np.arange(n))
    tree = O.KdTree(dst)
    O.set_threads(16)  # the independent kd queries of one search over host cores; results do not depend on it
    try:
        rc, oT, oidx_s, oinner = tree.estimate(np.ascontiguousarray(src[perm]), O.transform_identity(), 20,
                                               O.IcpOpts(1, 1, blocks, threads))
        assert rc == O.OK
        oidx = np.empty_like(oidx_s)
        oidx[perm] = oidx_s
        assert np.array_equal(T.as_array(), oT.as_array()), (T.as_array(), oT.as_array())
        assert np.array_equal(idx, oidx)
        assert np.array_equal(inner, oinner)
        # the same registration in the reference's summation order: north_star tolerance
        rc, rT, ridx, _ = tree.estimate(src, O.transform_identity(), 20)
        assert rc == O.OK
    finally:
        O.set_threads(1)
    want = rT.as_array()
    assert np.max(np.abs(T.as_array() - want)) <= 1e-5 * max(1.0, float(np.max(np.abs(want))))
    assert np.array_equal(idx, ridx)
    # 20 iterations from the identity are not enough to converge on this pair (independent samples of
    # the same surfaces: every iteration applies one small update); it must have moved towards the truth
    truth = I.Transform(synth.TRUTH_PARAM).as_array()
    assert np.max(np.abs(T.as_array() - truth)) < 0.6 * np.max(np.abs(I.Transform().as_array() - truth))


def test_map_of_10m_points_self_query_and_append_equals_fresh_create():
    import torch

    m0, k = 9_400_000, 600_000
    m = m0 + k
    cloud = synth.box_cloud(synth.SEED + 7, m)
    d_all = torch.from_numpy(cloud).cuda()
    fresh = I.Icp3d(d_all)
    assert I.lib().icp_get_nn_mode(fresh._h) == I.NN_GRID
    # (1) self-query: every target is its own nearest neighbour (an exact duplicate -> the lowest index)
    idx = torch.empty(m, dtype=torch.int32, device="cuda")
    fresh.nn_search_device(d_all, idx)
    fresh.synchronize()
    got = idx.cpu().numpy().view(np.uint32)
    bad = np.nonzero(got != np.arange(m, dtype=np.uint32))[0]
    assert len(bad) < 1000
    for i in bad:
        assert got[i] < i and np.array_equal(cloud[got[i]], cloud[i])
    # (2) a handle that grew to the same cloud answers every query with the same index
    grown = I.Icp3d(cloud[:m0])
    grown.append(d_all[m0:].contiguous())
    assert grown.target_count == m
    src, _ = synth.synthetic_pair(1_000_000, 1)
    d_q = torch.from_numpy(src).cuda()
    ia = torch.empty(len(src), dtype=torch.int32, device="cuda")
    ib = torch.empty_like(ia)
    fresh.nn_search_device(d_q, ia)
    grown.nn_search_device(d_q, ib)
    fresh.synchronize()
    grown.synchronize()
    assert torch.equal(ia, ib)
    # (3) ... and registers a scan to the same bits
    scan = d_q[:28_800].contiguous()
    init = I.Transform([0.05, -0.02, 0.004])
    Ta, xa, na = fresh.estimate(scan, init, 6, return_info=True)
    Tb, xb, nb = grown.estimate(scan, init, 6, return_info=True)
    assert np.array_equal(Ta.as_array(), Tb.as_array())
    assert np.array_equal(xa, xb) and np.array_equal(na, nb)
    # a sample of the queries against the oracle's brute force (the oracle at 10M targets x 2000 queries: seconds)
    O.set_threads(16)
    try:
        rc, want = O.nn_brute(cloud, src[:2000])
    finally:
        O.set_threads(1)
    assert rc == O.OK and np.array_equal(ia[:2000].cpu().numpy().view(np.uint32), want)
