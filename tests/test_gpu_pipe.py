"""The PIPELINED sharded registration (csrc/pipe.hip, csrc/gn_win.hip: k_win_pick_shard; include/icp_mi355x.h section 5c):
in the steady state of a registration every rank runs the one-GPU pipeline -- search -> paired first launches ->
finishing workgroups that meet across the ranks -- and the result must be ONE handle's, bit for bit: pose, inner
counts, correspondence indices (/root/reference/src/lib.rs:105-130, 148-173, 59-84)."""
import numpy as np
import pytest

import icp_rust_amd as I
import oracle_ffi as O
from parity_util import oracle_in_device_order
from icp_rust_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,n,m,iters", [(1, 150_000, 120_000, 8), (2, 150_000, 120_000, 8), (4, 300_000, 250_000, 9),
                                             (8, 150_000, 120_000, 8), (3, 70_001, 50_000, 7)])
def test_pipelined_virtual_ranks_equal_one_handle(world, n, m, iters):
    src, dst = synth.synthetic_pair(n, m)
    one = I.Icp3d(dst)
    T1, idx1, inner1 = one.estimate(src, I.Transform(), iters, return_info=True)
    multi = I.IcpMulti(dst, [0] * world)
    T, idx, inner = multi.estimate(src, I.Transform(), iters, return_info=True)
    assert np.array_equal(inner, inner1), (inner, inner1)
    assert np.array_equal(T.as_array(), T1.as_array())
    assert np.array_equal(idx, idx1)
    served = multi.pipe_iterations()
    # the benchmark-shaped pair applies one update per outer iteration from the start: all but the first two or three
    # iterations (no prediction for both kinds of evaluation yet) must have gone through the pipeline
    assert served >= iters - 4, (served, inner.tolist())
    # ... and a second call on the same object, from a non-identity pose (generations and buffers carry over)
    init = I.Transform([0.01, -0.02, 0.001])
    T1b, idx1b, inner1b = one.estimate(src, init, iters, return_info=True)
    Tb, idxb, innerb = multi.estimate(src, init, iters, return_info=True)
    assert np.array_equal(Tb.as_array(), T1b.as_array()) and np.array_equal(innerb, inner1b) and np.array_equal(idxb, idx1b)
    assert multi.pipe_iterations() > served
    multi.close()
    one.close()


def test_pipelined_ranks_against_the_oracle_and_at_the_full_size():
    """BASELINE configs[3] as far as one GPU can rehearse it: 1M x 1M over 8 virtual ranks, 20 iterations"""
    n = m = 1_000_000
    src, dst = synth.synthetic_pair(n, m)
    one = I.Icp3d(dst)
    T1, idx1, inner1 = one.estimate(src, I.Transform(), 20, return_info=True)
    multi = I.IcpMulti(dst, [0] * 8)
    T, idx, inner = multi.estimate(src, I.Transform(), 20, return_info=True)
    assert np.array_equal(T.as_array(), T1.as_array()) and np.array_equal(inner, inner1) and np.array_equal(idx, idx1)
    assert multi.pipe_iterations() >= 16
    multi.close()
    rc, oT, oidx, oinner = oracle_in_device_order(one, 3, dst, src, O.transform_identity(), 3)
    Tq, idxq, innerq = one.estimate(src, I.Transform(), 3, return_info=True)
    assert rc == O.OK and np.array_equal(Tq.as_array(), oT.as_array()) and np.array_equal(idxq, oidx)
    one.close()


@pytest.mark.parametrize("world,n", [(2, 2 * 1024 * 1024), (8, 8 * 1000 * 1000)])
def test_weak_scaling_sizes_go_through_the_pipeline(world, n):
    """world x 1M source points against a 1M target (BASELINE configs[3] scaled weakly): the tree has world x 256 blocks,
    every rank files and finishes its own 256 -- pose, indices, inner counts of ONE handle on the whole cloud"""
    src, dst = synth.synthetic_pair(n, 1_000_000)
    one = I.Icp3d(dst)
    T1, idx1, inner1 = one.estimate(src, I.Transform(), 8, return_info=True)
    one.close()
    multi = I.IcpMulti(dst, [0] * world)
    T, idx, inner = multi.estimate(src, I.Transform(), 8, return_info=True)
    assert np.array_equal(inner, inner1), (inner, inner1)
    assert np.array_equal(T.as_array(), T1.as_array()) and np.array_equal(idx, idx1)
    print("pipelined iterations:", multi.pipe_iterations(), "sharded / replicated:", multi.counters())
    multi.close()


def test_a_converging_pair_hands_back_and_comes_back():
    """inner loops of many updates, then of none: the pipeline must stay out of the way (same bits), whatever it serves"""
    src, dst, _ = synth.converging_pair(200_000, 200_000)
    one = I.Icp3d(dst)
    T1, idx1, inner1 = one.estimate(src, I.Transform(), 12, return_info=True)
    multi = I.IcpMulti(dst, [0] * 4)
    T, idx, inner = multi.estimate(src, I.Transform(), 12, return_info=True)
    assert np.array_equal(inner, inner1), (inner, inner1)
    assert np.array_equal(T.as_array(), T1.as_array()) and np.array_equal(idx, idx1)
    multi.close()
    one.close()
