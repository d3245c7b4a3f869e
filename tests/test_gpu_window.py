"""The three-launch "window" evaluation (gn_win.hip): it predicts the median and sigma of the
residuals from the previous evaluation of the same handle and must return the same bits as every
other pipeline -- on a hit, on a miss (repeat with the seven-launch pipeline), with the widened
windows after a miss, and when duplicates overload a fine bin.  The path counters of the C ABI
(icp_gn_path_counters) prove that each case took the path it is meant to test.

The checker is the oracle's tree variant (same association order): bit-exact comparisons."""
import numpy as np
import pytest

import icp_rust_amd as I
import oracle_ffi as O
from parity_util import oracle_in_device_order

pytestmark = pytest.mark.gpu


def opose(T):
    return O.Pose(*[float(x) for x in T.as_array()])


def pairs(n, seed, shift=(0.0, 0.0), spread=0.05, outliers=True):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(n, 2)) * 20
    Tt = O.transform_new(np.array([0.4, -0.3, 0.02]))
    b = O.transform_apply_many(Tt, a) + rng.normal(size=(n, 2)) * spread + np.asarray(shift)
    if outliers:
        k = rng.integers(0, n, size=n // 10)
        b[k] += rng.normal(size=(len(k), 2)) * 5
    return a, b


def check(T, a, b):
    got = I.weighted_gauss_newton_update(T, a, b)
    blocks, threads = I.reduce_geometry(len(a))
    rc, want, _ = O.weighted_gauss_newton_update_tree(opose(T), a, b, blocks, threads)
    assert rc == O.OK and got is not None
    assert np.array_equal(got, want), (got, want)


def delta(before, after):
    return tuple(y - x for x, y in zip(before, after))


@pytest.mark.parametrize("n", [40_000, 262_144, 1_000_003])
def test_window_hit_is_bit_exact(n):
    a, b = pairs(n, n)
    T = I.Transform([0.39, -0.31, 0.0199])
    check(T, a, b)  # whatever ran, the next evaluation has a prediction
    c0 = I.gn_path_counters()
    # a slightly different pose: the residuals move by ~1e-2 sigma, as between inner iterations
    for k in range(3):
        check(I.Transform([0.39 + 1e-4 * k, -0.31, 0.0199 + 1e-6 * k]), a, b)
    tried, missed, short, radix, _, _ = delta(c0, I.gn_path_counters())
    assert tried == 3 and missed == 0 and short == 0 and radix == 0


def test_window_miss_falls_back_and_recentres():
    n = 200_000
    a, b = pairs(n, 11)
    T = I.Transform([0.39, -0.31, 0.0199])
    check(T, a, b)
    c0 = I.gn_path_counters()
    # the whole residual distribution jumps by ~20 sigma: every order statistic leaves its window
    a2, b2 = pairs(n, 11, shift=(1.5, -2.0))
    check(T, a2, b2)
    tried, missed, short, radix, _, _ = delta(c0, I.gn_path_counters())
    assert tried == 1 and missed == 1 and short == 1 and radix == 0
    # the repeat re-centred the prediction: the next evaluations hit again (wide windows first)
    c1 = I.gn_path_counters()
    check(I.Transform([0.3901, -0.31, 0.0199]), a2, b2)
    check(I.Transform([0.3902, -0.31, 0.0199]), a2, b2)
    check(I.Transform([0.3903, -0.31, 0.0199]), a2, b2)
    tried, missed, short, radix, _, _ = delta(c1, I.gn_path_counters())
    assert tried == 3 and missed == 0


def test_window_with_changing_scale():
    """sigma doubles between evaluations: the MAD leaves its fine windows -> miss, repeat, then hits."""
    n = 150_000
    a, b = pairs(n, 5, spread=0.05, outliers=False)
    T = I.Transform([0.4, -0.3, 0.02])
    check(T, a, b)
    a2, b2 = pairs(n, 5, spread=0.11, outliers=False)
    c0 = I.gn_path_counters()
    check(T, a2, b2)
    check(T, a2, b2)
    tried, missed, _, _, _, _ = delta(c0, I.gn_path_counters())
    assert tried == 2 and missed == 1


def test_window_duplicates_overload_a_bin():
    """30 % of the x residuals are one exact value on the median: more candidates in the median
    bin than the window pipeline keeps -> it reports a miss; the seven-launch pipeline overflows
    too and the radix path serves the evaluation.  Same bits."""
    n = 200_001
    rng = np.random.default_rng(n)
    a = rng.normal(size=(n, 2)) * 10
    r = rng.normal(size=(n, 2)) * 0.2
    T = I.Transform()
    check(T, a, a - r)  # prediction from a clean distribution of the same scale
    k = int(0.3 * n)
    r[:k, 0] = 0.0
    r[k:k + (n - k) // 2, 0] = -np.abs(r[k:k + (n - k) // 2, 0]) - 1e-3
    r[k + (n - k) // 2:, 0] = np.abs(r[k + (n - k) // 2:, 0]) + 1e-3
    c0 = I.gn_path_counters()
    check(T, a, a - r)
    tried, missed, short, radix, _, _ = delta(c0, I.gn_path_counters())
    assert tried == 1 and missed == 1 and radix == 1
    # and the pipelines are back in their rest state afterwards
    a2, b2 = pairs(n, 3)
    check(I.Transform([0.39, -0.31, 0.0199]), a2, b2)
    check(I.Transform([0.39, -0.31, 0.0199]), a2, b2)


def test_window_even_and_odd_counts_and_signed_zero_medians():
    for n in (65_536, 65_537):
        rng = np.random.default_rng(n)
        a = rng.normal(size=(n, 2)) * 10
        r = rng.normal(size=(n, 2)) * 0.1
        # a run of +-0 sits exactly on the median of y, the rest splits evenly around it
        z = 301
        h = (n - z) // 2
        r[:z, 1] = np.where(np.arange(z) % 2 == 0, 0.0, -0.0)
        r[z:z + h, 1] = -np.abs(r[z:z + h, 1]) - 1e-6
        r[z + h:, 1] = np.abs(r[z + h:, 1]) + 1e-6
        T = I.Transform()
        c0 = I.gn_path_counters()
        check(T, a, a - r)
        check(T, a, a - r)
        check(T, a, a - r)
        tried, missed, _, _, _, _ = delta(c0, I.gn_path_counters())
        assert tried >= 2 and tried - missed >= 1  # at least one evaluation was served by the window pipeline


def test_estimate_uses_the_window_pipeline_and_stays_bit_exact():
    from icp_rust_amd import synth
    n = m = 120_000
    src, dst = synth.synthetic_pair(n, m)
    icp = I.Icp3d(dst)
    T, idx, inner = icp.estimate(src, I.Transform(), 8, return_info=True)
    tried, missed, short, radix, spec_hit, spec_miss = I.gn_path_counters(icp)
    assert tried > 8 and missed <= tried // 2
    launches, served, _ = I.gn_loop_counters(icp)
    # an outer iteration either runs its inner loop as one launch (gn_loop.hip) or, after an inner loop of exactly one
    # update, bets on the next pose (icp_estimate_device)
    assert launches + spec_hit + spec_miss >= 7
    rc, oT, oidx, oinner = oracle_in_device_order(icp, 3, dst, src, O.transform_identity(), 8)
    assert rc == O.OK
    assert np.array_equal(idx, oidx)
    assert np.array_equal(inner, oinner)
    assert np.array_equal(T.as_array(), oT.as_array())


def test_inner_loops_of_varying_length_on_the_device_leave_the_result_alone():
    """A re-observed cloud in millimetres (synth.converging_pair): inner loops of tens of updates at first, none at
    the end.  Every inner loop that did not follow a one-update loop is ONE launch (gn_loop.hip) whatever its length;
    indices, inner counts and pose equal the oracle's, bit for bit."""
    from icp_rust_amd import synth
    n = m = 60_000
    src, dst = synth.converging_pair(n, m)[:2]
    icp = I.Icp3d(dst)
    l0 = I.gn_loop_counters(icp)
    T, idx, inner = icp.estimate(src, I.Transform(), 12, return_info=True)
    launches, served, handbacks = delta(l0, I.gn_loop_counters(icp))
    assert max(int(x) for x in inner) >= 5, inner
    # (one launch per inner loop that did not follow a one-update loop, plus one per evaluation handed back; with the
    # widest windows for clouds of this size -- round 5 -- nothing is handed back here, and the loops after the pose has
    # stopped moving are not run at all: the count of loops with several updates is the floor)
    several = sum(1 for x in inner if int(x) >= 2)
    assert launches >= max(several, 4) and served >= launches and handbacks <= launches // 2, (launches, served, handbacks, inner)
    rc, oT, oidx, oinner = oracle_in_device_order(icp, 3, dst, src, O.transform_identity(), 12)
    assert rc == O.OK
    assert np.array_equal(idx, oidx)
    assert np.array_equal(inner, oinner)
    assert np.array_equal(T.as_array(), oT.as_array())


def test_speculative_search_hits_and_misses_leave_the_result_alone():
    """A small cloud whose inner loops mostly need one update: the bet on the next pose is sometimes wrong, the
    discarded search must not leak into the result (indices, inner counts and pose equal the oracle's, bit for bit)."""
    from icp_rust_amd import synth
    n, m = 28_000, 28_000
    src, dst = synth.synthetic_pair(n, m)
    icp = I.Icp3d(dst)
    T, idx, inner = icp.estimate(src, I.Transform(), 12, return_info=True)
    _, _, _, _, hit, miss = I.gn_path_counters(icp)
    assert hit + miss + I.gn_loop_counters(icp)[0] > 0
    rc, oT, oidx, oinner = oracle_in_device_order(icp, 3, dst, src, O.transform_identity(), 12)
    assert rc == O.OK
    assert np.array_equal(idx, oidx)
    assert np.array_equal(inner, oinner)
    assert np.array_equal(T.as_array(), oT.as_array())


def test_two_stream_estimate_equals_single_stream_stage_calls_repeatedly():
    """icp_estimate_device runs every inner loop as one launch whose workgroups hand histograms, candidates
    and block sums to each other across grid barriers; the stage calls run the same iteration launch by
    launch.  Both must agree bit for bit, every time (a hand-over that leaves state in an XCD's L2 shows
    up here as an occasional mismatch)."""
    import torch

    from icp_rust_amd import synth
    from icp_rust_amd.dist import HipStages, ShardedIcp

    n = m = 150_000
    src, dst = synth.synthetic_pair(n, m)
    d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
    staged = I.Icp3d(d_dst)
    # (the fused call folds its sums over the cell-sorted cloud, icp_last_fold_order: the stage calls get that
    # cloud as their source -- sorting it again is the identity)
    d_sorted, _ = staged.sort_source_device(d_src, I.Transform())
    T_ref, inner_ref = ShardedIcp(HipStages(staged), n).estimate(d_sorted, I.Transform(), 10)
    fused = I.Icp3d(d_dst)
    for rep in range(12):
        T, inner = fused.estimate(d_src, I.Transform(), 10, return_info="inner")
        assert np.array_equal(T.as_array(), T_ref.as_array()), rep
        assert inner.tolist() == inner_ref.tolist(), rep
    assert I.gn_loop_counters(fused)[0] + sum(I.gn_path_counters(fused)[4:]) > 0


def test_refined_windows_beyond_4m_points():
    """Above 4M pairs a window that survives a prediction error would overflow the candidate lists:
    the window is found in two passes instead (sample -> first histogram pass -> second pass), and
    resolved by the usual compaction / accumulate launches.  Same bits as the oracle's tree; several
    distributions, one of them heavy-tailed with a far-off median."""
    n = 4_500_000
    T = I.Transform([0.39, -0.31, 0.0199])
    for seed, kw in ((99, {}), (7, dict(shift=(3.0, -1.0), spread=0.5)), (8, dict(spread=1e-3, outliers=False))):
        a, b = pairs(n, seed, **kw)
        c0 = I.gn_path_counters()
        check(T, a, b)
        tried, missed, short, radix, _, _ = delta(c0, I.gn_path_counters())
        assert tried == 1 and missed == 0 and short == 0 and radix == 0, (seed, tried, missed, short, radix)


def test_refined_windows_fall_back_on_degenerate_residuals():
    """4.5M pairs whose residuals are almost all exactly equal (quantised noise: a few thousand
    distinct values, heavy runs of duplicates at the median): whatever the sample and the two passes
    make of it, the evaluation must end with the oracle's bits -- through the refined windows or
    through the pipelines they fall back to."""
    n = 4_500_000
    rng = np.random.default_rng(5)
    a = rng.normal(size=(n, 2)) * 20
    q = np.round(rng.normal(size=(n, 2)) * 8) / 64.0  # steps of 1/64: exact in binary
    b = a + q
    T = I.Transform()  # identity: the residuals are -q exactly
    c0 = I.gn_path_counters()
    check(T, a, b)
    tried, missed, short, radix, _, _ = delta(c0, I.gn_path_counters())
    assert tried + short + radix >= 1


def test_three_digit_radix_pipeline_beyond_4m_points():
    """With the refined windows switched off (ICP_GN_NO_REFINE, read once per process: a child
    process) the radix pipeline serves alone beyond 4M pairs and takes a third 12-bit digit per stage
    (candidate lists would overflow after two); it is also what a failed refinement falls back to."""
    import os
    import subprocess
    import sys

    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import icp_rust_amd as I\n"
        "import test_gpu_window as W\n"
        "a, b = W.pairs(4_500_000, 99)\n"
        "c0 = I.gn_path_counters()\n"
        "W.check(I.Transform([0.39, -0.31, 0.0199]), a, b)\n"
        "tried, missed, short, radix, _, _ = W.delta(c0, I.gn_path_counters())\n"
        "assert tried == 0 and short == 1 and radix == 0, (tried, short, radix)\n"
        "print('radix-ok')\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ICP_GN_NO_REFINE="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "radix-ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_pooled_handles_start_clean():
    """icp_destroy parks a handle's buffers for the next icp_create.  Whatever the previous owner
    did -- another size, another dimension, a window prediction, previous matches -- the next
    handle must behave like a fresh one (bit-equal to the oracle), before and after icp_trim_pool."""
    from icp_rust_amd import synth
    from icp_rust_amd.scans import load_scan2d
    import os

    def check3d(n, m, seed, iters):
        src, dst = synth.synthetic_pair(n, m, seed=synth.SEED + seed)
        icp = I.Icp3d(dst)
        T, idx, inner = icp.estimate(src, I.Transform(), iters, return_info=True)
        rc, oT, oidx, oinner = oracle_in_device_order(icp, 3, dst, src, O.transform_identity(), iters)
        icp.close()
        assert rc == O.OK
        assert np.array_equal(idx, oidx) and np.array_equal(inner, oinner)
        assert np.array_equal(T.as_array(), oT.as_array())

    def check2d():
        d = os.path.join(os.path.dirname(__file__), "golden", "scans2d")
        src, dst = load_scan2d(os.path.join(d, "001.txt")), load_scan2d(os.path.join(d, "002.txt"))
        icp = I.Icp2d(dst)
        T, idx, inner = icp.estimate(src, I.Transform(), 6, return_info=True)
        rc, oT, oidx, oinner = oracle_in_device_order(icp, 2, dst, src, O.transform_identity(), 6)
        icp.close()
        assert rc == O.OK
        assert np.array_equal(inner, oinner)
        assert np.array_equal(T.as_array(), oT.as_array())

    check3d(120_000, 90_000, 1, 5)   # leaves a big pooled handle with predictions and matches
    check2d()                         # tiny 2-D problem on the pooled buffers
    check3d(40_000, 150_000, 2, 4)    # more targets than the pooled grid held
    check3d(70_000, 20_000, 3, 4)
    I.lib().icp_trim_pool()
    check3d(50_000, 50_000, 4, 3)


def test_a_pooled_handle_starts_from_its_previous_owners_predictions_and_changes_nothing():
    """examples/scan3d.rs builds an Icp3d per frame: a handle taken from the pool adopts the window predictions of its
    previous owner for its first evaluations (common.hpp: hint_kind) -- fewer evaluations through the 7-launch
    pipeline, the same bits (a window is verified by exact counts whatever it was centred on)."""
    from icp_rust_amd import synth

    I.lib().icp_trim_pool()
    pk = synth.synthetic_scan3d_packets(75 * 4)
    frames = [synth.remove_invalid_values(pk[75 * k:75 * (k + 1)]) for k in range(4)]
    pulls = []
    for k in range(3):
        icp = I.Icp3d(frames[k])
        T, idx, inner = icp.estimate(frames[k + 1], I.Transform(), 6, return_info=True)
        rc, oT, oidx, oinner = oracle_in_device_order(icp, 3, frames[k], frames[k + 1], O.transform_identity(), 6)
        pulls.append(I.gn_path_counters(icp)[2])
        icp.close()
        assert rc == O.OK
        assert np.array_equal(idx, oidx) and np.array_equal(inner, oinner)
        assert np.array_equal(T.as_array(), oT.as_array())
    assert pulls[0] >= 2            # a handle without history: the first two kinds of evaluation have no prediction
    assert max(pulls[1:]) < pulls[0]  # its successors from the pool start from its predictions


def test_a_pooled_handle_changes_dimension():
    """regression (found by profiles/extended_fuzz.py as a device memory fault): the cell-sorted copy
    of the source cloud was sized n * dim doubles but its capacity remembered as n points, so a 3-D
    handle that took over a 2-D handle's buffers from the pool wrote past their end"""
    from icp_rust_amd import synth

    I.lib().icp_trim_pool()
    rng = np.random.default_rng(4)
    n, m = 30_000, 20_000
    for dim in (2, 3, 2, 3):
        dst = rng.normal(size=(m, dim)) * 10
        src = dst[rng.integers(0, m, size=n)] + rng.normal(size=(n, dim)) * 0.05
        icp = (I.Icp3d if dim == 3 else I.Icp2d)(dst)
        T, idx, inner = icp.estimate(src, I.Transform([0.05, -0.03, 0.01]), 3, return_info=True)
        rc, oT, oidx, oinner = oracle_in_device_order(icp, dim, dst, src, opose(I.Transform([0.05, -0.03, 0.01])), 3)
        icp.close()  # parked: the next handle, of the other dimension, reuses its buffers
        assert rc == O.OK
        assert np.array_equal(idx, oidx) and np.array_equal(inner, oinner)
        assert np.array_equal(T.as_array(), oT.as_array())
