"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bars (BASELINE.json north_star): correspondence indices bit-exact; order statistics
(median / MAD / sigma) bit-exact; the pose within 1e-5 relative of the oracle's
reference-order (left fold) result, and bit-exact against the oracle evaluated in the
device's documented reduction order.
"""
import os

import numpy as np
import pytest

import icp_rust_amd as I
import oracle_ffi as O
from icp_rust_amd import _lib, synth
from icp_rust_amd.scans import load_scan2d
from parity_util import oracle_in_device_order

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
POSE_RTOL = 1e-5  # north_star: "final SE(2)/SE(3) pose within 1e-5 relative"


def opose(T):
    return O.Pose(*T.pose.as_tuple())


def assert_pose_close(T, oT, rtol=POSE_RTOL):
    a, b = T.as_array(), oT.as_array()
    scale = max(1.0, float(np.max(np.abs(b))))
    assert np.max(np.abs(a - b)) <= rtol * scale, (a, b)


# ------------------------------------------------------------------ nearest neighbour --


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("n,m", [(1, 1), (7, 3), (1000, 1023), (1000, 1025), (5000, 7001), (3000, 50000)])
def test_nn_indices_bit_exact_vs_brute_force(dim, n, m):
    rng = np.random.default_rng(100 * dim + n + m)
    dst = rng.normal(size=(m, dim)) * 10
    q = rng.normal(size=(n, dim)) * 10
    icp = (I.Icp3d if dim == 3 else I.Icp2d)(dst)
    got = icp.nn_search(q)
    rc, want = O.nn_brute(dst, q)
    assert rc == O.OK
    assert np.array_equal(got, want)


@pytest.mark.parametrize("dim", [2, 3])
def test_nn_ties_resolve_to_lowest_index(dim):
    rng = np.random.default_rng(5)
    dst = rng.integers(-8, 8, size=(6000, dim)).astype(np.float64)  # duplicates + exact ties
    q = rng.integers(-9, 9, size=(4000, dim)).astype(np.float64) + 0.5 * rng.integers(0, 2, size=(4000, dim))
    icp = (I.Icp3d if dim == 3 else I.Icp2d)(dst)
    got = icp.nn_search(q)
    _, want = O.nn_brute(dst, q)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n", [300_000, 600_000, 1_100_000])
def test_nn_large_query_counts_vs_kdtree(n):
    """The sweep's 2 / 4 / 8 queries-per-lane instantiations (n >= 256k / 512k / 900k source points,
    nn_brute.hip:launch_nn_brute), FORCED to the sweep: with 6 000 targets AUTO would stay on it, with
    20 000 it resolves to the grid.  The oracle kd-tree equals its brute force
    (tests/test_oracle_kat.py::test_kdtree_equals_brute_force_including_ties)."""
    src, dst = synth.synthetic_pair(n, 6_000)
    got = _nn(dst, src, I.NN_BRUTE)
    rc, want = O.KdTree(dst).search(src)
    assert rc == O.OK
    assert np.array_equal(got, want)


def test_brute_force_nn_full_size_vs_kdtree():
    """BASELINE configs[2] as it is worded ("brute-force NN") at its own size: the LDS-tiled sweep
    (k_nn_brute_dot<3, 8>) on the 1M x 1M synthetic pair, every index against the oracle
    (search semantics: /root/reference/src/lib.rs:161-167)."""
    src, dst = synth.synthetic_pair(1_000_000, 1_000_000)
    got = _nn(dst, src, I.NN_BRUTE)
    rc, want = O.KdTree(dst).search(src)
    assert rc == O.OK
    assert np.array_equal(got, want)


def test_nn_self_query_is_identity_at_full_size():
    """size-independent property at BASELINE's 1M x 1M: every target is its own nearest
    neighbour (duplicates -> the lowest index of the duplicate group)."""
    import torch

    _, dst = synth.synthetic_pair(1, 1_000_000)
    icp = I.Icp3d(dst)
    d = torch.from_numpy(dst).cuda()
    idx = torch.empty(dst.shape[0], dtype=torch.int32, device="cuda")
    icp.nn_search_device(d, idx)
    icp.synchronize()
    got = idx.cpu().numpy().view(np.uint32)
    ar = np.arange(dst.shape[0], dtype=np.uint32)
    bad = np.nonzero(got != ar)[0]
    # any mismatch must be an exact duplicate with a lower index
    for i in bad:
        assert got[i] < i and np.array_equal(dst[got[i]], dst[i])
    assert len(bad) < 100


# --------------------------------------------------------------- medians / MAD / sigma --


def make_pairs(n, seed, dup=False, outliers=True):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(n, 2)) * 20
    Tt = O.transform_new(np.array([0.4, -0.3, 0.02]))
    b = O.transform_apply_many(Tt, a) + rng.normal(size=(n, 2)) * 0.05 if n else a.copy()
    if outliers and n > 10:
        k = rng.integers(0, n, size=n // 10)
        b[k] += rng.normal(size=(len(k), 2)) * 5
    if dup and n > 4:
        b[: n // 2] = O.transform_apply_many(O.transform_identity(), a[: n // 2])  # exact zeros
    return a, b


@pytest.mark.parametrize("n", [1, 2, 3, 4, 19, 20, 255, 256, 257, 1000, 4097, 100_001, 250_000])
@pytest.mark.parametrize("dup", [False, True])
def test_residual_stddevs_bit_exact(n, dup):
    a, b = make_pairs(n, n, dup=dup)
    T = I.Transform([0.01, -0.02, 0.001])
    got = I.residual_stddevs(T, a, b)
    res = np.array([O.residual(opose(T), s, d) for s, d in zip(a, b)]) if n <= 5000 else None
    if res is None:
        p = T.pose
        res = np.stack([(p.r00 * a[:, 0] + p.r01 * a[:, 1]) + p.tx - b[:, 0],
                        (p.r10 * a[:, 0] + p.r11 * a[:, 1]) + p.ty - b[:, 1]], axis=1)
    rc, want = O.calc_stddevs(res)
    assert rc == O.OK
    assert np.array_equal(got, want), (got, want)


def test_stddevs_heavy_duplicates_and_signed_zeros():
    a = np.zeros((1001, 2))
    b = np.zeros((1001, 2))
    b[::3, 0] = 1.0
    b[1::3, 1] = -2.0
    T = I.Transform()
    got = I.residual_stddevs(T, a, b)
    _, want = O.calc_stddevs(a - b)
    assert np.array_equal(got, want)


# ----------------------------------------------------------- weighted GN / estimator --


@pytest.mark.parametrize("n", [3, 19, 1000, 65_537, 300_000])
def test_weighted_gn_update_bit_exact_vs_tree_oracle(n):
    a, b = make_pairs(n, 7 * n + 1)
    T = I.Transform([0.02, 0.01, -0.003])
    got = I.weighted_gauss_newton_update(T, a, b)
    blocks, threads = I.reduce_geometry(n)
    rc, want, _ = O.weighted_gauss_newton_update_tree(opose(T), a, b, blocks, threads)
    assert rc == O.OK and got is not None
    assert np.array_equal(got, want), (got, want)
    # and within rounding of the reference's left fold
    rc, seq = O.weighted_gauss_newton_update(opose(T), a, b)
    assert np.allclose(got, seq, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("n", [19, 2000, 120_000])
def test_estimate_transform_matches_oracle(n):
    a, b = make_pairs(n, 3 * n + 5)
    got, inner = I.estimate_transform(a, b, return_inner_iters=True)
    want, want_inner = O.estimate_transform(a, b)
    assert_pose_close(got, want)
    assert inner == want_inner
    # same summation order => same bits
    blocks, threads = I.reduce_geometry(n)
    aa, bb = np.ascontiguousarray(a), np.ascontiguousarray(b)
    # replay src/lib.rs:59-84 with the oracle's tree-order update
    T = O.transform_identity()
    prev = np.finfo(np.float64).max
    applied = 0
    for _ in range(200):
        rc, d, err = O.weighted_gauss_newton_update_tree(T, aa, bb, blocks, threads)
        if rc != O.OK:
            break
        if (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2] < 1e-6:
            break
        if err > prev:
            break
        prev = err
        T = O.transform_mul(O.transform_new(d), T)
        applied += 1
    assert applied == inner
    assert np.array_equal(got.as_array(), T.as_array())


def test_plain_sums_match_oracle():
    a, b = make_pairs(5000, 42)
    T = I.Transform([0.3, 0.1, 0.01])
    assert np.isclose(I.error(T, a, b), O.error(opose(T), a, b), rtol=1e-12)
    assert np.isclose(I.huber_error(T, a, b), O.huber_error(opose(T), a, b), rtol=1e-12)
    got = I.gauss_newton_update(T, a, b)
    rc, want = O.gauss_newton_update(opose(T), a, b)
    assert rc == O.OK
    assert np.allclose(got, want, rtol=1e-9, atol=1e-12)


# ------------------------------------------------------------------------- ICP driver --


def icp_vs_oracle(dim, dst, src, init, max_iter, kd=True):
    icp = (I.Icp3d if dim == 3 else I.Icp2d)(dst)
    got, idx, inner = icp.estimate(src, init, max_iter, return_info=True)
    n = len(src)
    rc, want, oidx, oinner = O.icp_estimate(dim, dst, src, opose(init), max_iter, use_kdtree=kd)
    assert rc == O.OK
    rc, want_t, oidx_t, oinner_t = oracle_in_device_order(icp, dim, dst, src, opose(init), max_iter, use_kdtree=kd)
    assert rc == O.OK
    # bit-exact against the oracle in the device's summation order
    assert np.array_equal(got.as_array(), want_t.as_array())
    assert np.array_equal(idx, oidx_t)
    assert np.array_equal(inner, oinner_t)
    # and within the north_star tolerance of the reference-order oracle
    assert_pose_close(got, want)
    return got, idx, inner, (want, oidx, oinner)


def test_icp2d_on_reference_scans():
    src = load_scan2d(os.path.join(GOLDEN, "scans2d", "001.txt"))
    T = I.Transform()
    for k in (2, 3, 4, 5):  # examples/scan2d.rs:62-90: warm start from the previous frame
        dst = load_scan2d(os.path.join(GOLDEN, "scans2d", f"{k:03d}.txt"))
        T, idx, inner, (want, oidx, _) = icp_vs_oracle(2, dst, src, T, 20, kd=True)
        # indices equal the reference-order oracle's too, modulo identical-coordinate duplicates
        diff = np.nonzero(idx != oidx)[0]
        assert all(np.array_equal(dst[idx[i]], dst[oidx[i]]) for i in diff)


def test_icp3d_synthetic_config2_size():
    pk = synth.synthetic_scan3d_packets(150)
    src = synth.remove_invalid_values(pk[:75])
    dst = synth.remove_invalid_values(pk[75:150])
    icp_vs_oracle(3, dst, src, I.Transform(), 5, kd=True)


def test_icp3d_synthetic_box_200k():
    src, dst = synth.synthetic_pair(200_000, 150_000)
    got, idx, inner, _ = icp_vs_oracle(3, dst, src, I.Transform(), 3, kd=True)
    assert inner.sum() > 0


@pytest.mark.parametrize("n,m", [(90_000, 70_000), (30_000, 30_000)])
def test_every_iteration_count_around_the_run_ahead_search_matches_the_oracle(n, m):
    """icp_estimate_device enqueues the search of iteration k + 2 behind the pre-launched first evaluation of
    iteration k + 1 (pose derived on the device) and takes its pairs only if the host derives the same bits; which
    searches may run ahead, and which of them writes the caller's indices, depends on max_iter - it: one call per
    count, on a handle whose previous call ended with one update per iteration (so that the first iteration bets),
    for the one-lane-per-query search (90k points) and the four-lanes one (30k); bit-equal to the device-order oracle"""
    src, dst = synth.synthetic_pair(n, m)
    icp = I.Icp3d(dst)
    icp.estimate(src, I.Transform(), 6)
    for max_iter in (1, 2, 3, 4, 5, 7):
        got, idx, inner = icp.estimate(src, I.Transform(), max_iter, return_info=True)
        rc, want, oidx, oinner = oracle_in_device_order(icp, 3, dst, src, opose(I.Transform()), max_iter, use_kdtree=True)
        assert rc == O.OK
        assert np.array_equal(got.as_array(), want.as_array()), max_iter
        assert np.array_equal(idx, oidx), max_iter
        assert np.array_equal(inner, oinner), max_iter
    hits, misses = I.run_ahead_counters(icp)
    assert hits > 0, (hits, misses)


@pytest.mark.parametrize("case", ["a cloud against itself", "a registration that settles", "a 2-D scan pair in one launch"])
def test_a_pose_that_has_stopped_moving_ends_the_loop_with_the_bits_of_all_twenty_iterations(case):
    """An outer iteration that leaves the pose as it found it is a fixed point of src/lib.rs:105-130 / 148-173: the
    iterations after it repeat it.  icp_estimate[_device] and the single-launch kernel run only the last of them (the one
    that reports the correspondences); pose, indices and every inner count must be what the oracle gets by running all
    twenty -- and the counter must show that iterations were in fact left out where the host steps the loop."""
    if case == "a cloud against itself":  # the first frame of examples/scan3d.rs:104-131: every residual is exactly 0
        pk = synth.synthetic_scan3d_packets(75)
        src = dst = synth.remove_invalid_values(pk)
        dim, max_iter = 3, 20
    elif case == "a registration that settles":
        src, dst, _ = synth.converging_pair(100_000, 100_000)
        dim, max_iter = 3, 20
    else:
        src = load_scan2d(os.path.join(GOLDEN, "scans2d", "001.txt"))
        dst = load_scan2d(os.path.join(GOLDEN, "scans2d", "002.txt"))
        dim, max_iter = 2, 20
    icp = (I.Icp3d if dim == 3 else I.Icp2d)(dst)
    got, idx, inner = icp.estimate(src, I.Transform(), max_iter, return_info=True)
    rc, want, oidx, oinner = oracle_in_device_order(icp, dim, dst, src, opose(I.Transform()), max_iter, use_kdtree=True)
    assert rc == O.OK
    assert np.array_equal(got.as_array(), want.as_array())
    assert np.array_equal(idx, oidx)
    assert np.array_equal(inner, oinner), (inner, oinner)
    assert inner[-1] == 0 and inner[-3] == 0, inner  # (the case does reach a fixed point before the end)
    if dim == 3:
        assert I.fixed_point_skips(icp) > 0
    # the same call again, shorter and longer: the fixed point may come before, at or after the last iteration
    for k in (1, 2, int(np.argmin(inner > 0)) + 1, int(np.argmin(inner > 0)) + 2, max_iter + 5):
        g2, i2, n2 = icp.estimate(src, I.Transform(), k, return_info=True)
        rc, w2, oi2, on2 = oracle_in_device_order(icp, dim, dst, src, opose(I.Transform()), k, use_kdtree=True)
        assert rc == O.OK and np.array_equal(g2.as_array(), w2.as_array()), k
        assert np.array_equal(i2, oi2) and np.array_equal(n2, on2), (k, n2, on2)


def test_icp_is_run_to_run_deterministic():
    src, dst = synth.synthetic_pair(50_000, 40_000)
    icp = I.Icp3d(dst)
    a = icp.estimate(src, I.Transform(), 4, return_info=True)
    b = icp.estimate(src, I.Transform(), 4, return_info=True)
    assert np.array_equal(a[0].as_array(), b[0].as_array())
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_icp_recovers_known_pose_roundtrip():
    """size-independent property: dst = T_true (.) src exactly => the estimate maps src onto dst."""
    src, _ = synth.synthetic_pair(120_000, 1)
    Tt = I.Transform(synth.TRUTH_PARAM)
    p = Tt.pose
    dst = src.copy()
    dst[:, 0] = (p.r00 * src[:, 0] + p.r01 * src[:, 1]) + p.tx
    dst[:, 1] = (p.r10 * src[:, 0] + p.r11 * src[:, 1]) + p.ty
    icp = I.Icp3d(dst)
    got = icp.estimate(src, I.Transform(), 20)
    # the inner loop never applies an update with |delta|^2 < 1e-6 (src/lib.rs:71-73), so the
    # fixed point sits within ~1e-3 of the truth, not on it
    assert np.allclose(got.as_array(), Tt.as_array(), atol=2e-3)
    again = icp.estimate(src, got, 5)
    assert np.allclose(again.as_array(), got.as_array(), atol=2e-3)


def test_device_resident_inputs_equal_host_inputs():
    import torch

    src, dst = synth.synthetic_pair(30_000, 30_000)
    a = I.Icp3d(dst).estimate(src, I.Transform(), 3, return_info=True)
    d_dst = torch.from_numpy(dst).cuda()
    d_src = torch.from_numpy(src).cuda()
    b = I.Icp3d(d_dst).estimate(d_src, I.Transform(), 3, return_info=True)
    assert np.array_equal(a[0].as_array(), b[0].as_array())
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


# ----------------------------------------------------------------------- failure modes --


def test_empty_dst_is_reported():  # the reference panics, src/lib.rs:122,165
    icp = I.Icp2d(np.zeros((0, 2)))
    with pytest.raises(I.IcpError) as e:
        icp.estimate(np.array([[1.0, 2.0]]), I.Transform(), 1)
    assert e.value.status == _lib.EMPTY_DST
    # no query -> no panic
    T = icp.estimate(np.zeros((0, 2)), I.Transform([1.0, 2.0, 0.1]), 3)
    assert np.array_equal(T.as_array(), I.Transform([1.0, 2.0, 0.1]).as_array())


def test_nan_residual_is_reported():  # the reference panics, src/stats.rs:12
    a, b = make_pairs(100, 1)
    b[17, 1] = np.nan
    with pytest.raises(I.IcpError) as e:
        I.weighted_gauss_newton_update(I.Transform(), a, b)
    assert e.value.status == _lib.NAN_INPUT
    with pytest.raises(I.IcpError):
        I.estimate_transform(a, b)


def test_zero_max_iter_returns_initial_transform():
    src, dst = synth.synthetic_pair(100, 100)
    T0 = I.Transform([0.1, 0.2, 0.3])
    T = I.Icp3d(dst).estimate(src, T0, 0)
    assert np.array_equal(T.as_array(), T0.as_array())


# ------------------------------------------------- exact grid NN == brute force == oracle --


def _nn(dst, q, mode):
    dim = dst.shape[1]
    icp = (I.Icp3d if dim == 3 else I.Icp2d)(dst, nn_mode=mode)
    assert I.lib().icp_get_nn_mode(icp._h) == mode
    return icp.nn_search(q)


def _grid_cases():
    rng = np.random.default_rng(2024)
    cases = {}
    for dim in (2, 3):
        cases[f"normal{dim}"] = (rng.normal(size=(20_000, dim)) * 10, rng.normal(size=(15_000, dim)) * 10)
        # queries far outside the target bounding box, on every side
        far = rng.normal(size=(4000, dim)) * 300
        cases[f"far{dim}"] = (rng.normal(size=(9000, dim)), far)
        # integer lattice: exact ties and duplicates everywhere
        cases[f"lattice{dim}"] = (rng.integers(-8, 8, size=(12_000, dim)).astype(np.float64),
                                  rng.integers(-20, 20, size=(6000, dim)).astype(np.float64) * 0.5)
        # strongly non-uniform density (two tight clusters + sparse background)
        cl = np.concatenate([rng.normal(size=(8000, dim)) * 0.01, rng.normal(size=(8000, dim)) * 0.01 + 50,
                             rng.uniform(-100, 100, size=(2000, dim))])
        cases[f"clusters{dim}"] = (cl, np.concatenate([rng.uniform(-120, 120, size=(5000, dim)),
                                                       rng.normal(size=(3000, dim)) * 0.02]))
        # all targets identical / a single target
        cases[f"same{dim}"] = (np.ones((300, dim)) * 3.25, rng.normal(size=(500, dim)))
        cases[f"single{dim}"] = (np.array([[1.0, 2.0, 3.0][:dim]]), rng.normal(size=(100, dim)))
    # 3-D data that is flat in z, and on a line
    flat = rng.normal(size=(10_000, 3)) * 5
    flat[:, 2] = 1.5
    cases["flat3"] = (flat, rng.normal(size=(4000, 3)) * 5)
    line = np.zeros((5000, 3))
    line[:, 0] = rng.uniform(-10, 10, size=5000)
    cases["line3"] = (line, rng.normal(size=(3000, 3)) * 4)
    return cases


@pytest.mark.parametrize("name", sorted(_grid_cases()))
def test_grid_nn_equals_brute_force_and_oracle(name):
    dst, q = _grid_cases()[name]
    g = _nn(dst, q, I.NN_GRID)
    b = _nn(dst, q, I.NN_BRUTE)
    assert np.array_equal(g, b)
    rc, want = O.KdTree(dst).search(q)
    assert rc == O.OK
    assert np.array_equal(g, want)


@pytest.mark.parametrize("name", ["far3", "lattice3", "clusters3", "lattice2", "line3"])
def test_grid_nn_both_lane_layouts_agree(name):
    """up to 65536 queries the grid search gives every query four lanes (they share the rows of its
    cell box), beyond that one: the same queries, repeated past that size, must get the same indices"""
    dst, q = _grid_cases()[name]
    reps = 65536 // len(q) + 1
    big = np.ascontiguousarray(np.tile(q, (reps, 1)))
    assert len(q) <= 65536 < len(big)
    icp = (I.Icp3d if dst.shape[1] == 3 else I.Icp2d)(dst, nn_mode=I.NN_GRID)
    small = icp.nn_search(q)
    assert np.array_equal(icp.nn_search(big), np.tile(small, reps))
    rc, want = O.KdTree(dst).search(q)
    assert rc == O.OK and np.array_equal(small, want)


@pytest.mark.parametrize("name", ["far3", "lattice3", "clusters3", "lattice2", "line3", "same3", "single3", "flat3", "far2"])
def test_seeded_first_search_and_the_warm_search_after_it_on_hostile_clouds(name):
    """beyond 65536 queries the first search of a source snapshot is a seed pass (some nearby target per
    query) followed by the warm kernel, and later searches start from the previous matches: on clouds
    with empty regions, ties everywhere, a single target, queries far outside the grid -- and under a pose
    that moves the queries between the two searches -- every index must equal the kd-tree oracle's."""
    import torch

    dst, q = _grid_cases()[name]
    dim = dst.shape[1]
    reps = 65536 // len(q) + 1
    big = np.ascontiguousarray(np.tile(q, (reps, 1)))
    assert len(big) > 65536
    icp = (I.Icp3d if dim == 3 else I.Icp2d)(dst, nn_mode=I.NN_GRID)
    d_q = torch.from_numpy(big).cuda()
    idx = torch.empty(len(big), dtype=torch.int32, device="cuda")
    a = torch.empty((len(big), 2), dtype=torch.float64, device="cuda")
    b = torch.empty_like(a)
    tree = O.KdTree(dst)
    poses = [I.Transform(), I.Transform(np.array([0.3, -0.2, 0.05])), I.Transform(np.array([-2.0, 1.0, -0.4]))]
    icp.prepare_source_device(d_q, poses[0])
    for T in poses:  # the first: seeds + warm; the others: warm from the previous matches
        icp.correspond_device(d_q, T, a, b, idx)
        icp.synchronize()
        p = T.pose
        moved = q.copy()
        moved[:, 0] = (p.r00 * q[:, 0] + p.r01 * q[:, 1]) + p.tx  # Transform::transform, src/transform.rs:22-24
        moved[:, 1] = (p.r10 * q[:, 0] + p.r11 * q[:, 1]) + p.ty
        rc, want = tree.search(moved)
        assert rc == O.OK
        got = idx.cpu().numpy().view(np.uint32)
        assert np.array_equal(got, np.tile(want, reps))
        assert np.array_equal(a.cpu().numpy(), np.tile(moved[:, :2], (reps, 1)))
        assert np.array_equal(b.cpu().numpy(), np.tile(dst[want][:, :2], (reps, 1)))


@pytest.mark.parametrize("dim,kind", [(3, "uniform"), (3, "lattice"), (2, "uniform"), (3, "mm")])
def test_certified_matches_equal_the_kdtree_along_a_chain_of_small_steps(dim, kind):
    """once a registration has settled, a search proves most previous matches still nearest instead of walking the
    grid again (nn_grid.hip: k_nn_cert): along a chain of poses -- tiny steps, a repeated pose, one long jump, tiny
    steps again -- every index of every search must equal the kd-tree oracle's, whatever the certificates skipped."""
    import torch

    rng = np.random.default_rng(77 + dim + len(kind))
    m, n = 150_000, 200_000
    scale = 1000.0 if kind == "mm" else 1.0
    if kind == "lattice":  # ties everywhere: a certificate may only pass where the match is the unique nearest
        side = int(round(m ** (1.0 / dim)))
        axes = [np.arange(side, dtype=np.float64)] * dim
        dst = np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1).reshape(-1, dim) * 0.5
        src = dst[rng.integers(0, len(dst), size=n)] + np.round(rng.normal(size=(n, dim)) * 2) * 0.125
    else:
        box = np.array([40.0, 40.0, 4.0][:dim]) * scale
        dst = rng.uniform(-1, 1, size=(m, dim)) * box
        src = dst[rng.integers(0, m, size=n)] + rng.normal(size=(n, dim)) * 0.05 * scale
    spacing = float(np.prod(2 * (dst.max(0) - dst.min(0)) / 2 + 1e-9) / len(dst)) ** (1.0 / dim)
    icp = (I.Icp3d if dim == 3 else I.Icp2d)(dst, nn_mode=I.NN_GRID)
    d_q = torch.from_numpy(np.ascontiguousarray(src)).cuda()
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    a = torch.empty((n, 2), dtype=torch.float64, device="cuda")
    b = torch.empty_like(a)
    tree = O.KdTree(dst)
    reach = float(np.abs(src[:, :2]).max())
    params = [np.zeros(3)]
    steps = [1e-4, 1e-3, 0.0, 2e-3, 1e-2, 1e-4, 30.0, 1e-3, 1e-4, 0.0, 5e-4]  # in NN spacings
    for st in steps:
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        step = st * spacing * d
        step[2] = step[2] / max(reach, 1.0)  # the rotation moves the farthest query by about as much
        params.append(params[-1] + step)
    icp.prepare_source_device(d_q, I.Transform(params[0]))
    failed = []
    for prm in params:
        T = I.Transform(prm)
        icp.correspond_device(d_q, T, a, b, idx)
        icp.synchronize()
        failed.append(I.nn_cert_counters(icp))
        p = T.pose
        moved = src.copy()
        moved[:, 0] = (p.r00 * src[:, 0] + p.r01 * src[:, 1]) + p.tx  # Transform::transform, src/transform.rs:22-24
        moved[:, 1] = (p.r10 * src[:, 0] + p.r11 * src[:, 1]) + p.ty
        rc, want = tree.search(moved)
        assert rc == O.OK
        got = idx.cpu().numpy().view(np.uint32)
        assert np.array_equal(got, want), (prm, int((got != want).sum()))
        assert np.array_equal(a.cpu().numpy(), moved[:, :2])
        assert np.array_equal(b.cpu().numpy(), dst[want][:, :2])
    searches = failed[-1][0]
    assert searches >= 5, failed  # the small steps were checked (the long jump and the search after it were not)
    if kind != "lattice":
        assert min(f[1] for f in failed[2:] if f[0] > 0) < 0.2 * n, failed  # ... and mostly passed


def test_seeded_search_never_settles_on_a_target_with_a_nan_coordinate():
    """regression (profiles/extended_fuzz.py, seeds 10154 / 10654): the seed pass took the first record it saw,
    even one at a NaN distance, and a match at a NaN distance is never displaced (every comparison with it is
    false) -- two or three queries of 70 000 came back with the NaN target.  Such targets are still binned into
    the grid (by their finite coordinates); they must simply never win."""
    rng = np.random.default_rng(10154)
    m, n = 9000, 70_000
    dst = rng.normal(size=(m, 3)) * 5
    src = dst[rng.integers(0, m, size=n)] + rng.normal(size=(n, 3)) * 0.05
    dst[8603, 2] = np.nan
    dst[17, 0] = np.nan
    dst[4000, 1] = np.nan  # (an infinite coordinate leaves no grid at all: the sweep serves those clouds)
    icp = I.Icp3d(dst)
    assert I.lib().icp_get_nn_mode(icp._h) == I.NN_GRID
    T, idx, inner = icp.estimate(src, I.Transform(), 3, return_info=True)
    rc, oT, oidx, oinner = oracle_in_device_order(icp, 3, dst, src, O.transform_identity(), 3, use_kdtree=False)
    assert rc == O.OK
    assert not np.isin(idx, [8603, 17, 4000]).any()
    assert np.array_equal(idx, oidx) and np.array_equal(inner, oinner)
    assert np.array_equal(T.as_array(), oT.as_array())


def test_finite_queries_beyond_the_f32_screen_keep_their_exact_neighbour():
    """a finite query farther than ~1.8e19 from every target overflows the f32 screen of the seeded first
    search (every screened distance is +inf), which used to leave it without a seed -- "NaN query", index 0 -- for the
    rest of the call, while the f64 distance the contract is defined on is finite.  70 000 queries (the one-lane path)
    with a few at 1e20 and 1e30."""
    rng = np.random.default_rng(31)
    m, n = 30_000, 70_000
    dst = rng.normal(size=(m, 3)) * 5
    src = dst[rng.integers(0, m, size=n)] + rng.normal(size=(n, 3)) * 0.05
    far = rng.choice(n, size=12, replace=False)
    src[far[:6]] = rng.normal(size=(6, 3)) * 1e20
    src[far[6:]] = rng.normal(size=(6, 3)) * 1e30
    icp = I.Icp3d(dst)
    T, idx, inner = icp.estimate(src, I.Transform(), 3, return_info=True)
    rc, oT, oidx, oinner = oracle_in_device_order(icp, 3, dst, src, O.transform_identity(), 3, use_kdtree=False)
    assert rc == O.OK
    assert np.array_equal(idx, oidx), np.nonzero(idx != oidx)[0][:10]
    assert np.array_equal(inner, oinner) and np.array_equal(T.as_array(), oT.as_array())


def test_warm_search_with_four_lanes_per_query_tracks_the_oracle_over_a_large_motion():
    """the pose moves a lot in the first iterations: boxes of many rows, dealt to the four lanes"""
    pk = synth.synthetic_scan3d_packets(150)
    s3, d3 = synth.remove_invalid_values(pk[:75]), synth.remove_invalid_values(pk[75:150])
    icp = I.Icp3d(d3)
    T, idx, inner = icp.estimate(s3, I.Transform(), 8, return_info=True)
    rc, oT, oidx, oinner = oracle_in_device_order(icp, 3, d3, s3, O.transform_identity(), 8)
    assert rc == O.OK
    assert np.array_equal(idx, oidx) and np.array_equal(inner, oinner)
    assert np.array_equal(T.as_array(), oT.as_array())


def test_grid_nn_full_size_vs_kdtree():
    """BASELINE's 1M x 1M synthetic pair: every correspondence index equals the oracle's."""
    src, dst = synth.synthetic_pair(1_000_000, 1_000_000)
    got = _nn(dst, src, I.NN_GRID)
    rc, want = O.KdTree(dst).search(src)
    assert rc == O.OK
    assert np.array_equal(got, want)


def test_icp_grid_mode_is_bit_identical_to_brute_mode():
    src, dst = synth.synthetic_pair(60_000, 50_000)
    a = I.Icp3d(dst, nn_mode=I.NN_GRID).estimate(src, I.Transform(), 4, return_info=True)
    b = I.Icp3d(dst, nn_mode=I.NN_BRUTE).estimate(src, I.Transform(), 4, return_info=True)
    assert np.array_equal(a[0].as_array(), b[0].as_array())
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_auto_mode_picks_grid_for_large_targets_and_brute_for_small():
    big = I.Icp3d(synth.synthetic_pair(1, 20_000)[1])
    small = I.Icp2d(load_scan2d(os.path.join(GOLDEN, "scans2d", "002.txt")))
    assert I.lib().icp_get_nn_mode(big._h) == I.NN_GRID
    assert I.lib().icp_get_nn_mode(small._h) == I.NN_BRUTE


# ------------------------------------- fast selection pipeline: overflow -> radix fallback --


@pytest.mark.parametrize("n", [900, 10_000, 200_001])
def test_weighted_gn_with_a_run_of_duplicates_at_the_median(n):
    """30 % of the x residuals are one exact value sitting on the median: more equal-prefix
    keys than the candidate buffer holds, so the short pipeline must hand over to the general
    radix select -- same bits either way."""
    rng = np.random.default_rng(n)
    a = rng.normal(size=(n, 2)) * 10
    r = rng.normal(size=(n, 2)) * 0.2
    k = int(0.3 * n)
    r[:k, 0] = 0.0
    r[k:k + (n - k) // 2, 0] = -np.abs(r[k:k + (n - k) // 2, 0]) - 1e-3
    r[k + (n - k) // 2:, 0] = np.abs(r[k + (n - k) // 2:, 0]) + 1e-3
    b = a - r  # identity pose: residual = a - b = r exactly? (a - (a - r)) may round; fine, oracle sees the same
    T = I.Transform()
    got = I.weighted_gauss_newton_update(T, a, b)
    blocks, threads = I.reduce_geometry(n)
    rc, want, _ = O.weighted_gauss_newton_update_tree(opose(T), a, b, blocks, threads)
    assert rc == O.OK and got is not None
    assert np.array_equal(got, want)


def test_estimate_transform_small_inputs_use_the_single_launch_stages():
    for n in (2, 3, 5, 64, 1024, 1025):
        a, b = make_pairs(n, 11 * n)
        got, inner = I.estimate_transform(a, b, return_inner_iters=True)
        want, want_inner = O.estimate_transform(a, b)
        assert inner == want_inner
        assert_pose_close(got, want)


@pytest.mark.parametrize("n", [300, 700, 1000])
def test_single_launch_estimate_with_a_run_of_equal_residuals_at_the_median(n):
    """40 % or more of the source points sit exactly (0.5, 0.25) from their nearest target, the rest evenly
    either side: every median is a run of equal keys longer than the in-kernel selection lists.  The
    1024-thread workgroup (n > 768) sorts instead; the smaller workgroups hand the call back to the
    host-driven path.  Same bits as the oracle either way, for each of the three workgroup sizes."""
    rng = np.random.default_rng(n)
    side = int(np.ceil(np.sqrt(n)))
    gx, gy = np.meshgrid(np.arange(side), np.arange(side))
    dst = (np.stack([gx.ravel(), gy.ravel()], axis=1)[:n] * 4.0 + rng.integers(0, 8, size=(n, 2)) / 8.0).astype(np.float64)
    off = np.empty((n, 2))
    k = max(int(0.4 * n), 160)  # (the lists hold 128 keys)
    off[:k] = (0.5, 0.25)
    h = (n - k) // 2
    off[k:k + h] = (0.5, 0.25) - rng.integers(1, 64, size=(h, 2)) / 128.0
    off[k + h:] = (0.5, 0.25) + rng.integers(1, 64, size=(n - k - h, 2)) / 128.0
    src = dst + off  # exact: multiples of 1/128 of moderate size
    icp = I.Icp2d(dst)
    got, idx, inner = icp.estimate(src, I.Transform(), 3, return_info=True)
    served, evals, sorted_evals = icp.single_launch_counters()
    if n > 768:
        assert served == 1 and sorted_evals >= 1
    else:
        assert served == 0
    rc, want, oidx, oinner = oracle_in_device_order(icp, 2, dst, src, opose(I.Transform()), 3, use_kdtree=False)
    assert rc == O.OK
    assert np.array_equal(got.as_array(), want.as_array())
    assert np.array_equal(idx, oidx) and np.array_equal(inner, oinner)


def test_stage_calls_on_the_default_stream_are_ordered_and_shards_equal_the_whole():
    """Two handles each matching half of the source cloud into one pair buffer, then the
    replicated inner loop -- all on torch's default (NULL) stream with no explicit
    synchronisation, as the multi-GPU driver does around its all-gather.  Must equal the
    unsharded estimate bit for bit (regression: NULL used to mean "own stream")."""
    import torch

    from icp_rust_amd.dist import HipStages, ShardedIcp, shard_range

    n = m = 150_000
    src, dst = synth.synthetic_pair(n, m)
    d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
    whole = I.Icp3d(d_dst)
    ref = I.Icp3d(d_dst).estimate(d_src, I.Transform(), 4)
    # (the whole-call path folds its sums over the cell-sorted cloud, icp_last_fold_order; the stage calls fold over
    # the cloud they are handed: hand them the sorted one)
    d_src, _ = whole.sort_source_device(d_src, I.Transform())
    T1, inner1 = ShardedIcp(HipStages(whole), n).estimate(d_src, I.Transform(), 4)
    shards = [shard_range(n, r, 2) for r in range(2)]
    srcs = [d_src[lo:hi].contiguous() for lo, hi in shards]
    parts = [I.Icp3d(d_dst) for _ in shards]
    a = torch.empty((n, 2), dtype=torch.float64, device="cuda")
    b = torch.empty_like(a)
    T = I.Transform()
    for ic, s in zip(parts, srcs):
        ic.set_stream(torch.cuda.current_stream().cuda_stream)
        ic.prepare_source_device(s, T)
    inner2 = []
    for _ in range(4):
        for ic, s, (lo, hi) in zip(parts, srcs, shards):
            ic.correspond_device(s, T, a[lo:hi], b[lo:hi])
        dT, k = whole.estimate_transform_device(a, b)
        T = dT * T
        inner2.append(k)
    assert np.array_equal(T1.as_array(), T.as_array())
    assert inner1.tolist() == inner2
    assert np.array_equal(ref.as_array(), T.as_array())


def test_distinct_handles_are_independent_across_host_threads():
    """include/icp_mi355x.h: one in-flight call per handle, distinct handles are independent.  Four
    host threads (ctypes releases the GIL during a call) register different clouds at the same time,
    two of them creating and destroying their handles as they go (the handle pool is shared)."""
    import threading

    jobs = []
    for k, (n, m, dim) in enumerate([(30_000, 25_000, 3), (646, 668, 2), (70_000, 40_000, 3), (5_000, 9_000, 2)]):
        rng = np.random.default_rng(900 + k)
        dst = rng.normal(size=(m, dim)) * 8
        src = dst[rng.integers(0, m, size=n)] + rng.normal(size=(n, dim)) * 0.04
        probe = (I.Icp3d if dim == 3 else I.Icp2d)(dst)  # (one call up front: the order its sums are folded in)
        probe.estimate(src, I.Transform(), 4)
        rc, oT, oidx, oinner = oracle_in_device_order(probe, dim, dst, src, O.transform_identity(), 4)
        probe.close()
        assert rc == O.OK
        jobs.append((dim, dst, src, oT.as_array(), oidx, oinner))
    errors = []

    def work(k):
        dim, dst, src, want_T, want_idx, want_inner = jobs[k]
        cls = I.Icp3d if dim == 3 else I.Icp2d
        try:
            icp = cls(dst)
            for rep in range(6):
                if k % 2 and rep:  # handle turnover while the others are mid-call
                    icp.close()
                    icp = cls(dst)
                T, idx, inner = icp.estimate(src, I.Transform(), 4, return_info=True)
                if not (np.array_equal(T.as_array(), want_T) and np.array_equal(idx, want_idx)
                        and np.array_equal(inner, want_inner)):
                    errors.append((k, rep, "mismatch"))
            icp.close()
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not any(th.is_alive() for th in threads)
    assert errors == []


def test_device_tensor_arguments_are_validated():
    """a float32 tensor, a strided slice, the wrong column count or a short index buffer
    must be refused by the host mirror, not read out of bounds by a kernel."""
    import torch

    src, dst = synth.synthetic_pair(5000, 9000)
    d_dst, d_src = torch.from_numpy(dst).cuda(), torch.from_numpy(src).cuda()
    icp = I.Icp3d(d_dst)
    T = I.Transform()
    wide = torch.zeros((5000, 4), dtype=torch.float64, device="cuda")
    bad = [d_src.float(), wide[:, :3], d_src[:, :2].contiguous(), d_src.reshape(-1)]
    for t in bad:
        with pytest.raises(ValueError):
            icp.estimate(t, T, 1)
        with pytest.raises(ValueError):
            icp.prepare_source_device(t, T)
    with pytest.raises(ValueError):
        I.Icp3d(d_dst.float())
    with pytest.raises(ValueError):
        I.Icp2d(d_dst)
    idx_short = torch.empty(10, dtype=torch.int32, device="cuda")
    idx_wide = torch.empty(5000, dtype=torch.int64, device="cuda")
    for ix in (idx_short, idx_wide):
        with pytest.raises(ValueError):
            icp.nn_search_device(d_src, ix)
    a = torch.empty((5000, 2), dtype=torch.float64, device="cuda")
    with pytest.raises(ValueError):
        icp.correspond_device(d_src, T, a[:100], a)
    with pytest.raises(ValueError):
        icp.estimate_transform_device(a, a[:4000])
    # and the well-formed call still works
    assert icp.estimate(d_src, T, 2) is not None


def test_grid_on_an_elongated_cloud_keeps_its_cells_small():
    """a corridor-shaped cloud (20000 x 50 x 5) used to cap the x axis at 16384 cells of the
    isotropic size and pile everything beyond into the last cell of each row.  Results were exact
    either way; what is pinned here is that they still are and that the far end is searched as fast
    as the near end."""
    import time

    import torch

    rng = np.random.default_rng(77)
    m = 2_000_000
    dst = np.ascontiguousarray(rng.uniform(size=(m, 3)) * np.array([20000.0, 50.0, 5.0]))
    icp = I.Icp3d(dst, nn_mode=I.NN_GRID)
    tree = O.KdTree(dst)
    times = []
    for x0 in (100.0, 19000.0):
        q = np.ascontiguousarray(rng.uniform(size=(200_000, 3)) * np.array([900.0, 50.0, 5.0]) + np.array([x0, 0.0, 0.0]))
        d_q = torch.from_numpy(q).cuda()
        idx = torch.empty(len(q), dtype=torch.int32, device="cuda")
        icp.nn_search_device(d_q, idx)
        icp.synchronize()
        t0 = time.perf_counter()
        icp.nn_search_device(d_q, idx)
        icp.synchronize()
        times.append(time.perf_counter() - t0)
        O.set_threads(16)
        try:
            rc, want = tree.search(q[:20_000])
        finally:
            O.set_threads(1)
        assert rc == O.OK and np.array_equal(idx[:20_000].cpu().numpy().view(np.uint32), want)
    assert times[1] < 5 * times[0] + 1e-3, times
