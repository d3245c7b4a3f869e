"""EXTENSION beyond the reference (BASELINE.json configs[4], include/icp_mi355x.h section 6): a
target cloud that grows.  The reference defines the registration of a scan against any target
cloud, so the checkable contract is: after icp_append_targets the handle is, bit for bit, a fresh
Icp*::new on the concatenated cloud -- which the oracle can compute."""
import numpy as np
import pytest

import icp_rust_amd as I
import oracle_ffi as O
from parity_util import oracle_in_device_order
from icp_rust_amd import _lib, harness, synth


def moved(points, T):
    """Transform::transform on xy (transform.rs:22-24), z kept (lib.rs:52-57); numpy evaluates
    the same IEEE operations in the same order, no FMA."""
    p = np.array(points, dtype=np.float64, copy=True)
    r00, r10, r01, r11, tx, ty = T.pose.as_tuple()
    x, y = p[:, 0].copy(), p[:, 1].copy()
    p[:, 0] = (r00 * x + r01 * y) + tx
    p[:, 1] = (r10 * x + r11 * y) + ty
    return p


class OracleMap:
    """Map stand-in backed by the oracle (tests only): a plain concatenation."""

    def __init__(self, tree_order=False):
        self.tree_order = tree_order

    def __call__(self, dst):
        self.dst = np.ascontiguousarray(dst, dtype=np.float64)
        return self

    def estimate(self, src, transform, max_iter):
        kw = {}
        if self.tree_order:
            b, t = I.reduce_geometry(len(src))
            kw = dict(sum_mode=1, reduce_blocks=b, reduce_threads=t)
        rc, T, _, _ = O.icp_estimate(3, self.dst, src, O.Pose(*transform.pose.as_tuple()), max_iter,
                                     use_kdtree=True, **kw)
        assert rc == O.OK
        return I.Transform.from_pose(I.Pose(*[float(x) for x in T.as_array()]))

    def append(self, points, transform=None):
        p = moved(points, transform) if transform is not None else np.asarray(points, dtype=np.float64)
        self.dst = np.ascontiguousarray(np.concatenate([self.dst, p]))


def test_moved_equals_the_oracles_transform():
    rng = np.random.default_rng(3)
    T = I.Transform([0.3, -0.2, 0.4])
    p = rng.normal(size=(50, 3)) * 7
    want = np.array([list(O.transform_xy(O.Pose(*T.pose.as_tuple()), q)) for q in p])
    assert np.array_equal(moved(p, T), want)


def test_scan_to_map_loop_semantics():
    pk = synth.synthetic_scan3d_packets(32)
    Ts, path, world = harness.run_scan_to_map(pk, step=8, max_iter=3, icp_factory=OracleMap())
    assert len(Ts) == 3 and path.shape == (3, 2)
    # replay by hand: the map is frame 0 plus every registered frame at its pose
    dst = synth.remove_invalid_values(pk[0:8])
    T = O.transform_identity()
    for k, got in enumerate(Ts, start=1):
        scan = synth.remove_invalid_values(pk[8 * k:8 * k + 8])
        rc, T, _, _ = O.icp_estimate(3, dst, scan, T, 3, use_kdtree=True)
        assert rc == O.OK and np.array_equal(got.as_array(), T.as_array())
        dst = np.concatenate([dst, moved(scan, got)])
    assert np.array_equal(world.dst, dst)
    assert np.array_equal(path[-1], Ts[-1].t)
    # the sensor moves: so does the path
    assert np.linalg.norm(path[-1]) > np.linalg.norm(path[0]) > 0


def test_max_frames_limits_the_loop():
    pk = synth.synthetic_scan3d_packets(40)
    Ts, _, _ = harness.run_scan_to_map(pk, step=8, max_iter=1, icp_factory=OracleMap(), max_frames=2)
    assert len(Ts) == 2


# ------------------------------------------------------------------------------ GPU ----
gpu = pytest.mark.gpu


def _cloud(rng, m, dim=3):
    return np.ascontiguousarray(rng.normal(size=(m, dim)) * np.array([20.0, 20.0, 2.0][:dim]))


@gpu
@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("m0,k", [(500, 300), (8000, 500), (20000, 5000), (0, 4000), (1, 1)])
def test_append_equals_a_fresh_handle_on_the_concatenation(dim, m0, k):
    """covers: the sweep as engine (small), crossing the size where the grid takes over, the grid,
    a map that starts empty"""
    rng = np.random.default_rng(1000 * dim + m0 + k)
    base, extra = _cloud(rng, m0, dim), _cloud(rng, k, dim)
    q = _cloud(rng, 3000, dim)
    T = I.Transform([0.4, -0.3, 0.2])
    cls = I.Icp3d if dim == 3 else I.Icp2d
    grown = cls(base)
    grown.append(extra, T)
    cat = np.concatenate([base, moved(extra, T)])
    assert grown.target_count == m0 + k
    assert np.array_equal(grown.read_targets(), cat)
    fresh = cls(cat)
    assert _lib.lib().icp_get_nn_mode(grown._h) == _lib.lib().icp_get_nn_mode(fresh._h)
    got = grown.nn_search(q)
    assert np.array_equal(got, fresh.nn_search(q))
    rc, want = O.nn_brute(cat, q)
    assert rc == O.OK and np.array_equal(got, want)
    init = I.Transform([0.05, 0.02, -0.01])
    Tg, ig, ng = grown.estimate(q, init, 4, return_info=True)
    Tf, if_, nf = fresh.estimate(q, init, 4, return_info=True)
    assert np.array_equal(Tg.as_array(), Tf.as_array())
    assert np.array_equal(ig, if_) and np.array_equal(ng, nf)


@gpu
def test_repeated_appends_between_estimates_track_the_oracle():
    """a handle that has already searched (cell-sorted snapshot, previous matches, window
    predictions, speculation state) keeps returning the oracle's results as its cloud grows"""
    rng = np.random.default_rng(77)
    world = synth.box_cloud(synth.SEED + 5, 60000, synth.ROOM_LO, synth.ROOM_HI)
    parts = np.array_split(world, 6)
    grown = I.Icp3d(parts[0])
    dst = parts[0]
    scan = world[rng.choice(len(world), 20000, replace=False)] + rng.normal(size=(20000, 3)) * 0.01
    T = I.Transform([0.05, -0.04, 0.01])
    for part in parts[1:]:
        Tg, idx, inner = grown.estimate(scan, T, 3, return_info=True)
        rc, oT, oidx, oinner = oracle_in_device_order(grown, 3, dst, scan, O.Pose(*T.pose.as_tuple()), 3)
        assert rc == O.OK
        assert np.array_equal(idx, oidx) and np.array_equal(inner, oinner)
        assert np.array_equal(Tg.as_array(), oT.as_array())
        grown.append(part)  # no transform: plain concatenation
        dst = np.concatenate([dst, part])
    assert grown.target_count == len(world)
    assert np.array_equal(grown.read_targets(len(world) - 10, 10), world[-10:])


@gpu
def test_a_borrowed_device_cloud_moves_into_the_handle_on_append():
    import torch

    rng = np.random.default_rng(9)
    base, extra, q = _cloud(rng, 12000), _cloud(rng, 3000), _cloud(rng, 2000)
    d_base = torch.from_numpy(base).cuda()
    grown = I.Icp3d(d_base)
    grown.reserve(20000)
    d_base.zero_()  # the handle no longer reads the caller's buffer
    del d_base
    T = I.Transform([1.0, 2.0, -0.3])
    grown.append(torch.from_numpy(extra).cuda(), T)
    cat = np.concatenate([base, moved(extra, T)])
    assert np.array_equal(grown.read_targets(), cat)
    rc, want = O.nn_brute(cat, q)
    assert np.array_equal(grown.nn_search(q), want)


@gpu
def test_forced_sweep_on_a_grown_map_rebuilds_its_structures_on_demand():
    rng = np.random.default_rng(10)
    base, extra, q = _cloud(rng, 9000), _cloud(rng, 2500), _cloud(rng, 1500)
    grown = I.Icp3d(base)
    grown.append(extra)  # grid engine: the sweep's SoA / f32 screen are left stale
    _lib.check(_lib.lib().icp_set_nn_mode(grown._h, I.NN_BRUTE))
    rc, want = O.nn_brute(np.concatenate([base, extra]), q)
    assert np.array_equal(grown.nn_search(q), want)


@gpu
def test_a_pooled_handle_does_not_screen_with_the_previous_clouds_records():
    """regression: a handle taken from the pool kept the f32 screen of its previous target cloud
    when the new cloud has no grid (non-finite coordinates), and the sweep used it"""
    rng = np.random.default_rng(11)
    _lib.lib().icp_trim_pool()
    first = I.Icp3d(_cloud(rng, 9000) + 100.0)
    first.close()  # parked in the pool with its screen
    dst = _cloud(rng, 9000)
    dst[17, 0] = np.inf  # no bounding box -> no grid, no screen
    q = _cloud(rng, 1000)
    icp = I.Icp3d(dst)
    assert _lib.lib().icp_get_nn_mode(icp._h) == I.NN_BRUTE
    rc, want = O.nn_brute(dst, q)
    assert np.array_equal(icp.nn_search(q), want)


@gpu
def test_bad_arguments_and_limits():
    icp = I.Icp3d(np.zeros((4, 3)))
    L = _lib.lib()
    assert L.icp_append_targets(icp._h, None, 5, None) == _lib.BAD_ARGUMENT
    assert L.icp_append_targets(icp._h, None, 0, None) == _lib.OK
    assert L.icp_append_targets(None, None, 0, None) == _lib.BAD_ARGUMENT
    assert L.icp_read_targets(icp._h, 3, 2, None) == _lib.BAD_ARGUMENT
    assert L.icp_target_count(None) == 0
    assert icp.target_count == 4


@gpu
def test_scan_to_map_trajectory_on_gpu_matches_the_oracle_bit_for_bit():
    pk = synth.synthetic_scan3d_packets(5 * 30)
    Ts, path, world = harness.run_scan_to_map(pk, step=30, max_iter=5)
    Os, opath, oworld = harness.run_scan_to_map(pk, step=30, max_iter=5, icp_factory=OracleMap(tree_order=True))
    assert len(Ts) == len(Os) == 4
    for a, b in zip(Ts, Os):
        assert np.array_equal(a.as_array(), b.as_array())
    assert np.array_equal(path, opath)
    assert np.array_equal(world.read_targets(), oworld.dst)
    # and within the north_star tolerance of the reference-order (left fold) oracle
    Rs, _, _ = harness.run_scan_to_map(pk, step=30, max_iter=5, icp_factory=OracleMap())
    for a, b in zip(Ts, Rs):
        assert np.max(np.abs(a.as_array() - b.as_array())) <= 1e-5 * max(1.0, np.max(np.abs(b.as_array())))


@gpu
def test_incremental_and_rebuilding_appends_both_equal_a_fresh_handle():
    """The append moves the grid's sorted records instead of re-sorting the cloud while every new point lies within
    half a cell of the grid's box (include/icp_mi355x.h: icp_grid_append_counters); points farther out, or a cloud
    that has outgrown its cell size, rebuild.  Whatever path an append took, the handle must answer like a fresh one
    on the concatenated cloud: nearest neighbours of a query set, and a registration, bit for bit."""
    rng = np.random.default_rng(41)
    base = synth.box_cloud(synth.SEED + 21, 60_000, synth.ROOM_LO, synth.ROOM_HI)
    grown = I.Icp3d(base)
    cloud = base
    q = synth.box_cloud(synth.SEED + 22, 30_000, synth.ROOM_LO, synth.ROOM_HI) + rng.normal(size=(30_000, 3)) * 0.02
    scan = base[rng.choice(len(base), 20_000, replace=False)] + rng.normal(size=(20_000, 3)) * 0.01
    parts = [
        synth.box_cloud(synth.SEED + 23, 5_000, synth.ROOM_LO, synth.ROOM_HI),                 # inside: moved
        synth.box_cloud(synth.SEED + 24, 5_000, synth.ROOM_LO, synth.ROOM_HI) + rng.normal(size=(5_000, 3)) * 0.002,  # a hair outside (well within half a cell of 0.035): moved
        synth.box_cloud(synth.SEED + 25, 300, synth.ROOM_LO, synth.ROOM_HI) + np.array([4.0, 0.0, 0.0]),              # far outside: rebuilt
        synth.box_cloud(synth.SEED + 26, 7_000, synth.ROOM_LO, synth.ROOM_HI),                 # inside the new box: moved
        synth.box_cloud(synth.SEED + 27, 60_000, synth.ROOM_LO, synth.ROOM_HI),                # outgrows the cell size: rebuilt
    ]
    T0 = I.Transform([0.02, -0.01, 0.004])
    for part in parts:
        grown.append(part)
        cloud = np.concatenate([cloud, part])
        fresh = I.Icp3d(cloud, nn_mode=I.NN_GRID)
        assert I.lib().icp_get_nn_mode(grown._h) == I.NN_GRID
        assert np.array_equal(grown.nn_search(q), fresh.nn_search(q))
        a = grown.estimate(scan, T0, 3, return_info=True)
        b = fresh.estimate(scan, T0, 3, return_info=True)
        assert np.array_equal(a[0].as_array(), b[0].as_array()) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        fresh.close()
    rc, want = O.nn_brute(cloud, q[:3000])
    assert rc == O.OK and np.array_equal(grown.nn_search(q[:3000]), want)
    moved, rebuilt = grown.append_counters()
    assert moved >= 3 and rebuilt >= 2, (moved, rebuilt)


@gpu
def test_an_appended_handle_registers_a_large_source_cloud_like_its_own_fold_order_says():
    """a source cloud above 65 536 points folds its sums in the order of the HANDLE'S grid cells
    (include/icp_mi355x.h section 9a).  An incrementally appended handle keeps the grid of its last full build where a
    fresh handle on the concatenated cloud derives a new one, so the two fold in different orders: same
    correspondences, poses equal to the rounding of a re-ordered sum -- and each equal, bit for bit, to the oracle
    evaluated in ITS OWN fold order (icp_last_fold_order)."""
    rng = np.random.default_rng(77)
    base = synth.box_cloud(synth.SEED + 31, 90_000, synth.ROOM_LO, synth.ROOM_HI)
    extra = synth.box_cloud(synth.SEED + 32, 8_000, synth.ROOM_LO, synth.ROOM_HI)
    cloud = np.concatenate([base, extra])
    src = cloud[rng.choice(len(cloud), 70_000, replace=False)] + rng.normal(size=(70_000, 3)) * 0.01
    grown = I.Icp3d(base, nn_mode=I.NN_GRID)
    grown.append(extra)
    assert grown.append_counters()[0] == 1  # served incrementally: the grid of the 90 000-point build
    fresh = I.Icp3d(cloud, nn_mode=I.NN_GRID)
    T0 = I.Transform([0.02, -0.01, 0.004])
    init = O.transform_new(np.array([0.02, -0.01, 0.004]))
    for icp in (grown, fresh):
        T, idx, inner = icp.estimate(src, T0, 4, return_info=True)
        rc, oT, oidx, oinner = oracle_in_device_order(icp, 3, cloud, src, init, 4)
        assert rc == O.OK
        assert np.array_equal(idx, oidx) and np.array_equal(inner, oinner)
        assert np.array_equal(T.as_array(), oT.as_array())
    Tg, ig, _ = grown.estimate(src, T0, 4, return_info=True)
    Tf, i_f, _ = fresh.estimate(src, T0, 4, return_info=True)
    assert np.array_equal(ig, i_f)
    assert np.max(np.abs(Tg.as_array() - Tf.as_array())) <= 1e-10
