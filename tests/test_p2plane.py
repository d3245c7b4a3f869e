"""EXTENSION beyond the reference (BASELINE.json configs[4], include/icp_mi355x.h section 7):
point-to-plane residuals.  tier4/icp_rust has no normals and no plane residual, so there is NO
reference behaviour and NO parity claim here.  The checker is an independent CPU statement of the
documented definition (oracle/icp_oracle.c: orc_p2pl_*: brute-force k nearest neighbours, left-fold
sums), plus properties any point-to-plane ICP must have."""
import numpy as np
import pytest

import icp_rust_amd as I
import oracle_ffi as O
from icp_rust_amd import _lib


def room(rng, m):
    """points on the floor and two walls of a room, with a little noise off the planes"""
    k = m // 3
    fl = np.stack([rng.uniform(-3, 3, k), rng.uniform(-3, 3, k), rng.normal(0, 2e-3, k)], axis=1)
    w1 = np.stack([rng.normal(3, 2e-3, k), rng.uniform(-3, 3, k), rng.uniform(0, 2, k)], axis=1)
    w2 = np.stack([rng.uniform(-3, 3, m - 2 * k), rng.normal(-3, 2e-3, m - 2 * k), rng.uniform(0, 2, m - 2 * k)], axis=1)
    return np.ascontiguousarray(np.concatenate([fl, w1, w2]))


def moved(p, T):
    q = p.copy()
    r00, r10, r01, r11, tx, ty = T.pose.as_tuple()
    q[:, 0] = (r00 * p[:, 0] + r01 * p[:, 1]) + tx
    q[:, 1] = (r10 * p[:, 0] + r11 * p[:, 1]) + ty
    return q


def test_cpu_checker_normals_of_a_plane_and_its_sign_convention():
    rng = np.random.default_rng(1)
    p = np.stack([rng.uniform(-1, 1, 400), rng.uniform(-1, 1, 400), np.zeros(400)], axis=1)
    n = O.p2pl_normals(p, 8)
    assert np.allclose(n, [0, 0, 1], atol=1e-12)
    # a tilted plane: normal parallel to (1, 1, 1) / sqrt(3), leading component positive
    q = p.copy()
    q[:, 2] = -(q[:, 0] + q[:, 1])
    n = O.p2pl_normals(q, 10)
    assert np.allclose(n, np.ones(3) / np.sqrt(3), atol=1e-9)
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-12)


def test_cpu_checker_point_to_plane_recovers_a_pose_and_ignores_sliding():
    rng = np.random.default_rng(2)
    dst = room(rng, 3000)
    normals = O.p2pl_normals(dst, 10)
    tree = O.KdTree(dst)
    Tt = O.transform_new(np.array([0.04, -0.03, 0.02]))
    inv = O.transform_inverse(Tt)
    src = dst[rng.integers(0, len(dst), 1500)].copy()
    src[:, :2] = O.transform_apply_many(inv, src[:, :2])
    rc, T, _, inner = O.p2pl_estimate(tree, normals, src, O.transform_identity(), 10)
    assert rc == O.OK and inner.sum() > 0
    assert np.allclose(T.as_array(), Tt.as_array(), atol=5e-3)


gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize("m,k", [(5000, 10), (20000, 6), (9000, 16), (300, 3)])
def test_device_normals_equal_the_cpu_statement(m, k):
    rng = np.random.default_rng(10 * m + k)
    dst = room(rng, m)
    icp = I.Icp3d(dst, nn_mode=I.NN_GRID) if m >= 8192 else I.Icp3d(dst)
    icp.compute_normals(k)
    got = icp.read_normals()
    O.set_threads(16)
    try:
        want = O.p2pl_normals(dst, k)
    finally:
        O.set_threads(1)
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-12)
    # same neighbours (exact k-NN by (d^2, index)), same Jacobi sequence: equal to rounding
    assert np.max(np.abs(got - want)) < 1e-9


@gpu
def test_device_point_to_plane_registration_tracks_the_cpu_statement():
    rng = np.random.default_rng(77)
    dst = room(rng, 30000)
    src = dst[rng.integers(0, len(dst), 12000)] + rng.normal(0, 1e-3, (12000, 3))
    Tt = I.Transform([0.05, -0.04, 0.015])
    src = moved(src, Tt.inverse())
    icp = I.Icp3d(dst)
    with pytest.raises(I.IcpError):  # normals first
        icp.estimate_point_to_plane(src, I.Transform(), 3)
    icp.compute_normals(10)
    T, idx, inner = icp.estimate_point_to_plane(src, I.Transform(), 8, return_info=True)
    normals = icp.read_normals()
    tree = O.KdTree(dst)
    O.set_threads(16)
    try:
        rc, oT, oidx, oinner = O.p2pl_estimate(tree, normals, src, O.transform_identity(), 8)
    finally:
        O.set_threads(1)
    assert rc == O.OK
    assert np.array_equal(idx, oidx)          # correspondences are the reference's exact 3-D NN either way
    assert np.array_equal(inner, oinner)
    assert np.max(np.abs(T.as_array() - oT.as_array())) < 1e-9  # tree sums vs left folds
    assert np.allclose(T.as_array(), Tt.as_array(), atol=3e-3)
    # device-resident source: same bits as the host-buffer call
    import torch

    T2 = icp.estimate_point_to_plane(torch.from_numpy(src).cuda(), I.Transform(), 8)
    assert np.array_equal(T.as_array(), T2.as_array())


@gpu
def test_point_to_plane_on_independent_samples_and_the_degenerate_single_plane():
    """(1) scan and map are INDEPENDENT samples of the same planes (no point of the scan is in the map):
    nearest-neighbour pairs then differ by up to the sample spacing ALONG the planes; the plane residual
    does not see that, the point-to-point residual does, so for the same number of iterations the
    point-to-plane pose is the closer one.  (2) a single plane leaves y unconstrained: the normal
    equations are exactly singular and -- as the reference's inverse3x3 does (src/linalg.rs:12-14) -- no
    update is produced at all."""
    rng = np.random.default_rng(5)
    dst = room(rng, 60000)
    scan = room(np.random.default_rng(6), 20000)
    Tt = I.Transform([0.03, -0.02, 0.01])
    src = moved(scan, Tt.inverse())
    icp = I.Icp3d(dst)
    icp.compute_normals(10)
    Tp = icp.estimate_point_to_plane(src, I.Transform(), 6)
    Tq = icp.estimate(src, I.Transform(), 6)
    ep, eq = np.max(np.abs(Tp.as_array() - Tt.as_array())), np.max(np.abs(Tq.as_array() - Tt.as_array()))
    assert ep < 2e-3 and ep < eq, (ep, eq)
    m = 20000
    wall = np.ascontiguousarray(np.stack([np.full(m, 3.0), rng.uniform(-3, 3, m), rng.uniform(0, 2, m)], axis=1))
    s1 = wall[rng.integers(0, m, 6000)].copy()
    s1[:, 0] += rng.normal(-0.02, 2e-3, len(s1))
    one = I.Icp3d(wall)
    one.compute_normals(8)
    assert np.array_equal(one.estimate_point_to_plane(s1, I.Transform(), 3).as_array(), I.Transform().as_array())


@gpu
def test_normals_must_be_recomputed_after_an_append_and_3d_only():
    rng = np.random.default_rng(9)
    dst = room(rng, 9000)
    icp = I.Icp3d(dst)
    icp.compute_normals(8)
    icp.append(room(rng, 600))
    with pytest.raises(I.IcpError) as e:
        icp.estimate_point_to_plane(dst[:100], I.Transform(), 1)
    assert e.value.status == _lib.BAD_ARGUMENT
    icp.compute_normals(8)
    assert icp.read_normals().shape == (9600, 3)
    assert icp.estimate_point_to_plane(dst[:100], I.Transform(), 1) is not None
    with pytest.raises(I.IcpError):
        I.Icp2d(dst[:, :2]).compute_normals(8)


# ------------------------------------------------ a map that grows, registered point-to-plane ----
class OraclePlaneMap:
    """CPU statement of harness.run_scan_to_map(point_to_plane=k): concatenation, kd-tree rebuilt per frame,
    normals of the appended points from the cloud at insertion time (tests only)."""

    def __call__(self, dst):
        self.dst = np.ascontiguousarray(dst, dtype=np.float64)
        self.normals = None
        return self

    def compute_normals(self, k):
        self.k = k
        self.normals = O.p2pl_normals(self.dst, k)

    def update_normals(self, k):
        assert k == self.k
        self.normals = O.p2pl_normals_update(self.dst, len(self.normals), k, self.normals)

    def estimate_point_to_plane(self, src, transform, max_iter):
        assert len(self.normals) == len(self.dst)
        rc, T, _, _ = O.p2pl_estimate(O.KdTree(self.dst), self.normals, src, O.Pose(*transform.pose.as_tuple()), max_iter)
        assert rc == O.OK
        return I.Transform.from_pose(I.Pose(*[float(x) for x in T.as_array()]))

    def append(self, points, transform=None):
        p = moved(points, transform) if transform is not None else np.asarray(points, dtype=np.float64)
        self.dst = np.ascontiguousarray(np.concatenate([self.dst, p]))


@gpu
def test_normals_of_appended_targets_only_and_the_older_ones_kept():
    rng = np.random.default_rng(5)
    dst = room(rng, 6000)
    extra = room(rng, 1500)
    icp = I.Icp3d(dst)
    icp.compute_normals(8)
    before = icp.read_normals()
    icp.append(extra)
    with pytest.raises(I.IcpError):  # a different neighbourhood size for the new points
        icp.update_normals(10)
    icp.update_normals(8)
    got = icp.read_normals()
    assert np.array_equal(got[:6000], before)  # kept, bit for bit
    O.set_threads(16)
    try:
        want = O.p2pl_normals_update(np.concatenate([dst, extra]), 6000, 8, before)
    finally:
        O.set_threads(1)
    assert np.max(np.abs(got - want)) < 1e-9
    icp.estimate_point_to_plane(extra[:500], I.Transform(), 2)  # usable again


@gpu
def test_scan_to_map_point_to_plane_on_gpu_tracks_the_cpu_statement():
    from icp_rust_amd import harness, synth

    pk = synth.synthetic_scan3d_packets(4 * 12)
    O.set_threads(16)
    try:
        Os, opath, oworld = harness.run_scan_to_map(pk, step=12, max_iter=4, icp_factory=OraclePlaneMap(), point_to_plane=8)
    finally:
        O.set_threads(1)
    Ts, path, world = harness.run_scan_to_map(pk, step=12, max_iter=4, point_to_plane=8)
    assert len(Ts) == len(Os) == 3
    for a, b in zip(Ts, Os):
        assert np.max(np.abs(a.as_array() - b.as_array())) < 1e-8  # tree sums vs left folds, carried through the map
    assert np.max(np.abs(world.read_targets() - oworld.dst)) < 1e-7
    assert np.max(np.abs(world.read_normals() - oworld.normals)) < 1e-6


@gpu
@pytest.mark.parametrize("world", [1, 2, 4])
def test_point_to_plane_across_virtual_ranks_equals_one_handle(world):
    """icp_multi_estimate_point_to_plane: the search sharded over the ranks, every slice's indices
    handed to every rank, the inner loop replicated -- pose, indices and inner counts of one handle, bit for bit;
    also after an append whose targets got their normals 'at insertion time'."""
    rng = np.random.default_rng(100 + world)
    dst = room(rng, 60_000)
    Tt = I.Transform([0.03, -0.02, 0.01])
    src = moved(dst[rng.integers(0, len(dst), 20_000)] + rng.normal(size=(20_000, 3)) * 1e-3, Tt.inverse())
    one = I.Icp3d(dst, nn_mode=I.NN_GRID)
    one.compute_normals(8)
    T1, idx1, inner1 = one.estimate_point_to_plane(src, I.Transform(), 5, return_info=True)
    multi = I.IcpMulti(dst, [0] * world)
    multi.compute_target_normals(8)
    T, idx, inner = multi.estimate_point_to_plane(src, I.Transform(), 5, return_info=True)
    assert np.array_equal(T.as_array(), T1.as_array())
    assert np.array_equal(idx, idx1) and np.array_equal(inner, inner1)
    assert inner.sum() > 0
    extra = room(rng, 6_000) + np.array([0.0, 0.0, 0.001])
    one.append(extra)
    one.update_normals(8)
    multi.append(extra)
    multi.update_target_normals(8)
    T1, idx1, inner1 = one.estimate_point_to_plane(src, I.Transform(), 4, return_info=True)
    T, idx, inner = multi.estimate_point_to_plane(src, I.Transform(), 4, return_info=True)
    assert np.array_equal(T.as_array(), T1.as_array()) and np.array_equal(idx, idx1) and np.array_equal(inner, inner1)
    multi.close()


@gpu
def test_point_to_plane_against_a_map_of_ten_million_points():
    """BASELINE configs[4] at its size (point-to-plane had been tested at <= 20k targets): a 10.4M-point
    target cloud, normals of all of it on the device.  There is no reference behaviour; the checks are (a) the CPU
    statement on a SUB-BOX -- the k nearest neighbours of a target well inside the sub-box all lie inside it, so its
    normals computed from the sub-box alone must equal the device's normals from the whole cloud; (b) properties:
    unit length; a re-observed scan registers closer to the truth than it started; (c) two virtual ranks = one handle."""
    import torch

    from icp_rust_amd import synth

    m = 10_400_000
    cloud = synth.box_cloud(synth.SEED + 51, m)
    d_cloud = torch.from_numpy(cloud).cuda()
    icp = I.Icp3d(d_cloud)
    assert I.lib().icp_get_nn_mode(icp._h) == I.NN_GRID
    k = 8
    icp.compute_normals(k)
    # (a) sub-box [8, 14] x [8, 14] x [-2, -0.2] around a patch of the floor (z = -2; nothing lies below it): 455 targets
    # per square metre on the faces, 59 per cubic metre inside -- the 8 nearest neighbours of a target lie within
    # ~0.35 m; the core keeps 1 m from the sub-box's open sides
    lo, hi = np.array([8.0, 8.0, -2.1]), np.array([14.0, 14.0, -0.2])
    inside = np.all((cloud >= lo) & (cloud <= hi), axis=1)
    sub_idx = np.nonzero(inside)[0]
    sub = np.ascontiguousarray(cloud[sub_idx])
    assert 4_000 < len(sub) < 200_000
    core = np.all((sub >= lo + np.array([1.0, 1.0, 0.0])) & (sub <= hi - 1.0), axis=1)
    O.set_threads(16)
    try:
        want = O.p2pl_normals(sub, k)
    finally:
        O.set_threads(1)
    first, last = int(sub_idx[0]), int(sub_idx[-1]) + 1
    got_all = icp.read_normals(first, last - first)
    got = got_all[sub_idx - first]
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-12)
    assert core.sum() > 2_000
    assert np.max(np.abs(got[core] - want[core])) < 1e-9
    # floor points (z within a millimetre of -2... the faces are exact planes): normals are +-z
    floor = core & (np.abs(sub[:, 2] + 2.0) < 1e-9)
    assert floor.sum() > 1_000
    # (most of a floor target's neighbours are floor targets; a few interior points nearby tilt some normals)
    assert np.median(np.abs(got[floor][:, 2])) > 0.99
    # (b) a scan that re-observes map points, moved by the inverse of a frame-sized motion
    rng = np.random.default_rng(5)
    Tt = I.Transform([0.06, -0.04, 0.003])
    scan = moved(cloud[rng.integers(0, m, 28_800)] + rng.normal(size=(28_800, 3)) * 2e-3, Tt.inverse())
    T, idx, inner = icp.estimate_point_to_plane(scan, I.Transform(), 8, return_info=True)
    err0 = np.max(np.abs(I.Transform().as_array() - Tt.as_array()))
    err = np.max(np.abs(T.as_array() - Tt.as_array()))
    assert inner.sum() > 0 and err < 0.25 * err0, (err0, err)
    # (c) two virtual ranks (two replicas of the map and of its normals on this GPU)
    multi = I.IcpMulti(cloud, [0, 0])
    multi.compute_target_normals(k)
    Tm, idxm, innerm = multi.estimate_point_to_plane(scan, I.Transform(), 8, return_info=True)
    assert np.array_equal(Tm.as_array(), T.as_array()) and np.array_equal(idxm, idx) and np.array_equal(innerm, inner)
    multi.close()
    icp.close()
