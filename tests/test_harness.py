"""Frame-loop harnesses (examples/scan2d.rs, examples/scan3d.rs semantics)."""
import os
import shutil

import numpy as np
import pytest

import icp_rust_amd as I
import oracle_ffi as O
from icp_rust_amd import harness, synth
from icp_rust_amd.scans import load_scan2d

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scans2d")


class OracleIcp:
    """Icp{2,3}d stand-in backed by the oracle (tests only)."""

    def __init__(self, dim, tree_order=False):
        self.dim, self.tree_order = dim, tree_order

    def __call__(self, dst):
        self.dst = np.ascontiguousarray(dst, dtype=np.float64)
        return self

    def estimate(self, src, transform, max_iter):
        kw = {}
        if self.tree_order:
            b, t = I.reduce_geometry(len(src))
            kw = dict(sum_mode=1, reduce_blocks=b, reduce_threads=t)
        rc, T, _, _ = O.icp_estimate(self.dim, self.dst, src, O.Pose(*transform.pose.as_tuple()), max_iter,
                                     use_kdtree=True, **kw)
        assert rc == O.OK
        return I.Transform.from_pose(I.Pose(*[float(x) for x in T.as_array()]))


def test_scan2d_loop_semantics(tmp_path):
    d = tmp_path / "scans"
    d.mkdir()
    (d / "000.txt").write_text("this file must never be parsed\n")
    for k in (1, 2, 3):
        shutil.copy(os.path.join(GOLDEN, f"{k:03d}.txt"), d / f"{k:03d}.txt")
    # 004 missing -> the loop stops after frames 2 and 3
    shutil.copy(os.path.join(GOLDEN, "005.txt"), d / "005.txt")
    Ts, invs, path = harness.run_scan2d(str(d), max_iter=5, icp_factory=OracleIcp(2))
    assert len(Ts) == 2 and path.shape == (2, 2)
    # replay by hand: 001 is the fixed source, warm start from the previous frame
    src = load_scan2d(os.path.join(GOLDEN, "001.txt"))
    T = O.transform_identity()
    for k, got in zip((2, 3), Ts):
        rc, T, _, _ = O.icp_estimate(2, load_scan2d(os.path.join(GOLDEN, f"{k:03d}.txt")), src, T, 5,
                                     use_kdtree=True)
        assert np.array_equal(got.as_array(), T.as_array())
    assert np.array_equal(path[1], invs[1].t)
    assert np.array_equal(invs[1].as_array(), O.transform_inverse(T).as_array())


def test_scan3d_loop_semantics():
    pk = synth.synthetic_scan3d_packets(24)
    Ts, invs, path = harness.run_scan3d(pk, step=8, max_iter=2, icp_factory=OracleIcp(3))
    assert len(Ts) == 3
    # first frame registers the source against itself: nothing to correct
    assert np.allclose(Ts[0].as_array(), I.Transform().as_array(), atol=1e-12)
    assert np.linalg.norm(path[2]) > np.linalg.norm(path[0])


def test_invalid_value_filter():  # examples/scan3d.rs:63-69
    p = np.array([[0.0, 0.0, 0.0], [0.2, 0.0, 0.0], [0.0, 0.21, 0.0], [1.0, 1.0, 1.0]])
    assert np.array_equal(synth.remove_invalid_values(p), p[2:])


class CountingIcp2d:
    """Icp2d factory that keeps what the GPU handles report: inner-iteration counts per outer iteration
    and the speculative-search counters (confirmed, discarded) of every frame."""

    def __init__(self):
        self.inner, self.spec = [], [0, 0]

    def __call__(self, dst):
        self.icp = I.Icp2d(dst)
        return self

    def estimate(self, src, transform, max_iter):
        T, _, inner = self.icp.estimate(src, transform, max_iter, return_info=True)
        self.inner.append([int(x) for x in inner])
        c = I.gn_path_counters(self.icp)
        self.spec[0] += c[4]
        self.spec[1] += c[5]
        self.icp.close()
        return T


@pytest.mark.gpu
def test_scan2d_trajectory_on_gpu_matches_oracle_bit_for_bit():
    """examples/scan2d.rs:62-90 over the reference's scans 001 .. 040 (39 frames, 780 outer iterations,
    ~7 600 inner Gauss-Newton iterations with loops of up to 77): the whole trajectory bit-equal to the
    oracle in the device's summation order."""
    fac = CountingIcp2d()
    Ts, _, path = harness.run_scan2d(GOLDEN, max_iter=20, icp_factory=fac)
    Os, _, opath = harness.run_scan2d(GOLDEN, max_iter=20, icp_factory=OracleIcp(2, tree_order=True))
    assert len(Ts) == len(Os) == 39
    for a, b in zip(Ts, Os):
        assert np.array_equal(a.as_array(), b.as_array())
    assert np.array_equal(path, opath)
    # the run really exercised long inner loops and wrong speculative bets (a bet on "one update, then
    # the loop ends" that the loop did not honour: the search is discarded and repeated)
    assert max(max(f) for f in fac.inner) >= 30
    assert sum(sum(f) for f in fac.inner) > 5000
    assert fac.spec[0] >= 1 and fac.spec[1] >= 1, fac.spec
    # and within the north_star tolerance of the reference-order oracle
    Rs, _, _ = harness.run_scan2d(GOLDEN, max_iter=20, icp_factory=OracleIcp(2))
    for a, b in zip(Ts, Rs):
        assert np.max(np.abs(a.as_array() - b.as_array())) <= 1e-5 * max(1.0, np.max(np.abs(b.as_array())))


@pytest.mark.gpu
def test_scan3d_trajectory_on_gpu_matches_oracle_bit_for_bit():
    pk = synth.synthetic_scan3d_packets(4 * 75)
    Ts, _, path = harness.run_scan3d(pk, max_iter=20)
    Os, _, opath = harness.run_scan3d(pk, max_iter=20, icp_factory=OracleIcp(3, tree_order=True))
    assert len(Ts) == 4
    for a, b in zip(Ts, Os):
        assert np.array_equal(a.as_array(), b.as_array())


@pytest.mark.gpu
def test_cli_prints_the_trajectories(capsys):
    assert harness.main(["scan2d", GOLDEN, "--max-iter", "5"]) == 0
    out = [ln for ln in capsys.readouterr().out.splitlines() if ln and not ln.startswith("#")]
    Ts, _, path = harness.run_scan2d(GOLDEN, max_iter=5)
    assert len(out) == len(path) == 39
    assert [float(v) for v in out[-1].split()[1:]] == pytest.approx(list(path[-1]), abs=1e-9)
    assert harness.main(["scan2map", "--frames", "2", "--max-iter", "3"]) == 0
    out = capsys.readouterr().out.splitlines()
    assert out[0].startswith("# map:") and len(out) == 3
