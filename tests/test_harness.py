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

    def __init__(self, single_launch=True):
        self.inner, self.spec, self.single, self.single_launch = [], [0, 0], [0, 0, 0], single_launch

    def __call__(self, dst):
        self.icp = I.Icp2d(dst)
        self.icp.set_single_launch(self.single_launch)
        return self

    def estimate(self, src, transform, max_iter):
        T, _, inner = self.icp.estimate(src, transform, max_iter, return_info=True)
        self.inner.append([int(x) for x in inner])
        c = I.gn_path_counters(self.icp)
        self.spec[0] += c[4]
        self.spec[1] += c[5]
        for k, v in enumerate(self.icp.single_launch_counters()):
            self.single[k] += v
        self.icp.close()
        return T


@pytest.mark.gpu
def test_scan2d_trajectory_on_gpu_matches_oracle_bit_for_bit():
    """examples/scan2d.rs:62-90 over the reference's scans 001 .. 040 (39 frames, 780 outer iterations,
    ~7 600 inner Gauss-Newton iterations with loops of up to 77): the whole trajectory bit-equal to the
    oracle in the device's summation order."""
    Os, _, opath = harness.run_scan2d(GOLDEN, max_iter=20, icp_factory=OracleIcp(2, tree_order=True))
    # twice: every frame registered in ONE launch (the default at this size), and by the general
    # host-driven path (stage kernels, speculative searches)
    for single in (True, False):
        fac = CountingIcp2d(single_launch=single)
        Ts, _, path = harness.run_scan2d(GOLDEN, max_iter=20, icp_factory=fac)
        assert len(Ts) == len(Os) == 39
        for a, b in zip(Ts, Os):
            assert np.array_equal(a.as_array(), b.as_array())
        assert np.array_equal(path, opath)
        # the run really exercised long inner loops ...
        assert max(max(f) for f in fac.inner) >= 30
        assert sum(sum(f) for f in fac.inner) > 5000
        if single:
            assert fac.single[0] == 39 and fac.single[1] > 5000 + 39 * 20 - 1, fac.single
        else:
            # ... and wrong speculative bets (a bet on "one update, then the loop ends" that the loop did not
            # honour: the search is discarded and repeated)
            assert fac.single[0] == 0
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


# ---------------------------------------------------------------- on-disk formats (8(f) rank 4) --
def test_packet_container_round_trip(tmp_path):
    """the scans.hdf5 stand-in: one rank-3 f64 dataset of 24 x 16 x 3 values per packet
    (examples/scan3d.rs:9,21-23,45-49), written and read back bit for bit"""
    from icp_rust_amd import scans

    pk = synth.synthetic_scan3d_packets(20)
    f = tmp_path / "scans.icppkt"
    scans.write_packets(str(f), pk)
    assert os.path.getsize(f) == scans.file_size_for(20)
    assert scans.is_packet_file(str(f)) and not scans.is_packet_file(os.path.join(GOLDEN, "001.txt"))
    s = scans.PacketFile(str(f))
    assert s.size() == 20 and s.dims == (24, 16, 3) and s.names[3] == "000003"
    assert np.array_equal(s.get(7).view(np.uint64), pk[7].view(np.uint64))
    assert np.array_equal(s.get_range(5, 9), pk[5:9].reshape(-1, 3))
    assert s.get_range(4, 4).shape == (0, 3)
    assert np.array_equal(s.as_array(), pk)
    # packets handed over in the reference's 24 x 16 x 3 shape are the same bytes
    g = tmp_path / "b.icppkt"
    scans.write_packets(str(g), pk.reshape(20, 24, 16, 3), names=[f"packet_{k:04d}" for k in range(20)])
    assert np.array_equal(scans.PacketFile(str(g)).as_array(), pk)
    # malformed files are refused, not misread
    bad = tmp_path / "bad.icppkt"
    bad.write_bytes(open(f, "rb").read()[:5000])
    with pytest.raises(ValueError):
        scans.PacketFile(str(bad))
    bad.write_bytes(open(f, "rb").read()[:400])  # cut inside the dataset table
    with pytest.raises(ValueError):
        scans.PacketFile(str(bad))
    with pytest.raises(ValueError):
        scans.PacketFile(os.path.join(GOLDEN, "001.txt"))
    with pytest.raises(ValueError):
        scans.write_packets(str(bad), pk[:, :100])


def test_scan2d_text_round_trip(tmp_path):
    from icp_rust_amd import scans

    p = load_scan2d(os.path.join(GOLDEN, "017.txt"))
    scans.save_scan2d(str(tmp_path / "a.txt"), p)
    assert np.array_equal(load_scan2d(str(tmp_path / "a.txt")).view(np.uint64), p.view(np.uint64))


def test_scan3d_from_a_packet_file_equals_the_in_memory_stream_with_and_without_pipelining(tmp_path):
    from icp_rust_amd import scans

    pk = synth.synthetic_scan3d_packets(32)
    f = tmp_path / "s.icppkt"
    scans.write_packets(str(f), pk)
    a, _, pa = harness.run_scan3d(pk, step=8, max_iter=2, icp_factory=OracleIcp(3), pipeline=False)
    t = []
    b, _, pb = harness.run_scan3d(scans.PacketFile(str(f)), step=8, max_iter=2, icp_factory=lambda d: OracleIcp(3)(d),
                                  pipeline=True, timings=t)
    assert len(a) == len(b) == len(t) == 4
    for x, y in zip(a, b):
        assert np.array_equal(x.as_array(), y.as_array())
    assert np.array_equal(pa, pb)


@pytest.mark.gpu
def test_pipelined_scan3d_loop_on_gpu_equals_the_serial_loop_bit_for_bit(tmp_path):
    from icp_rust_amd import scans

    pk = synth.synthetic_scan3d_packets(6 * 75)
    f = tmp_path / "s.icppkt"
    scans.write_packets(str(f), pk)
    serial, _, _ = harness.run_scan3d(pk, max_iter=20, pipeline=False)
    piped, _, _ = harness.run_scan3d(scans.PacketFile(str(f)), max_iter=20, pipeline=True)
    assert len(serial) == len(piped) == 6
    for a, b in zip(serial, piped):
        assert np.array_equal(a.as_array(), b.as_array())
    assert harness.main(["write-synth", str(tmp_path / "w.icppkt"), "--frames", "3"]) == 0
    assert harness.main(["scan3d", str(tmp_path / "w.icppkt"), "--max-iter", "3"]) == 0
