"""Frozen oracle outputs (tests/golden/self_golden.json, made by tests/golden/make_self_golden.py)."""
import json
import os

import numpy as np
import pytest

import icp_rust_amd as I
import oracle_ffi as O
from icp_rust_amd.scans import load_scan2d

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
G = json.load(open(os.path.join(GOLDEN, "self_golden.json")))


def _checksum(idx):
    return int(np.bitwise_xor.reduce(idx.astype(np.uint64) * np.arange(1, len(idx) + 1, dtype=np.uint64)))


def test_oracle_reproduces_the_frozen_scan2d_trajectory():
    src = load_scan2d(os.path.join(GOLDEN, "scans2d", "001.txt"))
    T = O.transform_identity()
    for fr in G["scan2d"]:
        dst = load_scan2d(os.path.join(GOLDEN, "scans2d", f"{fr['frame']:03d}.txt"))
        rc, T, idx, inner = O.icp_estimate(2, dst, src, T, 20, use_kdtree=True)
        assert rc == O.OK
        assert [repr(float(x)) for x in T.as_array()] == fr["pose"]
        assert [int(x) for x in inner] == fr["inner_iters"]
        assert _checksum(idx) == fr["idx_checksum"]


@pytest.mark.gpu
def test_gpu_matches_the_frozen_scan2d_trajectory():
    src = load_scan2d(os.path.join(GOLDEN, "scans2d", "001.txt"))
    T = I.Transform()
    oT = O.transform_identity()
    for fr in G["scan2d"]:
        dst = load_scan2d(os.path.join(GOLDEN, "scans2d", f"{fr['frame']:03d}.txt"))
        T, idx, inner = I.Icp2d(dst).estimate(src, T, 20, return_info=True)
        want = np.array([float(x) for x in fr["pose"]])
        assert np.max(np.abs(T.as_array() - want)) <= 1e-5 * max(1.0, np.max(np.abs(want)))  # north_star
        assert [int(x) for x in inner] == fr["inner_iters"]
        # indices: the oracle's (whose checksum is the frozen one) up to identical-coordinate duplicates --
        # the scans hold repeated (0, 0) returns, which tie only with each other
        rc, oT, oidx, _ = O.icp_estimate(2, dst, src, oT, 20, use_kdtree=True)
        assert rc == O.OK and _checksum(oidx) == fr["idx_checksum"]
        diff = np.nonzero(idx != oidx)[0]
        assert len(diff) <= 8
        assert all(np.array_equal(dst[idx[i]], dst[oidx[i]]) for i in diff)


@pytest.mark.gpu
def test_gpu_matches_the_frozen_l_shape_cases():
    L2 = np.array([[0.0, v] for v in (0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0)] +
                  [[v, 0.0] for v in (0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0)])
    Tt = I.Transform([0.01, 0.01, -0.02])
    Ti = I.Transform([0.05, 0.010, 0.010]) * Tt
    dst2 = np.array([Tt.transform(p) for p in L2])
    got2 = I.Icp2d(dst2).estimate(L2, Ti, 20)
    want2 = np.array([float(x) for x in G["l_shape_2d"]["pose"]])
    assert np.max(np.abs(got2.as_array() - want2)) <= 1e-5
    src3 = np.concatenate([L2, np.array([[2.0]] * 11 + [[1.0]] * 10)], axis=1)
    dst3 = np.array([[*Tt.transform(p[:2]), p[2]] for p in src3])
    got3 = I.Icp3d(dst3).estimate(src3, Ti, 20)
    want3 = np.array([float(x) for x in G["l_shape_3d"]["pose"]])
    assert np.max(np.abs(got3.as_array() - want3)) <= 1e-5
