"""The reference's own unit tests for the path (src/lib.rs:267-595), restated against the
GPU implementation through the host mirror of its API: same inputs, same assertions."""
import numpy as np
import pytest

import icp_rust_amd as I
from icp_rust_amd import Icp2d, Icp3d, Transform
from test_oracle_kat import L_SHAPE_2D, WGN_NOISE, WGN_SRC

pytestmark = pytest.mark.gpu


def V(*v):
    return np.array(v, dtype=np.float64)


def apply_many(T, pts):
    return np.array([T.transform(p) for p in pts]).reshape(-1, 2)


def transform_xy(T, p):  # src/lib.rs:52-57
    d = T.transform(p[:2])
    return V(d[0], d[1], p[2])


def test_residual():  # lib.rs:267-274
    T = Transform.new(V(-10., 20., 0.01))
    src = V(7., 8.)
    dst = T.transform(src)
    assert np.array_equal(I.residual(T, src, dst), np.zeros(2))


def test_error():  # lib.rs:276-297 (the GPU sum uses a tree, not a left fold: 1 ulp-level)
    src = np.array([[-6., 9.], [-1., 9.], [-4., -4.]])
    dst = np.array([[-4., 4.], [0., 3.], [-3., -8.]])
    T = Transform.new(V(10., 20., 0.01))
    r = [I.residual(T, s, d) for s, d in zip(src, dst)]
    expected = r[0].dot(r[0]) + r[1].dot(r[1]) + r[2].dot(r[2])
    assert abs(I.error(T, src, dst) - expected) <= 4 * np.spacing(expected)


def test_gauss_newton_update_input_size():  # lib.rs:299-318
    T = Transform.new(V(10.0, 30.0, -0.15))
    assert I.gauss_newton_update(T, np.zeros((0, 2)), np.zeros((0, 2))) is None
    src = np.array([[-8.89304516, 0.54202289]])
    assert I.gauss_newton_update(T, src, apply_many(T, src)) is None
    src = np.array([[-8.89304516, 0.54202289], [-4.03198385, -2.81807802]])
    assert I.gauss_newton_update(T, src, apply_many(T, src)) is not None


def test_gauss_newton_update():  # lib.rs:320-351
    true_param = V(10.0, 30.0, -0.15)
    initial_param = true_param + V(0.3, -0.5, 0.001)
    Tt, Ti = Transform.new(true_param), Transform.new(initial_param)
    src = np.array([[-8.76116663, 3.50338231], [-5.21184804, -1.91561705], [6.63141168, 4.8915293],
                    [-2.29215281, -4.72658399], [6.81352587, -0.81624617]])
    dst = apply_many(Tt, src)
    update = I.gauss_newton_update(Ti, src, dst)
    assert update is not None
    Tu = Transform.new(initial_param + update)
    assert I.error(Tu, src, dst) < I.error(Ti, src, dst) * 0.01


def test_weighted_gauss_newton_update_input_size():  # lib.rs:353-401
    T = Transform.new(V(10.0, 30.0, -0.15))
    wgn = I.weighted_gauss_newton_update
    assert wgn(T, np.zeros((0, 2)), np.zeros((0, 2))) is None
    src = np.array([[-8.89304516, 0.54202289]])
    assert wgn(T, src, apply_many(T, src)) is None
    src = np.array([[-8.89304516, 0.54202289], [-4.03198385, -2.81807802]])
    assert wgn(T, src, apply_many(T, src)) is None
    src = np.array([[-8.89304516, 0.54202289], [-4.03198385, -2.81807802], [-4.03198385, -2.81807802]])
    assert wgn(T, src, apply_many(T, src)) is None
    src = np.array([[-8.89304516, 0.54202289], [-4.03198385, -2.81807802], [4.40356349, -9.43358563]])
    assert wgn(T, src, apply_many(T, src)) is None


def test_weighted_gauss_newton_update_zero_x_diff():  # lib.rs:403-427
    src = np.array([[0.0, 0.0], [0.0, 0.1], [0.0, 0.2], [0.0, 0.3], [0.0, 0.4], [0.0, 0.5]])
    dst = apply_many(Transform.new(V(0.00, 0.01, 0.00)), src)
    assert I.weighted_gauss_newton_update(Transform.new(V(0., 0., 0.)), src, dst) is None


def test_weighted_gauss_newton_update():  # lib.rs:429-507
    true_param = V(10.0, 30.0, -0.15)
    initial_param = true_param + V(0.3, -0.5, 0.001)
    Tt, Ti = Transform.new(true_param), Transform.new(initial_param)
    src = WGN_SRC
    dst = apply_many(Tt, src) + WGN_NOISE
    update = I.weighted_gauss_newton_update(Ti, src, dst)
    assert update is not None
    Tu = Transform.new(initial_param + update)
    e0 = I.error(Ti, src, dst)
    assert I.error(Tu, src, dst) < e0 * 0.1
    Te = I.estimate_transform(src, dst)
    assert I.error(Te, src, dst) < e0 * 0.001


def test_icp_3dscan():  # lib.rs:509-551
    src = np.concatenate([L_SHAPE_2D, np.array([[2.0]] * 11 + [[1.0]] * 10)], axis=1)
    Tt = Transform.new(V(0.01, 0.01, -0.02))
    dst = np.array([transform_xy(Tt, p) for p in src])
    noise = Transform.new(V(0.05, 0.010, 0.010))
    icp = Icp3d(dst)
    pred = icp.estimate(src, noise * Tt, 20)
    for sp, dp_true in zip(src, dst):
        assert I.norm(transform_xy(pred, sp) - dp_true) < 1e-3


def test_icp_2dscan():  # lib.rs:553-595
    src = L_SHAPE_2D
    Tt = Transform.new(V(0.01, 0.01, -0.02))
    dst = apply_many(Tt, src)
    noise = Transform.new(V(0.05, 0.010, 0.010))
    icp = Icp2d(dst)
    pred = icp.estimate(src, noise * Tt, 20)
    for sp, dp_true in zip(src, dst):
        assert I.norm(pred.transform(sp) - dp_true) < 1e-3
