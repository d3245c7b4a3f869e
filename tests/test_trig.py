"""include/icp_trig.h -- the sin / cos of the reference's no_std build (num-traits `libm` feature ->
the Rust libm crate, a port of musl's kernels), restated once and shared by the oracle, the library's
host code and its device code.  The crate's source is not under /root/reference, so the pin is the
published algorithm's accuracy: within 1 ulp of the C library on every branch, identical between the
three users, exact where exact values are known."""
import ctypes as C
import math

import numpy as np
import pytest

import icp_rust_amd as I
import oracle_ffi as O


def _ulps(a, b):
    ia = np.asarray(a, dtype=np.float64).view(np.int64).copy()
    ib = np.asarray(b, dtype=np.float64).view(np.int64).copy()
    ia[ia < 0] = np.iinfo(np.int64).min - ia[ia < 0]
    ib[ib < 0] = np.iinfo(np.int64).min - ib[ib < 0]
    return np.abs(ia - ib)


def _arguments():
    rng = np.random.default_rng(20240807)
    parts = [
        (rng.random(20000) - 0.5) * 2.0,                       # the kernels' own interval
        (rng.random(20000) - 0.5) * 20.0,                      # the one-round special cases up to 9 pi / 4
        (rng.random(20000) - 0.5) * 2000.0,                    # medium path
        (rng.random(20000) - 0.5) * 3.2e6,                     # up to the limit of the restated range
        np.ldexp(rng.random(5000) - 0.5, -rng.integers(0, 60, 5000)),  # tiny
        rng.integers(-1000, 1000, 20000) * (math.pi / 2) + (rng.random(20000) - 0.5) * 1e-6,   # cancellation
        rng.integers(-10, 10, 5000) * (math.pi / 4) + (rng.random(5000) - 0.5) * 1e-12,        # branch boundaries
        np.array([0.0, -0.0, 0.015, -0.015, 1e-9, 0.5, 1.0, math.pi / 4, math.pi / 2, math.pi, 3.0, 1e5, -1e6]),
    ]
    x = np.concatenate(parts)
    return x[np.abs(x) < 1.6e6]


def test_oracle_and_library_host_code_share_the_bits_and_stay_within_one_ulp_of_the_c_library():
    L, Lo = I.lib(), O.lib()
    x = _arguments()
    s_lib = np.array([L.icp_f64_sin(float(v)) for v in x])
    c_lib = np.array([L.icp_f64_cos(float(v)) for v in x])
    s_orc = np.array([Lo.orc_sin(float(v)) for v in x])
    c_orc = np.array([Lo.orc_cos(float(v)) for v in x])
    assert np.array_equal(s_lib.view(np.uint64), s_orc.view(np.uint64))
    assert np.array_equal(c_lib.view(np.uint64), c_orc.view(np.uint64))
    assert _ulps(s_lib, np.sin(x)).max() <= 1
    assert _ulps(c_lib, np.cos(x)).max() <= 1


def test_exactly_known_values_and_symmetries():
    L = I.lib()
    assert L.icp_f64_sin(0.0) == 0.0 and math.copysign(1.0, L.icp_f64_sin(-0.0)) == -1.0
    assert L.icp_f64_cos(0.0) == 1.0
    assert L.icp_f64_sin(1e-9) == 1e-9 and L.icp_f64_cos(1e-9) == 1.0  # |x| < 2^-26: sin x = x, cos x = 1
    for v in (0.015, 0.7, 2.0, 123.456, 9999.5):
        assert L.icp_f64_sin(-v) == -L.icp_f64_sin(v)
        assert L.icp_f64_cos(-v) == L.icp_f64_cos(v)
    assert math.isnan(L.icp_f64_sin(float("inf"))) and math.isnan(L.icp_f64_cos(float("nan")))
    # beyond the restated range the C library serves (documented deviation: Payne-Hanek is not restated)
    assert L.icp_f64_sin(1e9) == math.sin(1e9)


def test_transform_new_uses_it():
    T = I.Transform([0.3, -0.2, 0.015])
    L = I.lib()
    c, s = L.icp_f64_cos(0.015), L.icp_f64_sin(0.015)
    assert T.pose.r00 == c and T.pose.r10 == s and T.pose.r01 == -s and T.pose.r11 == c
    assert T.pose.tx == (s * 0.3 - (1.0 - c) * -0.2) / 0.015


@pytest.mark.gpu
def test_device_transform_new_equals_host_transform_new_bit_for_bit():
    rng = np.random.default_rng(5)
    n = 200_000
    p = np.ascontiguousarray(np.stack([rng.normal(size=n) * 3, rng.normal(size=n) * 3, _arguments()[:n] if len(_arguments()) >= n
                                       else np.resize(_arguments(), n)], axis=1))
    p[:10, 2] = 0.0  # the theta == 0 branch of se2::calc_rt
    out = np.zeros((n, 6))
    rc = I.lib().icp_transform_new_device(p.ctypes.data_as(C.POINTER(C.c_double)), n, C.c_void_p(out.ctypes.data), -1)
    assert rc == 0
    want = np.array([I.Transform(q).as_array() for q in p[:5000]])
    assert np.array_equal(out[:5000].view(np.uint64), want.view(np.uint64))
    # all of them against the oracle's Transform::new
    o = np.array([O.transform_new(q).as_array() for q in p[5000:25000]])
    assert np.array_equal(out[5000:25000].view(np.uint64), o.view(np.uint64))
