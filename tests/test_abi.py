"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header
declares, the host-only pose algebra agrees with the oracle bit for bit, and the compute
entry points fail loudly (no CPU fallback) when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import icp_rust_amd as I
import oracle_ffi as O
from icp_rust_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    I.build()


def header_symbols(names=("icp_mi355x.h", "icp_mi355x_debug.h")):
    """every function the C ABI declares: the drop-in boundary and its observability companion"""
    out = set()
    for name in names:
        text = open(os.path.join(ROOT, "include", name)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out |= set(re.findall(r"\b(icp_[a-z0-9_]+)\s*\(", text))
    return sorted(out)


def test_library_exports_every_declared_symbol():
    L = C.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/icp_mi355x.h but not exported"
    # and the Python binding covers exactly the headers
    assert sorted(_lib.SIGNATURES) == syms
    # ... and the library exports NOTHING beyond them: no C++ internals, no kernels' host stubs (csrc/exports.map)
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if " T " in line)
    assert exported == syms, sorted(set(exported) ^ set(syms))
    # the boundary proper carries no counters
    assert not [s for s in header_symbols(("icp_mi355x.h",)) if s.endswith("_counters") or s.startswith("icp_profile_")]


def test_abi_version_and_status_strings():
    import __graft_entry__ as G

    assert I.lib().icp_abi_version() == G.header_abi_version()
    for s in range(8):
        assert I.lib().icp_status_string(s)


def test_pose_algebra_matches_oracle_bit_for_bit():
    rng = np.random.default_rng(11)
    for _ in range(200):
        p, q = rng.normal(size=3) * [10, 10, 2], rng.normal(size=3)
        if rng.random() < 0.1:
            p[2] = 0.0
        T, U = I.Transform(p), I.Transform(q)
        oT, oU = O.transform_new(p), O.transform_new(q)
        assert np.array_equal(T.as_array(), oT.as_array())
        assert np.array_equal((T * U).as_array(), O.transform_mul(oT, oU).as_array())
        assert np.array_equal(T.inverse().as_array(), O.transform_inverse(oT).as_array())
        x = rng.normal(size=2) * 50
        assert np.array_equal(T.transform(x), O.transform_apply(oT, x))
        m = I.se2.exp(p)
        assert np.array_equal(I.se2.log(m), np.array(_orc_se2_log(m)))


def _orc_se2_log(m):
    out = np.zeros(3)
    mm = np.ascontiguousarray(m, dtype=np.float64)
    dp = C.POINTER(C.c_double)
    O.lib().orc_se2_log(mm.ctypes.data_as(dp), out.ctypes.data_as(dp))
    return out


def test_inverse3x3_none_cases():  # linalg.rs:52-60
    out = np.zeros(9)
    dp = C.POINTER(C.c_double)
    z = np.zeros(9)
    assert I.lib().icp_inverse3x3(z.ctypes.data_as(dp), out.ctypes.data_as(dp)) == _lib.NONE
    m = np.array([3.0, 1.0, 2.0, 6.0, 2.0, 4.0, 9.0, 9.0, 7.0])
    assert I.lib().icp_inverse3x3(m.ctypes.data_as(dp), out.ctypes.data_as(dp)) == _lib.NONE


def test_reduce_geometry():
    b1, t = I.reduce_geometry(1)
    assert b1 == 1 and t % 64 == 0
    assert I.reduce_geometry(t + 1) == (2, t)
    # up to 2^20 points: a block per `t` points, at most 256 (one per CU); beyond: a block per 8 t points (the tree grows
    # with the cloud -- a rank of N x 1M points owns 256 blocks), at most 2 048 (eight ranks' 1M each)
    assert I.reduce_geometry(10**6)[0] == min((10**6 + t - 1) // t, 256) and I.reduce_geometry(1 << 20)[0] == 256
    assert I.reduce_geometry((1 << 20) + 1)[0] == 257 and I.reduce_geometry(8 << 20)[0] == 2048
    assert I.reduce_geometry(10**9)[0] == 2048


@pytest.mark.skipif(I.lib().icp_device_count() > 0, reason="a GPU is present")
def test_compute_fails_loudly_without_a_gpu():
    pts = np.random.default_rng(0).normal(size=(10, 2))
    with pytest.raises(I.IcpError) as e:
        I.Icp2d(pts)
    assert e.value.status == _lib.NO_DEVICE
    T = I.Transform()
    for fn in (I.error, I.huber_error, I.gauss_newton_update, I.weighted_gauss_newton_update):
        with pytest.raises(I.IcpError) as e:
            fn(T, pts, pts)
        assert e.value.status == _lib.NO_DEVICE
    with pytest.raises(I.IcpError):
        I.estimate_transform(pts, pts)


def test_product_never_touches_the_oracle():
    """No file of the product package or the C/HIP sources mentions oracle/."""
    pkg = os.path.join(ROOT, "icp_rust_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                text = open(os.path.join(d, f)).read()
                assert "oracle_ffi" not in text and "icp_oracle" not in text and "orc_" not in text, f


def test_driver_entry_build_is_green():
    """The driver runs __graft_entry__.build() on the CPU box every round: run exactly that, so an
    ABI bump (or anything else build() asserts) cannot pass pytest and fail the driver."""
    import __graft_entry__ as G

    G.build()


@pytest.mark.gpu
def test_driver_entry_smoke_is_green():
    """The driver runs __graft_entry__.smoke() on the GPU box at round end (the one oracle
    comparison outside pytest; loop it checks: /root/reference/src/lib.rs:148-173)."""
    import __graft_entry__ as G

    G.smoke()
