"""icp_rust_amd -- MI355X-native ICP registration core behind tier4/icp_rust's public API.

The compute lives in icp_rust_amd/lib/libicp_mi355x.so (hand-written HIP for gfx950 +
the C ABI of include/icp_mi355x.h); this package is the thin host-side mirror of the
reference interface used by the tests, the benchmark and the multi-GPU driver.
"""
from . import _lib
from ._lib import IcpError, Pose, build, lib  # noqa: F401
from .api import (Icp2d, Icp3d, IcpMulti, Transform, error, estimate_transform, gauss_newton_update,  # noqa: F401
                  huber_error, gn_path_counters, gn_loop_counters, gn_loop_timeouts, fixed_point_skips, run_ahead_counters, nn_cert_counters, norm, reduce_geometry, residual, residual_stddevs, se2, so2,
                  weighted_gauss_newton_update)

HUBER_K = 1.345
NN_AUTO, NN_BRUTE, NN_GRID = _lib.NN_AUTO, _lib.NN_BRUTE, _lib.NN_GRID
