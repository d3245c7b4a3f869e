"""Host-side mirror of the reference crate's public interface (src/lib.rs:12-26) over the
C ABI: same names, same argument meaning, same failure behaviour.

    Transform, Icp2d, Icp3d, residual, error, huber_error, estimate_transform,
    gauss_newton_update, weighted_gauss_newton_update, norm, se2, so2

`Option::None` becomes Python `None`; the reference's two panics (empty `dst`, NaN
residual) become `IcpError`.  All compute runs in libicp_mi355x.so on the GPU.
Point sets are numpy arrays (n x 2 / n x 3 float64, C order = `&[Vector2]`/`&[Vector3]`)
or CUDA torch tensors of the same shape, which are used in place (no copy).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import IcpError, Pose, check, lib  # noqa: F401

_dp = C.POINTER(C.c_double)


def _is_device_tensor(x):
    return hasattr(x, "data_ptr") and getattr(x, "is_cuda", False)


def _host(a, dim):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.size == 0:
        a = a.reshape(0, dim)
    if a.ndim != 2 or a.shape[1] != dim:
        raise ValueError(f"expected an (n, {dim}) float64 array, got {a.shape}")
    return a


def _ptr(a):
    return C.c_void_p(a.ctypes.data if a.size else None)


def _dev_points(t, dim, device, what="points"):
    """A device-tensor argument of the C ABI: contiguous float64 (n, dim) on the handle's GPU.  Anything
    else (float32, a strided slice such as pts[:, :3], a 2-column tensor for Icp3d, another GPU) would be
    read out of bounds or misread by the kernels, so it is refused here."""
    if not _is_device_tensor(t):
        raise ValueError(f"{what}: expected a CUDA tensor")
    if t.dim() != 2 or t.shape[1] != dim or str(t.dtype) != "torch.float64" or not t.is_contiguous():
        raise ValueError(f"{what}: expected a contiguous (n, {dim}) float64 CUDA tensor, got "
                         f"{tuple(t.shape)} {t.dtype} contiguous={t.is_contiguous()}")
    if device is not None and t.device.index != device:
        raise ValueError(f"{what}: tensor lives on cuda:{t.device.index}, the handle on cuda:{device}")
    return t


def _dev_index(t, n, device, what="idx"):
    """A device index buffer: contiguous 4-byte integers, at least n of them, on the handle's GPU."""
    if not _is_device_tensor(t):
        raise ValueError(f"{what}: expected a CUDA tensor")
    if t.dim() != 1 or t.element_size() != 4 or t.dtype.is_floating_point or not t.is_contiguous() or t.shape[0] < n:
        raise ValueError(f"{what}: expected a contiguous int32/uint32 CUDA tensor of length >= {n}")
    if device is not None and t.device.index != device:
        raise ValueError(f"{what}: tensor lives on cuda:{t.device.index}, the handle on cuda:{device}")
    return t


def _vec(v, n):
    a = np.ascontiguousarray(v, dtype=np.float64).reshape(-1)
    if a.size != n:
        raise ValueError(f"expected {n} values")
    return a


class Transform:
    """`icp::Transform` (src/transform.rs:6-51)."""

    __slots__ = ("pose",)

    def __init__(self, param=None):
        """Transform::new(&param) (src/transform.rs:13-16); no argument = identity."""
        self.pose = Pose()
        if param is None:
            lib().icp_transform_identity(C.byref(self.pose))
        else:
            p = _vec(param, 3)
            lib().icp_transform_new(p.ctypes.data_as(_dp), C.byref(self.pose))

    @staticmethod
    def new(param):
        return Transform(param)

    @staticmethod
    def identity():
        return Transform()

    @staticmethod
    def from_rt(rot, t):
        """rot: 2x2 (row-major numpy), t: 2 (src/transform.rs:18-20)."""
        r = np.asarray(rot, dtype=np.float64).reshape(2, 2)
        cm = np.array([r[0, 0], r[1, 0], r[0, 1], r[1, 1]])
        tt = _vec(t, 2)
        o = Transform()
        lib().icp_transform_from_rt(cm.ctypes.data_as(_dp), tt.ctypes.data_as(_dp), C.byref(o.pose))
        return o

    @staticmethod
    def from_pose(pose):
        o = Transform()
        C.memmove(C.byref(o.pose), C.byref(pose), C.sizeof(Pose))
        return o

    @property
    def rot(self):
        p = self.pose
        return np.array([[p.r00, p.r01], [p.r10, p.r11]])

    @property
    def t(self):
        return np.array([self.pose.tx, self.pose.ty])

    def transform(self, landmark):
        p = _vec(landmark, 2)
        out = np.zeros(2)
        lib().icp_transform_apply(C.byref(self.pose), p.ctypes.data_as(_dp), out.ctypes.data_as(_dp))
        return out

    def inverse(self):
        o = Transform()
        lib().icp_transform_inverse(C.byref(self.pose), C.byref(o.pose))
        return o

    def __mul__(self, rhs):
        o = Transform()
        lib().icp_transform_mul(C.byref(self.pose), C.byref(rhs.pose), C.byref(o.pose))
        return o

    def as_array(self):
        return np.array(self.pose.as_tuple())

    def __repr__(self):
        return f"Transform(rot={self.rot.tolist()}, t={self.t.tolist()})"


class _Se2:
    """`icp::se2` (src/se2.rs)."""

    @staticmethod
    def exp(param):
        p = _vec(param, 3)
        m = np.zeros(9)
        lib().icp_se2_exp(p.ctypes.data_as(_dp), m.ctypes.data_as(_dp))
        return m.reshape(3, 3)

    @staticmethod
    def log(transform):
        m = _vec(transform, 9)
        p = np.zeros(3)
        lib().icp_se2_log(m.ctypes.data_as(_dp), p.ctypes.data_as(_dp))
        return p

    @staticmethod
    def get_rt(transform):
        m = _vec(transform, 9)
        r = np.zeros(4)
        t = np.zeros(2)
        lib().icp_se2_get_rt(m.ctypes.data_as(_dp), r.ctypes.data_as(_dp), t.ctypes.data_as(_dp))
        return r.reshape(2, 2), t

    @staticmethod
    def calc_rt(param):
        T = Transform(param)
        return T.rot, T.t


class _So2:
    """`icp::so2` (src/so2.rs)."""

    @staticmethod
    def exp(theta):
        m = np.zeros(4)
        lib().icp_so2_exp(float(theta), m.ctypes.data_as(_dp))
        return np.array([[m[0], m[2]], [m[1], m[3]]])

    new_rotation2 = exp

    @staticmethod
    def log(rotation):
        r = np.asarray(rotation, dtype=np.float64).reshape(2, 2)
        cm = np.array([r[0, 0], r[1, 0], r[0, 1], r[1, 1]])
        return lib().icp_so2_log(cm.ctypes.data_as(_dp))


se2 = _Se2()
so2 = _So2()


def norm(matrix):
    """`icp::norm` (src/norm.rs:19-21)."""
    m = np.asarray(matrix, dtype=np.float64)
    m = np.asfortranarray(m.reshape(m.shape[0], -1))
    return lib().icp_norm(m.ctypes.data_as(_dp), m.shape[0], m.shape[1])


def residual(transform, src, dst):
    """`icp::residual` (src/lib.rs:34-36)."""
    return transform.transform(src) - _vec(dst, 2)


def _pairs(src, dst):
    a = _host(src, 2)
    b = _host(dst, 2)
    if a.shape != b.shape:
        raise ValueError("src and dst differ in length")  # debug_assert_eq!, src/lib.rs:223
    return a, b


def error(transform, src, dst):
    """`icp::error` (src/lib.rs:38-43)."""
    a, b = _pairs(src, dst)
    out = C.c_double()
    check(lib().icp_error(C.byref(transform.pose), _ptr(a), _ptr(b), a.shape[0], C.byref(out)), "error")
    return out.value


def huber_error(transform, src, dst):
    """`icp::huber_error` (src/lib.rs:45-50)."""
    a, b = _pairs(src, dst)
    out = C.c_double()
    check(lib().icp_huber_error(C.byref(transform.pose), _ptr(a), _ptr(b), a.shape[0], C.byref(out)),
          "huber_error")
    return out.value


def gauss_newton_update(transform, src, dst):
    """`icp::gauss_newton_update` (src/lib.rs:191-216) -> Param or None."""
    a, b = _pairs(src, dst)
    d = np.zeros(3)
    rc = check(lib().icp_gauss_newton_update(C.byref(transform.pose), _ptr(a), _ptr(b), a.shape[0],
                                             d.ctypes.data_as(_dp)), "gauss_newton_update",
               allow=(_lib.NONE,))
    return None if rc == _lib.NONE else d


def weighted_gauss_newton_update(transform, src, dst):
    """`icp::weighted_gauss_newton_update` (src/lib.rs:218-261) -> Param or None."""
    a, b = _pairs(src, dst)
    d = np.zeros(3)
    rc = check(lib().icp_weighted_gauss_newton_update(C.byref(transform.pose), _ptr(a), _ptr(b),
                                                      a.shape[0], d.ctypes.data_as(_dp)),
               "weighted_gauss_newton_update", allow=(_lib.NONE,))
    return None if rc == _lib.NONE else d


def residual_stddevs(transform, src, dst):
    """stats::calc_stddevs over the residuals (src/stats.rs:49-60 at src/lib.rs:236)."""
    a, b = _pairs(src, dst)
    s = np.zeros(2)
    rc = check(lib().icp_residual_stddevs(C.byref(transform.pose), _ptr(a), _ptr(b), a.shape[0],
                                          s.ctypes.data_as(_dp)), "residual_stddevs", allow=(_lib.NONE,))
    return None if rc == _lib.NONE else s


def estimate_transform(src, dst, return_inner_iters=False):
    """`icp::estimate_transform` (src/lib.rs:59-84)."""
    a, b = _pairs(src, dst)
    o = Transform()
    inner = C.c_uint32(0)
    check(lib().icp_estimate_transform(_ptr(a), _ptr(b), a.shape[0], C.byref(o.pose), C.byref(inner)),
          "estimate_transform")
    return (o, inner.value) if return_inner_iters else o


def reduce_geometry(n):
    b, t = C.c_int(), C.c_int()
    lib().icp_reduce_geometry(n, C.byref(b), C.byref(t))
    return b.value, t.value


def gn_path_counters(icp=None):
    """(window started, window missed, short pipeline, radix path, speculative searches confirmed,
    discarded) counts of a handle (default: the scratch handle behind the free functions)."""
    out = (C.c_uint64 * 6)()
    check(lib().icp_gn_path_counters(icp._h if icp is not None else None, out), "icp_gn_path_counters")
    return tuple(int(x) for x in out)


def gn_loop_counters(icp=None):
    """(launches, evaluations served, launches that handed an evaluation back) of the one-launch inner loop"""
    out = (C.c_uint64 * 3)()
    check(lib().icp_gn_loop_counters(icp._h if icp is not None else None, out), "icp_gn_loop_counters")
    return tuple(int(x) for x in out)


def gn_loop_timeouts(icp):
    """launches of the one-launch inner loop that gave up their bounded wait (not fully resident)"""
    out = C.c_uint64(0)
    check(lib().icp_gn_loop_timeouts(icp._h, C.byref(out)), "icp_gn_loop_timeouts")
    return int(out.value)


def fixed_point_skips(icp):
    """outer iterations not run because the pose had stopped moving (their inner counts are 0, the result the same bits)"""
    out = C.c_uint64(0)
    check(lib().icp_fixed_point_skips(icp._h, C.byref(out)), "icp_fixed_point_skips")
    return int(out.value)


def run_ahead_counters(icp):
    """(run-ahead searches whose device-derived pose the host confirmed bit for bit, ... that it did not)"""
    out = (C.c_uint64 * 2)()
    check(lib().icp_run_ahead_counters(icp._h, out), "icp_run_ahead_counters")
    return tuple(int(x) for x in out)


def nn_cert_counters(icp):
    """(searches that checked certificates so far, queries whose certificate failed in the last of them)"""
    out = (C.c_uint64 * 2)()
    check(lib().icp_nn_cert_counters(icp._h, out), "icp_nn_cert_counters")
    return int(out[0]), int(out[1])


class _Icp:
    DIM = 0

    def __init__(self, dst, device=-1, nn_mode=_lib.NN_AUTO):
        """Icp2d::new / Icp3d::new (src/lib.rs:97-102, 139-144)."""
        self._h = C.c_void_p()
        self._keep = None
        self._device = None      # GPU index, known once a device tensor has been seen
        self._own_stream = True  # the handle runs on its private streams (icp_set_stream not called)
        if _is_device_tensor(dst):
            dev = dst.device.index if device < 0 else device
            _dev_points(dst, self.DIM, dev, "dst")
            self._keep = dst  # borrowed for the handle's lifetime, like `&'a [Vector]`
            self._device = dev
            self.m = dst.shape[0]
            self._after_producer(dst)
            check(lib().icp_create_device(C.byref(self._h), self.DIM, C.c_void_p(dst.data_ptr()),
                                          self.m, dev), "icp_create_device")
        else:
            d = _host(dst, self.DIM)
            self.m = d.shape[0]
            check(lib().icp_create(C.byref(self._h), self.DIM, _ptr(d), self.m, device), "icp_create")
        if nn_mode != _lib.NN_AUTO:
            check(lib().icp_set_nn_mode(self._h, nn_mode), "icp_set_nn_mode")

    def _after_producer(self, t):
        """The handle's private streams are not ordered against the torch stream that produced a device
        tensor (include/icp_mi355x.h: "device inputs must be complete when the call is made"): wait for
        torch's current stream on that device.  Not needed -- and not done -- once icp_set_stream has put
        the handle on the caller's stream, where everything is ordered."""
        if self._own_stream:
            import torch

            torch.cuda.current_stream(t.device).synchronize()

    def _dev(self, t, what):
        if self._device is None:
            self._device = t.device.index
        _dev_points(t, self.DIM, self._device, what)
        self._after_producer(t)
        return t

    def _dev_pairs(self, t, what):
        if self._device is None:
            self._device = t.device.index
        _dev_points(t, 2, self._device, what)
        self._after_producer(t)
        return t

    def last_fold_order(self, n, with_cells=False):
        """The order in which the last estimate() call on this handle folded its sums over the `n` source
        points (icp_last_fold_order): perm[k] = caller's index of the k-th folded point (identity when the
        call took no cell-sorted snapshot); with_cells=True also returns the sort keys."""
        perm = np.zeros(max(n, 1), dtype=np.uint32)
        cell = np.zeros(max(n, 1), dtype=np.uint32)
        check(lib().icp_last_fold_order(self._h, n, C.c_void_p(perm.ctypes.data), C.c_void_p(cell.ctypes.data)),
              "icp_last_fold_order")
        perm = perm[:n].astype(np.int64)
        return (perm, cell[:n]) if with_cells else perm

    def sort_source_device(self, d_src, transform):
        """(sorted copy, permutation) of a device-resident source cloud in the fold order of an estimate call
        that starts at `transform` (icp_sort_source_device)."""
        import torch

        self._dev(d_src, "src")
        out = torch.empty_like(d_src)
        perm = torch.empty(max(d_src.shape[0], 1), dtype=torch.int32, device=d_src.device)
        check(lib().icp_sort_source_device(self._h, C.c_void_p(d_src.data_ptr()), d_src.shape[0],
                                           C.byref(transform.pose), C.c_void_p(out.data_ptr()),
                                           C.c_void_p(perm.data_ptr())), "icp_sort_source_device")
        check(lib().icp_synchronize(self._h), "icp_synchronize")  # (produced on the handle's stream)
        return out, perm[:d_src.shape[0]]

    # -- the reference's method ------------------------------------------------------
    def estimate(self, src, initial_transform, max_iter, return_info=False):
        """Icp2d::estimate / Icp3d::estimate (src/lib.rs:105-130, 148-173).  return_info=True also
        returns the last correspondence indices and the inner-iteration counts; "inner" only the
        counts (no index buffer, no device-to-host copy)."""
        o = Transform()
        inner = np.zeros(max(max_iter, 1), dtype=np.uint32)
        if _is_device_tensor(src):
            import torch

            self._dev(src, "src")
            n = src.shape[0]
            want_idx = bool(return_info) and return_info != "inner"
            idx = torch.empty(max(n, 1), dtype=torch.int32, device=src.device) if want_idx else None
            check(lib().icp_estimate_device(self._h, C.c_void_p(src.data_ptr()), n,
                                            C.byref(initial_transform.pose), max_iter, C.byref(o.pose),
                                            C.c_void_p(idx.data_ptr()) if want_idx else None,
                                            C.c_void_p(inner.ctypes.data)), "icp_estimate_device")
            if return_info == "inner":
                return o, inner[:max_iter]
            if return_info:
                return o, idx[:n].cpu().numpy().view(np.uint32), inner[:max_iter]
            return o
        s = _host(src, self.DIM)
        n = s.shape[0]
        # (the reference returns the transform alone: the last correspondences are copied back only on request --
        # 113 KB and a stream synchronisation per 28k-point frame otherwise)
        want_idx = bool(return_info) and return_info != "inner"
        idx = np.zeros(max(n, 1), dtype=np.uint32) if want_idx else None
        check(lib().icp_estimate(self._h, _ptr(s), n, C.byref(initial_transform.pose), max_iter,
                                 C.byref(o.pose), C.c_void_p(idx.ctypes.data) if want_idx else None,
                                 C.c_void_p(inner.ctypes.data)), "icp_estimate")
        if return_info == "inner":
            return o, inner[:max_iter]
        if return_info:
            return o, idx[:n], inner[:max_iter]
        return o

    def set_single_launch(self, enable=True):
        """small clouds: whole estimate in one launch (default on); off = the general host-driven path"""
        check(lib().icp_set_single_launch(self._h, int(bool(enable))), "icp_set_single_launch")

    def single_launch_counters(self):
        out = (C.c_uint64 * 3)()
        check(lib().icp_single_launch_counters(self._h, out), "icp_single_launch_counters")
        return tuple(int(x) for x in out)

    # -- stage-level access (device tensors), used by the sharded driver and the bench --
    def set_stream(self, stream_ptr):
        check(lib().icp_set_stream(self._h, C.c_void_p(stream_ptr)), "icp_set_stream")
        self._own_stream = False

    def use_own_stream(self):
        check(lib().icp_use_own_stream(self._h), "icp_use_own_stream")
        self._own_stream = True

    def correspond_device(self, d_src, transform, d_a, d_b, d_idx=None):
        n = self._dev(d_src, "d_src").shape[0]
        for t, what in ((d_a, "d_a"), (d_b, "d_b")):
            if t is not None and self._dev_pairs(t, what).shape[0] < n:
                raise ValueError(f"{what}: needs {n} rows")
        if d_idx is not None:
            _dev_index(d_idx, n, self._device, "d_idx")
        check(lib().icp_correspond_device(self._h, C.c_void_p(d_src.data_ptr()), d_src.shape[0],
                                          C.byref(transform.pose),
                                          C.c_void_p(d_a.data_ptr()) if d_a is not None else None,
                                          C.c_void_p(d_b.data_ptr()) if d_b is not None else None,
                                          C.c_void_p(d_idx.data_ptr()) if d_idx is not None else None),
              "icp_correspond_device")

    def materialize_pairs_device(self, d_src, transform, d_idx, d_a, d_b):
        n = self._dev(d_src, "d_src").shape[0]
        _dev_index(d_idx, n, self._device, "d_idx")
        for t, what in ((d_a, "d_a"), (d_b, "d_b")):
            if self._dev_pairs(t, what).shape[0] < n:
                raise ValueError(f"{what}: needs {n} rows")
        check(lib().icp_materialize_pairs_device(self._h, C.c_void_p(d_src.data_ptr()), d_src.shape[0],
                                                 C.byref(transform.pose), C.c_void_p(d_idx.data_ptr()),
                                                 C.c_void_p(d_a.data_ptr()), C.c_void_p(d_b.data_ptr())),
              "icp_materialize_pairs_device")

    def prepare_source_device(self, d_src, transform):
        self._dev(d_src, "d_src")
        check(lib().icp_prepare_source_device(self._h, C.c_void_p(d_src.data_ptr()), d_src.shape[0],
                                              C.byref(transform.pose)), "icp_prepare_source_device")

    def estimate_transform_device(self, d_a, d_b):
        if self._dev_pairs(d_a, "d_a").shape[0] != self._dev_pairs(d_b, "d_b").shape[0]:
            raise ValueError("d_a and d_b differ in length")  # debug_assert_eq!, src/lib.rs:223
        o = Transform()
        inner = C.c_uint32(0)
        check(lib().icp_estimate_transform_device(self._h, C.c_void_p(d_a.data_ptr()),
                                                  C.c_void_p(d_b.data_ptr()), d_a.shape[0],
                                                  C.byref(o.pose), C.byref(inner)),
              "icp_estimate_transform_device")
        return o, inner.value

    def nn_search_device(self, d_q, d_idx):
        _dev_index(d_idx, self._dev(d_q, "d_q").shape[0], self._device, "d_idx")
        check(lib().icp_nn_search_device(self._h, C.c_void_p(d_q.data_ptr()), d_q.shape[0],
                                         C.c_void_p(d_idx.data_ptr())), "icp_nn_search_device")

    def nn_search(self, q):
        """exact NN indices of host points (test/observability helper)."""
        import torch

        qq = torch.from_numpy(_host(q, self.DIM)).cuda()
        idx = torch.empty(max(qq.shape[0], 1), dtype=torch.int32, device=qq.device)
        self.nn_search_device(qq, idx)
        self.synchronize()
        return idx[: qq.shape[0]].cpu().numpy().view(np.uint32)

    # -- EXTENSION (not in the reference): a target cloud that grows, for scan-to-map ------
    def append(self, points, transform=None):
        """Append `points` to the target cloud, moved by `transform` first if given (exactly
        Transform::transform on xy, z kept).  Afterwards the handle behaves like a fresh
        Icp*::new on the concatenated cloud (include/icp_mi355x.h section 6)."""
        tp = C.byref(transform.pose) if transform is not None else None
        if _is_device_tensor(points):
            self._dev(points, "points")
            check(lib().icp_append_targets_device(self._h, C.c_void_p(points.data_ptr()), points.shape[0], tp),
                  "icp_append_targets_device")
        else:
            p = _host(points, self.DIM)
            check(lib().icp_append_targets(self._h, _ptr(p), p.shape[0], tp), "icp_append_targets")
        self._keep = None  # the cloud now lives in the handle's own storage
        self.m = self.target_count

    def append_counters(self):
        """(appends served by moving the search grid's sorted records, appends that rebuilt the grid)"""
        out = (C.c_uint64 * 2)()
        check(lib().icp_grid_append_counters(self._h, out), "icp_grid_append_counters")
        return int(out[0]), int(out[1])

    def reserve(self, capacity):
        check(lib().icp_reserve_targets(self._h, int(capacity)), "icp_reserve_targets")
        self._keep = None if capacity > self.m else self._keep

    @property
    def target_count(self):
        return int(lib().icp_target_count(self._h))

    def read_targets(self, first=0, count=None):
        count = self.target_count - first if count is None else count
        out = np.empty((count, self.DIM), dtype=np.float64)
        check(lib().icp_read_targets(self._h, first, count, C.c_void_p(out.ctypes.data)), "icp_read_targets")
        return out

    # -- EXTENSION (not in the reference): point-to-plane residuals, include/icp_mi355x.h section 7 --
    def compute_normals(self, k=10):
        """Unit normals of the target points from their k nearest targets (3-D handles)."""
        check(lib().icp_compute_target_normals(self._h, int(k)), "icp_compute_target_normals")

    def update_normals(self, k=10):
        """Normals for the targets appended since compute_normals / update_normals (from the cloud as it is
        now); the older targets keep theirs."""
        check(lib().icp_update_target_normals(self._h, int(k)), "icp_update_target_normals")

    def read_normals(self, first=0, count=None):
        count = self.target_count - first if count is None else count
        out = np.empty((count, 3), dtype=np.float64)
        check(lib().icp_read_target_normals(self._h, first, count, C.c_void_p(out.ctypes.data)),
              "icp_read_target_normals")
        return out

    def estimate_point_to_plane(self, src, initial_transform, max_iter, return_info=False):
        """Icp3d::estimate with the residual n_q . (T p - q) (extension; needs compute_normals())."""
        o = Transform()
        inner = np.zeros(max(max_iter, 1), dtype=np.uint32)
        if _is_device_tensor(src):
            import torch

            self._dev(src, "src")
            n = src.shape[0]
            idx = torch.empty(max(n, 1), dtype=torch.int32, device=src.device) if return_info else None
            check(lib().icp_estimate_point_to_plane_device(self._h, C.c_void_p(src.data_ptr()), n,
                                                           C.byref(initial_transform.pose), max_iter, C.byref(o.pose),
                                                           C.c_void_p(idx.data_ptr()) if return_info else None,
                                                           C.c_void_p(inner.ctypes.data)),
                  "icp_estimate_point_to_plane_device")
            return (o, idx[:n].cpu().numpy().view(np.uint32), inner[:max_iter]) if return_info else o
        s = _host(src, self.DIM)
        n = s.shape[0]
        idx = np.zeros(max(n, 1), dtype=np.uint32)
        check(lib().icp_estimate_point_to_plane(self._h, _ptr(s), n, C.byref(initial_transform.pose), max_iter,
                                                C.byref(o.pose), C.c_void_p(idx.ctypes.data),
                                                C.c_void_p(inner.ctypes.data)), "icp_estimate_point_to_plane")
        return (o, idx[:n], inner[:max_iter]) if return_info else o

    def profile_enable(self, every=1):
        """Time every `every`-th NN search launch with HIP events (0 / False: off)."""
        check(lib().icp_profile_enable(self._h, int(every)), "icp_profile_enable")

    def profile_read(self):
        """(summed NN-kernel device time in ms, launches) since the last read."""
        ms, k = C.c_double(), C.c_uint64()
        check(lib().icp_profile_read(self._h, C.byref(ms), C.byref(k)), "icp_profile_read")
        return ms.value, k.value

    def synchronize(self):
        check(lib().icp_synchronize(self._h), "icp_synchronize")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().icp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Icp2d(_Icp):
    """`icp::Icp2d` (src/lib.rs:91-131)."""

    DIM = 2


class Icp3d(_Icp):
    """`icp::Icp3d` (src/lib.rs:133-174): 3-D nearest neighbours, SE(2) pose on the xy-plane."""

    DIM = 3


class IcpMulti:
    """Icp{2,3}d over several GPUs from one process (icp_create_multi): same result as one GPU, bit for
    bit.  `device_ids` may repeat a device (virtual ranks)."""

    def __init__(self, dst, device_ids, dim=3):
        self.dim = dim
        d = _host(dst, dim)
        ids = (C.c_int * len(device_ids))(*[int(x) for x in device_ids])
        self._h = C.c_void_p()
        check(lib().icp_create_multi(C.byref(self._h), dim, _ptr(d), d.shape[0], ids, len(device_ids)),
              "icp_create_multi")

    def estimate(self, src, initial_transform, max_iter, return_info=False):
        s = _host(src, self.dim)
        n = s.shape[0]
        o = Transform()
        idx = np.zeros(max(n, 1), dtype=np.uint32)
        inner = np.zeros(max(max_iter, 1), dtype=np.uint32)
        check(lib().icp_multi_estimate(self._h, _ptr(s), n, C.byref(initial_transform.pose), max_iter, C.byref(o.pose),
                                       C.c_void_p(idx.ctypes.data), C.c_void_p(inner.ctypes.data)), "icp_multi_estimate")
        return (o, idx[:n], inner[:max_iter]) if return_info else o

    def append(self, points, transform=None):
        """EXTENSION (icp_multi_append_targets): every rank appends the points, moved by `transform`, to its replica of
        the target cloud"""
        p = _host(points, self.dim)
        check(lib().icp_multi_append_targets(self._h, _ptr(p), p.shape[0],
                                             C.byref(transform.pose) if transform is not None else None),
              "icp_multi_append_targets")

    @property
    def target_count(self):
        return int(lib().icp_multi_target_count(self._h))

    def counters(self):
        out = (C.c_uint64 * 2)()
        check(lib().icp_multi_counters(self._h, out), "icp_multi_counters")
        return int(out[0]), int(out[1])

    def compute_target_normals(self, k=8):
        """EXTENSION (icp_multi_compute_target_normals): every rank computes the normals of its replica of the target cloud"""
        check(lib().icp_multi_compute_target_normals(self._h, int(k)), "icp_multi_compute_target_normals")

    def update_target_normals(self, k=8):
        check(lib().icp_multi_update_target_normals(self._h, int(k)), "icp_multi_update_target_normals")

    def estimate_point_to_plane(self, src, initial_transform, max_iter, return_info=False):
        """EXTENSION (icp_multi_estimate_point_to_plane): search sharded over the ranks, inner loop replicated"""
        s = _host(src, 3)
        n = s.shape[0]
        o = Transform()
        idx = np.zeros(max(n, 1), dtype=np.uint32)
        inner = np.zeros(max(max_iter, 1), dtype=np.uint32)
        check(lib().icp_multi_estimate_point_to_plane(self._h, _ptr(s), n, C.byref(initial_transform.pose), max_iter,
                                                      C.byref(o.pose), C.c_void_p(idx.ctypes.data), C.c_void_p(inner.ctypes.data)),
              "icp_multi_estimate_point_to_plane")
        return (o, idx[:n], inner[:max_iter]) if return_info else o

    def loop_counters(self):
        """(launches per rank, evaluations served, launches that handed an evaluation back) of the one-launch inner loop"""
        out = (C.c_uint64 * 3)()
        check(lib().icp_multi_loop_counters(self._h, out), "icp_multi_loop_counters")
        return tuple(int(x) for x in out)

    def pipe_iterations(self):
        """outer iterations served by the pipelined sharded evaluation (csrc/pipe.hip) over the life of the object"""
        out = C.c_uint64(0)
        check(lib().icp_multi_pipe_iterations(self._h, C.byref(out)), "icp_multi_pipe_iterations")
        return int(out.value)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().icp_destroy_multi(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
