"""ctypes binding of the CPU oracle (oracle/icp_oracle.{h,c}).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package (icp_rust_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_ORACLE_DIR, "build", "libicp_oracle.so")


class Pose(C.Structure):
    """orc_pose: Transform{rot, t} with the rotation column-major (src/transform.rs:6-10)."""

    _fields_ = [(n, C.c_double) for n in ("r00", "r10", "r01", "r11", "tx", "ty")]

    def as_array(self):
        return np.array([self.r00, self.r10, self.r01, self.r11, self.tx, self.ty])

    @staticmethod
    def from_array(a):
        return Pose(*[float(x) for x in a])


class IcpOpts(C.Structure):
    _fields_ = [("use_kdtree", C.c_int), ("sum_mode", C.c_int), ("reduce_blocks", C.c_int),
                ("reduce_threads", C.c_int)]


OK, NONE, EMPTY_DST, NAN = 0, 1, 2, 3


def build_oracle():
    src = [os.path.join(_ORACLE_DIR, f) for f in ("icp_oracle.c", "icp_oracle.h", "Makefile")]
    src.append(os.path.join(_ROOT, "include", "icp_trig.h"))
    if (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build_oracle())
    dp = C.POINTER(C.c_double)
    u32p = C.POINTER(C.c_uint32)
    pp = C.POINTER(Pose)
    sz = C.c_size_t

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("orc_sin", C.c_double, C.c_double)
    sig("orc_cos", C.c_double, C.c_double)
    sig("orc_so2_exp", None, C.c_double, dp)
    sig("orc_so2_log", C.c_double, dp)
    sig("orc_se2_calc_rt", None, dp, pp)
    sig("orc_se2_exp", None, dp, dp)
    sig("orc_se2_log", None, dp, dp)
    sig("orc_se2_get_rt", None, dp, dp, dp)
    sig("orc_transform_new", None, dp, pp)
    sig("orc_transform_identity", None, pp)
    sig("orc_transform_apply", None, pp, dp, dp)
    sig("orc_transform_inverse", None, pp, pp)
    sig("orc_transform_mul", None, pp, pp, pp)
    sig("orc_transform_xy", None, pp, dp, dp)
    sig("orc_norm_squared", C.c_double, dp, sz, sz)
    sig("orc_norm", C.c_double, dp, sz, sz)
    sig("orc_huber_rho", C.c_double, C.c_double, C.c_double)
    sig("orc_huber_drho", C.c_double, C.c_double, C.c_double)
    sig("orc_inverse3x3", C.c_int, dp, dp)
    sig("orc_median", C.c_int, dp, sz, dp)
    sig("orc_mad", C.c_int, dp, sz, dp)
    sig("orc_standard_deviation", C.c_int, dp, sz, dp)
    sig("orc_calc_stddevs", C.c_int, dp, sz, sz, dp)
    sig("orc_residual", None, pp, dp, dp, dp)
    sig("orc_error", C.c_double, pp, dp, dp, sz)
    sig("orc_huber_error", C.c_double, pp, dp, dp, sz)
    sig("orc_gauss_newton_update", C.c_int, pp, dp, dp, sz, dp)
    sig("orc_weighted_gauss_newton_update", C.c_int, pp, dp, dp, sz, dp)
    sig("orc_weighted_gauss_newton_update_tree", C.c_int, pp, dp, dp, sz, C.c_int, C.c_int, dp, dp)
    sig("orc_estimate_transform", C.c_int, dp, dp, sz, pp)
    sig("orc_nn_brute", C.c_int, dp, sz, C.c_int, dp, sz, u32p)
    sig("orc_kdtree_build", C.c_void_p, dp, sz, C.c_int)
    sig("orc_kdtree_free", None, C.c_void_p)
    sig("orc_kdtree_search", C.c_int, C.c_void_p, dp, sz, u32p)
    sig("orc_set_threads", None, C.c_int)
    sig("orc_icp_estimate", C.c_int, C.c_int, dp, sz, dp, sz, pp, sz, C.POINTER(IcpOpts), pp, u32p,
        u32p)
    sig("orc_icp_estimate_tree", C.c_int, C.c_void_p, dp, sz, dp, sz, pp, sz, C.POINTER(IcpOpts), pp,
        u32p, u32p)
    sig("orc_wgn_tree_partials", C.c_int, pp, dp, dp, sz, C.c_int, C.c_int, dp)
    sig("orc_wgn_tree_fold", C.c_int, dp, C.c_int, C.c_int, dp, dp, dp)
    sig("orc_p2pl_normals", C.c_int, dp, sz, C.c_int, dp)
    sig("orc_p2pl_normals_range", C.c_int, dp, sz, sz, C.c_int, dp)
    sig("orc_p2pl_estimate", C.c_int, C.c_void_p, dp, sz, dp, dp, sz, pp, sz, pp, u32p, u32p)
    _lib = L
    return L


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def _u32(n):
    a = np.zeros(max(int(n), 1), dtype=np.uint32)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint32))


# ---- thin pythonic wrappers (numpy in / numpy out) ---------------------------------


def transform_new(param):
    p = Pose()
    _, pp_ = _d(param)
    lib().orc_transform_new(pp_, C.byref(p))
    return p


def transform_identity():
    p = Pose()
    lib().orc_transform_identity(C.byref(p))
    return p


def transform_apply(T, pt):
    out = np.zeros(2)
    _, a = _d(pt)
    lib().orc_transform_apply(C.byref(T), a, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def transform_apply_many(T, pts):
    pts = np.asarray(pts, dtype=np.float64)
    return np.array([transform_apply(T, p) for p in pts]).reshape(-1, 2)


def transform_xy(T, pt):
    out = np.zeros(3)
    _, a = _d(pt)
    lib().orc_transform_xy(C.byref(T), a, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def transform_inverse(T):
    o = Pose()
    lib().orc_transform_inverse(C.byref(T), C.byref(o))
    return o


def transform_mul(a, b):
    o = Pose()
    lib().orc_transform_mul(C.byref(a), C.byref(b), C.byref(o))
    return o


def residual(T, s, d):
    out = np.zeros(2)
    _, sp = _d(s)
    _, dp_ = _d(d)
    lib().orc_residual(C.byref(T), sp, dp_, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def error(T, a, b):
    a, ap = _d(a)
    b, bp = _d(b)
    return lib().orc_error(C.byref(T), ap, bp, a.size // 2)


def huber_error(T, a, b):
    a, ap = _d(a)
    b, bp = _d(b)
    return lib().orc_huber_error(C.byref(T), ap, bp, a.size // 2)


def gauss_newton_update(T, a, b):
    a, ap = _d(a)
    b, bp = _d(b)
    out = np.zeros(3)
    rc = lib().orc_gauss_newton_update(C.byref(T), ap, bp, a.size // 2,
                                       out.ctypes.data_as(C.POINTER(C.c_double)))
    return rc, out


def weighted_gauss_newton_update(T, a, b):
    a, ap = _d(a)
    b, bp = _d(b)
    out = np.zeros(3)
    rc = lib().orc_weighted_gauss_newton_update(C.byref(T), ap, bp, a.size // 2,
                                                out.ctypes.data_as(C.POINTER(C.c_double)))
    return rc, out


def weighted_gauss_newton_update_tree(T, a, b, blocks, threads):
    a, ap = _d(a)
    b, bp = _d(b)
    out = np.zeros(3)
    err = C.c_double(0.0)
    rc = lib().orc_weighted_gauss_newton_update_tree(C.byref(T), ap, bp, a.size // 2, blocks, threads,
                                                     out.ctypes.data_as(C.POINTER(C.c_double)),
                                                     C.byref(err))
    return rc, out, err.value


def estimate_transform(a, b):
    a, ap = _d(a)
    b, bp = _d(b)
    o = Pose()
    n_applied = lib().orc_estimate_transform(ap, bp, a.size // 2, C.byref(o))
    return o, n_applied


def median(v):
    v, vp = _d(np.array(v, dtype=np.float64).copy())
    out = C.c_double()
    rc = lib().orc_median(vp, v.size, C.byref(out))
    return rc, out.value


def mad(v):
    v, vp = _d(np.array(v, dtype=np.float64).copy())
    out = C.c_double()
    rc = lib().orc_mad(vp, v.size, C.byref(out))
    return rc, out.value


def standard_deviation(v):
    v, vp = _d(np.array(v, dtype=np.float64).copy())
    out = C.c_double()
    rc = lib().orc_standard_deviation(vp, v.size, C.byref(out))
    return rc, out.value


def calc_stddevs(r):
    r, rp = _d(r)
    n, dim = r.shape
    out = np.zeros(dim)
    rc = lib().orc_calc_stddevs(rp, n, dim, out.ctypes.data_as(C.POINTER(C.c_double)))
    return rc, out


def inverse3x3(m):
    m, mp = _d(m)
    out = np.zeros((3, 3))
    rc = lib().orc_inverse3x3(mp, out.ctypes.data_as(C.POINTER(C.c_double)))
    return rc, out


def nn_brute(dst, q):
    dst, dp_ = _d(dst)
    q, qp = _d(q)
    dim = dst.shape[1] if dst.ndim == 2 else q.shape[1]
    idx, ip = _u32(q.shape[0])
    rc = lib().orc_nn_brute(dp_, dst.shape[0], dim, qp, q.shape[0], ip)
    return rc, idx[: q.shape[0]]


def set_threads(threads):
    """Split the queries of a kd search over `threads` host cores (1 = the reference's behaviour)."""
    lib().orc_set_threads(int(threads))


class KdTree:
    def __init__(self, dst):
        self.dst, dp_ = _d(dst)
        self.dim = self.dst.shape[1]
        self.h = lib().orc_kdtree_build(dp_, self.dst.shape[0], self.dim)

    def search(self, q):
        q, qp = _d(q)
        idx, ip = _u32(q.shape[0])
        rc = lib().orc_kdtree_search(self.h, qp, q.shape[0], ip)
        return rc, idx[: q.shape[0]]

    def estimate(self, src, init, max_iter, opts=None):
        src, sp = _d(src)
        _, dp_ = _d(self.dst)
        n = src.shape[0]
        idx, ip = _u32(n)
        inner, inp = _u32(max_iter)
        o = Pose()
        op = opts if opts is not None else IcpOpts(1, 0, 0, 0)
        rc = lib().orc_icp_estimate_tree(self.h, dp_, self.dst.shape[0], sp, n, C.byref(init), max_iter,
                                         C.byref(op), C.byref(o), ip, inp)
        return rc, o, idx[:n], inner[:max_iter]

    def __del__(self):
        try:
            lib().orc_kdtree_free(self.h)
        except Exception:
            pass


def icp_estimate(dim, dst, src, init, max_iter, use_kdtree=False, sum_mode=0, reduce_blocks=0,
                 reduce_threads=0):
    dst = np.ascontiguousarray(dst, dtype=np.float64).reshape(-1, dim)
    src = np.ascontiguousarray(src, dtype=np.float64).reshape(-1, dim)
    _, dp_ = _d(dst)
    _, sp = _d(src)
    n = src.shape[0]
    idx, ip = _u32(n)
    inner, inp = _u32(max_iter)
    o = Pose()
    opts = IcpOpts(int(use_kdtree), sum_mode, reduce_blocks, reduce_threads)
    rc = lib().orc_icp_estimate(dim, dp_, dst.shape[0], sp, n, C.byref(init), max_iter,
                                C.byref(opts), C.byref(o), ip, inp)
    return rc, o, idx[:n], inner[:max_iter]


# ---- EXTENSION checker (no reference counterpart): point-to-plane residuals -------------------
def p2pl_normals(dst, k):
    dst, dp_ = _d(dst)
    out = np.zeros((dst.shape[0], 3))
    rc = lib().orc_p2pl_normals(dp_, dst.shape[0], int(k), out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == OK
    return out


def p2pl_normals_update(dst, first, k, normals_prev):
    """normals of dst[first:] from the whole of dst; rows [0, first) copied from normals_prev"""
    dst, dp_ = _d(dst)
    out = np.zeros((dst.shape[0], 3))
    out[:first] = normals_prev[:first]
    rc = lib().orc_p2pl_normals_range(dp_, dst.shape[0], int(first), int(k), out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == OK
    return out


def p2pl_estimate(tree, normals, src, init, max_iter):
    src, sp = _d(src)
    normals, np_ = _d(normals)
    _, dp_ = _d(tree.dst)
    n = src.shape[0]
    idx, ip = _u32(n)
    inner, inp = _u32(max_iter)
    o = Pose()
    rc = lib().orc_p2pl_estimate(tree.h, dp_, tree.dst.shape[0], np_, sp, n, C.byref(init), max_iter, C.byref(o), ip, inp)
    return rc, o, idx[:n], inner[:max_iter]


# ---- halves of the tree-order evaluation (checking sharded evaluations) -------------------------
TREE_SUMS = 19  # per dimension 6 + 3 sums without 1 / sigma, + the Huber error (icp_oracle.c: NACC)


def wgn_tree_partials(T, a, b, blocks_local, threads):
    a, ap = _d(a)
    b, bp = _d(b)
    out = np.zeros((max(blocks_local, 1), TREE_SUMS))
    rc = lib().orc_wgn_tree_partials(C.byref(T), ap, bp, a.size // 2, blocks_local, threads,
                                     out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == OK
    return out[:blocks_local]


def wgn_tree_fold(partials, threads, stddevs):
    p, pp_ = _d(partials)
    sd, sp = _d(stddevs)
    delta = np.zeros(3)
    err = C.c_double(0.0)
    rc = lib().orc_wgn_tree_fold(pp_, p.shape[0], threads, sp, delta.ctypes.data_as(C.POINTER(C.c_double)), C.byref(err))
    return rc, delta, err.value
