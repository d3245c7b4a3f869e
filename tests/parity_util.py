"""Helpers of the -m gpu parity tests (test infrastructure)."""
import numpy as np

import icp_rust_amd as I
import oracle_ffi as O


def oracle_in_device_order(icp, dim, dst, src, init, max_iter, use_kdtree=True):
    """The oracle's Icp::estimate with its sums folded exactly as the LAST estimate call on `icp` folded
    them: the tree of icp_reduce_geometry over the source points in that call's fold order
    (icp_last_fold_order; the identity unless the call took a cell-sorted snapshot).  The oracle folds over
    the order of the cloud it is handed, so it is handed src[perm]; its indices go back to the caller's order.
    `init`: an oracle pose.  Returns (rc, pose, idx, inner) like O.icp_estimate."""
    src = np.ascontiguousarray(src, dtype=np.float64)
    n = len(src)
    perm, cell = icp.last_fold_order(n, with_cells=True)
    check_fold_order(perm, cell)
    blocks, threads = I.reduce_geometry(n)
    rc, oT, oidx_s, oinner = O.icp_estimate(dim, dst, np.ascontiguousarray(src[perm]), init, max_iter,
                                            use_kdtree=use_kdtree, sum_mode=1, reduce_blocks=blocks,
                                            reduce_threads=threads)
    oidx = np.empty_like(oidx_s)
    oidx[perm] = oidx_s
    return rc, oT, oidx, oinner


def check_fold_order(perm, cell):
    """a permutation, ascending by (sort key, original index): the stable order the header documents"""
    n = len(perm)
    assert np.array_equal(np.sort(perm), np.arange(n))
    if n > 1:
        c = cell.astype(np.int64)
        assert np.all(np.diff(c) >= 0)
        same = np.diff(c) == 0
        assert np.all(np.diff(perm)[same] > 0)
