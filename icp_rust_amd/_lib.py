"""Loader + ctypes signatures of libicp_mi355x.so (the C ABI in include/icp_mi355x.h).

The shared library is built in-tree by `make -C icp_rust_amd/csrc` (or
`__graft_entry__.build()`); it is never replaced by a Python/CPU implementation: if it
is missing, importing the compute API raises.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
# ICP_MI355X_LIB: load another build of the same library (e.g. the diagnostic one with counters)
LIB_PATH = os.environ.get("ICP_MI355X_LIB") or os.path.join(_HERE, "lib", "libicp_mi355x.so")

OK, NONE, EMPTY_DST, NAN_INPUT, BAD_ARGUMENT, NO_DEVICE, HIP_ERROR, OUT_OF_MEMORY, RETRY_REPLICATED, RETRY_SHARDED = range(10)
NN_AUTO, NN_BRUTE, NN_GRID = 0, 1, 2


class Pose(C.Structure):
    """`icp_pose` == Transform{rot: Rotation2 (column-major), t: Vector2}, src/transform.rs:6-10."""

    _fields_ = [(n, C.c_double) for n in ("r00", "r10", "r01", "r11", "tx", "ty")]

    def as_tuple(self):
        return (self.r00, self.r10, self.r01, self.r11, self.tx, self.ty)


def build(force=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC)]
    srcs += [os.path.join(os.path.dirname(_HERE), "include", f) for f in ("icp_mi355x.h", "icp_trig.h")]

    def stale():
        return (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)

    if force or stale():
        # one builder at a time: the ranks of a multi-process run all import this module together
        import fcntl

        os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
        with open(os.path.join(os.path.dirname(LIB_PATH), ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if force or stale():  # (another rank may have built it while this one waited)
                    subprocess.check_call(["make", "-C", _CSRC, "-j4", "-s"])
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


_dp = C.POINTER(C.c_double)
_u32p = C.POINTER(C.c_uint32)
_pp = C.POINTER(Pose)
_sz = C.c_size_t
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/icp_mi355x.h declares
SIGNATURES = {
    "icp_status_string": (C.c_char_p, [C.c_int]),
    "icp_abi_version": (C.c_int, []),
    "icp_device_count": (C.c_int, []),
    "icp_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p]),
    "icp_transform_new": (None, [_dp, _pp]),
    "icp_transform_from_rt": (None, [_dp, _dp, _pp]),
    "icp_transform_identity": (None, [_pp]),
    "icp_transform_apply": (None, [_pp, _dp, _dp]),
    "icp_transform_inverse": (None, [_pp, _pp]),
    "icp_transform_mul": (None, [_pp, _pp, _pp]),
    "icp_se2_exp": (None, [_dp, _dp]),
    "icp_se2_log": (None, [_dp, _dp]),
    "icp_se2_get_rt": (None, [_dp, _dp, _dp]),
    "icp_so2_exp": (None, [C.c_double, _dp]),
    "icp_so2_log": (C.c_double, [_dp]),
    "icp_norm": (C.c_double, [_dp, _sz, _sz]),
    "icp_inverse3x3": (C.c_int, [_dp, _dp]),
    "icp_f64_sin": (C.c_double, [C.c_double]),
    "icp_f64_cos": (C.c_double, [C.c_double]),
    "icp_transform_new_device": (C.c_int, [_vp, _sz, _vp, C.c_int]),
    "icp_create": (C.c_int, [C.POINTER(_vp), C.c_int, _vp, _sz, C.c_int]),
    "icp_create_device": (C.c_int, [C.POINTER(_vp), C.c_int, _vp, _sz, C.c_int]),
    "icp_destroy": (None, [_vp]),
    "icp_set_nn_mode": (C.c_int, [_vp, C.c_int]),
    "icp_set_single_launch": (C.c_int, [_vp, C.c_int]),
    "icp_single_launch_counters": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "icp_get_nn_mode": (C.c_int, [_vp]),
    "icp_set_stream": (C.c_int, [_vp, _vp]),
    "icp_use_own_stream": (C.c_int, [_vp]),
    "icp_estimate": (C.c_int, [_vp, _vp, _sz, _pp, _sz, _pp, _vp, _vp]),
    "icp_estimate_device": (C.c_int, [_vp, _vp, _sz, _pp, _sz, _pp, _vp, _vp]),
    "icp_estimate_transform": (C.c_int, [_vp, _vp, _sz, _pp, _vp]),
    "icp_weighted_gauss_newton_update": (C.c_int, [_pp, _vp, _vp, _sz, _dp]),
    "icp_gauss_newton_update": (C.c_int, [_pp, _vp, _vp, _sz, _dp]),
    "icp_error": (C.c_int, [_pp, _vp, _vp, _sz, _dp]),
    "icp_huber_error": (C.c_int, [_pp, _vp, _vp, _sz, _dp]),
    "icp_residual_stddevs": (C.c_int, [_pp, _vp, _vp, _sz, _dp]),
    "icp_correspond_device": (C.c_int, [_vp, _vp, _sz, _pp, _vp, _vp, _vp]),
    "icp_materialize_pairs_device": (C.c_int, [_vp, _vp, _sz, _pp, _vp, _vp, _vp]),
    "icp_prepare_source_device": (C.c_int, [_vp, _vp, _sz, _pp]),
    "icp_shard_prepare_source_device": (C.c_int, [_vp, _vp, _sz, _pp]),
    "icp_shard_sort_take_device": (C.c_int, [_vp, _vp, _sz, _pp, C.c_int, C.c_int, _vp, _vp]),
    "icp_estimate_transform_device": (C.c_int, [_vp, _vp, _vp, _sz, _pp, _vp]),
    "icp_nn_search_device": (C.c_int, [_vp, _vp, _sz, _vp]),
    "icp_weighted_gn_step_device": (C.c_int, [_vp, _vp, _vp, _sz, _pp, C.c_int, _dp, _dp]),
    "icp_shard_geometry": (C.c_int, [_sz, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                     C.POINTER(C.c_int), C.POINTER(_sz)]),
    "icp_shard_histogram_words": (_sz, []),
    "icp_shard_candidates_bytes": (_sz, []),
    "icp_shard_partials_bytes": (_sz, [C.c_int]),
    "icp_shard_exchange_bytes": (_sz, [C.c_int]),
    "icp_shard_take_device": (C.c_int, [_vp, _vp, _vp, _sz, C.c_int, C.c_int, _sz]),
    "icp_shard_put_device": (C.c_int, [_vp, _vp, _vp, _sz, C.c_int, C.c_int, _sz]),
    "icp_shard_eval_hist_device": (C.c_int, [_vp, _vp, _vp, _sz, C.c_int, C.c_int, _pp, C.c_int, C.c_int,
                                             C.POINTER(_vp)]),
    "icp_shard_eval_compact_device": (C.c_int, [_vp, _vp]),
    "icp_shard_eval_finish_device": (C.c_int, [_vp, _vp, _dp, _dp]),
    "icp_create_multi": (C.c_int, [C.POINTER(_vp), C.c_int, _vp, _sz, C.POINTER(C.c_int), C.c_int]),
    "icp_multi_estimate": (C.c_int, [_vp, _vp, _sz, _pp, _sz, _pp, _vp, _vp]),
    "icp_multi_counters": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "icp_multi_loop_counters": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "icp_multi_pipe_iterations": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "icp_pipe_counters": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "icp_debug_sort_cells": (C.c_int, [_vp, _sz, C.c_uint, _vp, _vp]),
    "icp_set_fixed_point_exit": (C.c_int, [_vp, C.c_int]),
    "icp_shard_pipe_run_device": (C.c_int, [_vp, _vp, _sz, _sz, C.c_int, C.c_int, _pp, C.POINTER(_sz), _sz, _vp, _vp,
                                            C.POINTER(C.c_int)]),
    "icp_loop_inbox_bytes": (_sz, []),
    "icp_loop_inbox": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_void_p)]),
    "icp_loop_inbox_ipc_handle": (C.c_int, [_vp, C.c_char_p]),
    "icp_loop_ipc_open": (C.c_int, [C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]),
    "icp_loop_ipc_close": (C.c_int, [_vp]),
    "icp_loop_inbox_shm_name": (C.c_int, [_vp, C.c_char_p]),
    "icp_loop_inbox_shm_unlink": (C.c_int, [_vp]),
    "icp_loop_shm_open": (C.c_int, [C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]),
    "icp_loop_shm_close": (C.c_int, [_vp]),
    "icp_loop_transport_probe": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int)]),
    "icp_reset_window_predictions": (C.c_int, [_vp]),
    "icp_shard_loop_connect": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "icp_shard_loop_launch_device": (C.c_int, [_vp, _vp, _vp, _sz, C.c_uint, C.c_uint, C.c_int, C.c_uint32, _pp, C.c_double,
                                               C.c_int, C.c_int]),
    "icp_shard_loop_wait": (C.c_int, [_vp, _pp, C.POINTER(C.c_double), _u32p, C.POINTER(C.c_int), C.POINTER(C.c_int), _u32p]),
    "icp_destroy_multi": (None, [_vp]),
    "icp_synchronize": (C.c_int, [_vp]),
    "icp_profile_enable": (C.c_int, [_vp, C.c_int]),
    "icp_profile_read": (C.c_int, [_vp, _dp, C.POINTER(C.c_uint64)]),
    "icp_reduce_geometry": (None, [_sz, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "icp_nn_cert_counters": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "icp_multi_append_targets": (C.c_int, [_vp, _vp, _sz, _pp]),
    "icp_multi_target_count": (_sz, [_vp]),
    "icp_shard_eval_status": (C.c_int, [_vp, _vp, C.c_int]),
    "icp_shard_eval_abort_device": (C.c_int, [_vp]),
    "icp_last_fold_order": (C.c_int, [_vp, _sz, _vp, _vp]),
    "icp_sort_source_device": (C.c_int, [_vp, _vp, _sz, _pp, _vp, _vp]),
    "icp_gn_path_counters": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "icp_gn_loop_counters": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "icp_gn_loop_timeouts": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "icp_fixed_point_skips": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "icp_run_ahead_counters": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "icp_trim_pool": (None, []),
    "icp_append_targets": (C.c_int, [_vp, _vp, _sz, _pp]),
    "icp_grid_append_counters": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "icp_append_targets_device": (C.c_int, [_vp, _vp, _sz, _pp]),
    "icp_reserve_targets": (C.c_int, [_vp, _sz]),
    "icp_target_count": (_sz, [_vp]),
    "icp_read_targets": (C.c_int, [_vp, _sz, _sz, _vp]),
    "icp_compute_target_normals": (C.c_int, [_vp, C.c_int]),
    "icp_update_target_normals": (C.c_int, [_vp, C.c_int]),
    "icp_read_target_normals": (C.c_int, [_vp, _sz, _sz, _vp]),
    "icp_estimate_point_to_plane": (C.c_int, [_vp, _vp, _sz, _pp, _sz, _pp, _vp, _vp]),
    "icp_estimate_point_to_plane_device": (C.c_int, [_vp, _vp, _sz, _pp, _sz, _pp, _vp, _vp]),
    "icp_multi_compute_target_normals": (C.c_int, [_vp, C.c_int]),
    "icp_multi_update_target_normals": (C.c_int, [_vp, C.c_int]),
    "icp_multi_estimate_point_to_plane": (C.c_int, [_vp, _vp, _sz, _pp, _sz, _pp, _vp, _vp]),
}

_lib = None


def lib():
    """The loaded C-ABI library.  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C icp_rust_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 and this
        # library links the same SONAME from /opt/rocm.  Whichever is loaded first serves both,
        # and torch does not work on the system copy, so when torch is installed let it load
        # its runtime first (device pointers and streams are shared with torch anyway).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


class IcpError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = status
        msg = lib().icp_status_string(status).decode()
        super().__init__(f"{where}: {msg} (status {status})" if where else f"{msg} (status {status})")


def check(status, where="", allow=()):
    if status != OK and status not in allow:
        raise IcpError(status, where)
    return status
