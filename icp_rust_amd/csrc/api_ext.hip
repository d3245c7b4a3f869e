// C ABI, EXTENSIONS beyond the reference (include/icp_mi355x.h sections 6 and 7): a target cloud that grows
// (scan-to-map, BASELINE.json configs[4]) and point-to-plane residuals.  Split out of api.hip in round 5; the hot path
// of the reference (Icp{2,3}d::new / estimate and the stage calls) lives in api.hip.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <new>

#include "api_internal.hpp"

using namespace icp;
using namespace icp::api;

// ------------------------------------------- EXTENSION: a growing target cloud ----
// Scan-to-map (BASELINE.json configs[4]; not in the reference, see include/icp_mi355x.h section 6).
// After an append the handle is indistinguishable from a fresh icp_create on the concatenated
// cloud: same target indices, same search results, same poses.
namespace {

// Transform::transform on every appended point (transform.rs:22-24; products, add, then + t; the
// library is compiled with -ffp-contract=off), z carried through as in transform_xy (lib.rs:52-57)
template <int DIM>
__global__ void k_append_targets(const double *__restrict__ pts, unsigned k, Pose T, bool xform,
                                 double *__restrict__ out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  double x = pts[(size_t)i * DIM], y = pts[(size_t)i * DIM + 1];
  if (xform) {
    const double nx = (T.r00 * x + T.r01 * y) + T.tx;
    const double ny = (T.r10 * x + T.r11 * y) + T.ty;
    x = nx;
    y = ny;
  }
  out[(size_t)i * DIM] = x;
  out[(size_t)i * DIM + 1] = y;
  if (DIM == 3) out[(size_t)i * DIM + 2] = pts[(size_t)i * DIM + 2];
}

int quiesce(icp_handle *h) {
  HIP_TRY(hipSetDevice(h->device));
  if (h->own_stream) HIP_TRY(hipStreamSynchronize(h->own_stream));
  if (h->stream != h->own_stream) HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->ws.spec_stream) HIP_TRY(hipStreamSynchronize(h->ws.spec_stream));
  return ICP_OK;
}

// make the target cloud live in storage the handle owns, with room for `points` points
int own_targets(icp_handle *h, size_t points) {
  const size_t need = points * (size_t)h->dim;
  if (h->owns_dst && need <= h->cap_dst_own) return ICP_OK;
  if (!h->owns_dst && need <= h->cap_dst_own && h->d_dst_own) {
    // a pooled buffer is large enough: move the borrowed cloud in
    if (h->m > 0)
      HIP_TRY(hipMemcpyAsync(h->d_dst_own, h->d_dst, h->m * h->dim * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  } else {
    // grow geometrically: a map that gains one scan per frame is copied O(log) times
    size_t cap = h->owns_dst ? h->cap_dst_own * 2 : 0;
    if (cap < need) cap = need + need / 8 + 1;
    double *p = nullptr;
    HIP_TRY(hipMalloc(&p, cap * sizeof(double)));
    if (h->m > 0) {
      const hipError_t e =
          hipMemcpyAsync(p, h->d_dst, h->m * h->dim * sizeof(double), hipMemcpyDeviceToDevice, h->stream);
      if (e != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) {
        (void)hipFree(p);
        return ICP_HIP_ERROR;
      }
    }
    (void)hipFree(h->d_dst_own);
    h->d_dst_own = p;
    h->cap_dst_own = cap;
  }
  h->d_dst = h->d_dst_own;
  h->owns_dst = true;
  return ICP_OK;
}

int append_common(icp_handle *h, const double *pts, size_t k, const icp_pose *T, bool on_device) {
  if (!h || (k > 0 && !pts) || h->m + k >= 0xffffffffull) return ICP_BAD_ARGUMENT;
  if (k == 0) return ICP_OK;
  int rc = quiesce(h);
  if (rc != ICP_OK) return rc;
  const double *d_pts = pts;
  if (!on_device) {
    // stage through the handle's source buffer (the same one icp_estimate stages a scan in)
    HIP_TRY(ensure_workspace(h, k, true));
    HIP_TRY(hipMemcpyAsync(h->ws.d_src, pts, k * h->dim * sizeof(double), hipMemcpyHostToDevice, h->stream));
    d_pts = h->ws.d_src;
  }
  if ((rc = own_targets(h, h->m + k)) != ICP_OK) return rc;
  double *tail = h->d_dst_own + h->m * h->dim;
  const Pose P = T ? *T : transform_identity();
  const unsigned blocks = (unsigned)((k + 255) / 256);
  if (h->dim == 3)
    hipLaunchKernelGGL(k_append_targets<3>, dim3(blocks), dim3(256), 0, h->stream, d_pts, (unsigned)k, P, T != nullptr, tail);
  else
    hipLaunchKernelGGL(k_append_targets<2>, dim3(blocks), dim3(256), 0, h->stream, d_pts, (unsigned)k, P, T != nullptr, tail);
  HIP_TRY(hipGetLastError());
  const size_t m_before = h->m;
  h->m += k;
  // (extension) the normals of the targets that were there stay; the new ones have none until
  // icp_update_target_normals / icp_compute_target_normals (normals_m < m: point-to-plane calls refuse)
  h->qsort.valid = false;  // snapshots and previous matches refer to the old grid
  h->qsort.have_prev = false;
  h->brute_valid = h->screen_valid = false;
  // the grid is what a map-sized cloud is searched with; the sweep's structures (SoA + f32 screen,
  // 36 B per target) are rebuilt right away only where the sweep is the engine in use
  bool appended = false;
  hipError_t e = append_grid(h, m_before, k, &appended);  // (the sorted records move, nothing is re-sorted: nn_grid.hip)
  if (e == hipSuccess && !appended) e = build_grid(h);
  if (e == hipSuccess) ++(appended ? h->grid.appends_moved : h->grid.appends_rebuilt);
  if (e == hipSuccess && resolved_nn_mode(h) == ICP_NN_BRUTE) {
    if ((e = build_target_soa(h)) == hipSuccess) e = build_target_screen(h);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);  // host `pts` may be freed; later calls may use another stream
  if (e != hipSuccess) {
    // back to the cloud as it was: the old points are untouched, the search structures are rebuilt
    // for them (if even that fails the handle has no grid and the sweep rebuilds its copies on demand)
    h->m = m_before;
    (void)build_grid(h);
    (void)hipStreamSynchronize(h->stream);
    return map_hip(e);
  }
  return ICP_OK;
}

}  // namespace

extern "C" int icp_append_targets(icp_handle *h, const double *pts, size_t k, const icp_pose *T) {
  return append_common(h, pts, k, T, false);
}
extern "C" int icp_append_targets_device(icp_handle *h, const double *d_pts, size_t k, const icp_pose *T) {
  return append_common(h, d_pts, k, T, true);
}
extern "C" int icp_reserve_targets(icp_handle *h, size_t capacity) {
  if (!h || capacity >= 0xffffffffull) return ICP_BAD_ARGUMENT;
  if (capacity <= h->m) return ICP_OK;
  const int rc = quiesce(h);
  if (rc != ICP_OK) return rc;
  return own_targets(h, capacity);
}
extern "C" size_t icp_target_count(const icp_handle *h) { return h ? h->m : 0; }
// Observability: out[0] = appends served by moving the sorted records (append_grid), out[1] = appends that rebuilt the grid
extern "C" int icp_grid_append_counters(const icp_handle *h, uint64_t out[2]) {
  if (!h || !out) return ICP_BAD_ARGUMENT;
  out[0] = h->grid.appends_moved;
  out[1] = h->grid.appends_rebuilt;
  return ICP_OK;
}
extern "C" int icp_read_targets(icp_handle *h, size_t first, size_t k, double *out) {
  if (!h || first > h->m || k > h->m - first || (k > 0 && !out)) return ICP_BAD_ARGUMENT;
  if (k == 0) return ICP_OK;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMemcpyAsync(out, h->d_dst + first * h->dim, k * h->dim * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return ICP_OK;
}

// ------------------------------------------- EXTENSION: point-to-plane residuals -----
// Not in the reference (no normals anywhere in src/); definition and CPU restatement: p2plane.hip,
// the CPU checker under tests (tests/test_p2plane.py).  Everything around the residual is the reference's: exact 3-D
// nearest neighbour, SE(2) pose on xy, Huber / MAD Gauss-Newton, the inner loop's break tests.
extern "C" int icp_compute_target_normals(icp_handle *h, int k) {
  if (!h || h->dim != 3 || k < 3 || k > 16) return ICP_BAD_ARGUMENT;
  if (h->m == 0) return ICP_EMPTY_DST;
  if (!h->grid.built) return ICP_BAD_ARGUMENT;  // non-finite targets: no grid to search neighbourhoods with
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(reserve(h->d_normals, h->cap_normals, h->m * 3));
  HIP_TRY(launch_target_normals(h, k, h->d_normals));
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->normals_m = h->m;
  h->normals_k = k;
  return ICP_OK;
}

// The targets appended since the normals were last computed get theirs (from their k nearest targets in the cloud
// as it is NOW); the older targets keep the normals they have -- "normals at insertion time", the definition a map that
// grows frame by frame uses (a full icp_compute_target_normals re-derives all of them from the current cloud).
extern "C" int icp_update_target_normals(icp_handle *h, int k) {
  if (!h || h->dim != 3 || k < 3 || k > 16) return ICP_BAD_ARGUMENT;
  if (h->m == 0) return ICP_EMPTY_DST;
  if (!h->grid.built || h->normals_m > h->m) return ICP_BAD_ARGUMENT;
  if (h->normals_m > 0 && h->normals_k != k) return ICP_BAD_ARGUMENT;  // one neighbourhood size per cloud
  HIP_TRY(hipSetDevice(h->device));
  if (h->m * 3 > h->cap_normals || !h->d_normals) {  // grow, keeping the normals that exist
    double *old = h->d_normals;
    const size_t keep = h->normals_m * 3;
    h->d_normals = nullptr;
    h->cap_normals = 0;
    hipError_t e = reserve(h->d_normals, h->cap_normals, h->m * 3);
    if (e == hipSuccess && old && keep)
      e = hipMemcpyAsync(h->d_normals, old, keep * sizeof(double), hipMemcpyDeviceToDevice, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(old);
    if (e != hipSuccess) {
      h->normals_m = 0;
      return map_hip(e);
    }
  }
  HIP_TRY(launch_target_normals(h, k, h->d_normals, h->normals_m));
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->normals_m = h->m;
  h->normals_k = k;
  return ICP_OK;
}

extern "C" int icp_read_target_normals(icp_handle *h, size_t first, size_t count, double *out) {
  if (!h || h->normals_m != h->m || h->m == 0 || first > h->m || count > h->m - first || (count > 0 && !out))
    return ICP_BAD_ARGUMENT;
  if (count == 0) return ICP_OK;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMemcpyAsync(out, h->d_normals + first * 3, count * 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return ICP_OK;
}

// the per-pair scratch of a point-to-plane inner loop
static int ensure_plane_buffers(icp_handle *h, size_t n) {
  if (n > h->cap_plane) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    (void)hipFree(h->d_plane_pairs);
    (void)hipFree(h->d_plane_fa);
    (void)hipFree(h->d_plane_fb);
    h->d_plane_pairs = nullptr;
    h->d_plane_fa = h->d_plane_fb = nullptr;
    h->cap_plane = 0;
    HIP_TRY(hipMalloc(&h->d_plane_pairs, n * p2pl_pair_bytes()));
    HIP_TRY(hipMalloc(&h->d_plane_fa, n * 2 * sizeof(double)));
    HIP_TRY(hipMalloc(&h->d_plane_fb, n * 2 * sizeof(double)));
    h->cap_plane = n;
  }
  return ICP_OK;
}

// One outer iteration's inner loop (src/lib.rs:59-84 around the plane residual) for given correspondences d_idx of the
// WHOLE source cloud under pose T: the pose update dT and the updates applied.  (icp_multi_estimate_point_to_plane runs
// this on every rank after the ranks have exchanged the indices of their slices.)
int icp_p2pl_inner_loop_device(icp_handle *h, const double *d_src, size_t n, const icp_pose *T, const uint32_t *d_idx,
                               icp_pose *dT, uint32_t *applied_out) {
  if (!h || h->dim != 3 || !T || !dT || (n > 0 && (!d_src || !d_idx)) || h->normals_m != h->m || h->m == 0) return ICP_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(ensure_workspace(h, n, false));
  ICP_TRY_RC(ensure_plane_buffers(h, n));
  Workspace &w = h->ws;
  HIP_TRY(launch_p2pl_gather(h, d_src, n, *T, d_idx, h->d_normals, h->d_plane_pairs));
  Pose Ti = transform_identity();
  uint32_t applied = 0;
  if (n >= 2) {
    double prev_error = DBL_MAX;
    for (int k = 0; k < ICP_INNER_MAX_ITER; ++k) {
      HIP_TRY(launch_p2pl_eval(h, h->d_plane_pairs, n, Ti, h->d_plane_fa, h->d_plane_fb));
      HIP_TRY(hipStreamSynchronize(h->stream));
      const GnResult &r = *w.h_res;
      if (r.nan_flag) return ICP_NAN_INPUT;
      double delta[3];
      if (!solve_update(r.acc, r.acc + 9, delta)) break;
      if ((delta[0] * delta[0] + delta[1] * delta[1]) + delta[2] * delta[2] < ICP_DELTA_NORM_THRESHOLD) break;
      if (r.acc[12] > prev_error) break;
      prev_error = r.acc[12];
      Ti = transform_mul(transform_new(delta), Ti);
      ++applied;
    }
  }
  *dT = Ti;
  if (applied_out) *applied_out = applied;
  return ICP_OK;
}

extern "C" int icp_estimate_point_to_plane_device(icp_handle *h, const double *d_src, size_t n, const icp_pose *init,
                                                  size_t max_iter, icp_pose *out, uint32_t *d_last_idx,
                                                  uint32_t *inner_iters) {
  if (!h || h->dim != 3 || !init || !out || (n > 0 && !d_src) || n >= 0xffffffffull) return ICP_BAD_ARGUMENT;
  if (h->m == 0) {  // index.unwrap() on an empty tree, src/lib.rs:165 -- only when a search would run
    if (n > 0 && max_iter > 0) return ICP_EMPTY_DST;
    *out = *init;
    return ICP_OK;
  }
  if (h->normals_m != h->m) return ICP_BAD_ARGUMENT;  // icp_compute_target_normals first (again after an append)
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(ensure_workspace(h, n, false));
  ICP_TRY_RC(ensure_plane_buffers(h, n));
  Workspace &w = h->ws;
  Pose T = *init;
  if (max_iter > 0) {
    const int prc = icp_prepare_source_device(h, d_src, n, init);
    if (prc != ICP_OK) return prc;
  }
  struct Quiesce {
    icp_handle *h;
    ~Quiesce() {
      (void)hipStreamSynchronize(h->stream);
      h->qsort.valid = false;
      h->qsort.have_prev = false;
    }
  } quiesce_on_exit{h};
  for (size_t it = 0; it < max_iter; ++it) {
    uint32_t *idx = (it + 1 == max_iter && d_last_idx) ? d_last_idx : w.d_idx;
    int rc = icp_correspond_device(h, d_src, n, &T, nullptr, nullptr, idx);  // exact 3-D NN, src/lib.rs:161-167
    if (rc != ICP_OK) return rc;
    Pose Ti;
    uint32_t applied = 0;
    rc = icp_p2pl_inner_loop_device(h, d_src, n, &T, idx, &Ti, &applied);
    if (rc != ICP_OK) return rc;
    if (inner_iters) inner_iters[it] = applied;
    T = transform_mul(Ti, T);
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  *out = T;
  return ICP_OK;
}

extern "C" int icp_estimate_point_to_plane(icp_handle *h, const double *src, size_t n, const icp_pose *init,
                                           size_t max_iter, icp_pose *out, uint32_t *last_idx, uint32_t *inner_iters) {
  if (!h || h->dim != 3 || !init || !out || (n > 0 && !src) || n >= 0xffffffffull) return ICP_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(ensure_workspace(h, n, true));
  if (n > 0)
    HIP_TRY(hipMemcpyAsync(h->ws.d_src, src, n * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  uint32_t *d_li = nullptr;
  if (last_idx && n > 0) HIP_TRY(hipMalloc(&d_li, n * sizeof(uint32_t)));
  int rc = icp_estimate_point_to_plane_device(h, h->ws.d_src, n, init, max_iter, out, d_li, inner_iters);
  if (rc == ICP_OK && d_li && max_iter > 0) {
    if (hipMemcpy(last_idx, d_li, n * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) rc = ICP_HIP_ERROR;
  }
  (void)hipFree(d_li);
  return rc;
}

