// C ABI of the MI355X ICP core (include/icp_mi355x.h): handle lifecycle, the outer ICP
// loop (src/lib.rs:105-130, 148-173) and the inner robust Gauss-Newton loop
// (src/lib.rs:59-84) driven from the host over the device stages in nn_brute.hip /
// gn.hip.  Nothing here falls back to a CPU computation: without a HIP device every
// compute entry point fails with ICP_NO_DEVICE.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <new>

#include "api_internal.hpp"

using namespace icp;
using namespace icp::api;

namespace icp {
namespace api {
int map_hip(hipError_t e) {
  switch (e) {
    case hipSuccess: return ICP_OK;
    case hipErrorOutOfMemory: return ICP_OUT_OF_MEMORY;
    case hipErrorNoDevice:
    case hipErrorInvalidDevice:
    case hipErrorInsufficientDriver:
    case hipErrorNotInitialized: return ICP_NO_DEVICE;
    default: return ICP_HIP_ERROR;
  }
}
}  // namespace api
}  // namespace icp

namespace {

template <typename Tp>
hipError_t grow(Tp *&p, size_t count) {
  if (p) {
    hipError_t e = hipFree(p);
    p = nullptr;
    if (e != hipSuccess) return e;
  }
  return hipMalloc(&p, (count ? count : 1) * sizeof(Tp));
}

void free_ctx(GnCtx &c) {
  (void)hipFree(c.d_rx);
  (void)hipFree(c.d_ry);
  (void)hipFree(c.d_hist);
  (void)hipFree(c.d_cand);
  (void)hipFree(c.d_ctl);
  (void)hipFree(c.d_sel);
  (void)hipFree(c.d_scal);
  (void)hipFree(c.d_partials);
  (void)hipFree(c.d_whist);
  (void)hipFree(c.d_wstate);
  (void)hipFree(c.d_wmed);
  (void)hipFree(c.d_wring);
  (void)hipFree(c.d_bkt);
  (void)hipFree(c.d_bkt_dir);
  if (c.h_res) (void)hipHostFree(c.h_res);
}

void free_workspace(Workspace &w) {
  (void)hipFree(w.d_src);
  (void)hipFree(w.d_a);
  (void)hipFree(w.d_b);
  (void)hipFree(w.d_a2);
  (void)hipFree(w.d_b2);
  (void)hipFree(w.d_a3);
  (void)hipFree(w.d_b3);
  (void)hipFree(w.d_ahead);
  (void)hipFree(w.d_idx);
  (void)hipFree(w.d_idx_slot);
  (void)hipFree(w.d_sa);
  (void)hipFree(w.d_sb);
  if (w.h_whist) (void)hipHostFree(w.h_whist);
  if (w.h_tiny) (void)hipHostFree(w.h_tiny);
  (void)hipFree(w.d_rlist);
  (void)hipFree(w.d_rlist_len);
  (void)hipFree(w.d_part_d);
  (void)hipFree(w.d_part_i);
  free_loop_inbox(w);
  free_loop_plan(w.loop_plan);
  (void)hipFree(w.d_loop_ctl);
  (void)hipFree(w.d_loop_hist);
  (void)hipFree(w.d_loop_part);
  if (w.h_loop_res) (void)hipHostFree(w.h_loop_res);
  free_ctx(w);
  free_ctx(w.alt);
  if (w.spec_stream) {
    (void)hipStreamSynchronize(w.spec_stream);
    (void)hipStreamDestroy(w.spec_stream);
  }
  w = Workspace();
}

hipError_t alloc_ctx(GnCtx &c, hipStream_t s) {
  hipError_t e;
  const size_t hist_bytes = (size_t)kSelRoles * kSelProblems * kSelBins * sizeof(uint32_t);
  if ((e = hipMalloc(&c.d_hist, hist_bytes)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(c.d_hist, 0, hist_bytes, s)) != hipSuccess) return e;
  if ((e = hipMalloc(&c.d_cand, (size_t)2 * kSelProblems * kSelCap * sizeof(unsigned long long))) != hipSuccess) return e;
  if ((e = hipMalloc(&c.d_ctl, sizeof(SelCtl))) != hipSuccess) return e;
  if ((e = hipMemsetAsync(c.d_ctl, 0, sizeof(SelCtl), s)) != hipSuccess) return e;
  if ((e = hipMalloc(&c.d_sel, 2 * kSelProblems * sizeof(SelState))) != hipSuccess) return e;
  if ((e = hipMalloc(&c.d_scal, sizeof(GnScalars))) != hipSuccess) return e;
  // (+ one row: the folded totals k_win_finish's first workgroups leave for its last one)
  if ((e = hipMalloc(&c.d_partials, (size_t)(kTreeMaxBlocks + 1) * (kNSum + 1) * sizeof(double))) != hipSuccess) return e;
  const size_t whist_bytes = ((size_t)2 * kWinBins + kShardStatusWords) * sizeof(uint32_t);  // (+ the sharded status words)
  if ((e = hipMalloc(&c.d_whist, whist_bytes)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(c.d_whist, 0, whist_bytes, s)) != hipSuccess) return e;
  if ((e = hipMalloc(&c.d_wstate, sizeof(WinState))) != hipSuccess) return e;
  if ((e = hipMemsetAsync(c.d_wstate, 0, sizeof(WinState), s)) != hipSuccess) return e;
  if ((e = hipMalloc(&c.d_wmed, (size_t)2 * kWinCapMed * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMalloc(&c.d_wring, (size_t)2 * kWinCapRing * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMalloc(&c.d_bkt, (size_t)kReduceMaxBlocks * kBktStage * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMalloc(&c.d_bkt_dir, (size_t)kReduceMaxBlocks * kBktDir * sizeof(unsigned short))) != hipSuccess) return e;
  if ((e = hipHostMalloc(&c.h_res, sizeof(GnResult), hipHostMallocCoherent)) != hipSuccess) return e;
  memset(c.h_res, 0, sizeof(GnResult));
  return hipSuccess;
}

}  // namespace

namespace icp {

namespace {
using PushFn = int (*)(const char *);
using PopFn = int (*)();
PushFn g_roctx_push = nullptr;
PopFn g_roctx_pop = nullptr;
std::once_flag g_roctx_once;
void roctx_lookup() {
  for (const char *lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
    void *hnd = dlopen(lib, RTLD_LAZY | RTLD_GLOBAL);
    if (!hnd) continue;
    g_roctx_push = reinterpret_cast<PushFn>(dlsym(hnd, "roctxRangePushA"));
    g_roctx_pop = reinterpret_cast<PopFn>(dlsym(hnd, "roctxRangePop"));
    if (g_roctx_push && g_roctx_pop) return;
    g_roctx_push = nullptr;
    g_roctx_pop = nullptr;
  }
}
}  // namespace
Range::Range(const char *name) {
  std::call_once(g_roctx_once, roctx_lookup);
  if (g_roctx_push) (void)g_roctx_push(name);
}
Range::~Range() {
  if (g_roctx_pop) (void)g_roctx_pop();
}

hipError_t ensure_workspace(icp_handle *h, size_t n, bool need_src) {
  Workspace &w = h->ws;
  hipError_t e;
  if (!w.d_hist) {
    if ((e = alloc_ctx(w, h->stream)) != hipSuccess) return e;
    if ((e = alloc_ctx(w.alt, h->stream)) != hipSuccess) return e;
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    // ICP_EVAL_PRIORITY=low|normal: A/B switch for the evaluation stream's priority (default: highest)
    int prio = prio_greatest;
    if (const char *pe = exp_env("ICP_EVAL_PRIORITY")) prio = pe[0] == 'l' ? prio_least : (pe[0] == 'n' ? 0 : prio_greatest);
    if ((e = hipStreamCreateWithPriority(&w.spec_stream, hipStreamNonBlocking, prio)) != hipSuccess) return e;
    if ((e = hipMalloc(&w.d_ahead, sizeof(AheadPose))) != hipSuccess) return e;
    if ((e = hipMemsetAsync(w.d_ahead, 0, sizeof(AheadPose), h->stream)) != hipSuccess) return e;
    // the memsets above must have landed before either stream uses the scratch
    if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return e;
  }
  if (n > w.cap_n) {
    // in-flight work may still read the old buffers
    if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(w.spec_stream)) != hipSuccess) return e;
    size_t cap = n + n / 8;
    (void)hipFree(w.d_src);
    w.d_src = nullptr;
    if ((e = grow(w.d_a, cap * 2)) != hipSuccess) return e;
    if ((e = grow(w.d_b, cap * 2)) != hipSuccess) return e;
    if ((e = grow(w.d_a2, cap * 2)) != hipSuccess) return e;
    if ((e = grow(w.d_b2, cap * 2)) != hipSuccess) return e;
    if ((e = grow(w.d_a3, cap * 2)) != hipSuccess) return e;
    if ((e = grow(w.d_b3, cap * 2)) != hipSuccess) return e;
    if ((e = grow(w.d_rx, cap)) != hipSuccess) return e;
    if ((e = grow(w.d_ry, cap)) != hipSuccess) return e;
    if ((e = grow(w.alt.d_rx, cap)) != hipSuccess) return e;
    if ((e = grow(w.alt.d_ry, cap)) != hipSuccess) return e;
    if ((e = grow(w.d_idx, cap)) != hipSuccess) return e;
    if ((e = grow(w.d_idx_slot, cap)) != hipSuccess) return e;
    w.cap_n = cap;
  }
  if (need_src && !w.d_src) {
    if ((e = hipMalloc(&w.d_src, (w.cap_n ? w.cap_n : 1) * 3 * sizeof(double))) != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace icp

// ------------------------------------------------------------------ misc ---------
extern "C" const char *icp_status_string(int s) {
  switch (s) {
    case ICP_OK: return "ok";
    case ICP_NONE: return "none (the reference returns None)";
    case ICP_EMPTY_DST: return "empty dst (the reference panics on index.unwrap())";
    case ICP_NAN_INPUT: return "NaN residual (the reference panics on partial_cmp().unwrap())";
    case ICP_BAD_ARGUMENT: return "bad argument";
    case ICP_NO_DEVICE: return "no usable HIP device (there is no CPU fallback)";
    case ICP_HIP_ERROR: return "HIP error";
    case ICP_OUT_OF_MEMORY: return "out of device memory";
    case ICP_RETRY_REPLICATED: return "sharded evaluation: evaluate this one on the gathered pairs";
    case ICP_RETRY_SHARDED: return "sharded evaluation: the window missed; again with the refined window";
    default: return "unknown status";
  }
}

extern "C" int icp_abi_version(void) { return ICP_ABI_VERSION; }

// what tells two devices of one node apart, whatever ordinal a process sees them under (HIP_VISIBLE_DEVICES)
extern "C" int icp_device_pci_bus_id(int device, char out[64]) {
  if (!out) return ICP_BAD_ARGUMENT;
  out[0] = 0;
  if (hipDeviceGetPCIBusId(out, 64, device) != hipSuccess) {
    (void)hipGetLastError();
    return ICP_NO_DEVICE;
  }
  return ICP_OK;
}

extern "C" int icp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ------------------------------------------------------- host pose algebra -------
extern "C" void icp_transform_new(const double p[3], icp_pose *out) { *out = transform_new(p); }
extern "C" void icp_transform_from_rt(const double r[4], const double t[2], icp_pose *out) {
  *out = icp_pose{r[0], r[1], r[2], r[3], t[0], t[1]};
}
extern "C" void icp_transform_identity(icp_pose *out) { *out = transform_identity(); }
extern "C" void icp_transform_apply(const icp_pose *T, const double p[2], double out[2]) {
  double o[2];
  transform_apply(*T, p, o);
  out[0] = o[0];
  out[1] = o[1];
}
extern "C" void icp_transform_inverse(const icp_pose *T, icp_pose *out) { *out = transform_inverse(*T); }
extern "C" void icp_transform_mul(const icp_pose *l, const icp_pose *r, icp_pose *out) {
  *out = transform_mul(*l, *r);
}
extern "C" void icp_se2_exp(const double p[3], double m[9]) { se2_exp(p, m); }
extern "C" void icp_se2_log(const double m[9], double p[3]) { se2_log(m, p); }
extern "C" void icp_se2_get_rt(const double m[9], double rot[4], double t[2]) { se2_get_rt(m, rot, t); }
extern "C" void icp_so2_exp(double theta, double m[4]) { so2_exp(theta, m); }
extern "C" double icp_so2_log(const double m[4]) { return so2_log(m); }
extern "C" double icp_norm(const double *m, size_t nrows, size_t ncols) { return icp::norm(m, nrows, ncols); }
extern "C" int icp_inverse3x3(const double m[9], double out[9]) {
  return inverse3x3(m, out) ? ICP_OK : ICP_NONE;
}
extern "C" void icp_reduce_geometry(size_t n, int *blocks, int *threads) { reduce_geometry(n, blocks, threads); }
extern "C" double icp_f64_sin(double x) { return ref_sin(x); }
extern "C" double icp_f64_cos(double x) { return ref_cos(x); }

// Transform::new on the device (observability: host and device must agree to the bit)
__global__ void k_transform_new(const double *__restrict__ params, unsigned n, icp_pose *__restrict__ out,
                                int *__restrict__ out_of_range) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double p[3] = {params[3 * (size_t)i], params[3 * (size_t)i + 1], params[3 * (size_t)i + 2]};
  bool ok;
  out[i] = transform_new_in_range(p, &ok);
  if (!ok) atomicOr(out_of_range, 1);
}

extern "C" int icp_transform_new_device(const double *params, size_t n, icp_pose *out, int device) {
  if ((n > 0 && (!params || !out)) || n >= 0xffffffffull) return ICP_BAD_ARGUMENT;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return ICP_NO_DEVICE;
  if (n == 0) return ICP_OK;
  if (device >= 0) HIP_TRY(hipSetDevice(device));
  double *d_p = nullptr;
  icp_pose *d_o = nullptr;
  int *d_f = nullptr;
  int rc = ICP_OK, flag = 0;
  do {
    hipError_t e;
    if ((e = hipMalloc(&d_p, n * 3 * sizeof(double))) != hipSuccess) { rc = map_hip(e); break; }
    if ((e = hipMalloc(&d_o, n * sizeof(icp_pose))) != hipSuccess) { rc = map_hip(e); break; }
    if ((e = hipMalloc(&d_f, sizeof(int))) != hipSuccess) { rc = map_hip(e); break; }
    if ((e = hipMemcpy(d_p, params, n * 3 * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) { rc = map_hip(e); break; }
    if ((e = hipMemset(d_f, 0, sizeof(int))) != hipSuccess) { rc = map_hip(e); break; }
    hipLaunchKernelGGL(k_transform_new, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, d_p, (unsigned)n, d_o, d_f);
    if ((e = hipGetLastError()) != hipSuccess) { rc = map_hip(e); break; }
    if ((e = hipMemcpy(out, d_o, n * sizeof(icp_pose), hipMemcpyDeviceToHost)) != hipSuccess) { rc = map_hip(e); break; }
    if ((e = hipMemcpy(&flag, d_f, sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess) { rc = map_hip(e); break; }
  } while (0);
  (void)hipFree(d_p);
  (void)hipFree(d_o);
  (void)hipFree(d_f);
  if (rc != ICP_OK) return rc;
  return flag ? ICP_BAD_ARGUMENT : ICP_OK;  // a theta beyond the restated range of sin / cos: host only
}

// ---------------------------------------------------------------- handle ---------
static icp_handle *pool_take(int device);
static void destroy_failed(icp_handle *h);

static int create_common(icp_handle **out, int dim, const double *dst, size_t m, int device, bool dst_on_device) {
  if (!out || (dim != 2 && dim != 3) || (m > 0 && !dst) || m >= 0xffffffffull) return ICP_BAD_ARGUMENT;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return ICP_NO_DEVICE;
  if (device < 0) HIP_TRY(hipGetDevice(&device));
  if (device >= count) return ICP_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(device));
  // the reference builds a new Icp per frame (examples/scan3d.rs:130): a destroyed handle's device
  // buffers, streams and pinned memory wait in a small pool for the next create on the same device
  icp_handle *h = pool_take(device);
  if (!h) h = new (std::nothrow) icp_handle();
  if (!h) return ICP_OUT_OF_MEMORY;
  h->dim = dim;
  h->m = m;
  h->device = device;
  int rc = ICP_OK;
  do {
    hipError_t e;
    if (!h->own_stream &&
        (e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking)) != hipSuccess) { rc = map_hip(e); break; }
    h->stream = h->own_stream;
    if (dst_on_device || m == 0) {
      h->d_dst = dst;
      h->owns_dst = false;
    } else {
      if ((e = reserve(h->d_dst_own, h->cap_dst_own, m * dim)) != hipSuccess) { rc = map_hip(e); break; }
      h->d_dst = h->d_dst_own;
      h->owns_dst = true;
      if ((e = hipMemcpyAsync(h->d_dst_own, dst, m * dim * sizeof(double), hipMemcpyHostToDevice, h->stream)) != hipSuccess) { rc = map_hip(e); break; }
    }
    if ((e = build_grid(h)) != hipSuccess) { rc = map_hip(e); break; }
    // the sweep's structures (SoA + f32 screen, 36 B per target) only where the sweep is the engine
    // this cloud resolves to; otherwise launch_nn_brute builds them if it is ever asked
    h->brute_valid = h->screen_valid = false;
    if (resolved_nn_mode(h) == ICP_NN_BRUTE) {
      if ((e = build_target_soa(h)) != hipSuccess) { rc = map_hip(e); break; }
      if ((e = build_target_screen(h)) != hipSuccess) { rc = map_hip(e); break; }
    }
    if ((e = ensure_workspace(h, 0, false)) != hipSuccess) { rc = map_hip(e); break; }
    // the host buffer may be freed by the caller as soon as we return
    if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) { rc = map_hip(e); break; }
  } while (0);
  if (rc != ICP_OK) {
    // a half-built handle must not reach the pool (its scratch may be partly allocated): release it
    destroy_failed(h);
    return rc;
  }
  *out = h;
  return ICP_OK;
}

extern "C" int icp_create(icp_handle **out, int dim, const double *dst, size_t m, int device) {
  return create_common(out, dim, dst, m, device, false);
}
extern "C" int icp_create_device(icp_handle **out, int dim, const double *d_dst, size_t m, int device) {
  return create_common(out, dim, d_dst, m, device, true);
}

namespace {

void free_handle(icp_handle *h) {  // really release everything
  (void)hipSetDevice(h->device);
  free_workspace(h->ws);
  (void)hipFree(h->d_dst_own);
  (void)hipFree(h->d_dst_soa);
  (void)hipFree(h->d_dst_f32);
  (void)hipFree(h->grid.d_start);
  (void)hipFree(h->grid.d_pts);
  (void)hipFree(h->grid.t_cell_of);
  (void)hipFree(h->grid.t_cnt);
  (void)hipFree(h->grid.t_btot);
  (void)hipFree(h->grid.t_part);
  (void)hipFree(h->grid.d_rcell);
  (void)hipFree(h->grid.d_rcell2);
  (void)hipFree(h->grid.d_start2);
  (void)hipFree(h->grid.d_pts2);
  (void)hipFree(h->grid.t_shift);
  (void)hipFree(h->grid.t_cell_new);
  (void)hipFree(h->grid.d_flag);
  (void)hipFree(h->qsort.d_cell_of);
  (void)hipFree(h->qsort.d_tmp);
  (void)hipFree(h->qsort.d_cert_lists);
  (void)hipFree(h->qsort.d_cert_ctr);
  (void)hipFree(h->qsort.d_perm);
  (void)hipFree(h->qsort.d_sorted);
  (void)hipFree(h->qsort.d_prev);
  (void)hipFree(h->shard.d_ordered);
  (void)hipFree(h->d_normals);
  (void)hipFree(h->d_plane_pairs);
  (void)hipFree(h->d_plane_fa);
  (void)hipFree(h->d_plane_fb);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
}

constexpr size_t kPoolMax = 2;  // per process; a frame loop needs one
std::mutex g_pool_mu;
std::vector<icp_handle *> g_pool;

}  // namespace

static void destroy_failed(icp_handle *h) {
  (void)hipSetDevice(h->device);
  if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
  if (h->ws.spec_stream) (void)hipStreamSynchronize(h->ws.spec_stream);
  free_handle(h);
}

static icp_handle *pool_take(int device) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (size_t i = 0; i < g_pool.size(); ++i)
    if (g_pool[i]->device == device) {
      icp_handle *h = g_pool[i];
      g_pool.erase(g_pool.begin() + i);
      return h;
    }
  return nullptr;
}

extern "C" void icp_destroy(icp_handle *h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
  if (h->stream != h->own_stream) (void)hipStreamSynchronize(h->stream);  // work enqueued on a caller's stream
  if (h->ws.spec_stream) (void)hipStreamSynchronize(h->ws.spec_stream);
  for (auto *v : {&h->prof_events, &h->prof_free}) {
    for (auto &ev : *v) {
      (void)hipEventDestroy(ev.first);
      (void)hipEventDestroy(ev.second);
    }
    v->clear();
  }
  if (exp_env("ICP_DBG_WIN") && h->ws.win_tried)
    fprintf(stderr,
            "[icp] window evaluations: %llu tried, %llu missed; speculative searches: %llu hit, %llu missed; "
            "first evaluations launched ahead: %llu; refined windows (n > 4M): %llu tried, %llu missed\n",
            h->ws.win_tried, h->ws.win_missed, h->ws.spec_hits, h->ws.spec_misses, h->ws.pre_evals,
            h->ws.refine_tried, h->ws.refine_missed);
  static const bool no_pool = getenv("ICP_NO_POOL") != nullptr;
  if (!no_pool) {
    // back to the state of a fresh handle, buffers kept (everything logical is reset here; the
    // evaluation scratch is in its rest state unless gn_dirty says otherwise)
    Workspace &w = h->ws;
    h->m = 0;
    h->d_dst = nullptr;
    h->owns_dst = false;
    h->nn_mode = ICP_NN_AUTO;
    h->single_launch = true;
    h->fixed_point_exit = true;
    h->stream = h->own_stream;
    h->profile = 0;
    h->prof_seen = 0;
    h->grid.built = false;
    h->grid.appends_moved = h->grid.appends_rebuilt = 0;
    h->qsort.valid = false;
    h->qsort.have_prev = false;
    h->qsort.slot_order = false;
    h->qsort.fold_n = 0;
    h->qsort.last_cert_ctr = nullptr;  // (observability of the previous owner's searches)
    h->qsort.cert_searches = 0;
    h->qsort.have_certs = false;
    h->shard.active = false;
    h->shard.refined_ready = h->shard.attempt_refined = false;
    h->normals_m = 0;
    h->normals_k = 0;
    static const bool no_hints = exp_env("ICP_NO_POOL_HINTS") != nullptr;
    for (int k = 0; k < 5; ++k) {  // what the next owner may start from (common.hpp: hint_kind)
      w.hint_kind[k] = no_hints ? Workspace::WinPred() : w.win_kind[k];
      w.hint_kind[k].wide = false;
    }
    w.hint_last_inner = no_hints ? 0xffffffffu : w.last_inner;
    w.win_valid = w.win_wide = false;
    for (auto &wk : w.win_kind) wk = Workspace::WinPred();
    // whatever the previous owner's last evaluations left in the selection scratch (a parked flag, a
    // half-consumed list) must not steer the next owner's first evaluation: one small launch per
    // scratch at its first use
    w.gn_dirty = w.alt.gn_dirty = true;
    w.win_tried = w.win_missed = w.short_evals = w.radix_evals = 0;
    w.spec_hits = w.spec_misses = w.pre_evals = 0;
    w.ahead_hits = w.ahead_misses = 0;
    w.ahead_on = w.ahead_seen_valid = false;
    w.loop_launches = w.loop_evals = w.loop_handbacks = 0;
    w.loop_off = false;
    w.loop_rank = -1;
    w.loop_world = 0;
    w.tiny_calls = w.tiny_evals = w.tiny_sorted = 0;
    w.refine_tried = w.refine_missed = 0;
    w.last_inner = 0xffffffffu;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_pool.size() < kPoolMax) {
      g_pool.push_back(h);
      return;
    }
  }
  free_handle(h);
}

extern "C" void icp_trim_pool(void) {
  std::vector<icp_handle *> take;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    take.swap(g_pool);
  }
  for (icp_handle *h : take) free_handle(h);
}

extern "C" int icp_set_nn_mode(icp_handle *h, int mode) {
  if (!h || mode < ICP_NN_AUTO || mode > ICP_NN_GRID) return ICP_BAD_ARGUMENT;
  if (mode == ICP_NN_GRID && !h->grid.built && h->m > 0) return ICP_BAD_ARGUMENT;  // non-finite targets
  h->nn_mode = mode;
  return ICP_OK;
}

// AUTO: the grid pays off once the target cloud is large enough to amortise the
// scattered cell reads; tiny clouds (2-D LiDAR scans, ~650 points) stay on the sweep.
int icp::api::resolved_nn_mode(const icp_handle *h) {
  static const long grid_min_m = exp_env("ICP_NN_GRID_MIN_M") ? atol(exp_env("ICP_NN_GRID_MIN_M")) : 8192;
  if (h->nn_mode == ICP_NN_BRUTE || !h->grid.built) return ICP_NN_BRUTE;
  if (h->nn_mode == ICP_NN_GRID) return ICP_NN_GRID;
  return (long)h->m >= grid_min_m ? ICP_NN_GRID : ICP_NN_BRUTE;
}

static hipError_t launch_nn(icp_handle *h, const double *d_src, size_t n, const Pose *T, double *d_a,
                            double *d_b, uint32_t *d_idx) {
  Range range("icp: search (transform + exact nearest neighbour + pairs)");
  if (resolved_nn_mode(h) == ICP_NN_GRID) return launch_nn_grid(h, d_src, n, T, d_a, d_b, d_idx);
  return launch_nn_brute(h, d_src, n, T, d_a, d_b, d_idx);
}

extern "C" int icp_set_single_launch(icp_handle *h, int enable) {
  if (!h) return ICP_BAD_ARGUMENT;
  h->single_launch = enable != 0;
  return ICP_OK;
}
// Observability (include/icp_mi355x_debug.h): enable = 0 makes icp_estimate[_device] run every one of its max_iter outer
// iterations, also those behind a fixed point (same outputs either way: what the benchmark's `all_twenty_run` line times)
extern "C" int icp_set_fixed_point_exit(icp_handle *h, int enable) {
  if (!h) return ICP_BAD_ARGUMENT;
  h->fixed_point_exit = enable != 0;
  return ICP_OK;
}
extern "C" int icp_single_launch_counters(icp_handle *h, uint64_t out[3]) {
  if (!h || !out) return ICP_BAD_ARGUMENT;
  out[0] = h->ws.tiny_calls;
  out[1] = h->ws.tiny_evals;
  out[2] = h->ws.tiny_sorted;
  return ICP_OK;
}
extern "C" int icp_get_nn_mode(const icp_handle *h) {
  if (!h) return ICP_BAD_ARGUMENT;
  return resolved_nn_mode(h);
}
extern "C" int icp_set_stream(icp_handle *h, void *s) {
  if (!h) return ICP_BAD_ARGUMENT;
  h->stream = (hipStream_t)s;  // NULL = the HIP default stream, a legitimate choice
  return ICP_OK;
}
extern "C" int icp_use_own_stream(icp_handle *h) {
  if (!h) return ICP_BAD_ARGUMENT;
  h->stream = h->own_stream;
  return ICP_OK;
}
extern "C" int icp_synchronize(icp_handle *h) {
  if (!h) return ICP_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return ICP_OK;
}

extern "C" int icp_profile_enable(icp_handle *h, int enable) {
  if (!h) return ICP_BAD_ARGUMENT;
  h->profile = enable < 0 ? 0 : enable;
  h->prof_seen = 0;
  if (h->profile > 0) {  // event pairs ahead of the region they will time (creating one costs tens of microseconds)
    HIP_TRY(hipSetDevice(h->device));
    while (h->prof_free.size() < 32) {
      hipEvent_t a = nullptr, b = nullptr;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) break;
      h->prof_free.emplace_back(a, b);
    }
  }
  return ICP_OK;
}
extern "C" int icp_profile_read(icp_handle *h, double *ms, uint64_t *launches) {
  if (!h) return ICP_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  double total = 0.;
  for (auto &ev : h->prof_events) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, ev.first, ev.second) == hipSuccess) total += t;
    h->prof_free.push_back(ev);  // (creating an event pair costs more host time than recording it: reused)
  }
  if (ms) *ms = total;
  if (launches) *launches = h->prof_events.size();
  h->prof_events.clear();
  return ICP_OK;
}

// ------------------------------------------------------------ stage calls --------
extern "C" int icp_correspond_device(icp_handle *h, const double *d_src, size_t n, const icp_pose *T,
                                     double *d_a, double *d_b, uint32_t *d_idx) {
  if (!h || !T || (n > 0 && !d_src) || n >= 0xffffffffull) return ICP_BAD_ARGUMENT;
  if (n == 0) return ICP_OK;
  if (h->m == 0) return ICP_EMPTY_DST;  // index.unwrap() on an empty tree, src/lib.rs:122,165
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(launch_nn(h, d_src, n, T, d_a, d_b, d_idx));
  return ICP_OK;
}

extern "C" int icp_materialize_pairs_device(icp_handle *h, const double *d_src, size_t n, const icp_pose *T,
                                            const uint32_t *d_idx, double *d_a, double *d_b) {
  if (!h || !T || (n > 0 && (!d_src || !d_idx || !d_a || !d_b)) || n >= 0xffffffffull) return ICP_BAD_ARGUMENT;
  if (n == 0) return ICP_OK;
  if (h->m == 0) return ICP_EMPTY_DST;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(launch_materialize(h, d_src, n, *T, d_idx, d_a, d_b));
  return ICP_OK;
}

extern "C" int icp_prepare_source_device(icp_handle *h, const double *d_src, size_t n, const icp_pose *T) {
  if (!h || !T || (n > 0 && !d_src) || n >= 0xffffffffull) return ICP_BAD_ARGUMENT;
  h->qsort.valid = false;
  static const long min_n = exp_env("ICP_QSORT_MIN_N") ? atol(exp_env("ICP_QSORT_MIN_N")) : 16384;
  if (resolved_nn_mode(h) != ICP_NN_GRID || (long)n < min_n) return ICP_OK;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(prepare_queries(h, d_src, n, *T));
  return ICP_OK;
}

namespace icp {
__global__ void k_iota_u32(uint32_t *p, unsigned n) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}
}  // namespace icp

// The fold order of an estimate call that starts at pose T (icp_last_fold_order), applied: d_sorted[k] =
// d_src[perm[k]].  For hosts that drive the stage calls themselves (the sharded drivers) and want the bits of
// icp_estimate_device: sort first, then treat the sorted cloud as the source.  Where icp_estimate_device would
// take no snapshot or keeps the caller's order (sweep engine; up to grid_coop_max() = 65 536 points, ICP_NN_COOP_MAX_N) this
// is a plain copy and the identity permutation.
extern "C" int icp_sort_source_device(icp_handle *h, const double *d_src, size_t n, const icp_pose *T,
                                      double *d_sorted, uint32_t *d_perm) {
  if (!h || !T || (n > 0 && (!d_src || !d_sorted)) || n >= 0xffffffffull) return ICP_BAD_ARGUMENT;
  if (n == 0) return ICP_OK;
  HIP_TRY(hipSetDevice(h->device));
  bool sorted = false;
  if ((long)n > grid_coop_max()) {  // exactly where icp_estimate_device folds in snapshot order
    const int prc = icp_prepare_source_device(h, d_src, n, T);
    if (prc != ICP_OK) return prc;
    sorted = h->qsort.valid && h->qsort.src == d_src && h->qsort.n == n;
  }
  HIP_TRY(hipMemcpyAsync(d_sorted, sorted ? h->qsort.d_sorted : d_src, n * h->dim * sizeof(double),
                         hipMemcpyDeviceToDevice, h->stream));
  if (d_perm) {
    if (sorted)
      HIP_TRY(hipMemcpyAsync(d_perm, h->qsort.d_perm, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, h->stream));
    else
      hipLaunchKernelGGL(icp::k_iota_u32, dim3(((unsigned)n + 255) / 256), dim3(256), 0, h->stream, d_perm, (unsigned)n);
  }
  h->qsort.valid = false;  // (the snapshot belonged to this call)
  h->qsort.have_prev = false;
  return ICP_OK;
}

extern "C" int icp_nn_search_device(icp_handle *h, const double *d_q, size_t n, uint32_t *d_idx) {
  if (!h || (n > 0 && (!d_q || !d_idx)) || n >= 0xffffffffull) return ICP_BAD_ARGUMENT;
  if (n == 0) return ICP_OK;
  if (h->m == 0) return ICP_EMPTY_DST;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(launch_nn(h, d_q, n, nullptr, nullptr, nullptr, d_idx));
  return ICP_OK;
}

// Wait for the fast pipeline's result: poll the sequence number its last workgroup releases
// into pinned memory (a few us earlier than the stream's completion signal, three times per
// outer iteration).  The poll is bounded in TIME, not in spins: an evaluation at the sizes this
// path serves lasts 30-60 us from its launch, so after 250 us something larger is running (a 64M-pair
// evaluation, a cold search ahead of it) and the host thread blocks in hipStreamSynchronize instead
// of burning a core.
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __asm__ __volatile__("pause" ::: "memory");
#elif defined(__aarch64__) || defined(__arm__)
  __asm__ __volatile__("yield" ::: "memory");
#else
  __asm__ __volatile__("" ::: "memory");
#endif
}
hipError_t icp::api::wait_seq(icp_handle *h, volatile unsigned *seq, unsigned want, hipStream_t stream) {
  static const bool no_poll = exp_env("ICP_NO_POLL") != nullptr;
  if (!no_poll) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      for (int spins = 0; spins < 256; ++spins) {
        if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) == want) return hipSuccess;
        cpu_relax();
      }
      if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(250)) break;
    }
  }
  return hipStreamSynchronize(stream ? stream : h->stream);
}
// (stream: where the evaluation was enqueued, if not on h->stream)
hipError_t icp::api::wait_result(icp_handle *h, hipStream_t stream) {
  return wait_seq(h, &h->ws.h_res->seq, h->ws.seq, stream);
}


// what the next evaluation's window is centred on: this evaluation's exact median and sigma, kept per
// kind of evaluation (common.hpp, Workspace::win_kind) and as "the most recent one"
void icp::api::record_statistics(Workspace &w, int kind, bool has_median, const GnResult &r) {
  w.win_valid = has_median;
  if (Workspace::kind_has_slot(kind)) w.win_kind[kind].valid = has_median;
  if (kind == 3 || kind == 4) w.win_kind[kind - 3].valid = has_median;  // (the next iteration's evaluations follow on from these)
  if (has_median)
    for (int d = 0; d < 2; ++d) {
      w.win_med[d] = r.median[d];
      w.win_sigma[d] = r.sigma[d];
      if (Workspace::kind_has_slot(kind)) {
        w.win_kind[kind].med[d] = r.median[d];
        w.win_kind[kind].sigma[d] = r.sigma[d];
      }
      if (kind == 3 || kind == 4) {
        w.win_kind[kind - 3].med[d] = r.median[d];
        w.win_kind[kind - 3].sigma[d] = r.sigma[d];
      }
    }
}

// A handle fresh from the pool: the previous owner's prediction for this kind of evaluation, adopted once.  Only the
// library's own loops call this (estimate_transform_loop): a handle that serves as a rank of a sharded evaluation goes
// through the stage calls, where a rank-local prediction (and the `wide` flag a miss leaves behind) would give the
// ranks different windows for the same histogram sum.
// (Whether the handle has evaluated anything else yet does not matter: the second evaluation of a frame's first iteration
// is predicted far better by the previous frame's second evaluation than by this frame's first -- their medians differ by
// 0.6 sigma on the scan3d frames, a miss even with the widest windows: profiles/r05_frame_trace_fresh.txt.)
static void adopt_pool_hint(Workspace &w, int kind) {
  if (Workspace::kind_has_slot(kind) && !w.win_kind[kind].valid && w.hint_kind[kind].valid) {
    w.win_kind[kind] = w.hint_kind[kind];
    w.hint_kind[kind].valid = false;
  }
}

// weighted_gauss_newton_update on device pairs (src/lib.rs:218-261); also yields the
// Huber error of the same T (src/lib.rs:75), which shares the pass.
// `after_launch` (optional) runs once, right after the first attempt's kernels have been enqueued
// and before the host waits for them: the place to enqueue work that does not depend on the result.
// `pre_launched`: the window pipeline for exactly this evaluation is already in flight on this
// stream (icp_estimate_device enqueued it behind the speculative search); only its result is awaited.
template <typename Hook>
static int wgn_step(icp_handle *h, const double *d_a, const double *d_b, size_t n, const Pose &T,
                    double delta[3], double *huber_err, Hook &&after_launch, bool pre_launched = false,
                    int kind = 2) {
  static const bool force_radix = exp_env("ICP_GN_RADIX") != nullptr;
  Range range("icp: evaluation (weighted_gauss_newton_update + huber_error)");
  Workspace &w = h->ws;
  bool done = false, has_median = false, hooked = false;
  hipStream_t on_stream = nullptr;  // where the evaluation in hand was enqueued, if not on h->stream
  // the prediction this evaluation's window is (or, pre-launched, was) centred on: its own kind's
  // previous evaluation if there is one, else the most recent evaluation (window_usable)
  const bool own = Workspace::kind_has_slot(kind) && w.win_kind[kind].valid;
  bool &wide = own ? w.win_kind[kind].wide : w.win_wide;
  double p_med[2], p_sigma[2];
  for (int d = 0; d < 2; ++d) {
    p_med[d] = own ? w.win_kind[kind].med[d] : w.win_med[d];
    p_sigma[d] = own ? w.win_kind[kind].sigma[d] : w.win_sigma[d];
  }
  if (w.gn_dirty) {  // first use, or the radix path / a NaN left its state behind (the short pipelines clean up after themselves)
    HIP_TRY(launch_sel_init(h, n));
    w.gn_dirty = false;
  }
  if (!force_radix) {
    WinParams P;
    if (pre_launched || w.bkt_pair_launched || window_usable(h, n, &P, kind)) {  // launches around the predicted median and sigma
      if (!pre_launched && w.bkt_pair_launched) {
        // this evaluation went out on the search stream, side by side with the next iteration's first evaluation
        // (launch_bkt_pair): only its result is awaited
        w.bkt_pair_launched = false;
        on_stream = h->own_stream;
        HIP_TRY(after_launch());
        hooked = true;
      } else if (!pre_launched) {
        ++w.win_tried;
        HIP_TRY(launch_weighted_gn_win(h, d_a, d_b, n, T, P));
        HIP_TRY(after_launch());
        hooked = true;
      }
#ifdef ICP_EXPERIMENTS
      const auto tw0 = std::chrono::steady_clock::now();
#endif
      HIP_TRY(wait_result(h, on_stream));
#ifdef ICP_EXPERIMENTS
      (pre_launched ? w.dbg_wait_pre_us : w.dbg_wait_other_us) +=
          std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tw0).count();
#endif
      done = has_median = !w.h_res->overflow;
      if (pre_launched) {  // (the run-ahead search behind it read the same pose from the device)
        w.ahead_seen_valid = w.h_res->next_valid != 0;
        w.ahead_seen_pose = w.h_res->next_pose;
      }
      if (!done && w.h_res->overflow == 3) {  // the buckets were too small, not the window wrong: no wider windows,
        ++w.bkt_misses;                       // the next evaluations of this handle take the second pass instead
        w.bkt_off = 64;
      } else if (!done) {
        if (exp_env("ICP_WIN_TRACE")) fprintf(stderr, "[win] kind %d: the window missed\n", kind);
        ++w.win_missed;
        wide = true;
      } else if (wide) {  // back to narrow windows once the statistics have settled
        double shift = 0.;
        for (int d = 0; d < 2; ++d)
          shift = fmax(shift, (fabs(w.h_res->median[d] - p_med[d]) + fabs(w.h_res->sigma[d] - p_sigma[d])) / p_sigma[d]);
        if (shift < 0.01) wide = false;
      }
    }
    if (!done && refine_applies(n)) {
      // beyond 4M points: windows refined in two passes (gn_win.hip).  The exact statistics of a
      // strided 1M-pair sample centre the first pass ...
      ++w.refine_tried;
      HIP_TRY(launch_sample_pairs(h, d_a, d_b, n));
      HIP_TRY(launch_weighted_gn_pull(h, w.d_sa, w.d_sb, kRefineSample, T));
      HIP_TRY(wait_result(h));
      WinParams P1, P2;
      bool ok = !w.h_res->overflow && !w.h_res->nan_flag;
      if (ok) {
        const double m[2] = {w.h_res->median[0], w.h_res->median[1]}, sg[2] = {w.h_res->sigma[0], w.h_res->sigma[1]};
        ok = make_window(m, sg, 0.02, &P1);
      }
      if (ok) {  // ... whose counts place the windows of the second
        HIP_TRY(launch_win_first_pass(h, d_a, d_b, n, T, P1));
        HIP_TRY(hipStreamSynchronize(h->stream));
        ok = refine_window(w.h_whist, n, P1, &P2);
      }
      if (ok) {
        ++w.win_tried;
        HIP_TRY(launch_win_second_pass(h, d_a, n, T, P2));
        if (!hooked) HIP_TRY(after_launch());
        hooked = true;
        HIP_TRY(wait_result(h));
        done = has_median = !w.h_res->overflow;
        if (!done) ++w.win_missed;
      }
      if (!done) ++w.refine_missed;
    }
    if (!done) {
      ++w.short_evals;
      HIP_TRY(launch_weighted_gn_fast(h, d_a, d_b, n, T));
      if (!hooked) HIP_TRY(after_launch());
      hooked = true;
      HIP_TRY(wait_result(h));
      done = !w.h_res->overflow;
      has_median = done && n > 1024;  // gn_pull.hip reports the median, the single-workgroup kernel does not
    }
  }
  if (!done) {  // heavy duplicates around a median: the general 6-pass radix select
    ++w.radix_evals;
    HIP_TRY(launch_sel_init(h, n));
    HIP_TRY(launch_weighted_gn(h, d_a, d_b, n, T));
    if (!hooked) HIP_TRY(after_launch());
    HIP_TRY(hipStreamSynchronize(h->stream));
    w.gn_dirty = true;
  }
  const GnResult &r = *w.h_res;
  static const bool win_trace = exp_env("ICP_WIN_TRACE") != nullptr;
  if (win_trace)
    fprintf(stderr, "[win] kind %d own %d predicted med %.6g %.6g sigma %.6g %.6g -> med %.6g %.6g sigma %.6g %.6g%s\n", kind,
            (int)own, p_med[0], p_med[1], p_sigma[0], p_sigma[1], r.median[0], r.median[1], r.sigma[0], r.sigma[1],
            has_median ? "" : " (no statistics)");
  record_statistics(w, kind, has_median, r);
  if (r.nan_flag) {
    w.gn_dirty = true;
    return ICP_NAN_INPUT;
  }
  if (huber_err) *huber_err = r.acc[12];
  Range solve("icp: solve (inverse3x3, host)");
  return solve_update(r.acc, r.acc + 9, delta) ? ICP_OK : ICP_NONE;
}
int icp::api::wgn_step(icp_handle *h, const double *d_a, const double *d_b, size_t n, const Pose &T,
                       double delta[3], double *huber_err, bool pre_launched, int kind) {
  return wgn_step(h, d_a, d_b, n, T, delta, huber_err, [] { return hipSuccess; }, pre_launched, kind);
}

// ---- the inner loop in one launch (gn_loop.hip) --------------------------------------------------------------
// One resident loop launch per DEVICE and process at a time: two of them on one device, each waiting at its grid barrier
// for workgroups the other one keeps off the CUs, would only end by timeout.  (Launches of other processes, and the
// sharded launches of this one -- whose launch and wait are two calls of the caller -- are not covered: such a
// collision ends in a bounded wait, the handle steps from the host for the next kLoopOffCalls loops and tries again.)
constexpr int kLoopMutexes = 64;
static std::mutex g_loop_mu[kLoopMutexes];
static std::mutex &loop_mutex(const icp_handle *h) { return g_loop_mu[(unsigned)h->device % kLoopMutexes]; }
// A launch that was not resident switches the one-launch loop off for the handle's next kLoopOffCalls inner loops (they
// are stepped from the host), then the handle tries again: a co-tenant that has left costs nothing further, one that
// stays costs one bounded wait (2 ms) in kLoopOffCalls loops.
constexpr unsigned kLoopOffCalls = 64;
void icp::api::loop_timed_out(Workspace &w) {
  w.loop_off = true;
  w.loop_off_calls = kLoopOffCalls;
  ++w.loop_timeouts;
}
static bool loop_allowed(Workspace &w) {
  if (!w.loop_off) return true;
  if (w.loop_off_calls > 0 && --w.loop_off_calls == 0) {
    w.loop_off = false;
    return true;
  }
  return false;
}

hipError_t icp::api::ensure_loop(icp_handle *h) {
  Workspace &w = h->ws;
  if (w.d_loop_ctl) return hipSuccess;
  hipError_t e;
  const size_t hist_bytes = (size_t)4 * kWinBins * sizeof(uint32_t);
  if ((e = hipMalloc(&w.d_loop_ctl, sizeof(LoopCtl))) != hipSuccess) return e;
  if ((e = hipMalloc(&w.d_loop_hist, hist_bytes)) != hipSuccess) return e;
  if ((e = hipMalloc(&w.d_loop_part, gn_loop_partials_doubles() * sizeof(double))) != hipSuccess) return e;
  if ((e = hipHostMalloc(&w.h_loop_res, sizeof(LoopResult), hipHostMallocCoherent)) != hipSuccess) return e;
  memset(w.h_loop_res, 0, sizeof(LoopResult));
  if ((e = hipMemsetAsync(w.d_loop_ctl, 0, sizeof(LoopCtl), h->stream)) != hipSuccess) return e;
  return hipMemsetAsync(w.d_loop_hist, 0, hist_bytes, h->stream);
}

static void record_values(Workspace &w, int kind, const double med[2], const double sigma[2]) {
  GnResult r = {};
  for (int d = 0; d < 2; ++d) {
    r.median[d] = med[d];
    r.sigma[d] = sigma[d];
  }
  record_statistics(w, kind, true, r);
}

// false: the handle has no prediction for evaluation `it` (nothing is launched).  `hints`: a handle fresh from the pool
// may adopt its previous owner's predictions -- never a rank of a sharded registration, whose windows must be the
// other ranks' (DESIGN.md section 7).
bool icp::api::loop_plan(icp_handle *h, size_t n, int it, int first_kind, int second_kind, bool hints, LoopPlan *pl) {
  Workspace &w = h->ws;
  pl->first_kind = first_kind;
  pl->second_kind = second_kind;
  pl->it0 = it;
  pl->A = LoopArgs{};
  int kind0 = pl->kind_of(it);
  const int kind1 = pl->kind_of(it + 1);
  if (hints) adopt_pool_hint(w, kind0);
  // Whose statistics predict a loop's FIRST evaluation?  After a loop of one update (a settled registration) the
  // previous loop's first evaluation: the populations "new correspondences" / "after the update" alternate (common.hpp:
  // Workspace::win_kind).  After a loop of several updates the previous loop's LAST evaluation: same outer pose, and
  // the re-matched pairs differ little from the old ones, while that loop's first evaluation lies many updates back
  // (the converging pair: three of these windows per call missed even at 0.2 sigma, each costing a handed-back
  // evaluation).  The statistics are still RECORDED under the evaluation's own kind.
  if (hints && it == 0 && w.win_valid && w.last_inner != 0xffffffffu && w.last_inner >= 2u) kind0 = 2;
  if (!window_usable(h, n, &pl->A.PA, kind0, !hints)) return false;
  pl->own0 = Workspace::kind_has_slot(kind0) && w.win_kind[kind0].valid;
  for (int d = 0; d < 2; ++d) {
    pl->p_med[0][d] = pl->own0 ? w.win_kind[kind0].med[d] : w.win_med[d];
    pl->p_sigma[0][d] = pl->own0 ? w.win_kind[kind0].sigma[d] : w.win_sigma[d];
  }
  // the second evaluation of the launch: the host's own prediction for that KIND of evaluation if it has one (the
  // population after the first update differs from the first one's, common.hpp: Workspace::win_kind), else the launch
  // centres it on its first evaluation like every later one
  if (hints) adopt_pool_hint(w, kind1);
  const bool own1 = kind1 != 2 && Workspace::kind_has_slot(kind1) && w.win_kind[kind1].valid;
  pl->A.pb_valid = own1 && window_usable(h, n, &pl->A.PB, kind1, !hints) ? 1 : 0;
  if (pl->A.pb_valid)
    for (int d = 0; d < 2; ++d) {
      pl->p_med[1][d] = w.win_kind[kind1].med[d];
      pl->p_sigma[1][d] = w.win_kind[kind1].sigma[d];
    }
  pl->A.f_next = window_half_width(n, w.win_wide);
  return true;
}

// The launch's result into the loop's state and the handle's prediction history.  *finished, or evaluation *it is the
// caller's to serve (a window missed, a rotation beyond the restated sin / cos).
int icp::api::loop_finish(icp_handle *h, const LoopPlan &pl, const LoopResult *res, Pose *T, double *prev_error,
                       uint32_t *applied, int *it, bool *finished) {
  Workspace &w = h->ws;
  *finished = false;
  ++w.loop_launches;
  w.loop_evals += res->evals;
  w.win_tried += res->evals + (res->status == 1 ? 1u : 0u);
  // the statistics the next predictions are made from, in the order the evaluations ran
  for (unsigned e = 0; e < res->evals && e < 2u; ++e) {
    const int kind = pl.kind_of(pl.it0 + (int)e);
    record_values(w, kind, res->med[e], res->sigma[e]);
    if (e == 0 || pl.A.pb_valid) {  // back to narrow windows once a kind's statistics have settled (wgn_step)
      bool &wide = (e == 0 ? pl.own0 : true) ? w.win_kind[kind].wide : w.win_wide;
      double shift = 0.;
      for (int d = 0; d < 2; ++d)
        shift = fmax(shift, (fabs(res->med[e][d] - pl.p_med[e][d]) + fabs(res->sigma[e][d] - pl.p_sigma[e][d])) / pl.p_sigma[e][d]);
      if (wide && shift < 0.01) wide = false;
    }
  }
  if (res->evals > 2u) {
    record_values(w, 2, res->med[2], res->sigma[2]);
    w.win_wide = false;
  }
  *T = res->Ti;
  *prev_error = res->prev_error;
  *applied = res->applied;
  if (res->status == 3) {
    w.gn_dirty = true;
    return ICP_NAN_INPUT;
  }
  if (res->finished) {
    *finished = true;
    return ICP_OK;
  }
  ++w.loop_handbacks;
  *it = (int)res->it;
  if (res->status == 1) {  // evaluation *it missed its window: what a miss leaves behind (wgn_step)
    ++w.win_missed;
    const int kind = pl.kind_of(*it);
    const bool own = Workspace::kind_has_slot(kind) && w.win_kind[kind].valid;
    (own ? w.win_kind[kind].wide : w.win_wide) = true;
  }
  return ICP_OK;
}

// From evaluation *it on, as far as the device gets by itself.  *served = false: nothing was launched (no window
// prediction for evaluation *it) -- the caller steps once from the host.
static int gn_loop_run(icp_handle *h, const double *d_a, const double *d_b, size_t n, int first_kind, int second_kind,
                       Pose *T, double *prev_error, uint32_t *applied, int *it, bool *finished, bool *served) {
  Workspace &w = h->ws;
  *served = *finished = false;
  LoopPlan pl;
  if (!loop_plan(h, n, *it, first_kind, second_kind, true, &pl)) return ICP_OK;
  Range range("icp: inner loop (one launch: evaluations + solve + break tests on the device)");
  LoopArgs &A = pl.A;
  HIP_TRY(ensure_loop(h));
  if (w.gn_dirty) {  // (the host-driven pipelines' rest state; the launch itself does not touch it)
    HIP_TRY(launch_sel_init(h, n));
    w.gn_dirty = false;
  }
  A.a = (const double2 *)d_a;
  A.b = (const double2 *)d_b;
  A.n = (unsigned)n;
  A.it0 = (unsigned)*it;
  A.applied0 = *applied;
  A.T0 = *T;
  A.prev_error0 = *prev_error;
  A.whist = w.d_loop_hist;
  A.wmed = w.d_wmed;
  A.wring = w.d_wring;
  A.partials = w.d_loop_part;
  A.ctl = reinterpret_cast<LoopCtl *>(w.d_loop_ctl);
  LoopResult *res = reinterpret_cast<LoopResult *>(w.h_loop_res);
  A.res = res;
  A.seq = ++w.loop_seq;
  {
    std::lock_guard<std::mutex> lk(loop_mutex(h));
    HIP_TRY(launch_gn_loop(h, A));
    HIP_TRY(wait_seq(h, &res->seq, A.seq));
  }
  *served = true;
  if (res->status == 5) {  // not resident: put the scratch back into its rest state and step from the host from now on
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemsetAsync(w.d_loop_ctl, 0, sizeof(LoopCtl), h->stream));
    HIP_TRY(hipMemsetAsync(w.d_loop_hist, 0, (size_t)4 * kWinBins * sizeof(uint32_t), h->stream));
    loop_timed_out(w);
  }
  return loop_finish(h, pl, res, T, prev_error, applied, it, finished);
}

void icp::api::free_loop_plan(void *p) { delete reinterpret_cast<LoopPlan *>(p); }

// estimate_transform (src/lib.rs:59-84) on device pairs.  Pair sets of up to 2^20 run the loop on the device
// (gn_loop.hip: one launch, gn_loop_run above); the host steps only where that launch hands an evaluation back, and
// for larger sets.  `second_eval_hook(T1)` (optional, host-stepped loops only) is
// called when the evaluation at the once-updated pose T1 has been enqueued: if that evaluation
// ends the loop -- the usual case once a registration has settled -- T1 is the result, so the
// caller may start work for it while the device is still evaluating.
// With `eval_stream` set, the evaluations after the first run on that stream (see
// icp_estimate_device); `hook_first` calls the hook before that evaluation is enqueued instead of
// after.
template <typename Hook>
static int estimate_transform_loop(icp_handle *h, const double *d_a, const double *d_b, size_t n, Pose *out,
                                   uint32_t *inner_iters, Hook &&second_eval_hook,
                                   hipStream_t eval_stream = nullptr, bool hook_first = false,
                                   bool first_pre_launched = false, int first_kind = 0, int second_kind = 1,
                                   bool allow_device_loop = true) {
  Pose T = transform_identity();
  uint32_t applied = 0;
  if (input_size_ok(n)) {
    double prev_error = DBL_MAX;  // f64::MAX, src/lib.rs:63
    const bool device_loop = allow_device_loop && !first_pre_launched && gn_loop_applies(n);
    for (int it = 0; it < ICP_INNER_MAX_ITER; ++it) {
      if (device_loop && loop_allowed(h->ws)) {
        bool finished = false, served = false;
        const int rc = gn_loop_run(h, d_a, d_b, n, first_kind, second_kind, &T, &prev_error, &applied, &it, &finished,
                                   &served);
        if (rc != ICP_OK) return rc;
        if (finished) break;
        if (it >= ICP_INNER_MAX_ITER) break;
        (void)served;  // either way evaluation `it` is stepped from the host now
      }
      double delta[3], err = 0.;
      hipStream_t first_stream = h->stream;
      const bool on_eval_stream = it >= 1 && eval_stream;
      if (on_eval_stream) {  // the other stream and its own evaluation scratch
        h->stream = eval_stream;
        h->ws.swap_ctx();
      }
#ifdef ICP_EXPERIMENTS
      // (timing experiment only: what an outer iteration costs WITHOUT its deciding evaluation -- the loop is taken to end
      // after one update, as it does on the benchmark pair)
      if (it == 1 && exp_env("ICP_HACK_SKIP_E2")) {
        (void)second_eval_hook(T);
        if (on_eval_stream) {
          h->stream = first_stream;
          h->ws.swap_ctx();
        }
        break;
      }
#endif
      if (it == 1 && hook_first) {
        const hipError_t he = second_eval_hook(T);
        if (he != hipSuccess) {
          if (on_eval_stream) {
            h->stream = first_stream;
            h->ws.swap_ctx();
          }
          return map_hip(he);
        }
      }
      {
        const int kind_now = it == 0 ? first_kind : (it == 1 ? second_kind : 2);
        if (!(it == 0 && first_pre_launched)) adopt_pool_hint(h->ws, kind_now);
      }
      const int rc = (it == 1 && !hook_first)
                         ? wgn_step(h, d_a, d_b, n, T, delta, &err, [&] { return second_eval_hook(T); }, false, second_kind)
                         : wgn_step(h, d_a, d_b, n, T, delta, &err, it == 0 && first_pre_launched,
                                    it == 0 ? first_kind : (it == 1 ? second_kind : 2));
      if (on_eval_stream) {
        h->stream = first_stream;
        h->ws.swap_ctx();
      }
      if (rc == ICP_NONE) break;           // src/lib.rs:67-69
      if (rc != ICP_OK) return rc;
      if ((delta[0] * delta[0] + delta[1] * delta[1]) + delta[2] * delta[2] < ICP_DELTA_NORM_THRESHOLD)
        break;                             // src/lib.rs:71-73
      if (err > prev_error) break;         // src/lib.rs:75-78
      prev_error = err;
      T = transform_mul(transform_new(delta), T);  // src/lib.rs:81
      ++applied;
    }
  }
  *out = T;
  if (inner_iters) *inner_iters = applied;
  return ICP_OK;
}

extern "C" int icp_estimate_transform_device(icp_handle *h, const double *d_a, const double *d_b, size_t n,
                                             icp_pose *out, uint32_t *inner_iters) {
  if (!h || !out || (n > 0 && (!d_a || !d_b)) || n >= 0xffffffffull) return ICP_BAD_ARGUMENT;
  if (input_size_ok(n)) {
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(ensure_workspace(h, n, false));
  }
  return estimate_transform_loop(h, d_a, d_b, n, out, inner_iters, [](const Pose &) { return hipSuccess; });
}

// Icp{2,3}d::estimate (src/lib.rs:105-130, 148-173) on a device-resident source cloud.
//
// Speculative search: once the inner loop has needed exactly one update in the previous outer
// iteration, the next pose is known as soon as that update is (T_next = Exp(delta_1) * T, provided
// the evaluation at the updated pose ends the loop).  The search for T_next is then enqueued
// together with that evaluation instead of after the host has seen its result, into the second
// pair buffer.  The guess is verified bit for bit (the pose the loop really returns must equal
// the speculated one); on a mismatch the speculative pairs are ignored and the search runs again
// for the true pose.  Results cannot depend on it.
//
// Two streams with fixed roles and no device-side dependency between them (resolving a
// cross-queue event costs ~12 us here; every hand-over below goes through a host wait that the
// loop needs anyway):  the handle's stream runs search -> first evaluation -> search -> ...; the
// second, high-priority stream runs the later evaluations of an inner loop.  The search
// (latency-bound, 4 waves/SIMD) and the evaluation it bets on (one workgroup per CU) then share
// the CUs instead of queueing behind each other.
extern "C" int icp_estimate_device(icp_handle *h, const double *d_src, size_t n, const icp_pose *init,
                                   size_t max_iter, icp_pose *out, uint32_t *d_last_idx,
                                   uint32_t *inner_iters) {
  if (!h || !init || !out || (n > 0 && !d_src) || n >= 0xffffffffull) return ICP_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(ensure_workspace(h, n, false));
  // Whatever way this call ends, nothing of it may still be in flight afterwards: a speculative
  // search or a pre-launched evaluation reads the caller's d_src and writes its d_last_idx, and the
  // cell-sorted snapshot is keyed on (pointer, n) of a buffer the caller may now rewrite.
  struct Quiesce {
    icp_handle *h;
    ~Quiesce() {
      h->ws.bkt_pair_launched = h->ws.alt.bkt_pair_launched = false;  // (both halves of a pair are in flight or done: nothing stays behind)
      (void)hipStreamSynchronize(h->stream);
      if (h->ws.spec_stream) (void)hipStreamSynchronize(h->ws.spec_stream);
      h->qsort.valid = false;
      h->qsort.have_prev = false;
      h->qsort.slot_order = false;
    }
  } quiesce_on_exit{h};
  h->qsort.fold_n = 0;  // (identity, until this call takes a snapshot)
  // the reference's own sizes (2-D scans of ~650 points): the whole call in one launch on one CU
  if (n > 0 && max_iter > 0 && h->m > 0) {
    int tiny_status = -1;
    HIP_TRY(launch_tiny_estimate(h, d_src, n, *init, max_iter, out, d_last_idx, inner_iters, &tiny_status));
    if (tiny_status == 0) {
      ++h->ws.tiny_calls;
      return ICP_OK;
    }
    if (tiny_status == 3) return ICP_NAN_INPUT;
  }
  static const bool no_spec = getenv("ICP_NO_SPECULATION") != nullptr;
  static const bool one_stream_env = exp_env("ICP_SPEC_SAME_STREAM") != nullptr;
  static const bool nn_first = exp_env("ICP_SPEC_NN_LAST") == nullptr;
  Workspace &w = h->ws;
  // Two ways through an outer iteration, chosen per iteration from how the previous inner loop went:
  //  * it applied exactly ONE update (a settled registration; the benchmark pair): the host steps the two evaluations
  //    and bets on the pose after the first update -- the search for it runs beside the second evaluation (below);
  //  * anything else: the inner loop is ONE launch (gn_loop.hip), an iteration is search -> loop launch -> one host
  //    wait, however many updates the loop applies; no bet.
  const bool loop_ok = gn_loop_applies(n);
  // with a caller-supplied stream everything stays on that stream
  const bool two_streams = !no_spec && !one_stream_env && h->stream == h->own_stream;
  Pose T = *init;
  if (max_iter > 0) {
    const int prc = icp_prepare_source_device(h, d_src, n, init);
    if (prc != ICP_OK) return prc;
  }
  // With a snapshot, the whole call lives in its order (QuerySort::slot_order): searches store the
  // pairs of slot k at k, the evaluations fold them as they lie (the reduction order of DESIGN.md
  // section 3 applied to the sorted cloud; icp_last_fold_order hands the permutation to whoever wants to
  // reproduce the sums), and only the indices of the last search go back to the caller's order.
  // (clouds small enough for the four-lanes-per-query search keep the caller's order: their scatter is cheap,
  // and frame-sized registrations stay comparable with stage-call drivers point for point)
  const bool slot = h->qsort.valid && h->qsort.src == d_src && h->qsort.n == n && (long)n > grid_coop_max();
  h->qsort.slot_order = slot;
  if (slot) h->qsort.fold_n = n;
  uint32_t *const idx_target = slot ? w.d_idx_slot : d_last_idx;
  double *A[3] = {w.d_a, w.d_a2, w.d_a3}, *B[3] = {w.d_b, w.d_b2, w.d_b3};
  int cur = 0;
  bool spec_valid = false, pre_valid = false, first_pre_launched = false;
  // Run-ahead search: behind the pre-launched first evaluation of the NEXT iteration the search of the iteration
  // after it is enqueued at once, with its pose read from the device (k_win_finish solves the update there) -- the
  // search stream then runs search -> evaluation -> search -> ... without waiting for the host between them.  The
  // host still derives every pose itself and takes the pairs only if the device's pose has the same bits.
  static const bool no_ahead = exp_env("ICP_NO_RUN_AHEAD") != nullptr;  // (ICP_NO_SPECULATION switches it off with the bet: no second stream)
  const bool can_ahead = two_streams && !no_ahead && resolved_nn_mode(h) == ICP_NN_GRID;
  bool ahead_issued = false;  // an ahead search into the buffers after next is in flight behind the current pre-evaluation
  static const bool no_pre = exp_env("ICP_NO_PRE_EVAL") != nullptr;
  static const bool pair_evals = !(exp_env("ICP_PAIR_EVALS") && atoi(exp_env("ICP_PAIR_EVALS")) == 0);
  bool hooked_first = false;  // this iteration's hook runs in front of the deciding evaluation's launches
  Pose spec_pose = T;
  // the bet needs "the inner loop took exactly one update last time"; across calls the handle
  // remembers how its previous call ended (a new frame usually behaves like the last one)
  uint32_t prev_inner = w.last_inner != 0xffffffffu ? w.last_inner : w.hint_last_inner;  // (a pooled handle: its previous owner's)
  hipStream_t search_stream = h->stream;
  for (size_t it = 0; it < max_iter; ++it) {
    if (spec_valid && memcmp(&spec_pose, &T, sizeof(Pose)) == 0) {
      cur = (cur + 1) % 3;  // the pairs of this pose are already in (or on their way into) the next buffers
      ++w.spec_hits;
      first_pre_launched = pre_valid;
    } else {
      first_pre_launched = false;  // (a pre-launched evaluation of discarded pairs just runs out; nobody reads it)
      ahead_issued = false;        // (... and so does a run-ahead search behind it)
      if (spec_valid) ++w.spec_misses;  // the discarded search precedes this one on the same stream
      uint32_t *idx_out = (it + 1 == max_iter && d_last_idx) ? idx_target : nullptr;
      const int rc = icp_correspond_device(h, d_src, n, &T, A[cur], B[cur], idx_out);
      if (rc != ICP_OK) return rc;
    }
    spec_valid = false;
    pre_valid = false;
    const bool device_loop = loop_ok && !first_pre_launched && !(two_streams && prev_inner == 1) && loop_allowed(w);
    const bool speculate = !device_loop && !no_spec && n > 0 && h->m > 0 && it + 1 < max_iter && prev_inner == 1;
    auto launch_spec = [&](const Pose &T1) -> hipError_t {
      spec_pose = transform_mul(T1, T);  // src/lib.rs:127, 170 -- what the outer loop will compute
      uint32_t *idx_out = (it + 2 == max_iter && d_last_idx) ? idx_target : nullptr;
      const int nxt = (cur + 1) % 3, nxt2 = (cur + 2) % 3;
      spec_valid = true;
      pre_valid = false;
      hipStream_t eval_stream = h->stream;  // the hook runs inside the second evaluation
      h->stream = search_stream;
      hipError_t e = hipSuccess;
      // the search for spec_pose: already behind this iteration's pre-launched evaluation if the device derived the
      // same pose (its pairs go where this search's would)
      const bool have_search = ahead_issued && w.ahead_seen_valid && memcmp(&w.ahead_seen_pose, &spec_pose, sizeof(Pose)) == 0;
      // (a run-ahead search behind an evaluation that MISSED its window never ran -- the device left no pose for it --
      // and is neither: the misses count the searches whose device pose the host did not reproduce)
      if (ahead_issued && (have_search || w.ahead_seen_valid)) ++(have_search ? w.ahead_hits : w.ahead_misses);
      ahead_issued = false;
      if (!have_search) e = launch_nn(h, d_src, n, &spec_pose, A[nxt], B[nxt], idx_out);
      if (e == hipSuccess && two_streams && !no_pre) {
        // ... and the next iteration's FIRST evaluation (inner pose = identity) right behind it,
        // in the search stream's own evaluation scratch: if the bet holds, its result is waiting
        // when the host gets there
        WinParams P;
        w.swap_ctx();
        if (w.gn_dirty) {
          e = launch_sel_init(h, n);
          w.gn_dirty = false;
        }
        if (e == hipSuccess && window_usable(h, n, &P, 0)) {
          ++w.win_tried;
          ++w.pre_evals;
          const bool go_ahead = can_ahead && it + 2 < max_iter;
          // Round 5: where both evaluations can file their candidates, the launch that FINISHES this pre-launched one
          // (one workgroup, a serial chain the next search waits for) also carries the first launch of the DECIDING
          // evaluation of the current iteration (pairs A[cur] at T1) -- the chip is idle during that chain, and the
          // deciding evaluation no longer shares the CUs with the search (it cost 13 us of a 123-us step there).
          // `hooked_first`: the hook runs before the deciding evaluation's own launches, which then find their first
          // launch done (wgn_step: bkt_pair_launched).
          WinParams P2;
          const int kind2 = it == 0 ? 4 : 1;
          bool pair = pair_evals && hooked_first && w.bkt_off == 0 && !w.alt.gn_dirty && !w.alt.bkt_pair_launched && bkt_fits(n, P);
          if (pair) {
            adopt_pool_hint(w, kind2);  // (estimate_transform_loop would, in front of that evaluation)
            pair = window_usable(h, n, &P2, kind2) && bkt_fits(n, P2);
          }
          if (pair) {
            ++w.win_tried;
            e = launch_bkt_pair(h, h->stream, w, A[nxt], B[nxt], P, go_ahead, spec_pose, w.alt, A[cur], B[cur], T1, P2, n);
          } else {
            w.ahead_on = go_ahead;
            w.ahead_outer = spec_pose;
            e = launch_weighted_gn_win(h, A[nxt], B[nxt], n, transform_identity(), P);
            w.ahead_on = false;
          }
          pre_valid = true;
          if (e == hipSuccess && go_ahead) {  // the search of the iteration after next, behind that evaluation
            uint32_t *idx_out2 = (it + 3 == max_iter && d_last_idx) ? idx_target : nullptr;
            e = launch_nn_grid_ahead(h, d_src, n, w.d_ahead, A[nxt2], B[nxt2], idx_out2, &ahead_issued);
          }
        }
        w.swap_ctx();
      }
      h->stream = eval_stream;
      return e;
    };
    auto hook = [&](const Pose &T1) -> hipError_t { return speculate ? launch_spec(T1) : hipSuccess; };
    Pose dT;
    uint32_t inner = 0;
    // two streams: the search is enqueued first; the evaluation's workgroups arrive on the
    // high-priority stream and are placed as soon as a CU has room
    w.search_beside_eval = speculate;
    // (the hook -- next search / pre-evaluation / run-ahead search -- goes in front of the deciding evaluation's launches
    // unless the next search is in flight already and the cloud is frame-sized: there the host's launches are the
    // critical path)
    hooked_first = two_streams && nn_first && !device_loop && !(ahead_issued && (long)n <= grid_coop_max());
    const int rc = estimate_transform_loop(h, A[cur], B[cur], n, &dT, &inner, hook,
                                           two_streams && !device_loop ? w.spec_stream : nullptr, hooked_first,
                                           first_pre_launched, it == 0 ? 3 : 0,
                                           it == 0 ? 4 : 1, device_loop);
    w.search_beside_eval = false;
    if (rc != ICP_OK) return rc;
    if (inner_iters) inner_iters[it] = inner;
    prev_inner = inner;
    w.last_inner = inner;
    const Pose T_next = transform_mul(dT, T);  // src/lib.rs:127, 170
    // An outer iteration that leaves the pose as it found it, bit for bit, is a fixed point of the loop: every later
    // iteration repeats it (correspondences and updates are functions of the pose and the two clouds -- whichever
    // kernels serve them return the same bits).  Only the last one still runs: it reports the correspondences.  A
    // settled registration stops paying for the iterations the reference spends re-deriving the same zero update, and
    // a cloud registered against itself (the first frame of examples/scan3d.rs) for twenty rounds of its slowest path.
    if (h->fixed_point_exit && inner == 0 && it + 2 < max_iter && memcmp(&T_next, &T, sizeof(Pose)) == 0) {
      if (inner_iters)
        for (size_t k = it + 1; k + 1 < max_iter; ++k) inner_iters[k] = 0;
      w.fixed_point_skips += max_iter - 2 - it;
      it = max_iter - 2;
    }
    T = T_next;
  }
#ifdef ICP_EXPERIMENTS
  if (exp_env("ICP_STEP_TRACE")) {
    fprintf(stderr, "[step trace] %zu iterations: waiting for pre-launched first evaluations %.1f us, for other evaluations %.1f us\n",
            max_iter, w.dbg_wait_pre_us, w.dbg_wait_other_us);
    w.dbg_wait_pre_us = w.dbg_wait_other_us = 0.;
  }
#endif
  if (slot && d_last_idx && max_iter > 0 && n > 0) HIP_TRY(launch_unpermute_idx(h, w.d_idx_slot, n, d_last_idx));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (two_streams) HIP_TRY(hipStreamSynchronize(w.spec_stream));
  *out = T;
  return ICP_OK;  // (quiesce_on_exit invalidates the snapshot: the caller may reuse or rewrite the source buffer)
}

// The order in which the last icp_estimate[_device] call on `h` folded its sums (QuerySort::slot_order):
// perm[k] = original index of the k-th point of that order, cell[k] = its sort key (the target-grid cell of
// init * src[perm[k]]); the identity (cells 0) when that call took no snapshot.  Host buffers, either may be null.
extern "C" int icp_last_fold_order(icp_handle *h, size_t n, uint32_t *perm, uint32_t *cell) {
  if (!h || n >= 0xffffffffull) return ICP_BAD_ARGUMENT;
  if (n == 0) return ICP_OK;
  if (h->qsort.fold_n == n && h->qsort.d_perm && h->qsort.d_cell_of) {
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    // (the sort leaves the permutation alone: the sorted keys are the keys gathered through it)
    std::vector<uint32_t> p(n), c(cell ? n : 0);
    HIP_TRY(hipMemcpy(p.data(), h->qsort.d_perm, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (cell) {
      HIP_TRY(hipMemcpy(c.data(), h->qsort.d_cell_of, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
      for (size_t k = 0; k < n; ++k) cell[k] = p[k] < n ? c[p[k]] : 0u;
    }
    if (perm) memcpy(perm, p.data(), n * sizeof(uint32_t));
    return ICP_OK;
  }
  for (size_t i = 0; i < n; ++i) {
    if (perm) perm[i] = (uint32_t)i;
    if (cell) cell[i] = 0;
  }
  return ICP_OK;
}

// Observability: the certified searches of `h` (nn_grid.hip, k_nn_cert) -- out[0] = searches that checked
// certificates since the handle was created, out[1] = queries whose certificate failed in the last of them (they
// were searched as before).
extern "C" int icp_nn_cert_counters(icp_handle *h, uint64_t out[2]) {
  if (!h || !out) return ICP_BAD_ARGUMENT;
  out[0] = h->qsort.cert_searches;
  out[1] = 0;
  if (!h->qsort.last_cert_ctr) return ICP_OK;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  unsigned ctr[16 * 32];
  HIP_TRY(hipMemcpy(ctr, h->qsort.last_cert_ctr, sizeof(ctr), hipMemcpyDeviceToHost));
  for (int l = 0; l < 16; ++l) out[1] += ctr[l * 32];
  return ICP_OK;
}

extern "C" int icp_estimate(icp_handle *h, const double *src, size_t n, const icp_pose *init, size_t max_iter,
                            icp_pose *out, uint32_t *last_idx, uint32_t *inner_iters) {
  if (!h || !init || !out || (n > 0 && !src) || n >= 0xffffffffull) return ICP_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(ensure_workspace(h, n, true));
  if (n > 0)
    HIP_TRY(hipMemcpyAsync(h->ws.d_src, src, n * h->dim * sizeof(double), hipMemcpyHostToDevice, h->stream));
  const int rc = icp_estimate_device(h, h->ws.d_src, n, init, max_iter, out,
                                     last_idx ? h->ws.d_idx : nullptr, inner_iters);
  if (rc != ICP_OK) return rc;
  if (last_idx && n > 0 && max_iter > 0) {
    HIP_TRY(hipMemcpyAsync(last_idx, h->ws.d_idx, n * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  return ICP_OK;
}

// ------------------------------------------- free functions on host buffers ------
namespace {

std::mutex g_scratch_mu;
icp_handle *g_scratch = nullptr;  // dim 2, no targets; lives for the process

int scratch_handle(icp_handle **out) {
  if (!g_scratch) {
    const int rc = icp_create(&g_scratch, 2, nullptr, 0, -1);
    if (rc != ICP_OK) return rc;
  }
  *out = g_scratch;
  return ICP_OK;
}

// stage host pairs into the scratch handle's a/b buffers
int stage_pairs(icp_handle *h, const double *a, const double *b, size_t n) {
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(ensure_workspace(h, n, false));
  if (n > 0) {
    HIP_TRY(hipMemcpyAsync(h->ws.d_a, a, n * 2 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->ws.d_b, b, n * 2 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  }
  return ICP_OK;
}

}  // namespace

extern "C" int icp_gn_path_counters(icp_handle *h, uint64_t out[6]) {
  if (!out) return ICP_BAD_ARGUMENT;
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!h) {
    const int rc = scratch_handle(&h);
    if (rc != ICP_OK) return rc;
  }
  out[0] = h->ws.win_tried;
  out[1] = h->ws.win_missed;
  out[2] = h->ws.short_evals;
  out[3] = h->ws.radix_evals;
  out[4] = h->ws.spec_hits;
  out[5] = h->ws.spec_misses;
  return ICP_OK;
}

extern "C" int icp_run_ahead_counters(icp_handle *h, uint64_t out[2]) {
  if (!h || !out) return ICP_BAD_ARGUMENT;
  out[0] = h->ws.ahead_hits;
  out[1] = h->ws.ahead_misses;
  return ICP_OK;
}

// launches of the one-launch inner loop (single handle or sharded) that were not resident and gave up their bounded wait
extern "C" int icp_gn_loop_timeouts(icp_handle *h, uint64_t *out) {
  if (!h || !out) return ICP_BAD_ARGUMENT;
  *out = h->ws.loop_timeouts;
  return ICP_OK;
}

// outer iterations icp_estimate[_device] did not run because the pose had stopped moving (a fixed point repeats)
extern "C" int icp_fixed_point_skips(icp_handle *h, uint64_t *out) {
  if (!h || !out) return ICP_BAD_ARGUMENT;
  *out = h->ws.fixed_point_skips;
  return ICP_OK;
}

extern "C" int icp_gn_loop_counters(icp_handle *h, uint64_t out[3]) {
  if (!out) return ICP_BAD_ARGUMENT;
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!h) {
    const int rc = scratch_handle(&h);
    if (rc != ICP_OK) return rc;
  }
  out[0] = h->ws.loop_launches;
  out[1] = h->ws.loop_evals;
  out[2] = h->ws.loop_handbacks;
  return ICP_OK;
}

extern "C" int icp_estimate_transform(const double *a, const double *b, size_t n, icp_pose *out,
                                      uint32_t *inner_iters) {
  if (!out || (n > 0 && (!a || !b))) return ICP_BAD_ARGUMENT;
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  icp_handle *h;
  int rc = scratch_handle(&h);
  if (rc != ICP_OK) return rc;
  if ((rc = stage_pairs(h, a, b, n)) != ICP_OK) return rc;
  return icp_estimate_transform_device(h, h->ws.d_a, h->ws.d_b, n, out, inner_iters);
}

extern "C" int icp_weighted_gauss_newton_update(const icp_pose *T, const double *a, const double *b, size_t n,
                                                double delta[3]) {
  if (!T || !delta || (n > 0 && (!a || !b))) return ICP_BAD_ARGUMENT;
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  icp_handle *h;
  int rc = scratch_handle(&h);  // a missing device is reported before any result
  if (rc != ICP_OK) return rc;
  if (!input_size_ok(n)) return ICP_NONE;  // src/lib.rs:225-228
  if ((rc = stage_pairs(h, a, b, n)) != ICP_OK) return rc;
  return wgn_step(h, h->ws.d_a, h->ws.d_b, n, *T, delta, nullptr);
}

static int plain_pass(const icp_pose *T, const double *a, const double *b, size_t n, GnResult *res) {
  icp_handle *h;
  int rc = scratch_handle(&h);
  if (rc != ICP_OK) return rc;
  if ((rc = stage_pairs(h, a, b, n)) != ICP_OK) return rc;
  HIP_TRY(launch_plain_gn(h, h->ws.d_a, h->ws.d_b, n, *T));
  HIP_TRY(hipStreamSynchronize(h->stream));
  *res = *h->ws.h_res;
  return ICP_OK;
}

extern "C" int icp_gauss_newton_update(const icp_pose *T, const double *a, const double *b, size_t n,
                                       double delta[3]) {
  if (!T || !delta || (n > 0 && (!a || !b))) return ICP_BAD_ARGUMENT;
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  icp_handle *h;
  int rc = scratch_handle(&h);
  if (rc != ICP_OK) return rc;
  if (!input_size_ok(n)) return ICP_NONE;  // src/lib.rs:196-199
  GnResult r;
  if ((rc = plain_pass(T, a, b, n, &r)) != ICP_OK) return rc;
  return solve_update(r.acc, r.acc + 9, delta) ? ICP_OK : ICP_NONE;
}

extern "C" int icp_error(const icp_pose *T, const double *a, const double *b, size_t n, double *out) {
  if (!T || !out || (n > 0 && (!a || !b))) return ICP_BAD_ARGUMENT;
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  GnResult r;
  const int rc = plain_pass(T, a, b, n, &r);
  if (rc != ICP_OK) return rc;
  *out = r.acc[13];
  return ICP_OK;
}

extern "C" int icp_huber_error(const icp_pose *T, const double *a, const double *b, size_t n, double *out) {
  if (!T || !out || (n > 0 && (!a || !b))) return ICP_BAD_ARGUMENT;
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  GnResult r;
  const int rc = plain_pass(T, a, b, n, &r);
  if (rc != ICP_OK) return rc;
  *out = r.acc[12];
  return ICP_OK;
}

extern "C" int icp_residual_stddevs(const icp_pose *T, const double *a, const double *b, size_t n,
                                    double sigma[2]) {
  if (!T || !sigma || (n > 0 && (!a || !b))) return ICP_BAD_ARGUMENT;
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  icp_handle *h;
  int rc = scratch_handle(&h);
  if (rc != ICP_OK) return rc;
  if (n == 0) return ICP_NONE;  // mutable_median of an empty input, src/stats.rs:15-17
  if ((rc = stage_pairs(h, a, b, n)) != ICP_OK) return rc;
  HIP_TRY(launch_sel_init(h, n));
  HIP_TRY(launch_stddevs(h, h->ws.d_a, h->ws.d_b, n, *T));
  h->ws.gn_dirty = true;
  GnScalars s;
  HIP_TRY(hipMemcpyAsync(&s, h->ws.d_scal, sizeof(s), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (s.nan_flag) return ICP_NAN_INPUT;
  sigma[0] = s.sigma[0];
  sigma[1] = s.sigma[1];
  return ICP_OK;
}

