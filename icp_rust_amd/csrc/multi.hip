// icp_create_multi / icp_multi_estimate: ONE host process driving the GPUs of a node (SURVEY.md 8(b)
// sketch `icp_create(..., device_ids, n_devices)`, 8(e) options 2 and 3).
//
// Rank r = a handle on device_ids[r]; the source cloud is sharded by reduction-tree block and every
// evaluation is the sharded evaluation of shard.hip, so the pose equals the one-GPU pose bit for bit.
// The three exchanges per evaluation need no collective library: a rank EXPORTS a stage's bytes
// (histogram / candidates / block sums) into a buffer its peers have mapped (hipDeviceEnablePeerAccess:
// xGMI), bumps a flag behind them, and the consumers -- after a bounded wait on every peer's flag --
// read the peers' exports IN PLACE: the histograms are summed, the candidate lists merged, the block
// sums folded straight out of peer memory (a few KB each).  One host thread enqueues everything; it
// waits once per evaluation, for the folded result, exactly as the one-GPU loop does.
//
// Ranks may share a device ("virtual ranks": device_ids = {0, 0, 0, 0}) -- that is how the N-rank path
// is tested on a one-GPU box.  Ranks on one device share ONE stream and are enqueued stage by stage,
// so every wait is already satisfied when the queue reaches it (two spinning kernels on one hardware
// queue could otherwise wait for each other).  STATUS: the peer-memory path itself (distinct devices)
// has not run on hardware -- no multi-GPU box is available to this build; its flags are bounded spins,
// so a visibility problem would surface as ICP_HIP_ERROR, not as a hang.
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "common.hpp"

using namespace icp;

extern "C" int icp_loop_inbox(icp_handle *h, int fine_grained, void **d_inbox);
extern "C" int icp_shard_loop_connect(icp_handle *h, int rank, int world, void *const *inboxes);
extern "C" int icp_shard_loop_launch_device(icp_handle *h, const double *d_a, const double *d_b, size_t n_total, unsigned launch_no,
                                            unsigned eval_base, int it0, uint32_t applied0, const icp_pose *Ti, double prev_error,
                                            int first_kind, int second_kind);
extern "C" int icp_shard_loop_wait(icp_handle *h, icp_pose *Ti, double *prev_error, uint32_t *applied, int *it, int *finished,
                                   uint32_t *evals);
int icp_p2pl_inner_loop_device(icp_handle *h, const double *d_src, size_t n, const icp_pose *T, const uint32_t *d_idx, icp_pose *dT,
                               uint32_t *applied_out);
int icp_shard_loop_launch_fused(icp_handle *const *hs, int world, const double *const *d_a, const double *const *d_b, size_t n_total,
                                unsigned launch_no, unsigned eval_base, int it0, uint32_t applied0, const icp_pose *Ti,
                                double prev_error, int first_kind, int second_kind);

struct icp_multi {
  int world = 0, dim = 0;
  size_t m = 0;
  bool one_device = true;
  struct Rank {
    icp_handle *h = nullptr;
    int device = 0;
    size_t cap_n = 0;       // local points the buffers below hold
    size_t cap_full = 0;
    double *d_src = nullptr;            // local source cloud
    double *d_a = nullptr, *d_b = nullptr;
    uint32_t *d_idx = nullptr;
    double *d_a_full = nullptr, *d_b_full = nullptr;  // replicated fallback
    // exports (peer-visible) and this rank's flag words {hist, candidates, partials, pairs}
    unsigned char *x_hist = nullptr, *x_exch = nullptr;  // exported: histograms; candidates + block sums
    unsigned *x_flags = nullptr;
    unsigned *d_err = nullptr;
    // EXTENSION, point-to-plane across the ranks: the whole source cloud and every rank's indices (peer-written)
    double *d_p_src = nullptr;
    uint32_t *d_p_idx = nullptr;
    size_t cap_p = 0;
  };
  std::vector<Rank> r;
  // rank 0's device: the whole source cloud, its sorted copy and the permutation (the fold order of the call)
  double *d_sort_in = nullptr, *d_sort_out = nullptr;
  uint32_t *d_sort_perm = nullptr;
  uint32_t *d_idx_full = nullptr, *d_idx_out = nullptr;  // the last search's indices: fold order / the caller's order
  size_t cap_sort = 0;
  unsigned seq = 0;  // generation of the exchanges
  uint64_t sharded = 0, replicated = 0;
  // the inner loop as one launch per rank (gn_loop.hip: k_gn_loop_shard; api.hip: icp_shard_loop_*)
  bool loop_ok = false;
  bool pipe_ok = false;     // the pipelined evaluation across the ranks (pipe.hip) may serve the steady state
  unsigned pipe_skip = 0;
  uint64_t pipe_served = 0;
  uint32_t first_applied = 0xffffffffu;  // updates the first inner loop of the previous call applied
  unsigned loop_launch = 0, loop_evals = 0;
  uint64_t loop_launches = 0, loop_served = 0, loop_handbacks = 0;
};

namespace {

int map_hip(hipError_t e) {
  switch (e) {
    case hipSuccess: return ICP_OK;
    case hipErrorOutOfMemory: return ICP_OUT_OF_MEMORY;
    case hipErrorNoDevice:
    case hipErrorInvalidDevice: return ICP_NO_DEVICE;
    default: return ICP_HIP_ERROR;
  }
}
#define HIP_TRY(expr)                                                                                  \
  do {                                                                                                 \
    hipError_t e__ = (expr);                                                                           \
    if (e__ != hipSuccess) {                                                                           \
      if (getenv("ICP_MULTI_DEBUG")) fprintf(stderr, "[multi] %s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return map_hip(e__);                                                                             \
    }                                                                                                  \
  } while (0)
#define MULTI_FAIL(what)                                                                          \
  do {                                                                                            \
    if (getenv("ICP_MULTI_DEBUG")) fprintf(stderr, "[multi] %s:%d %s\n", __FILE__, __LINE__, what); \
    return ICP_HIP_ERROR;                                                                         \
  } while (0)
#define ICP_TRY(expr)              \
  do {                             \
    const int rc__ = (expr);       \
    if (rc__ != ICP_OK) return rc__; \
  } while (0)

// memory a peer device reads while the owner's kernels are still running must be fine-grained
hipError_t alloc_export(void **p, size_t bytes, bool peers) {
  if (peers) return hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained);
  return hipMalloc(p, bytes);
}

int ensure_rank_buffers(icp_multi *M, icp_multi::Rank &R, size_t n_local, size_t n_total) {
  HIP_TRY(hipSetDevice(R.device));
  if (n_local > R.cap_n) {
    (void)hipFree(R.d_src);
    (void)hipFree(R.d_a);
    (void)hipFree(R.d_b);
    (void)hipFree(R.d_idx);
    R.d_src = R.d_a = R.d_b = nullptr;
    R.d_idx = nullptr;
    R.cap_n = 0;
    const size_t cap = n_local + n_local / 8 + 1;
    // the pairs are read by peers (replicated fallback): exported too
    HIP_TRY(hipMalloc(&R.d_src, cap * 3 * sizeof(double)));
    HIP_TRY(alloc_export((void **)&R.d_a, cap * 2 * sizeof(double), !M->one_device));
    HIP_TRY(alloc_export((void **)&R.d_b, cap * 2 * sizeof(double), !M->one_device));
    HIP_TRY(hipMalloc(&R.d_idx, cap * sizeof(uint32_t)));
    R.cap_n = cap;
  }
  if (n_total > R.cap_full) {
    (void)hipFree(R.d_a_full);
    (void)hipFree(R.d_b_full);
    R.d_a_full = R.d_b_full = nullptr;
    R.cap_full = 0;
    HIP_TRY(hipMalloc(&R.d_a_full, (n_total + 1) * 2 * sizeof(double)));
    HIP_TRY(hipMalloc(&R.d_b_full, (n_total + 1) * 2 * sizeof(double)));
    R.cap_full = n_total + 1;
  }
  return ICP_OK;
}

enum { kFlagHist = 0, kFlagCand = 1, kFlagPairs = 3 };

// every rank bumps flag `which` to `value` behind what it has enqueued so far
int signal_all(icp_multi *M, int which, unsigned value) {
  for (auto &R : M->r) {
    HIP_TRY(hipSetDevice(R.device));
    HIP_TRY(multi_signal(R.h->stream, R.x_flags + 32 * which, value));
  }
  return ICP_OK;
}
int wait_all(icp_multi *M, icp_multi::Rank &R, int which, unsigned value) {
  const unsigned *flags[kShardMaxWorld];
  for (int q = 0; q < M->world; ++q) flags[q] = M->r[q].x_flags + 32 * which;
  HIP_TRY(multi_wait(R.h->stream, flags, M->world, value, R.d_err));
  return ICP_OK;
}

// weighted_gauss_newton_update at inner pose T on every rank; all ranks return the same status / delta
int evaluate(icp_multi *M, size_t n_total, const Pose &T, int kind, double delta[3], double *err, int refined = 0) {
  const int W = M->world;
  uint32_t *hist[kShardMaxWorld];
  int rc0 = ICP_OK;
  for (int q = 0; q < W; ++q) {
    auto &R = M->r[q];
    const int rc = icp_shard_eval_hist_device(R.h, R.d_a, R.d_b, n_total, q, W, &T, kind, refined, &hist[q]);
    static const bool debug = getenv("ICP_MULTI_DEBUG") != nullptr;  // (read once: this runs per rank and evaluation)
    if (debug)
      fprintf(stderr, "[multi] hist rank %d kind %d refined %d -> rc %d (win_valid %d, kinds %d %d %d %d)\n", q, kind, refined, rc,
              (int)R.h->ws.win_valid, (int)R.h->ws.win_kind[0].valid, (int)R.h->ws.win_kind[1].valid,
              (int)R.h->ws.win_kind[3].valid, (int)R.h->ws.win_kind[4].valid);
    if (q == 0) rc0 = rc;
    else if (rc != rc0) MULTI_FAIL("the ranks' prediction state diverged");  // cannot happen
  }
  if (rc0 == ICP_OK) {
    const unsigned gen = ++M->seq;
    // 1. histograms: export, signal, (wait) sum over peers in place
    for (int q = 0; q < W; ++q) {
      auto &R = M->r[q];
      HIP_TRY(hipSetDevice(R.device));
      HIP_TRY(hipMemcpyAsync(R.x_hist, hist[q], icp_shard_histogram_words() * 4, hipMemcpyDeviceToDevice, R.h->stream));
    }
    ICP_TRY(signal_all(M, kFlagHist, gen));
    const void *ptrs[kShardMaxWorld];
    for (int q = 0; q < W; ++q) {
      auto &R = M->r[q];
      HIP_TRY(hipSetDevice(R.device));
      ICP_TRY(wait_all(M, R, kFlagHist, gen));
      for (int p = 0; p < W; ++p) ptrs[p] = M->r[p].x_hist;
      HIP_TRY(multi_sum_hist(R.h->stream, ptrs, W, hist[q]));
      // 2. candidates + block sums
      ICP_TRY(icp_shard_eval_compact_device(R.h, R.x_exch));
    }
    ICP_TRY(signal_all(M, kFlagCand, gen));
    for (int q = 0; q < W; ++q) {
      auto &R = M->r[q];
      HIP_TRY(hipSetDevice(R.device));
      ICP_TRY(wait_all(M, R, kFlagCand, gen));
    }
    int rcf = ICP_OK;
    for (int q = 0; q < W; ++q) {
      auto &R = M->r[q];
      double dq[3], eq = 0.;
      // (finish = selection + fold from the peers' exports + wait + bookkeeping; the contiguous-buffer entry
      // point of the ABI is not used here: the peers' blocks are read in place)
      for (int p = 0; p < W; ++p) ptrs[p] = M->r[p].x_exch;
      const int rc = icp_shard_eval_finish_ptrs(R.h, ptrs, dq, &eq);
      if (q == 0) {
        rcf = rc;
        for (int k = 0; k < 3; ++k) delta[k] = dq[k];
        if (err) *err = eq;
      } else if (rc != rcf) {
        MULTI_FAIL("the ranks finished an evaluation differently");
      }
    }
    for (auto &R : M->r)  // a peer that never arrived?
      if (__atomic_load_n(R.d_err, __ATOMIC_ACQUIRE)) MULTI_FAIL("a wait on a peer's flag gave up");
    if (rcf == ICP_RETRY_SHARDED) return evaluate(M, n_total, T, kind, delta, err, 1);  // refined window
    if (rcf != ICP_RETRY_REPLICATED) {
      ++M->sharded;
      return rcf;
    }
  } else if (rc0 != ICP_RETRY_REPLICATED) {
    return rc0;
  }
  // replicated: every rank assembles the pairs of all ranks in global order and evaluates them
  ++M->replicated;
  const unsigned gen = ++M->seq;
  ICP_TRY(signal_all(M, kFlagPairs, gen));
  int rcr = ICP_OK;
  for (int q = 0; q < W; ++q) {
    auto &R = M->r[q];
    HIP_TRY(hipSetDevice(R.device));
    ICP_TRY(wait_all(M, R, kFlagPairs, gen));
    for (int p = 0; p < W; ++p)
      HIP_TRY(multi_put_pairs(R.h->stream, M->r[p].d_a, M->r[p].d_b, n_total, p, W, R.d_a_full, R.d_b_full));
  }
  for (int q = 0; q < W; ++q) {
    auto &R = M->r[q];
    double dq[3], eq = 0.;
    const int rc = icp_weighted_gn_step_device(R.h, R.d_a_full, R.d_b_full, n_total, &T, kind, dq, &eq);
    if (q == 0) {
      rcr = rc;
      for (int k = 0; k < 3; ++k) delta[k] = dq[k];
      if (err) *err = eq;
    } else if (rc != rcr) {
      MULTI_FAIL("the ranks finished a replicated evaluation differently");
    }
  }
  return rcr;
}

// The inner loop from evaluation *k on as ONE launch per rank: two enqueues per rank and OUTER iteration (search, loop),
// no host wait inside the loop; the ranks exchange histograms, candidates and block sums through their inboxes from
// inside the launches.  *served = false: nothing was launched (every rank said so: no window prediction yet, or the
// pair set does not fit) -- evaluate() steps evaluation *k.
int multi_loop(icp_multi *M, size_t n_total, Pose *Ti, double *prev_error, uint32_t *applied, int *k, bool *finished, bool *served,
               int first_kind, int second_kind) {
  const int W = M->world;
  *served = *finished = false;
  const unsigned launch_no = M->loop_launch + 1;
  int launched = 0;
  if (M->one_device) {
    // ranks that share a device: ALL of them in one launch on their shared stream, behind the searches (the launches of
    // ranks wait for each other: as separate launches they would need a hardware queue each)
    icp_handle *hs[kShardMaxWorld];
    const double *as[kShardMaxWorld], *bs[kShardMaxWorld];
    for (int q = 0; q < W; ++q) {
      hs[q] = M->r[q].h;
      as[q] = M->r[q].d_a;
      bs[q] = M->r[q].d_b;
    }
    const int rc = icp_shard_loop_launch_fused(hs, W, as, bs, n_total, launch_no, M->loop_evals, *k, *applied, Ti, *prev_error, first_kind,
                                               second_kind);
    if (rc == ICP_OK) launched = W;
    else if (rc != ICP_RETRY_SHARDED) return rc;
  } else {
    for (int q = 0; q < W; ++q) {
      auto &R = M->r[q];
      const int rc = icp_shard_loop_launch_device(R.h, R.d_a, R.d_b, n_total, launch_no, M->loop_evals, *k, *applied, Ti, *prev_error,
                                                  first_kind, second_kind);
      if (rc == ICP_OK) ++launched;
      else if (rc != ICP_RETRY_SHARDED) return rc;
    }
  }
  if (launched == 0) return ICP_OK;
  if (launched != W) {  // cannot happen: the answer depends on replicated state only
    M->loop_ok = false;
    MULTI_FAIL("the ranks disagreed about launching the inner loop");
  }
  M->loop_launch = launch_no;
  Pose T0 = *Ti;
  double pe0 = *prev_error;
  uint32_t ap0 = *applied, ev0 = 0;
  int k0 = *k, fin0 = 0, rc0 = ICP_OK;
  bool gave_up = false, differ = false, have_ref = false;
  for (int q = 0; q < W; ++q) {  // (every rank's result is awaited whatever the others say: nothing may still be running)
    auto &R = M->r[q];
    Pose Tq = *Ti;
    double pe = *prev_error;
    uint32_t ap = *applied, ev = 0;
    int kq = *k, fin = 0;
    const int rc = icp_shard_loop_wait(R.h, &Tq, &pe, &ap, &kq, &fin, &ev);
    if (rc == ICP_HIP_ERROR) {
      gave_up = true;
      continue;
    }
    if (!have_ref) {
      have_ref = true;
      T0 = Tq;
      pe0 = pe;
      ap0 = ap;
      ev0 = ev;
      k0 = kq;
      fin0 = fin;
      rc0 = rc;
    } else if (rc != rc0 || memcmp(&Tq, &T0, sizeof(Pose)) != 0 || memcmp(&pe, &pe0, sizeof(double)) != 0 || ap != ap0 || ev != ev0 ||
               kq != k0 || fin != fin0) {
      differ = true;
    }
  }
  if (gave_up) {
    // A launch gave up waiting (its workgroups, or a peer's, were not all running at once: a device that is shared, fewer
    // CUs than the launches of the ranks on it need).  Nothing of the launch is used: the loop's state is the one it
    // was started with, the ranks' prediction histories may have diverged and are dropped, and the stage calls serve
    // from here on (bit-identical either way).
    M->loop_ok = false;
    for (auto &R : M->r) {
      R.h->ws.win_valid = false;
      for (auto &wk : R.h->ws.win_kind) wk = Workspace::WinPred();
    }
    if (getenv("ICP_MULTI_DEBUG")) fprintf(stderr, "[multi] an inner-loop launch gave up waiting: stage calls from now on\n");
    return ICP_OK;
  }
  if (differ) {
    M->loop_ok = false;
    MULTI_FAIL("the ranks finished an inner-loop launch differently");
  }
  ++M->loop_launches;
  *served = true;
  M->loop_evals += ev0;
  M->loop_served += ev0;
  if (rc0 != ICP_OK) return rc0;
  *Ti = T0;
  *prev_error = pe0;
  *applied = ap0;
  *k = k0;
  *finished = fin0 != 0;
  if (!*finished) ++M->loop_handbacks;
  return ICP_OK;
}

}  // namespace

extern "C" void icp_destroy_multi(icp_multi *M) {
  if (!M) return;
  // ranks that share a device run on rank 0's stream: quiesce everything and hand every handle its own stream
  // back BEFORE any handle (and with it, possibly, that stream) is released
  for (auto &R : M->r)
    if (R.h) {
      (void)hipSetDevice(R.device);
      (void)hipStreamSynchronize(R.h->stream);
    }
  for (auto &R : M->r)
    if (R.h) (void)icp_use_own_stream(R.h);
  for (auto &R : M->r) {
    (void)hipSetDevice(R.device);
    if (R.h) icp_destroy(R.h);
    (void)hipFree(R.d_src);
    (void)hipFree(R.d_a);
    (void)hipFree(R.d_b);
    (void)hipFree(R.d_idx);
    (void)hipFree(R.d_a_full);
    (void)hipFree(R.d_b_full);
    (void)hipFree(R.x_hist);
    (void)hipFree(R.x_exch);
    (void)hipFree(R.x_flags);
    if (R.d_err) (void)hipHostFree(R.d_err);
    (void)hipFree(R.d_p_src);
    (void)hipFree(R.d_p_idx);
  }
  if (!M->r.empty()) (void)hipSetDevice(M->r[0].device);
  (void)hipFree(M->d_sort_in);
  (void)hipFree(M->d_sort_out);
  (void)hipFree(M->d_sort_perm);
  (void)hipFree(M->d_idx_full);
  (void)hipFree(M->d_idx_out);
  delete M;
}

extern "C" int icp_create_multi(icp_multi **out, int dim, const double *dst, size_t m, const int *device_ids,
                                int n_devices) {
  if (!out || (dim != 2 && dim != 3) || (m > 0 && !dst) || !device_ids || n_devices < 1 || n_devices > kShardMaxWorld)
    return ICP_BAD_ARGUMENT;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return ICP_NO_DEVICE;
  for (int q = 0; q < n_devices; ++q)
    if (device_ids[q] < 0 || device_ids[q] >= count) return ICP_BAD_ARGUMENT;
  icp_multi *M = new (std::nothrow) icp_multi();
  if (!M) return ICP_OUT_OF_MEMORY;
  M->world = n_devices;
  M->dim = dim;
  M->m = m;
  M->r.resize(n_devices);
  M->one_device = true;
  for (int q = 1; q < n_devices; ++q) M->one_device = M->one_device && device_ids[q] == device_ids[0];
  if (!M->one_device)  // all ranks on one device, or every rank on a device of its own (nothing in between)
    for (int q = 0; q < n_devices; ++q)
      for (int p = 0; p < q; ++p)
        if (device_ids[p] == device_ids[q]) {
          delete M;
          return ICP_BAD_ARGUMENT;
        }
  int rc = ICP_OK;
  for (int q = 0; q < n_devices && rc == ICP_OK; ++q) {
    auto &R = M->r[q];
    R.device = device_ids[q];
    hipError_t e = hipSetDevice(R.device);
    if (e != hipSuccess) { rc = map_hip(e); break; }
    if (!M->one_device)
      for (int p = 0; p < n_devices; ++p)
        if (device_ids[p] != R.device) {
          e = hipDeviceEnablePeerAccess(device_ids[p], 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { rc = map_hip(e); break; }
          (void)hipGetLastError();
        }
    if (rc != ICP_OK) break;
    rc = icp_create(&R.h, dim, dst, m, R.device);  // the target cloud is replicated
    if (rc != ICP_OK) break;
    const bool peers = !M->one_device;
    if ((e = alloc_export((void **)&R.x_hist, icp_shard_histogram_words() * 4, peers)) != hipSuccess ||
        (e = alloc_export((void **)&R.x_exch, icp_shard_exchange_bytes(n_devices), peers)) != hipSuccess ||
        (e = alloc_export((void **)&R.x_flags, 4 * 32 * sizeof(unsigned), peers)) != hipSuccess ||
        (e = hipHostMalloc((void **)&R.d_err, sizeof(unsigned), hipHostMallocCoherent)) != hipSuccess ||
        (e = hipMemset(R.x_flags, 0, 4 * 32 * sizeof(unsigned))) != hipSuccess) {
      rc = map_hip(e);
      break;
    }
    *R.d_err = 0;  // (pinned, device-visible: a wait kernel that gives up raises it, the host reads it directly)
  }
  if (rc == ICP_OK && M->one_device)  // ranks on one device share one stream: lockstep order, no waiting
    for (int q = 1; q < n_devices; ++q) rc = rc == ICP_OK ? icp_set_stream(M->r[q].h, M->r[0].h->stream) : rc;
  // the one-launch inner loop: every rank's inbox mapped on every rank (plain pointers: one process)
  if (rc == ICP_OK && getenv("ICP_NO_GN_LOOP") == nullptr) {
    void *boxes[kShardMaxWorld] = {};
    for (int q = 0; q < n_devices && rc == ICP_OK; ++q) rc = icp_loop_inbox(M->r[q].h, M->one_device ? 0 : 1, &boxes[q]);
    for (int q = 0; q < n_devices && rc == ICP_OK; ++q) rc = icp_shard_loop_connect(M->r[q].h, q, n_devices, boxes);
    M->loop_ok = rc == ICP_OK;
    // (ranks that share a device ride in one finishing launch: its arguments hold eight of them)
    M->pipe_ok = M->loop_ok && getenv("ICP_NO_SPECULATION") == nullptr && (!M->one_device || n_devices <= 8);
  }
  if (rc != ICP_OK) {
    icp_destroy_multi(M);
    return rc;
  }
  *out = M;
  return ICP_OK;
}

// EXTENSION (BASELINE configs[4], the growing target cloud of include/icp_mi355x.h section 6, across the ranks): the
// target cloud is replicated, so every rank appends the same k points (moved by T) and rebuilds its search grid;
// afterwards the object is, bit for bit, a fresh icp_create_multi on the concatenated cloud.
extern "C" int icp_multi_append_targets(icp_multi *M, const double *pts, size_t k, const icp_pose *T) {
  if (!M || (k > 0 && !pts)) return ICP_BAD_ARGUMENT;
  for (auto &R : M->r) {
    const int rc = icp_append_targets(R.h, pts, k, T);
    if (rc != ICP_OK) return rc;  // (a rank that failed leaves the replicas different: the object is then unusable)
  }
  M->m += k;
  return ICP_OK;
}
extern "C" size_t icp_multi_target_count(const icp_multi *M) { return M ? M->m : 0; }

extern "C" int icp_multi_counters(const icp_multi *M, uint64_t out[2]) {
  if (!M || !out) return ICP_BAD_ARGUMENT;
  // evaluations that ran sharded: stepped from the host, inside a loop launch, or pipelined (two per outer iteration)
  out[0] = M->sharded + M->loop_served + 2 * M->pipe_served;
  out[1] = M->replicated;
  return ICP_OK;
}
extern "C" int icp_multi_loop_counters(const icp_multi *M, uint64_t out[3]) {
  if (!M || !out) return ICP_BAD_ARGUMENT;
  out[0] = M->loop_launches;
  out[1] = M->loop_served;
  out[2] = M->loop_handbacks;
  return ICP_OK;
}

// outer iterations the pipelined evaluation (pipe.hip) served over the life of `M`
extern "C" int icp_multi_pipe_iterations(const icp_multi *M, uint64_t *out) {
  if (!M || !out) return ICP_BAD_ARGUMENT;
  *out = M->pipe_served;
  return ICP_OK;
}

// Icp{2,3}d::estimate (src/lib.rs:105-130, 148-173) over all the devices of `M`; host buffers
extern "C" int icp_multi_estimate(icp_multi *M, const double *src, size_t n, const icp_pose *init, size_t max_iter,
                                  icp_pose *out, uint32_t *last_idx, uint32_t *inner_iters) {
  if (!M || !init || !out || (n > 0 && !src) || n >= 0xffffffffull) return ICP_BAD_ARGUMENT;
  const int W = M->world, dim = M->dim;
  Pose T = *init;
  if (n > 0 && max_iter > 0 && M->m == 0) return ICP_EMPTY_DST;  // index.unwrap(), src/lib.rs:122,165
  // whatever way this call ends (an error half-way through an evaluation leaves kernels of the other
  // ranks enqueued, some of them waiting on flags), nothing of it is in flight afterwards and no
  // rank keeps a cell-sorted snapshot of a source buffer that the next call overwrites
  struct Quiesce {
    icp_multi *M;
    ~Quiesce() {
      for (auto &R : M->r)
        if (R.h) {
          (void)hipSetDevice(R.device);
          (void)hipStreamSynchronize(R.h->stream);
          R.h->qsort.valid = false;
          R.h->qsort.have_prev = false;
        }
    }
  } quiesce_on_exit{M};
  for (auto &R : M->r)  // (a wait that gave up in an earlier call must not fail this one)
    if (R.d_err) __atomic_store_n(R.d_err, 0u, __ATOMIC_RELEASE);
  // One handle folds its sums over the source cloud in FOLD ORDER (icp_last_fold_order: the cell-sorted
  // snapshot of the call); the ranks shard THAT order, so that N ranks return the bits of one.  The whole
  // cloud goes to rank 0's device once, is sorted there (icp_sort_source_device: the same sort, the identity
  // where one handle would keep the caller's order), and every rank compacts its blocks' points out of the
  // sorted copy on its own device -- a peer read over xGMI between devices; nothing returns to the host
  // (a first version dealt the shards on the host: two more 24 MB pageable transfers per call).
  bool sorted = false;
  std::vector<size_t> n_local(W);
  const double *d_full = nullptr;
  if (n > 0) {
    auto &R0 = M->r[0];
    HIP_TRY(hipSetDevice(R0.device));
    if (n > M->cap_sort) {
      (void)hipFree(M->d_sort_in);
      (void)hipFree(M->d_sort_out);
      (void)hipFree(M->d_sort_perm);
      (void)hipFree(M->d_idx_full);
      (void)hipFree(M->d_idx_out);
      M->d_sort_in = M->d_sort_out = nullptr;
      M->d_sort_perm = M->d_idx_full = M->d_idx_out = nullptr;
      M->cap_sort = 0;
      const size_t cap = n + n / 8 + 1;
      HIP_TRY(hipMalloc(&M->d_sort_in, cap * 3 * sizeof(double)));
      HIP_TRY(hipMalloc(&M->d_sort_out, cap * 3 * sizeof(double)));
      HIP_TRY(hipMalloc(&M->d_sort_perm, cap * sizeof(uint32_t)));
      HIP_TRY(alloc_export((void **)&M->d_idx_full, cap * sizeof(uint32_t), !M->one_device));
      HIP_TRY(hipMalloc(&M->d_idx_out, cap * sizeof(uint32_t)));
      M->cap_sort = cap;
    }
    HIP_TRY(hipMemcpyAsync(M->d_sort_in, src, n * dim * sizeof(double), hipMemcpyHostToDevice, R0.h->stream));
    d_full = M->d_sort_in;
    if (max_iter > 0) {
      ICP_TRY(icp_sort_source_device(R0.h, M->d_sort_in, n, init, M->d_sort_out, M->d_sort_perm));
      d_full = M->d_sort_out;
      sorted = true;
    }
    HIP_TRY(hipStreamSynchronize(R0.h->stream));  // the peers read the sorted cloud
  }
  for (int q = 0; q < W; ++q) {
    int b0, b1, B;
    shard_geometry(n, q, W, &b0, &b1, &B, &n_local[q]);
    auto &R = M->r[q];
    ICP_TRY(ensure_rank_buffers(M, R, n_local[q], n));
    if (n_local[q] == 0) continue;
    HIP_TRY(hipSetDevice(R.device));
    HIP_TRY(launch_shard_copy(R.h, d_full, R.d_src, n, q, W, (unsigned)(dim * 2), true));
    // (the rank's points are runs of the sorted cloud: its search snapshot needs no sort of its own)
    R.h->qsort.presorted = sorted;
    if (max_iter > 0) ICP_TRY(icp_prepare_source_device(R.h, R.d_src, n_local[q], init));
    R.h->qsort.presorted = false;
  }
  // (the bet needs "the inner loop applied exactly one update last time": a call's first iteration goes by what the previous
  // call's first iteration did -- the next frame, or the same cloud again, usually starts like the last one; its two
  // evaluations are predicted from the previous call's first two, kinds 3 and 4 of common.hpp: Workspace::win_kind)
  uint32_t prev_applied = M->first_applied;
  for (size_t it = 0; it < max_iter; ++it) {
    // Round 6: once an inner loop has applied exactly one update (a settled registration; the benchmark pair from its
    // second iteration on), the ranks run the one-GPU pipeline -- search -> paired first launches -> finishing workgroups
    // that meet across the ranks, the next search already behind them (pipe.hip) -- until something else happens, and
    // this loop takes over again at the start of that iteration.
    if (M->pipe_ok && M->loop_ok && prev_applied == 1u && M->pipe_skip == 0) {
      PipeRank pr[kShardMaxWorld];
      bool all = true;
      for (int q = 0; q < W; ++q) {
        int b0, b1, B;
        size_t nl;
        shard_geometry(n, q, W, &b0, &b1, &B, &nl);
        pr[q] = PipeRank{M->r[q].h, M->r[q].d_src, n_local[q], q, b0, b1 - b0, M->r[q].d_idx};
        all = all && n_local[q] > 0;
      }
      if (all) {
        size_t it2 = it;
        int why = 1;
        ICP_TRY(pipe_run(pr, W, W, n, &T, &it2, max_iter, inner_iters, &why));
        M->pipe_served += it2 - it;
        if (it == 0 && it2 > 0) M->first_applied = 1u;
        if (why == 5) {  // a wait for a peer ran out: the inboxes carry a raised abort word, the stage calls serve from here
          M->pipe_ok = M->loop_ok = false;
          if (getenv("ICP_MULTI_DEBUG")) fprintf(stderr, "[multi] the pipelined evaluation gave up waiting: stage calls from now on\n");
        }
        if (it2 == it) M->pipe_skip = 2;  // (handed back at once: this loop serves a few iterations before the next try)
        it = it2;
        if (it >= max_iter) break;
      }
    } else if (M->pipe_skip > 0) {
      --M->pipe_skip;
    }
    for (int q = 0; q < W; ++q) {
      auto &R = M->r[q];
      if (n_local[q]) ICP_TRY(icp_correspond_device(R.h, R.d_src, n_local[q], &T, R.d_a, R.d_b, R.d_idx));
    }
    // estimate_transform, src/lib.rs:59-84
    Pose Ti = transform_identity();
    uint32_t applied = 0;
    if (n >= 2) {
      double prev_error = DBL_MAX;
      for (int k = 0; k < ICP_INNER_MAX_ITER; ++k) {
        const int kind_a = it == 0 ? 3 : 0, kind_b = it == 0 ? 4 : 1;  // (the first / second evaluation of this inner loop)
        if (M->loop_ok) {
          bool finished = false, served = false;
          ICP_TRY(multi_loop(M, n, &Ti, &prev_error, &applied, &k, &finished, &served, kind_a, kind_b));
          if (finished || k >= ICP_INNER_MAX_ITER) break;
        }
        double delta[3], err = 0.;
        const int rc = evaluate(M, n, Ti, k == 0 ? kind_a : (k == 1 ? kind_b : 2), delta, &err);
        if (rc == ICP_NONE) break;
        if (rc != ICP_OK) return rc;
        if ((delta[0] * delta[0] + delta[1] * delta[1]) + delta[2] * delta[2] < ICP_DELTA_NORM_THRESHOLD) break;
        if (err > prev_error) break;
        prev_error = err;
        Ti = transform_mul(transform_new(delta), Ti);
        ++applied;
      }
    }
    if (inner_iters) inner_iters[it] = applied;
    prev_applied = applied;
    if (it == 0) M->first_applied = applied;
    const Pose T_next = transform_mul(Ti, T);
    // (a fixed point of the loop, as in icp_estimate_device: the iterations after it repeat it -- and here every search
    // leaves its indices, so not even the last one has to run)
    if (applied == 0 && memcmp(&T_next, &T, sizeof(Pose)) == 0) {
      if (inner_iters)
        for (size_t k = it + 1; k < max_iter; ++k) inner_iters[k] = 0;
      break;
    }
    T = T_next;
  }
  // the last search's indices back to the caller's point order, on the devices: every rank puts its slice into rank 0's
  // fold-order array (a peer write between devices), rank 0 un-permutes, one transfer to the host
  const bool want_idx = last_idx && max_iter > 0 && n > 0;
  for (int q = 0; q < W; ++q) {
    auto &R = M->r[q];
    HIP_TRY(hipSetDevice(R.device));
    if (want_idx && n_local[q]) HIP_TRY(launch_shard_copy(R.h, R.d_idx, M->d_idx_full, n, q, W, 1u, false));
    HIP_TRY(hipStreamSynchronize(R.h->stream));
    R.h->qsort.valid = false;
    R.h->qsort.have_prev = false;
  }
  if (want_idx) {
    auto &R0 = M->r[0];
    HIP_TRY(hipSetDevice(R0.device));
    const uint32_t *d_res = M->d_idx_full;
    if (sorted) {
      HIP_TRY(multi_unpermute(R0.h->stream, M->d_idx_full, M->d_sort_perm, n, M->d_idx_out));
      d_res = M->d_idx_out;
    }
    HIP_TRY(hipMemcpyAsync(last_idx, d_res, n * sizeof(uint32_t), hipMemcpyDeviceToHost, R0.h->stream));
    HIP_TRY(hipStreamSynchronize(R0.h->stream));
  }
  *out = T;
  return ICP_OK;
}

// ---- EXTENSION (include/icp_mi355x.h section 7 across the ranks; BASELINE configs[4]: scan-to-map, point-to-plane) ------
// The normals belong to the TARGET cloud, which is replicated: every rank computes (or updates) the same normals from
// the same cloud.  A registration shards the SEARCH -- contiguous slices of the source cloud, the part that grows with
// the map -- then every rank receives every slice's indices (4 bytes per point, written into the peers' arrays) and runs
// the same inner loop on the whole cloud: SURVEY 8(e) option 1, gather-then-replicate.  The result is, bit for bit, one
// handle's icp_estimate_point_to_plane (same indices, same pairs in the caller's order, same kernels).
extern "C" int icp_multi_compute_target_normals(icp_multi *M, int k) {
  if (!M) return ICP_BAD_ARGUMENT;
  for (auto &R : M->r) ICP_TRY(icp_compute_target_normals(R.h, k));
  return ICP_OK;
}
extern "C" int icp_multi_update_target_normals(icp_multi *M, int k) {
  if (!M) return ICP_BAD_ARGUMENT;
  for (auto &R : M->r) ICP_TRY(icp_update_target_normals(R.h, k));
  return ICP_OK;
}

extern "C" int icp_multi_estimate_point_to_plane(icp_multi *M, const double *src, size_t n, const icp_pose *init, size_t max_iter,
                                                 icp_pose *out, uint32_t *last_idx, uint32_t *inner_iters) {
  if (!M || M->dim != 3 || !init || !out || (n > 0 && !src) || n >= 0xffffffffull) return ICP_BAD_ARGUMENT;
  const int W = M->world;
  if (M->m == 0) {
    if (n > 0 && max_iter > 0) return ICP_EMPTY_DST;
    *out = *init;
    return ICP_OK;
  }
  struct Quiesce {
    icp_multi *M;
    ~Quiesce() {
      for (auto &R : M->r)
        if (R.h) {
          (void)hipSetDevice(R.device);
          (void)hipStreamSynchronize(R.h->stream);
          R.h->qsort.valid = false;
          R.h->qsort.have_prev = false;
        }
    }
  } quiesce_on_exit{M};
  Pose T = *init;
  std::vector<size_t> lo(W + 1);
  for (int q = 0; q <= W; ++q) lo[q] = n * (size_t)q / (size_t)W;
  for (int q = 0; q < W; ++q) {
    auto &R = M->r[q];
    HIP_TRY(hipSetDevice(R.device));
    if (n > R.cap_p) {
      (void)hipFree(R.d_p_src);
      (void)hipFree(R.d_p_idx);
      R.d_p_src = nullptr;
      R.d_p_idx = nullptr;
      R.cap_p = 0;
      const size_t cap = n + n / 8 + 1;
      HIP_TRY(hipMalloc(&R.d_p_src, cap * 3 * sizeof(double)));
      HIP_TRY(alloc_export((void **)&R.d_p_idx, cap * sizeof(uint32_t), !M->one_device));
      R.cap_p = cap;
    }
    if (n > 0) HIP_TRY(hipMemcpyAsync(R.d_p_src, src, n * 3 * sizeof(double), hipMemcpyHostToDevice, R.h->stream));
    if (max_iter > 0 && lo[q + 1] > lo[q]) ICP_TRY(icp_prepare_source_device(R.h, R.d_p_src + lo[q] * 3, lo[q + 1] - lo[q], init));
  }
  for (size_t it = 0; it < max_iter; ++it) {
    for (int q = 0; q < W; ++q) {  // every rank searches its slice ...
      auto &R = M->r[q];
      const size_t cnt = lo[q + 1] - lo[q];
      if (cnt == 0) continue;
      ICP_TRY(icp_correspond_device(R.h, R.d_p_src + lo[q] * 3, cnt, &T, nullptr, nullptr, R.d_p_idx + lo[q]));
      HIP_TRY(hipSetDevice(R.device));
      for (int p = 0; p < W; ++p)  // ... and hands its indices to every other rank
        if (p != q)
          HIP_TRY(hipMemcpyAsync(M->r[p].d_p_idx + lo[q], R.d_p_idx + lo[q], cnt * sizeof(uint32_t), hipMemcpyDeviceToDevice,
                                 R.h->stream));
    }
    for (auto &R : M->r) {
      HIP_TRY(hipSetDevice(R.device));
      HIP_TRY(hipStreamSynchronize(R.h->stream));
    }
    Pose Ti0 = transform_identity();
    uint32_t ap0 = 0;
    for (int q = 0; q < W; ++q) {  // the same inner loop on every rank
      auto &R = M->r[q];
      Pose Ti;
      uint32_t ap = 0;
      ICP_TRY(icp_p2pl_inner_loop_device(R.h, R.d_p_src, n, &T, R.d_p_idx, &Ti, &ap));
      if (q == 0) {
        Ti0 = Ti;
        ap0 = ap;
      } else if (memcmp(&Ti, &Ti0, sizeof(Pose)) != 0 || ap != ap0) {
        MULTI_FAIL("the ranks finished a point-to-plane inner loop differently");
      }
    }
    if (inner_iters) inner_iters[it] = ap0;
    T = transform_mul(Ti0, T);
  }
  if (last_idx && max_iter > 0 && n > 0) {
    auto &R0 = M->r[0];
    HIP_TRY(hipSetDevice(R0.device));
    HIP_TRY(hipMemcpy(last_idx, R0.d_p_idx, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  }
  *out = T;
  return ICP_OK;
}
