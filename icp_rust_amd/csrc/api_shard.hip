// C ABI of the SHARDED registration (include/icp_mi355x.h sections 5 and 5b): the stage calls of a block-sharded
// evaluation and the one-launch inner loop over the ranks (inboxes, their transports, launch and wait).  Split out of
// api.hip in round 5.
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

namespace icp {
__global__ void k_iota_u32(uint32_t *p, unsigned n);  // (api.hip)
long grid_coop_max();
}

// ---- ... and over the ranks of a sharded registration (include/icp_mi355x.h section 5b) ---------------------------------
extern "C" size_t icp_loop_inbox_bytes(void) { return sizeof(LoopInbox); }

// (the inbox of a handle, whatever it is made of, released)
void icp::api::free_loop_inbox(Workspace &w) {
  if (!w.d_loop_inbox) return;
  if (w.loop_inbox_kind == ICP_INBOX_HOST) {
    (void)hipHostUnregister(w.loop_inbox_host);
    (void)munmap(w.loop_inbox_host, sizeof(LoopInbox));
    if (w.loop_shm_name[0]) (void)shm_unlink(w.loop_shm_name);
    w.loop_shm_name[0] = 0;
    w.loop_inbox_host = nullptr;
  } else {
    (void)hipFree(w.d_loop_inbox);
  }
  w.d_loop_inbox = nullptr;
}

// kind (include/icp_mi355x.h: ICP_INBOX_*): what the inbox is made of -- ordinary device memory (ranks on ONE device:
// virtual ranks, processes sharing a GPU), fine-grained device memory (peer devices write it while this device's
// kernels poll it), or pinned host memory in a POSIX shared-memory object that every process of the node can map and
// register (coherent by construction; the exchange then crosses the host link instead of xGMI).
extern "C" int icp_loop_inbox(icp_handle *h, int kind, void **d_inbox) {
  if (!h || !d_inbox || kind < ICP_INBOX_DEVICE || kind > ICP_INBOX_HOST) return ICP_BAD_ARGUMENT;
  Workspace &w = h->ws;
  HIP_TRY(hipSetDevice(h->device));
  if (w.d_loop_inbox && w.loop_inbox_kind != kind) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    free_loop_inbox(w);
  }
  if (!w.d_loop_inbox) {
    if (kind == ICP_INBOX_HOST) {
      static std::atomic<unsigned> serial{0};
      snprintf(w.loop_shm_name, sizeof(w.loop_shm_name), "/icp_inbox_%d_%u", (int)getpid(), serial.fetch_add(1u));
      const int fd = shm_open(w.loop_shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0) {
        w.loop_shm_name[0] = 0;
        return ICP_HIP_ERROR;
      }
      void *p = MAP_FAILED;
      if (ftruncate(fd, (off_t)sizeof(LoopInbox)) == 0) p = mmap(nullptr, sizeof(LoopInbox), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      (void)close(fd);
      void *d = nullptr;
      if (p == MAP_FAILED || hipHostRegister(p, sizeof(LoopInbox), hipHostRegisterPortable | hipHostRegisterMapped) != hipSuccess) {
        if (p != MAP_FAILED) (void)munmap(p, sizeof(LoopInbox));
        (void)shm_unlink(w.loop_shm_name);
        w.loop_shm_name[0] = 0;
        (void)hipGetLastError();
        return ICP_HIP_ERROR;
      }
      if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess) {
        (void)hipHostUnregister(p);
        (void)munmap(p, sizeof(LoopInbox));
        (void)shm_unlink(w.loop_shm_name);
        w.loop_shm_name[0] = 0;
        (void)hipGetLastError();
        return ICP_HIP_ERROR;
      }
      memset(p, 0, sizeof(LoopInbox));
      w.loop_inbox_host = p;
      w.d_loop_inbox = d;
    } else {
      if (kind == ICP_INBOX_FINE) HIP_TRY(hipExtMallocWithFlags(&w.d_loop_inbox, sizeof(LoopInbox), hipDeviceMallocFinegrained));
      else HIP_TRY(hipMalloc(&w.d_loop_inbox, sizeof(LoopInbox)));
      HIP_TRY(hipMemset(w.d_loop_inbox, 0, sizeof(LoopInbox)));
    }
    w.loop_inbox_kind = kind;
  }
  *d_inbox = w.d_loop_inbox;
  return ICP_OK;
}

// ICP_INBOX_HOST: the name of the shared-memory object behind this handle's inbox (what a peer process hands to
// icp_loop_shm_open), and its removal from the name space once every peer has opened it (the mappings live on)
extern "C" int icp_loop_inbox_shm_name(icp_handle *h, char out[64]) {
  if (!h || !out || !h->ws.d_loop_inbox || h->ws.loop_inbox_kind != ICP_INBOX_HOST) return ICP_BAD_ARGUMENT;
  memcpy(out, h->ws.loop_shm_name, 64);
  return ICP_OK;
}
extern "C" int icp_loop_inbox_shm_unlink(icp_handle *h) {
  if (!h) return ICP_BAD_ARGUMENT;
  if (h->ws.loop_shm_name[0]) (void)shm_unlink(h->ws.loop_shm_name);
  h->ws.loop_shm_name[0] = 0;
  return ICP_OK;
}
namespace {
std::mutex g_shm_mu;
std::map<void *, void *> g_shm_maps;  // device pointer -> host mapping of a peer's inbox opened here
}  // namespace
extern "C" int icp_loop_shm_open(int device, const char *name, void **d_ptr) {
  if (!name || !d_ptr) return ICP_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(device));
  const int fd = shm_open(name, O_RDWR, 0600);
  if (fd < 0) return ICP_HIP_ERROR;
  void *p = mmap(nullptr, sizeof(LoopInbox), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  (void)close(fd);
  if (p == MAP_FAILED) return ICP_HIP_ERROR;
  void *d = nullptr;
  if (hipHostRegister(p, sizeof(LoopInbox), hipHostRegisterPortable | hipHostRegisterMapped) != hipSuccess ||
      hipHostGetDevicePointer(&d, p, 0) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipHostUnregister(p);
    (void)hipGetLastError();
    (void)munmap(p, sizeof(LoopInbox));
    return ICP_HIP_ERROR;
  }
  std::lock_guard<std::mutex> lk(g_shm_mu);
  g_shm_maps[d] = p;
  *d_ptr = d;
  return ICP_OK;
}
extern "C" int icp_loop_shm_close(void *d_ptr) {
  if (!d_ptr) return ICP_OK;
  void *p = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_shm_mu);
    auto it = g_shm_maps.find(d_ptr);
    if (it == g_shm_maps.end()) return ICP_BAD_ARGUMENT;
    p = it->second;
    g_shm_maps.erase(it);
  }
  (void)hipHostUnregister(p);
  (void)munmap(p, sizeof(LoopInbox));
  return ICP_OK;
}

extern "C" int icp_loop_inbox_ipc_handle(icp_handle *h, unsigned char out[64]) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the ABI hands an IPC handle over as 64 bytes");
  if (!h || !out || !h->ws.d_loop_inbox) return ICP_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(h->device));
  hipIpcMemHandle_t mh;
  HIP_TRY(hipIpcGetMemHandle(&mh, h->ws.d_loop_inbox));
  memcpy(out, &mh, 64);
  return ICP_OK;
}
extern "C" int icp_loop_ipc_open(int device, const unsigned char handle[64], void **d_ptr) {
  if (!handle || !d_ptr) return ICP_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(device));
  hipIpcMemHandle_t mh;
  memcpy(&mh, handle, 64);
  HIP_TRY(hipIpcOpenMemHandle(d_ptr, mh, hipIpcMemLazyEnablePeerAccess));
  return ICP_OK;
}
extern "C" int icp_loop_ipc_close(void *d_ptr) {
  if (!d_ptr) return ICP_OK;
  HIP_TRY(hipIpcCloseMemHandle(d_ptr));
  return ICP_OK;
}

// every rank's inbox as mapped in this process, this rank's own among them; empties the own inbox: the ranks must meet
// (a barrier of the driver) between their connects and the first launch
extern "C" int icp_shard_loop_connect(icp_handle *h, int rank, int world, void *const *inboxes) {
  if (!h || !inboxes || world < 1 || world > kShardMaxWorld || rank < 0 || rank >= world) return ICP_BAD_ARGUMENT;
  Workspace &w = h->ws;
  if (!w.d_loop_inbox || inboxes[rank] != w.d_loop_inbox) return ICP_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemset(w.d_loop_inbox, 0, sizeof(LoopInbox)));
  for (int q = 0; q < world; ++q) {
    if (!inboxes[q]) return ICP_BAD_ARGUMENT;
    w.loop_peers[q] = inboxes[q];
  }
  w.loop_rank = rank;
  w.loop_world = world;
  HIP_TRY(ensure_loop(h));  // (the pinned result block)
  w.loop_seq = 0;           // launch numbers restart with the connection
  w.loop_probe_gen = 0;     // ... the tokens of the transport probe (every rank connects, then probes: the same tokens)
  w.pipe_gen = 0;           // ... and the generations of the pipelined evaluation's exchanges
  w.pipe_off = 0;
  memset(w.h_loop_res, 0, sizeof(LoopResult));
  return ICP_OK;
}

// Ping-pong over the connected inboxes (gn_loop.hip: k_loop_probe): collective -- every rank calls it at about the same
// time (a barrier of the driver in front); *ok = 1 when this rank saw every token of every peer.  A transport whose
// probe fails on ANY rank must not carry the loop (the driver agrees on that and tries the next one).
extern "C" int icp_loop_transport_probe(icp_handle *h, int rounds, int *ok) {
  if (!h || !ok || rounds < 1 || rounds > 1024) return ICP_BAD_ARGUMENT;
  Workspace &w = h->ws;
  if (!w.d_loop_inbox || w.loop_world < 1 || w.loop_rank < 0) return ICP_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(h->device));
  unsigned *d_ok = nullptr;
  HIP_TRY(hipMalloc(&d_ok, sizeof(unsigned)));
  HIP_TRY(hipMemsetAsync(d_ok, 0, sizeof(unsigned), h->stream));
  const unsigned base = (++w.loop_probe_gen) * 2048u;
  hipError_t e = launch_loop_probe(h, w.loop_rank, w.loop_world, w.loop_peers, base, (unsigned)rounds, d_ok);
  unsigned got = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&got, d_ok, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(d_ok);
  HIP_TRY(e);
  *ok = got == 1u ? 1 : 0;
  return ICP_OK;
}

// Forget every window prediction of this handle (the next evaluations take the pipelines that need none and re-centre).
// What a rank of a sharded registration does when a one-launch inner loop gave up (icp_shard_loop_wait: ICP_HIP_ERROR):
// the ranks' prediction histories may have diverged inside the abandoned launch, and the stage calls that serve from
// there on must see the same windows on every rank.
extern "C" int icp_reset_window_predictions(icp_handle *h) {
  if (!h) return ICP_BAD_ARGUMENT;
  Workspace &w = h->ws;
  w.win_valid = false;
  w.win_wide = false;
  for (auto &wk : w.win_kind) wk = Workspace::WinPred();
  for (auto &wk : w.hint_kind) wk = Workspace::WinPred();
  return ICP_OK;
}

// The inner loop of a sharded registration from evaluation `it0` on, in one launch per rank (gn_loop.hip:
// k_gn_loop_shard).  d_a / d_b: this rank's pairs (icp_shard_geometry: compact, fold order); the state is the loop's
// (src/lib.rs:62-82).  launch_no: 1, 2, ... -- the same number on every rank for the same launch, growing over the life
// of the connection; eval_base: evaluations the earlier launches of the connection served (every rank's results say the
// same).  ICP_RETRY_SHARDED: not launched -- no window prediction for evaluation it0, a pair set beyond the launch's
// size, fewer tree blocks than ranks: the stage calls (icp_shard_eval_*) serve that evaluation; the answer depends on
// replicated state only, so every rank gives it.
// nh = 1: the rank of hs[0]; nh = world: ALL ranks (hs[q] = rank q, one device) in one launch on hs[0]'s stream.
static int shard_loop_launch_common(icp_handle *const *hs, int nh, const double *const *d_a, const double *const *d_b,
                                    size_t n_total, unsigned launch_no, unsigned eval_base, int it0, uint32_t applied0,
                                    const icp_pose *Ti, double prev_error, int first_kind, int second_kind) {
  icp_handle *h0 = hs[0];
  Workspace &w0 = h0->ws;
  if (w0.loop_rank < 0 || !w0.d_loop_inbox || (nh != 1 && nh != w0.loop_world)) return ICP_BAD_ARGUMENT;
  const int world = w0.loop_world;
  if (!gn_loop_shard_applies(n_total, world)) return ICP_RETRY_SHARDED;  // (before the pointers: a rank without points has none)
  for (int q = 0; q < nh; ++q) {
    Workspace &w = hs[q]->ws;
    if (!d_a[q] || !d_b[q] || w.loop_world != world || (nh > 1 && w.loop_rank != q) || !w.d_loop_inbox) return ICP_BAD_ARGUMENT;
    // (w.loop_off -- "this handle's own one-launch loops were not resident lately" -- is rank-local state and must not
    // decide here: every rank has to give the same answer, ADVICE r5.  A sharded launch that gives up is reported by
    // every rank, and the drivers drop the loop for the connection.)
  }
  HIP_TRY(hipSetDevice(h0->device));
  LoopRankPtrs ptrs = {};
  for (int q = 0; q < nh; ++q) {
    Workspace &w = hs[q]->ws;
    if (!w.loop_plan) w.loop_plan = new (std::nothrow) LoopPlan();
    if (!w.loop_plan) return ICP_OUT_OF_MEMORY;
    LoopPlan &pl = *reinterpret_cast<LoopPlan *>(w.loop_plan);
    if (!loop_plan(hs[q], n_total, it0, first_kind, second_kind, false, &pl)) {
      if (q > 0) return ICP_HIP_ERROR;  // (the ranks' prediction state diverged: cannot happen)
      return ICP_RETRY_SHARDED;
    }
    if (q > 0) {  // every rank must bin with the same windows
      const LoopPlan &p0 = *reinterpret_cast<LoopPlan *>(w0.loop_plan);
      if (memcmp(&pl.A.PA, &p0.A.PA, sizeof(WinParams)) != 0 || pl.A.pb_valid != p0.A.pb_valid ||
          (pl.A.pb_valid && memcmp(&pl.A.PB, &p0.A.PB, sizeof(WinParams)) != 0) ||
          memcmp(&pl.A.f_next, &p0.A.f_next, sizeof(double)) != 0)
        return ICP_HIP_ERROR;
    }
    const int r = nh > 1 ? q : w.loop_rank;
    ptrs.a[r] = (const double2 *)d_a[q];
    ptrs.b[r] = (const double2 *)d_b[q];
    ptrs.res[r] = reinterpret_cast<LoopResult *>(w.h_loop_res);
    w.loop_seq = launch_no;
  }
  LoopPlan &pl = *reinterpret_cast<LoopPlan *>(w0.loop_plan);
  LoopArgs A = pl.A;
  LoopShardArgs S = {};
  S.rank = nh > 1 ? 0 : w0.loop_rank;
  S.world = world;
  int B = 0;
  for (int q = 0; q < world; ++q) {
    int b0, b1;
    size_t nl;
    shard_geometry(n_total, q, world, &b0, &b1, &B, &nl);
    S.first_block[q] = b0;
    S.first_block[q + 1] = b1;
    S.inbox[q] = reinterpret_cast<LoopInbox *>(w0.loop_peers[q]);
  }
  S.blocks_total = B;
  S.gen_base = launch_no * 1024u;  // (a launch runs at most 200 evaluations, each at most twice: rounds < 1024)
  S.eval_base = eval_base;
  A.n = (unsigned)n_total;
  A.it0 = (unsigned)it0;
  A.applied0 = applied0;
  A.T0 = *Ti;
  A.prev_error0 = prev_error;
  A.seq = launch_no;
  HIP_TRY(launch_gn_loop_shard(h0, A, S, ptrs, nh));
  return ICP_OK;
}

extern "C" int icp_shard_loop_launch_device(icp_handle *h, const double *d_a, const double *d_b, size_t n_total, unsigned launch_no,
                                            unsigned eval_base, int it0, uint32_t applied0, const icp_pose *Ti, double prev_error,
                                            int first_kind, int second_kind) {
  if (!h || !Ti || n_total >= 0xffffffffull || it0 < 0 || launch_no == 0) return ICP_BAD_ARGUMENT;
  return shard_loop_launch_common(&h, 1, &d_a, &d_b, n_total, launch_no, eval_base, it0, applied0, Ti, prev_error, first_kind,
                                  second_kind);
}

// (multi.hip) all the ranks of one device in one launch on rank 0's stream
int icp_shard_loop_launch_fused(icp_handle *const *hs, int world, const double *const *d_a, const double *const *d_b, size_t n_total,
                                unsigned launch_no, unsigned eval_base, int it0, uint32_t applied0, const icp_pose *Ti,
                                double prev_error, int first_kind, int second_kind) {
  if (!hs || world < 1 || world > kShardMaxWorld || !Ti || !d_a || !d_b || n_total >= 0xffffffffull || it0 < 0 || launch_no == 0)
    return ICP_BAD_ARGUMENT;
  for (int q = 0; q < world; ++q)
    if (!hs[q]) return ICP_BAD_ARGUMENT;
  return shard_loop_launch_common(hs, world, d_a, d_b, n_total, launch_no, eval_base, it0, applied0, Ti, prev_error, first_kind,
                                  second_kind);
}

// ... its result (blocks until this rank's launch has published it): the loop's state, *finished, or the evaluation
// *it that the stage calls must serve before the next launch.  ICP_HIP_ERROR: the launch gave up waiting for a peer.
extern "C" int icp_shard_loop_wait(icp_handle *h, icp_pose *Ti, double *prev_error, uint32_t *applied, int *it, int *finished,
                                   uint32_t *evals) {
  if (!h || !Ti || !prev_error || !applied || !it || !finished || !h->ws.loop_plan) return ICP_BAD_ARGUMENT;
  Workspace &w = h->ws;
  HIP_TRY(hipSetDevice(h->device));
  LoopResult *res = reinterpret_cast<LoopResult *>(w.h_loop_res);
  HIP_TRY(wait_seq(h, &res->seq, w.loop_seq));
  if (evals) *evals = res->rounds;  // (what the connection's eval_base advances by)
  if (res->status == 5) {
    ++w.loop_timeouts;  // (observability; the single-handle loop's back-off, Workspace::loop_off, is not touched)
    return ICP_HIP_ERROR;
  }
  bool fin = false;
  const int rc = loop_finish(h, *reinterpret_cast<LoopPlan *>(w.loop_plan), res, Ti, prev_error, applied, it, &fin);
  *finished = fin ? 1 : 0;
  return rc;
}

// ------------------------------------------------ sharded evaluation (stage calls) -----
// shard.hip has the design.  One evaluation = hist (residuals, histograms, block sums) -> [sum the histograms over
// ranks] -> compact -> [gather every rank's candidates + block sums] -> finish.  ICP_RETRY_REPLICATED
// from hist (no prediction yet) or finish (the window missed) means: gather the pairs of all ranks in
// global order and call icp_weighted_gn_step_device on them -- same bits, and it seeds the prediction.
extern "C" int icp_shard_geometry(size_t n_total, int rank, int world, int *b0, int *b1, int *blocks, size_t *n_local) {
  if (world < 1 || rank < 0 || rank >= world || !b0 || !b1 || !n_local || n_total >= 0xffffffffull) return ICP_BAD_ARGUMENT;
  shard_geometry(n_total, rank, world, b0, b1, blocks, n_local);
  return ICP_OK;
}
extern "C" size_t icp_shard_histogram_words(void) { return (size_t)2 * kWinBins + kShardStatusWords; }
extern "C" size_t icp_shard_candidates_bytes(void) { return shard_cand_bytes(); }
extern "C" size_t icp_shard_partials_bytes(int world) { return world >= 1 && world <= kShardMaxWorld ? shard_part_bytes(world) : 0; }
extern "C" size_t icp_shard_exchange_bytes(int world) { return world >= 1 && world <= kShardMaxWorld ? shard_exchange_bytes(world) : 0; }

static int shard_copy(icp_handle *h, const void *src, void *dst, size_t n_total, int rank, int world, size_t elem_bytes,
                      bool take) {
  if (!h || world < 1 || rank < 0 || rank >= world || elem_bytes == 0 || elem_bytes % 4 || n_total >= 0xffffffffull)
    return ICP_BAD_ARGUMENT;
  int b0, b1, blocks;
  size_t n_local;
  shard_geometry(n_total, rank, world, &b0, &b1, &blocks, &n_local);
  if (n_local == 0) return ICP_OK;  // (more ranks than reduction blocks: this rank owns no point, its buffers may be empty)
  if (!src || !dst) return ICP_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(launch_shard_copy(h, src, dst, n_total, rank, world, (unsigned)(elem_bytes / 4), take));
  return ICP_OK;
}
// icp_prepare_source_device for a rank's slice of the fold order (icp_shard_take_device): its points are runs of the
// cell-sorted cloud already, so the search snapshot keeps their order -- no second sort, pairs and indices stored as
// they lie (what icp_multi_estimate does for its ranks)
extern "C" int icp_shard_prepare_source_device(icp_handle *h, const double *d_src_local, size_t n_local, const icp_pose *T) {
  if (!h) return ICP_BAD_ARGUMENT;
  h->qsort.presorted = true;
  const int rc = icp_prepare_source_device(h, d_src_local, n_local, T);
  h->qsort.presorted = false;
  return rc;
}

// A rank's points of a sharded registration in ONE call: the fold order of the whole cloud under pose T (the sort of
// icp_sort_source_device: every rank computes the same permutation from the same inputs) and this rank's points gathered
// through it -- no sorted copy of the whole cloud (at N x 1M points that copy and its gather were a third of a call's head).
// d_perm (nullable, n_total words): the permutation, for whoever wants to put results back into the caller's order.
// Where one handle keeps the caller's order (sweep engine, up to 65 536 points) the permutation is the identity.
extern "C" int icp_shard_sort_take_device(icp_handle *h, const double *d_src_full, size_t n_total, const icp_pose *T, int rank, int world,
                                          double *d_local, uint32_t *d_perm) {
  if (!h || !T || world < 1 || world > kShardMaxWorld || rank < 0 || rank >= world || n_total >= 0xffffffffull ||
      (n_total > 0 && !d_src_full))
    return ICP_BAD_ARGUMENT;
  if (n_total == 0) return ICP_OK;
  {
    int b0, b1, blocks;
    size_t n_local;
    shard_geometry(n_total, rank, world, &b0, &b1, &blocks, &n_local);
    if (n_local > 0 && !d_local) return ICP_BAD_ARGUMENT;  // (more ranks than tree blocks: a rank without points has no buffer)
  }
  HIP_TRY(hipSetDevice(h->device));
  const unsigned words = (unsigned)(h->dim * 2);
  bool sorted = false;
  if ((long)n_total > grid_coop_max() && resolved_nn_mode(h) == ICP_NN_GRID) {  // (exactly where icp_estimate_device folds in snapshot order)
    h->qsort.sort_only = true;
    const int prc = icp_prepare_source_device(h, d_src_full, n_total, T);
    sorted = prc == ICP_OK && !h->qsort.sort_only && h->qsort.src == d_src_full && h->qsort.n == n_total && !h->qsort.valid;
    h->qsort.sort_only = false;
    if (prc != ICP_OK) return prc;
  }
  if (sorted) {
    HIP_TRY(launch_shard_take_perm(h, d_src_full, h->qsort.d_perm, d_local, n_total, rank, world, words));
    if (d_perm) HIP_TRY(hipMemcpyAsync(d_perm, h->qsort.d_perm, n_total * sizeof(uint32_t), hipMemcpyDeviceToDevice, h->stream));
  } else {
    HIP_TRY(launch_shard_copy(h, d_src_full, d_local, n_total, rank, world, words, true));
    if (d_perm) hipLaunchKernelGGL(icp::k_iota_u32, dim3(((unsigned)n_total + 255) / 256), dim3(256), 0, h->stream, d_perm, (unsigned)n_total);
  }
  h->qsort.valid = false;
  h->qsort.have_prev = false;
  return ICP_OK;
}

extern "C" int icp_shard_take_device(icp_handle *h, const void *d_full, void *d_local, size_t n_total, int rank, int world,
                                     size_t elem_bytes) {
  return shard_copy(h, d_full, d_local, n_total, rank, world, elem_bytes, true);
}
extern "C" int icp_shard_put_device(icp_handle *h, const void *d_local, void *d_full, size_t n_total, int rank, int world,
                                    size_t elem_bytes) {
  return shard_copy(h, d_local, d_full, n_total, rank, world, elem_bytes, false);
}

static int shard_eval_hist_impl(icp_handle *h, const double *d_a, const double *d_b, size_t n_total, int rank, int world,
                                const icp_pose *T, int kind, int refined, uint32_t **d_hist);

// Whatever this returns (short of ICP_BAD_ARGUMENT / ICP_NO_DEVICE), *d_hist is the buffer to sum over the ranks --
// histograms (all zero unless ICP_OK) followed by four status words, one-hot by the answer -- and EVERY rank is
// expected to take part in that sum: icp_shard_eval_status then tells every rank the same four counts.
extern "C" int icp_shard_eval_hist_device(icp_handle *h, const double *d_a, const double *d_b, size_t n_total, int rank,
                                          int world, const icp_pose *T, int kind, int refined, uint32_t **d_hist) {
  // (the last stage indexes its per-rank tables with at most kShardMaxWorld entries)
  if (!h || !T || !d_hist || world < 1 || world > kShardMaxWorld || rank < 0 || rank >= world || n_total >= 0xffffffffull)
    return ICP_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(ensure_workspace(h, 1, false));
  const int rc = shard_eval_hist_impl(h, d_a, d_b, n_total, rank, world, T, kind, refined, d_hist);
  if (rc == ICP_BAD_ARGUMENT) return rc;
  *d_hist = h->ws.d_whist;
  // the status words: written by the hist launch itself when this rank answers OK and owns blocks, else by a launch of their own
  if (!(rc == ICP_OK && h->shard.b1 - h->shard.b0 >= 1) && shard_launch_status(h, rc) != hipSuccess) return ICP_HIP_ERROR;
  return rc;
}

// The four counts {ranks that answered OK, RETRY_REPLICATED, NONE, anything else} of the evaluation in flight.
// from_device = 0: as the fold kernel of icp_shard_eval_finish_device left them in host memory (no wait: valid once
// finish has returned); 1: read from the summed buffer behind the stream (a rank whose own answer was not OK and
// which therefore ran no finish).
extern "C" int icp_shard_eval_status(icp_handle *h, uint32_t out[4], int from_device) {
  if (!h || !out) return ICP_BAD_ARGUMENT;
  if (!from_device) {
    if (!h->ws.h_res) return ICP_BAD_ARGUMENT;
    for (int k = 0; k < kShardStatusWords; ++k) out[k] = h->ws.h_res->status[k];
    return ICP_OK;
  }
  if (!h->ws.d_whist) return ICP_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMemcpyAsync(out, h->ws.d_whist + 2 * kWinBins, kShardStatusWords * sizeof(uint32_t), hipMemcpyDeviceToHost,
                         h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return ICP_OK;
}

// After an evaluation that the ranks did NOT all answer with ICP_OK: the summed buffer of a rank that had no
// histogram of its own holds its peers' counts; back to the all-zero rest state the next evaluation expects.
extern "C" int icp_shard_eval_abort_device(icp_handle *h) {
  if (!h || !h->ws.d_whist) return ICP_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMemsetAsync(h->ws.d_whist, 0, ((size_t)2 * kWinBins + kShardStatusWords) * sizeof(uint32_t), h->stream));
  h->shard.active = false;
  h->ws.gn_dirty = true;  // (the stages that did run may have left selection state behind)
  return ICP_OK;
}

static int shard_eval_hist_impl(icp_handle *h, const double *d_a, const double *d_b, size_t n_total, int rank, int world,
                                const icp_pose *T, int kind, int refined, uint32_t **d_hist) {
  icp_handle::ShardEval &S = h->shard;
  S.active = false;
  if (refined && !S.refined_ready) return ICP_BAD_ARGUMENT;  // only right after ICP_RETRY_SHARDED
  if (!input_size_ok(n_total)) return ICP_NONE;  // check_input_size, src/lib.rs:225-228
  shard_geometry(n_total, rank, world, &S.b0, &S.b1, &S.blocks, &S.n_local);
  if (S.blocks < world || (S.n_local > 0 && (!d_a || !d_b))) return S.blocks < world ? ICP_RETRY_REPLICATED : ICP_BAD_ARGUMENT;
  HIP_TRY(ensure_workspace(h, S.n_local, false));
  Workspace &w = h->ws;
  if (refined) {
    S.P = S.P2;
  } else if (!window_usable(h, n_total, &S.P, kind, true, n_total > 1000000 ? 0.2 : 0.)) {
    // (beyond 1M points a window narrow enough for the candidate lists would have to be predicted to
    // ~0.01 sigma: the first attempt is instead as WIDE as the layout allows -- it tolerates a
    // prediction that is off by 0.2 sigma -- and serves as the counting pass whose exact, global
    // counts place the narrow window of the second attempt: two sharded passes, like the one-GPU path
    // beyond 4M points, instead of a gather of all pairs)
    return ICP_RETRY_REPLICATED;
  }
  S.attempt_refined = refined != 0;
  S.refined_ready = false;
  if (!S.d_ordered) HIP_TRY(hipMalloc(&S.d_ordered, (size_t)kTreeMaxBlocks * (kNSum + 1) * sizeof(double)));
  if (!w.h_whist) HIP_TRY(hipHostMalloc(&w.h_whist, (size_t)2 * kWinBins * sizeof(uint32_t), hipHostMallocDefault));
  if (w.gn_dirty) {
    HIP_TRY(launch_sel_init(h, S.n_local));
    w.gn_dirty = false;
  }
  S.kind = kind;
  S.rank = rank;
  S.world = world;
  S.n_total = n_total;
  S.d_a = d_a;
  S.T = *T;
  ++w.win_tried;
  HIP_TRY(shard_launch_hist(h, d_a, d_b, S.n_local, S.T, S.P, S.b1 - S.b0));
  *d_hist = w.d_whist;
  S.active = true;
  return ICP_OK;
}

extern "C" int icp_shard_eval_compact_device(icp_handle *h, void *d_exchange_out) {
  if (!h || !h->shard.active || !d_exchange_out) return ICP_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(h->device));
  // (the global counts go to the host only if this window misses -- they place the next attempt's: k_shard_finish)
  const icp_handle::ShardEval &S = h->shard;
  HIP_TRY(shard_launch_compact(h, S.n_local, S.n_total, S.P, S.world, S.b1 - S.b0, d_exchange_out));
  return ICP_OK;
}

// the host half of `finish`: wait for the folded result, keep the prediction state, solve
static int shard_finish_common(icp_handle *h, double delta[3], double *huber_err) {
  icp_handle::ShardEval &S = h->shard;
  Workspace &w = h->ws;
  HIP_TRY(wait_result(h));
  S.active = false;
  const GnResult &r = *w.h_res;
  const int kind = S.kind;
  const bool own = Workspace::kind_has_slot(kind) && w.win_kind[kind].valid;
  bool &wide = own ? w.win_kind[kind].wide : w.win_wide;
  if (r.nan_flag) {
    w.gn_dirty = true;
    return ICP_NAN_INPUT;
  }
  if (r.overflow) {
    // The window missed (an order statistic outside its fine bins, or more candidates than the lists
    // hold).  Its counts are still exact counts of ALL ranks' residuals: they place the median and the
    // MAD to within a bin, and windows as narrow as the lists require go around them (refine_window,
    // the host half of the one-GPU path beyond 4M points) -- the next attempt, still sharded, then
    // hits.  Only a refined attempt that misses too goes back to the gathered pairs.
    ++w.win_missed;
    wide = true;
    if (!S.attempt_refined && refine_window(w.h_whist, S.n_total, S.P, &S.P2)) {
      S.refined_ready = true;
      return ICP_RETRY_SHARDED;
    }
    return ICP_RETRY_REPLICATED;
  }
  if (wide) {
    double shift = 0.;
    for (int d = 0; d < 2; ++d) {
      const double pm = own ? w.win_kind[kind].med[d] : w.win_med[d], ps = own ? w.win_kind[kind].sigma[d] : w.win_sigma[d];
      shift = fmax(shift, (fabs(r.median[d] - pm) + fabs(r.sigma[d] - ps)) / ps);
    }
    if (shift < 0.01) wide = false;
  }
  record_statistics(w, kind, true, r);
  if (huber_err) *huber_err = r.acc[12];
  return solve_update(r.acc, r.acc + 9, delta) ? ICP_OK : ICP_NONE;
}

extern "C" int icp_shard_eval_finish_device(icp_handle *h, const void *d_exchange_all, double delta[3], double *huber_err) {
  if (!h || !h->shard.active || !d_exchange_all || !delta) return ICP_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(shard_launch_finish(h, d_exchange_all, h->shard.world, h->shard.n_total, h->shard.blocks, h->shard.d_ordered));
  return shard_finish_common(h, delta, huber_err);
}

// (multi.hip) the same with the block of every rank read where it lies: one pointer per rank
int icp_shard_eval_finish_ptrs(icp_handle *h, const void *const *exch_ptrs, double delta[3], double *huber_err) {
  if (!h || !h->shard.active || !exch_ptrs || !delta) return ICP_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(shard_launch_finish_ptrs(h, exch_ptrs, h->shard.world, h->shard.n_total, h->shard.blocks, h->shard.d_ordered));
  return shard_finish_common(h, delta, huber_err);
}

// One evaluation of weighted_gauss_newton_update (+ the Huber error of the same pose) on device pairs,
// through whichever pipeline serves it -- what the inner loop of icp_estimate_transform_device calls per
// iteration, exposed for hosts that drive that loop themselves (the sharded driver's replicated fallback).
// kind: 0 first evaluation on new correspondences, 1 the one after the first update, 2 later ones.
extern "C" int icp_weighted_gn_step_device(icp_handle *h, const double *d_a, const double *d_b, size_t n,
                                           const icp_pose *T, int kind, double delta[3], double *huber_err) {
  if (!h || !T || !delta || (n > 0 && (!d_a || !d_b)) || n >= 0xffffffffull) return ICP_BAD_ARGUMENT;
  if (!input_size_ok(n)) return ICP_NONE;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(ensure_workspace(h, n, false));
  return wgn_step(h, d_a, d_b, n, *T, delta, huber_err, false, kind);
}

