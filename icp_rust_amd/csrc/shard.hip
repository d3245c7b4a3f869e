// Sharded evaluation of the robust Gauss-Newton step (src/lib.rs:218-261 + :45-50) across ranks --
// GPUs of one node, or "virtual ranks" on one GPU in tests -- with results that are, bit for bit, those
// of one GPU.  SURVEY.md 8(e) options 2/3; VERDICT r1 item 2.
//
// The N-term sums are folded in a fixed tree (DESIGN.md "GN reduction order"): global thread g of
// `blocks x 512` folds points g, g + G, g + 2G, ...; a block folds its threads; a second stage folds
// the block sums in block order.  Rank r OWNS the blocks [B r / W, B (r + 1) / W): it holds exactly the
// points those blocks fold (`n_local` of them, compacted in fold order: chunk `it` of the local arrays
// is the global range [it G + b0 512, it G + b1 512)), searches their nearest neighbours and produces
// their block sums with the SAME kernels the one-GPU path runs -- launched over its own blocks, on its
// local arrays, every per-thread and per-block sum is the one the global launch would have produced.
// What crosses ranks is small and exact, in TWO exchanges per evaluation (three until round 3: the block sums
// used to depend on sigma and so could only be made after the candidates had been exchanged; since the sums are
// kept per dimension without 1 / sigma -- common.hpp, kNSum -- they ride with the residual pass):
//   1. the window histograms (2 x 2048 u32): integer sums, order-free;
//   2. one block per rank (ShardCandHeader ...): the candidate lists around the median and the MAD ring (a few
//      hundred doubles: order statistics do not depend on the order of the list) and the rank's block sums
//      (19 doubles per block), which the last stage places in block order -- it then folds the same numbers in
//      the same order as one GPU does.
// The exchange itself is the caller's (RCCL through torch.distributed in icp_rust_amd/dist.py, peer
// copies inside icp_create_multi); these are the stage-level calls between the exchanges.
#include "common.hpp"
#include "gn_device.hpp"

namespace icp {

hipError_t launch_sel_init(icp_handle *h, size_t n);
__global__ void k_win_hist_sums(const double2 *__restrict__ a, const double2 *__restrict__ b, Pose T,
                                double *__restrict__ rx, double *__restrict__ ry, unsigned n, WinParams P, uint32_t *whist,
                                WinState *st, GnScalars *scal, double *partials, int status_cls);
template <bool LISTS>
__global__ void k_win_compact(const double *__restrict__ rx, const double *__restrict__ ry, unsigned n_local, unsigned n,
                              WinParams P, const uint32_t *__restrict__ whist, WinState *st, double *wmed, double *wring,
                              const unsigned *__restrict__ llen, unsigned lcap);
// (gn_win.hip, beside the selection code it shares with the one-GPU pipelines) the last stage: candidates of every
// rank -> exact statistics; block sums of every rank, in block order -> the result
__global__ void k_shard_finish(ShardPtrs srcs, int world, unsigned n_total, int blocks_total, const WinState *st,
                               double *ordered, uint32_t *whist, GnResult *res, unsigned seq, uint32_t *h_counts);

void shard_geometry(size_t n_total, int rank, int world, int *b0, int *b1, int *blocks, size_t *n_local) {
  int B, threads;
  reduce_geometry(n_total, &B, &threads);
  const int lo = (int)((long long)B * rank / world), hi = (int)((long long)B * (rank + 1) / world);
  const size_t G = (size_t)B * kReduceThreads, c0 = (size_t)lo * kReduceThreads, c1 = (size_t)hi * kReduceThreads;
  size_t cnt = 0;
  for (size_t base = 0; base < n_total; base += G) {
    const size_t s = base + c0, e = base + c1 < n_total ? base + c1 : n_total;
    if (e > s) cnt += e - s;
  }
  *b0 = lo;
  *b1 = hi;
  if (blocks) *blocks = B;
  *n_local = cnt;
}

// local slot l of rank (b0, b1) <-> global point index (fold order: see the header comment)
__device__ __forceinline__ size_t shard_global_index(size_t l, unsigned b0, unsigned b1, unsigned B) {
  const size_t row = (size_t)(b1 - b0) * kReduceThreads;
  const size_t it = l / row, off = l % row;
  return it * ((size_t)B * kReduceThreads) + (size_t)b0 * kReduceThreads + off;
}

// words = 4-byte words per element (6 for a 3-D point, 4 for a 2-D point, 1 for an index)
__global__ void k_shard_take(const uint32_t *__restrict__ full, uint32_t *__restrict__ local, size_t n_local,
                             unsigned b0, unsigned b1, unsigned B, unsigned words) {
  const size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n_local) return;
  const size_t i = shard_global_index(l, b0, b1, B);
  for (unsigned k = 0; k < words; ++k) local[l * words + k] = full[i * words + k];
}
__global__ void k_shard_put(const uint32_t *__restrict__ local, uint32_t *__restrict__ full, size_t n_local,
                            unsigned b0, unsigned b1, unsigned B, unsigned words) {
  const size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n_local) return;
  const size_t i = shard_global_index(l, b0, b1, B);
  for (unsigned k = 0; k < words; ++k) full[i * words + k] = local[l * words + k];
}

// ... straight out of the UNSORTED cloud through the fold order's permutation: local[l] = src[perm[global index of l]]
__global__ void k_shard_take_perm(const uint32_t *__restrict__ full, const uint32_t *__restrict__ perm, uint32_t *__restrict__ local,
                                  size_t n_local, unsigned b0, unsigned b1, unsigned B, unsigned words) {
  const size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n_local) return;
  const size_t i = perm[shard_global_index(l, b0, b1, B)];
  for (unsigned k = 0; k < words; ++k) local[l * words + k] = full[i * words + k];
}
hipError_t launch_shard_take_perm(icp_handle *h, const void *src, const uint32_t *perm, void *dst, size_t n_total, int rank, int world,
                                  unsigned words) {
  int b0, b1, B;
  size_t n_local;
  shard_geometry(n_total, rank, world, &b0, &b1, &B, &n_local);
  if (n_local == 0) return hipSuccess;
  hipLaunchKernelGGL(k_shard_take_perm, dim3((unsigned)((n_local + 255) / 256)), dim3(256), 0, h->stream, (const uint32_t *)src, perm,
                     (uint32_t *)dst, n_local, (unsigned)b0, (unsigned)b1, (unsigned)B, words);
  return hipGetLastError();
}

hipError_t launch_shard_copy(icp_handle *h, const void *src, void *dst, size_t n_total, int rank, int world,
                             unsigned words, bool take) {
  int b0, b1, B;
  size_t n_local;
  shard_geometry(n_total, rank, world, &b0, &b1, &B, &n_local);
  if (n_local == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((n_local + 255) / 256);
  if (take)
    hipLaunchKernelGGL(k_shard_take, dim3(blocks), dim3(256), 0, h->stream, (const uint32_t *)src, (uint32_t *)dst, n_local,
                       (unsigned)b0, (unsigned)b1, (unsigned)B, words);
  else
    hipLaunchKernelGGL(k_shard_put, dim3(blocks), dim3(256), 0, h->stream, (const uint32_t *)src, (uint32_t *)dst, n_local,
                       (unsigned)b0, (unsigned)b1, (unsigned)B, words);
  return hipGetLastError();
}

// ---- what a rank hands to the others ---------------------------------------------------
// one block per rank and evaluation (common.hpp: ShardCandHeader has the layout):
//   [ShardCandHeader][med x: kWinCapMed][med y][ring x: kWinCapRing][ring y] doubles,
//   then shard_part_rows(world) rows of (kNSum + 1) doubles: the rank's block sums first (unused rows zero), the
//   last row = {nan flag}
size_t shard_cand_bytes() { return sizeof(ShardCandHeader) + (size_t)(2 * kWinCapMed + 2 * kWinCapRing) * sizeof(double); }
int shard_part_rows(int world) { return (kTreeMaxBlocks + world - 1) / world + 1; }
size_t shard_part_bytes(int world) { return (size_t)shard_part_rows(world) * (kNSum + 1) * sizeof(double); }
size_t shard_exchange_bytes(int world) { return shard_cand_bytes() + shard_part_bytes(world); }

__global__ void k_shard_pack(const WinState *__restrict__ st, const double *__restrict__ wmed,
                             const double *__restrict__ wring, const double *__restrict__ partials, int blocks_local,
                             int rows, const GnScalars *__restrict__ scal, unsigned char *__restrict__ out) {
  ShardCandHeader *hd = reinterpret_cast<ShardCandHeader *>(out);
  double *body = reinterpret_cast<double *>(out + sizeof(ShardCandHeader));
  double *prow = body + (size_t)(2 * kWinCapMed + 2 * kWinCapRing);
  const unsigned cm[2] = {min(st->list_cnt[0][0], (unsigned)kWinCapMed), min(st->list_cnt[1][0], (unsigned)kWinCapMed)};
  const unsigned cr[2] = {min(st->list_cnt[2][0], (unsigned)kWinCapRing), min(st->list_cnt[3][0], (unsigned)kWinCapRing)};
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    hd->cnt[0] = st->list_cnt[0][0];
    hd->cnt[1] = st->list_cnt[1][0];
    hd->cnt[2] = st->list_cnt[2][0];
    hd->cnt[3] = st->list_cnt[3][0];
    hd->fail = st->fail;
  }
  const unsigned G = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int d = 0; d < 2; ++d) {
    for (unsigned e = t; e < cm[d]; e += G) body[(size_t)d * kWinCapMed + e] = wmed[(size_t)d * kWinCapMed + e];
    for (unsigned e = t; e < cr[d]; e += G)
      body[(size_t)2 * kWinCapMed + (size_t)d * kWinCapRing + e] = wring[(size_t)d * kWinCapRing + e];
  }
  const int W = kNSum + 1;
  for (unsigned e = t; e < (unsigned)(rows * W); e += G) {
    const int row = (int)e / W, k = (int)e % W;
    double v = 0.;
    if (row < blocks_local) v = k < kNSum ? partials[(size_t)row * W + k] : 0.;
    else if (row == rows - 1 && k == 0) v = (double)scal->nan_flag;
    prow[e] = v;
  }
}

// ---- stage launchers (api.hip drives them; each enqueues on h->stream) ---------------------------
// residuals + histograms + the block sums of this rank's blocks: the launch of the one-GPU pipeline over the
// rank's own blocks, on its local arrays
hipError_t shard_launch_hist(icp_handle *h, const double *d_a, const double *d_b, size_t n_local, const Pose &T,
                             const WinParams &P, int blocks_local) {
  Workspace &w = h->ws;
  if (blocks_local < 1) return hipSuccess;
  hipLaunchKernelGGL(k_win_hist_sums, dim3(blocks_local), dim3(kReduceThreads), 0, h->stream, (const double2 *)d_a,
                     (const double2 *)d_b, T, w.d_rx, w.d_ry, (unsigned)n_local, P, w.d_whist, w.d_wstate, w.d_scal,
                     w.d_partials, 0 /* this rank answers OK: the status words ride with the launch */);
  return hipGetLastError();
}

hipError_t shard_launch_compact(icp_handle *h, size_t n_local, size_t n_total, const WinParams &P, int world,
                                int blocks_local, void *d_out) {
  Workspace &w = h->ws;
  const unsigned n = (unsigned)n_local;
  const unsigned per = 512 * 4;
  unsigned hb = (n + per - 1) / per;
  if (hb > (unsigned)kWinBlocks) hb = kWinBlocks;
  if (hb < 1) hb = 1;
  hipLaunchKernelGGL(k_win_compact<false>, dim3(hb), dim3(512), 0, h->stream, (const double *)w.d_rx,
                     (const double *)w.d_ry, n, (unsigned)n_total, P, (const uint32_t *)w.d_whist, w.d_wstate, w.d_wmed,
                     w.d_wring, (const unsigned *)nullptr, 0u);
  hipLaunchKernelGGL(k_shard_pack, dim3(16), dim3(256), 0, h->stream, (const WinState *)w.d_wstate,
                     (const double *)w.d_wmed, (const double *)w.d_wring, (const double *)w.d_partials, blocks_local,
                     shard_part_rows(world), (const GnScalars *)w.d_scal, (unsigned char *)d_out);
  return hipGetLastError();
}

static ShardPtrs slices(const void *base, size_t stride, int world) {
  ShardPtrs t;
  for (int q = 0; q < kShardMaxWorld; ++q) t.p[q] = (const unsigned char *)base + (size_t)(q < world ? q : 0) * stride;
  return t;
}
static ShardPtrs table(const void *const *ptrs, int world) {
  ShardPtrs t;
  for (int q = 0; q < kShardMaxWorld; ++q) t.p[q] = (const unsigned char *)ptrs[q < world ? q : 0];
  return t;
}

static hipError_t finish_from(icp_handle *h, const ShardPtrs &blocks_of, int world, size_t n_total, int blocks_total,
                              double *d_ordered) {
  Workspace &w = h->ws;
  hipLaunchKernelGGL(k_shard_finish, dim3(1), dim3(kReduceThreads), 0, h->stream, blocks_of, world, (unsigned)n_total,
                     blocks_total, (const WinState *)w.d_wstate, d_ordered, w.d_whist, w.h_res, ++w.seq, w.h_whist);
  return hipGetLastError();
}
hipError_t shard_launch_finish(icp_handle *h, const void *d_exch_all, int world, size_t n_total, int blocks_total,
                               double *d_ordered) {
  return finish_from(h, slices(d_exch_all, shard_exchange_bytes(world), world), world, n_total, blocks_total, d_ordered);
}
hipError_t shard_launch_finish_ptrs(icp_handle *h, const void *const *exch_ptrs, int world, size_t n_total,
                                    int blocks_total, double *d_ordered) {
  return finish_from(h, table(exch_ptrs, world), world, n_total, blocks_total, d_ordered);
}

// The status words behind the histograms: one-hot by what this rank's hist stage answered.  They travel with the
// histograms through the ranks' sum, so that after it every rank holds the same four counts and takes the same
// branch -- also when a rank-local condition (an error, a stale prediction) made one rank answer differently.
__global__ void k_shard_status(uint32_t *st, int cls) {
  if (threadIdx.x < kShardStatusWords) st[threadIdx.x] = (int)threadIdx.x == cls ? 1u : 0u;
}
hipError_t shard_launch_status(icp_handle *h, int rc) {
  const int cls = rc == ICP_OK ? 0 : (rc == ICP_RETRY_REPLICATED ? 1 : (rc == ICP_NONE ? 2 : 3));
  hipLaunchKernelGGL(k_shard_status, dim3(1), dim3(64), 0, h->stream, h->ws.d_whist + 2 * kWinBins, cls);
  return hipGetLastError();
}

// ---- in-library exchange of icp_create_multi: flags + peer reads ---------------------------------
// A rank publishes a stage by bumping its own flag (system scope) behind the kernels that wrote the
// exported bytes; a consumer first runs k_multi_wait, one lane per peer, BOUNDED (a peer that never
// arrives raises the error word instead of hanging the queue).  Ranks that share a device run on ONE
// stream in lockstep order, so their waits are already satisfied when they are reached.
__global__ void k_multi_signal(unsigned *flag, unsigned value) {
  __threadfence_system();
  __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
struct FlagPtrs {
  const unsigned *p[kShardMaxWorld];
};
__global__ void k_multi_wait(FlagPtrs flags, int world, unsigned value, unsigned *err) {
  const int q = threadIdx.x;
  if (q < world) {
    bool ok = false;
    for (unsigned spin = 0; spin < 4000000u; ++spin) {
      if ((int)(__hip_atomic_load(flags.p[q], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - value) >= 0) {
        ok = true;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    if (!ok) atomicOr(err, 1u);
  }
  __threadfence_system();
}
__global__ void k_multi_sum_hist(ShardPtrs srcs, int world, uint32_t *__restrict__ out, unsigned words) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= words) return;
  uint32_t s = 0;
  for (int q = 0; q < world; ++q) s += reinterpret_cast<const uint32_t *>(srcs.p[q])[i];
  out[i] = s;
}
// pairs of rank q (its local a | b, n_q each) into this rank's full arrays, global order
__global__ void k_multi_put_pairs(const double2 *__restrict__ a_loc, const double2 *__restrict__ b_loc, size_t n_q,
                                  unsigned b0, unsigned b1, unsigned B, double2 *__restrict__ a_full,
                                  double2 *__restrict__ b_full) {
  const size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n_q) return;
  const size_t i = shard_global_index(l, b0, b1, B);
  a_full[i] = a_loc[l];
  b_full[i] = b_loc[l];
}

__global__ void k_multi_unpermute(const uint32_t *__restrict__ in, const uint32_t *__restrict__ perm, unsigned n,
                                  uint32_t *__restrict__ out) {
  const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[perm[k]] = in[k];
}
// out[perm[k]] = in[k]: indices in fold order back to the caller's order
hipError_t multi_unpermute(hipStream_t s, const uint32_t *in, const uint32_t *perm, size_t n, uint32_t *out) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_multi_unpermute, dim3(((unsigned)n + 255) / 256), dim3(256), 0, s, in, perm, (unsigned)n, out);
  return hipGetLastError();
}

hipError_t multi_signal(hipStream_t s, unsigned *flag, unsigned value) {
  hipLaunchKernelGGL(k_multi_signal, dim3(1), dim3(1), 0, s, flag, value);
  return hipGetLastError();
}
hipError_t multi_wait(hipStream_t s, const unsigned *const *flags, int world, unsigned value, unsigned *err) {
  FlagPtrs f;
  for (int q = 0; q < kShardMaxWorld; ++q) f.p[q] = flags[q < world ? q : 0];
  hipLaunchKernelGGL(k_multi_wait, dim3(1), dim3(64), 0, s, f, world, value, err);
  return hipGetLastError();
}
hipError_t multi_sum_hist(hipStream_t s, const void *const *hists, int world, uint32_t *out) {
  const unsigned words = 2 * kWinBins + kShardStatusWords;
  hipLaunchKernelGGL(k_multi_sum_hist, dim3((words + 255) / 256), dim3(256), 0, s, table(hists, world), world, out, words);
  return hipGetLastError();
}
hipError_t multi_put_pairs(hipStream_t s, const double *a_loc, const double *b_loc, size_t n_total, int rank, int world,
                           double *a_full, double *b_full) {
  int b0, b1, B;
  size_t nq;
  shard_geometry(n_total, rank, world, &b0, &b1, &B, &nq);
  if (!nq) return hipSuccess;
  hipLaunchKernelGGL(k_multi_put_pairs, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, (const double2 *)a_loc,
                     (const double2 *)b_loc, nq, (unsigned)b0, (unsigned)b1, (unsigned)B, (double2 *)a_full,
                     (double2 *)b_full);
  return hipGetLastError();
}

}  // namespace icp
