// "Window" variant of one inner Gauss-Newton evaluation (src/lib.rs:218-261 + :45-50): three
// launches instead of the seven of gn_pull.hip,
//     W (residuals + histograms)   C (compaction of the candidates)   A (accumulate)
//
// The radix pipelines spend two histogram passes and a compaction on the median
// (src/stats.rs:11-28) and the same again on the MAD (:30-37), because the MAD keys
// |r - median| do not exist before the median does.  Here the host predicts where the order
// statistics of this evaluation lie from the previous evaluation's median and sigma (they move
// by ~1e-2 sigma between evaluations of a converging registration) and W counts the residuals
// of each dimension into ONE histogram whose bins are fine in three windows -- around the
// predicted median and around median -+ MAD -- and coarse in between.  Any monotone bin
// function keeps order statistics exact; from the counts alone C derives
//   * the bins [jlo, jhi] holding the two middle order statistics -> median candidates;
//   * knowing only that the median lies in those bins: a set I of bins whose points are surely
//     closer to the median than the MAD, and a ring of bins around I that surely contains every
//     point at MAD distance (the bracket is derived at resolve_window below);
// and appends both candidate sets while re-reading the residuals once.  A ranks the median
// candidates, turns the ring into distances to that exact median, ranks those behind |I|, and
// accumulates.  Every order statistic is the exact one, so the results are bit-identical to
// gn_pull.hip / gn.hip and to the oracle's tree variant.  When the prediction
// is off (an order statistic outside its fine window, too many candidates) the evaluation
// reports `overflow = 2` and the host repeats it with gn_pull.hip, which also re-centres the
// windows.  (A single fine histogram over the whole range needs ~2 scattered global atomics per
// point: 92 us at 1M points, measured -- scattered atomics run at ~20 G/s on this chip.  The
// piecewise bins keep the histogram at 2 x 2048 words, small enough for LDS privatisation and a
// dense flush.)
#include "common.hpp"
#include "gn_device.hpp"
#include "gn_win_device.hpp"
#include "gn_loop.hpp"

namespace icp {


// ---- W ------------------------------------------------------------------------------
// SUMS: the running sums of the normal equations ride along (they do not depend on the order statistics any more,
// common.hpp: kNSum); the launch then has the geometry of the reduction tree and leaves one block sum per workgroup.
template <bool SUMS, int BATCH = kWinBatch>
__device__ __forceinline__ void win_hist_body(const double2 *__restrict__ a, const double2 *__restrict__ b,
                                              const Pose &T, double *__restrict__ rx, double *__restrict__ ry,
                                              unsigned n, const WinParams &P, uint32_t *whist, WinState *st,
                                              GnScalars *scal, double *partials, int status_cls = -1) {
  __shared__ uint32_t lh[2 * kWinBins];
  // (a sharded hist stage that answers OK: the rank's one-hot status words behind the histograms, shard.hip)
  if (SUMS && status_cls >= 0 && blockIdx.x == 0 && threadIdx.x < (unsigned)kShardStatusWords)
    whist[2 * kWinBins + threadIdx.x] = (int)threadIdx.x == status_cls ? 1u : 0u;
  double acc[SUMS ? kNSum : 1];
#pragma unroll
  for (int k = 0; k < (SUMS ? kNSum : 1); ++k) acc[k] = 0.;
#ifdef ICP_WIN_DEBUG
  long long wst[6];
  wst[0] = wall_clock64();
#endif
  for (unsigned i = threadIdx.x; i < 2u * kWinBins; i += kWinThreads) lh[i] = 0;
  __syncthreads();
#ifdef ICP_WIN_DEBUG
  wst[1] = wall_clock64();
#endif
  unsigned edge[4] = {0u, 0u, 0u, 0u};  // {below, above} x {x, y}: one word each, kept out of the LDS atomics
  bool saw_nan = false;
  const unsigned G = gridDim.x * kWinThreads;
  for (unsigned base = blockIdx.x * kWinThreads + threadIdx.x; base < n; base += G * BATCH) {
    double2 s[BATCH], d[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const unsigned i = base + u * G;
      if (i < n) {
        s[u] = a[i];
        d[u] = b[i];
      }
    }
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const unsigned i = base + u * G;
      if (i >= n) continue;
      // residual(), src/lib.rs:34-36
      const double v0 = ((T.r00 * s[u].x + T.r01 * s[u].y) + T.tx) - d[u].x;
      const double v1 = ((T.r10 * s[u].x + T.r11 * s[u].y) + T.ty) - d[u].y;
      rx[i] = v0;
      ry[i] = v1;
      saw_nan |= (v0 != v0) | (v1 != v1);
      const unsigned j0 = wbin(v0, P.d[0]), j1 = wbin(v1, P.d[1]);
      if (j0 == 0u) ++edge[0];
      else if (j0 == (unsigned)(kWinBins - 1)) ++edge[1];
      else atomicAdd(&lh[j0], 1u);
      if (j1 == 0u) ++edge[2];
      else if (j1 == (unsigned)(kWinBins - 1)) ++edge[3];
      else atomicAdd(&lh[kWinBins + j1], 1u);
      if (SUMS) accumulate_pair<true>(s[u], v0, v1, T, acc);  // (this thread's points in index order: the tree's first level)
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unsigned v = edge[k];  // the wave's total in its lane 63, by DPP (an inclusive scan's last lane): no LDS crossbar
    v = wave_scan_inclusive(v);
    if ((threadIdx.x & 63) == 63 && v) atomicAdd(&lh[(k >> 1) * kWinBins + ((k & 1) ? kWinBins - 1 : 0)], v);
  }
  if (saw_nan) atomicOr(&scal->nan_flag, 1);
  __shared__ double s_wsum[kWinThreads / 64][SUMS ? kNSum : 1];
  if (SUMS) block_reduce_waves<SUMS ? kNSum : 1>(acc, s_wsum);  // (in front of the barrier: gn_device.hpp)
#ifdef ICP_WIN_DEBUG
  wst[2] = wall_clock64();
#endif
  __syncthreads();
#ifdef ICP_WIN_DEBUG
  wst[3] = wall_clock64();
#endif
  for (unsigned i = threadIdx.x; i < 2u * kWinBins; i += kWinThreads) {  // dense flush: contiguous words
    const uint32_t c = lh[i];
    if (c) atomicAdd(&whist[i], c);
  }
  if (blockIdx.x == 0 && threadIdx.x < 4) st->list_cnt[threadIdx.x][0] = 0;  // the previous evaluation has read them
#ifdef ICP_WIN_DEBUG
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wst[4] = wall_clock64();
  if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 200) && st->med_cnt[0] % 16 == 3)
    printf("[W blk %d] zero %lld stream %lld barrier %lld flush %lld (x10ns)\n", blockIdx.x, wst[1] - wst[0],
           wst[2] - wst[1], wst[3] - wst[2], wst[4] - wst[3]);
#endif
  if (SUMS) block_reduce_finish<SUMS ? kNSum : 1, true>(s_wsum, partials + (size_t)blockIdx.x * (kNSum + 1));
}

#ifdef ICP_EXPERIMENTS  // (the histograms alone: first launch of the four-launch forms of rounds 1-3)
__global__ __launch_bounds__(kWinThreads) void k_win_hist(const double2 *__restrict__ a,
                                                          const double2 *__restrict__ b, Pose T,
                                                          double *__restrict__ rx, double *__restrict__ ry,
                                                          unsigned n, WinParams P, uint32_t *whist, WinState *st,
                                                          GnScalars *scal) {
  win_hist_body<false>(a, b, T, rx, ry, n, P, whist, st, scal, nullptr);
}
#endif

// launched with reduce_geometry(n): workgroup i leaves block sum i of the fixed reduction tree in `partials`
__global__ __launch_bounds__(kWinThreads) void k_win_hist_sums(const double2 *__restrict__ a,
                                                               const double2 *__restrict__ b, Pose T,
                                                               double *__restrict__ rx, double *__restrict__ ry,
                                                               unsigned n, WinParams P, uint32_t *whist, WinState *st,
                                                               GnScalars *scal, double *partials, int status_cls) {
  win_hist_body<true>(a, b, T, rx, ry, n, P, whist, st, scal, partials, status_cls);
}

#ifndef ICP_DEEP_BATCH
#define ICP_DEEP_BATCH 8
#endif
// beyond 4M points the same launch runs against HBM bandwidth: eight pairs in flight per lane (the geometry of the
// reduction tree allows one workgroup per CU only; 8 x 32 B x 512 lanes = 128 KB in flight per CU)
__global__ __launch_bounds__(kWinThreads) void k_win_hist_sums_deep(const double2 *__restrict__ a,
                                                                    const double2 *__restrict__ b, Pose T,
                                                                    double *__restrict__ rx, double *__restrict__ ry,
                                                                    unsigned n, WinParams P, uint32_t *whist,
                                                                    WinState *st, GnScalars *scal, double *partials) {
  win_hist_body<true, ICP_DEEP_BATCH>(a, b, T, rx, ry, n, P, whist, st, scal, partials, -1);
}

// ---- W + filed candidates (round 5) ---------------------------------------------------
// k_win_hist_sums that also FILES the candidates: every residual that lands in a fine bin is staged in LDS beside the
// count; once the counts of the workgroup are complete, an exclusive scan over its fine bins (directory order:
// dimension, window, bin) sorts the members by bin into the workgroup's OWN segment, and the scan itself -- where
// each fine bin's members start -- goes out as the segment's directory.  Whatever bins the complete histogram
// resolves to afterwards (k_win_pick), their residuals are lying in the 256 segments: the second pass over the
// points (k_win_finish's 16 MB at 1M pairs, its 343 instructions per point, its ticket and its 256 workgroups) is
// gone, and so are the stores of rx / ry.  A window of 0.05 sigma puts a tenth of the residuals of each dimension
// into fine bins: ~800 members per workgroup at 1M pairs.  (Measured and dropped: one bucket per bin for all
// workgroups, slots handed out by returning atomics -- on the histogram words themselves 6.5 us per launch, 256
// workgroups serialise on each of its 128 lines; on one padded counter per bin 14 us.)
// directory index f (dimension, window, bin) -> the histogram word
__host__ __device__ __forceinline__ unsigned fine_to_word(unsigned f) {
  const unsigned d = f / (3u * kWinFine), r = f % (3u * kWinFine), w = r / (unsigned)kWinFine, k = r % (unsigned)kWinFine;
  return d * (unsigned)kWinBins + (w == 0 ? (unsigned)kF0 : (w == 1 ? (unsigned)kF1 : (unsigned)kF2)) + k;
}
// ... and back, for a regular bin j of dimension d that IS fine
__device__ __forceinline__ unsigned word_to_fine(unsigned d, unsigned j) {
  const unsigned w = j >= (unsigned)kF2 ? 2u : (j >= (unsigned)kF1 ? 1u : 0u);
  return d * 3u * kWinFine + w * (unsigned)kWinFine + (j - (w == 0 ? (unsigned)kF0 : (w == 1 ? (unsigned)kF1 : (unsigned)kF2)));
}

// the arguments of the two launches of such an evaluation (kernel arguments by value; one definition for the plain
// launches and for the launch that carries one evaluation's second and another's first)
struct HistBktArgs {
  const double2 *a, *b;
  Pose T;
  unsigned n;
  WinParams P;
  uint32_t *whist;
  WinState *st;
  GnScalars *scal;
  double *partials;
  double *seg_all;
  unsigned short *dir_all;
};
struct HistBktLds {
  uint32_t lh[2 * kWinBins];
  double mv[kBktStage];
  unsigned short mk[kBktStage];
  WinRegion tab[2][8];  // (five rows each: wbin_tab)
  double wsum[kWinThreads / 64][kNSum];  // the waves' sums (block_reduce_store's `sm`)
  unsigned nmem;
};

// blk of nblk: this workgroup's place in the reduction tree's geometry (a launch may carry other work in front)
__device__ __forceinline__ void win_hist_sums_bkt_body(const HistBktArgs &A, const unsigned blk, const unsigned nblk,
                                                       HistBktLds &S) {
#ifndef ICP_BKT_BATCH
#define ICP_BKT_BATCH 2
#endif
  constexpr int BATCH = ICP_BKT_BATCH;
  const double2 *__restrict__ a = A.a, *__restrict__ b = A.b;
  const Pose T = A.T;
  const unsigned n = A.n;
  const WinParams &P = A.P;
  uint32_t *const whist = A.whist;
  WinState *const st = A.st;
  GnScalars *const scal = A.scal;
  double *const partials = A.partials;
  double *const seg_all = A.seg_all;
  unsigned short *const dir_all = A.dir_all;
  auto &lh = S.lh;
  auto &s_mv = S.mv;
  auto &s_mk = S.mk;
  unsigned &s_nmem = S.nmem;
  double acc[kNSum];
#pragma unroll
  for (int k = 0; k < kNSum; ++k) acc[k] = 0.;
#ifdef ICP_WIN_DEBUG
  long long bst[8];
  bst[0] = wall_clock64();
#endif
  for (unsigned i = threadIdx.x; i < 2u * kWinBins; i += kWinThreads) lh[i] = 0;
  if (threadIdx.x == 0) s_nmem = 0;
  if (threadIdx.x < 10) win_region_table(P.d[threadIdx.x / 5], S.tab[threadIdx.x / 5], (int)(threadIdx.x % 5));
  __syncthreads();
  const unsigned G = nblk * kWinThreads;
  const unsigned lane = threadIdx.x & 63u;
  // The loop's condition is the WAVE's (every lane stays active for the scan below).
  // The pairs of the NEXT batch are in flight while this one is worked on: the workgroups of a launch start together,
  // and with load-then-compute batches the whole chip alternated between a burst of loads and a burst of arithmetic.
  double2 ns[BATCH], nd[BATCH];
  auto fetch = [&](unsigned from) {
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const unsigned i = from + u * G;
      if (i < n) {
        ns[u] = a[i];
        nd[u] = b[i];
      }
    }
  };
#ifdef ICP_WIN_DEBUG
  bst[5] = wall_clock64();
  bst[6] = bst[7] = 0;
#endif
  fetch(blk * kWinThreads + threadIdx.x);
  for (unsigned base = blk * kWinThreads + threadIdx.x; base - lane < n; base += G * BATCH) {
    double2 s[BATCH], d[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      s[u] = ns[u];
      d[u] = nd[u];
    }
#ifdef ICP_WIN_DEBUG
    if (!bst[6]) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      bst[6] = wall_clock64();
    }
#endif
    if (base + G * BATCH - lane < n) fetch(base + G * BATCH);
#ifdef ICP_WIN_DEBUG
    if (!bst[7] && !(base + G * BATCH - lane < n)) bst[7] = wall_clock64();  // (start of the last batch's arithmetic)
#endif
    // the members of the fine windows among this thread's 2 x BATCH residuals: ONE reservation in the stage per wave
    // and batch (a returning LDS atomic per member made the wave wait eight times per batch)
    double mv[2 * BATCH];
    unsigned mj[BATCH];  // both bins of a pair, 16 bits each
    unsigned mem = 0;    // bit 2 u + dim
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const unsigned i = base + u * G;
      mv[2 * u] = mv[2 * u + 1] = 0.;
      mj[u] = 0u;
      if (i >= n) continue;
      // residual(), src/lib.rs:34-36
      const double v0 = ((T.r00 * s[u].x + T.r01 * s[u].y) + T.tx) - d[u].x;
      const double v1 = ((T.r10 * s[u].x + T.r11 * s[u].y) + T.ty) - d[u].y;
      bool lo0, hi0, f0, lo1, hi1, f1;
      const unsigned j0 = wbin_tab(v0, P.d[0], S.tab[0], lo0, hi0, f0), j1 = wbin_tab(v1, P.d[1], S.tab[1], lo1, hi1, f1);
      // (the catch-all bins 0 and kWinBins - 1 are counted like any other: kept out of the LDS atomics they cost four
      // selects, four adds and two exec masks per pair of a launch that is bound by instruction issue; a window that holds
      // sends them a few lanes of a wave, and one that does not is repeated anyway)
      atomicAdd(&lh[j0], 1u);
      atomicAdd(&lh[kWinBins + j1], 1u);
      mv[2 * u] = v0;
      mv[2 * u + 1] = v1;
      mj[u] = j0 | (j1 << 16);
      mem |= ((unsigned)f0 << (2 * u)) | ((unsigned)f1 << (2 * u + 1));
      accumulate_pair<true>(s[u], v0, v1, T, acc);  // (this thread's points in index order: the tree's first level)
    }
    const unsigned mine = (unsigned)__popc(mem), incl = wave_scan_inclusive(mine);
    const unsigned wave_total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
    if (wave_total) {
      unsigned got = 0;
      if (lane == 63u) got = atomicAdd(&s_nmem, wave_total);
      unsigned pos = (unsigned)__builtin_amdgcn_readlane((int)got, 63) + incl - mine;
#pragma unroll
      for (int q = 0; q < 2 * BATCH; ++q) {
        if (mem & (1u << q)) {
          if (pos < (unsigned)kBktStage) {
            s_mv[pos] = mv[q];
            s_mk[pos] = (unsigned short)((q & 1) ? (unsigned)kWinBins + (mj[q >> 1] >> 16) : (mj[q >> 1] & 0xffffu));
          }
          ++pos;
        }
      }
    }
  }
  // (a NaN residual makes the Huber sum NaN -- rho(NaN) is NaN on either branch -- and nothing else can: its terms are
  // >= 0; no test per pair)
  if (acc[kNSum - 1] != acc[kNSum - 1]) atomicOr(&scal->nan_flag, 1);
  // the wave's part of the block sums (block_reduce_store's: same tree, same order) BEFORE the barrier: a wave that is
  // through with its pairs folds while the slower ones finish, and the barrier that completes the histograms also
  // completes the wave sums
  block_reduce_waves<kNSum>(acc, S.wsum);
#ifdef ICP_WIN_DEBUG
  bst[1] = wall_clock64();
#endif
  __syncthreads();
#ifdef ICP_WIN_DEBUG
  bst[2] = wall_clock64();
#endif
  block_reduce_finish<kNSum, true>(S.wsum, partials + (size_t)blk * (kNSum + 1));
  for (unsigned i = threadIdx.x; i < 2u * kWinBins; i += kWinThreads) {  // dense flush: contiguous words, nobody waits
    const uint32_t c = lh[i];
    if (c) atomicAdd(&whist[i], c);
  }
  // the directory: eight consecutive fine bins per thread (they never straddle a window), one scan over the workgroup
  constexpr unsigned kDirPer = 8;
  static_assert(kWinFine % kDirPer == 0 && kBktFine / kDirPer <= kWinThreads, "directory slices");
  const unsigned f0 = threadIdx.x * kDirPer;
  const bool has_slice = f0 < (unsigned)kBktFine;
  const unsigned w0 = has_slice ? fine_to_word(f0) : 0u;
  unsigned cnt[kDirPer], sum = 0;
#pragma unroll
  for (unsigned q = 0; q < kDirPer; ++q) {
    cnt[q] = has_slice ? lh[w0 + q] : 0u;
    sum += cnt[q];
  }
  unsigned total;
  unsigned run = block_excl_scan(sum, &total);  // (two barriers: every count above is read before any is overwritten)
  unsigned short *const dir = dir_all + (size_t)blk * kBktDir;
  double *const seg = seg_all + (size_t)blk * kBktStage;
  if (has_slice) {
    unsigned o[kDirPer];
#pragma unroll
    for (unsigned q = 0; q < kDirPer; ++q) {
      o[q] = run;
      lh[w0 + q] = run;  // the bin's word now hands out the positions of its members
      run += cnt[q];
    }
    uint4 pk;
    pk.x = o[0] | (o[1] << 16);
    pk.y = o[2] | (o[3] << 16);
    pk.z = o[4] | (o[5] << 16);
    pk.w = o[6] | (o[7] << 16);
    *reinterpret_cast<uint4 *>(dir + f0) = pk;
  }
  if (threadIdx.x == 0) dir[kBktFine] = (unsigned short)(total < 0xffffu ? total : 0xffffu);
  __syncthreads();  // (the directory's offsets are in LDS)
#ifdef ICP_WIN_DEBUG
  bst[3] = wall_clock64();
#endif
  const unsigned nm_all = s_nmem;
  if (nm_all > (unsigned)kBktStage) {  // more members than the stage holds: nothing of this evaluation's files is usable
    if (threadIdx.x == 0) atomicOr(&st->stage_overflow, 1u);
  } else {
    for (unsigned e = threadIdx.x; e < nm_all; e += kWinThreads) {
      const unsigned key = s_mk[e];
      const unsigned pos = atomicAdd(&lh[key], 1u);  // (returning LDS atomic: directory offset + rank inside the bin)
      if (pos < (unsigned)kBktStage) seg[pos] = s_mv[e];
    }
  }
#ifdef ICP_WIN_DEBUG
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  bst[4] = wall_clock64();
  if (threadIdx.x == 0 && (blk == 0 || blk == 200) && (bst[0] & 0x1f0) == 0)
    printf("[B blk %d] stream %lld (prologue %lld first pairs %lld batches but the last %lld last %lld) barrier %lld flush+dir+reduce %lld "
           "scatter %lld (x10ns) members %u\n", blk, bst[1] - bst[0], bst[5] - bst[0], bst[6] - bst[5], bst[7] - bst[6],
           bst[1] - bst[7], bst[2] - bst[1], bst[3] - bst[2], bst[4] - bst[3], nm_all);
#endif
}

__global__ __launch_bounds__(kWinThreads, 4) void k_win_hist_sums_bkt(HistBktArgs A) {
  __shared__ HistBktLds S;
  win_hist_sums_bkt_body(A, blockIdx.x, gridDim.x, S);
}

// ---- W' (refined windows, n > 4M): the histograms again for new windows, from the stored residuals
// ... and the points inside the three fine windows of each dimension (where every candidate of the
// compaction will be) go to two dense lists: staged in LDS, one reservation per workgroup and list.
constexpr unsigned kRehistStage = 1024;
__global__ __launch_bounds__(kWinThreads) void k_win_rehist(const double *__restrict__ rx,
                                                            const double *__restrict__ ry, unsigned n, WinParams P,
                                                            uint32_t *whist, WinState *st, double *lx, double *ly,
                                                            unsigned lcap, unsigned *llen) {
  __shared__ uint32_t lh[2 * kWinBins];
  __shared__ double s_stage[2][kRehistStage];
  __shared__ unsigned s_n[2], s_base[2];
  for (unsigned i = threadIdx.x; i < 2u * kWinBins; i += kWinThreads) lh[i] = 0;
  if (threadIdx.x < 2) s_n[threadIdx.x] = 0;
  __syncthreads();
  auto fine = [](unsigned j) {
    return (j >= (unsigned)kF0 && j < (unsigned)kC0) || (j >= (unsigned)kF1 && j < (unsigned)kC1) ||
           (j >= (unsigned)kF2 && j < (unsigned)(kWinBins - 1));
  };
  auto keep = [&](int d, double v) {
    const unsigned pos = atomicAdd(&s_n[d], 1u);
    if (pos < kRehistStage) {
      s_stage[d][pos] = v;
    } else {  // more than the stage holds: one by one
      const unsigned g = atomicAdd(&llen[d], 1u);
      if (g < lcap) (d ? ly : lx)[g] = v;
    }
  };
  unsigned edge[4] = {0u, 0u, 0u, 0u};
  const unsigned G = gridDim.x * kWinThreads;
  for (unsigned base = blockIdx.x * kWinThreads + threadIdx.x; base < n; base += G * kWinBatch) {
    double v0[kWinBatch], v1[kWinBatch];
#pragma unroll
    for (int u = 0; u < kWinBatch; ++u) {
      const unsigned i = base + u * G;
      if (i < n) {
        v0[u] = rx[i];
        v1[u] = ry[i];
      }
    }
#pragma unroll
    for (int u = 0; u < kWinBatch; ++u) {
      const unsigned i = base + u * G;
      if (i >= n) continue;
      const unsigned j0 = wbin(v0[u], P.d[0]), j1 = wbin(v1[u], P.d[1]);
      if (j0 == 0u) ++edge[0];
      else if (j0 == (unsigned)(kWinBins - 1)) ++edge[1];
      else atomicAdd(&lh[j0], 1u);
      if (j1 == 0u) ++edge[2];
      else if (j1 == (unsigned)(kWinBins - 1)) ++edge[3];
      else atomicAdd(&lh[kWinBins + j1], 1u);
      if (fine(j0)) keep(0, v0[u]);
      if (fine(j1)) keep(1, v1[u]);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unsigned v = edge[k];  // the wave's total in its lane 63, by DPP (an inclusive scan's last lane): no LDS crossbar
    v = wave_scan_inclusive(v);
    if ((threadIdx.x & 63) == 63 && v) atomicAdd(&lh[(k >> 1) * kWinBins + ((k & 1) ? kWinBins - 1 : 0)], v);
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < 2u * kWinBins; i += kWinThreads) {
    const uint32_t c = lh[i];
    if (c) atomicAdd(&whist[i], c);
  }
  if (threadIdx.x < 2) {
    const unsigned c = s_n[threadIdx.x] < kRehistStage ? s_n[threadIdx.x] : kRehistStage;
    s_base[threadIdx.x] = c ? atomicAdd(&llen[threadIdx.x], c) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const unsigned c = s_n[d] < kRehistStage ? s_n[d] : kRehistStage;
    for (unsigned e = threadIdx.x; e < c; e += kWinThreads)
      if (s_base[d] + e < lcap) (d ? ly : lx)[s_base[d] + e] = s_stage[d][e];
  }
  if (blockIdx.x == 0 && threadIdx.x < 4) st->list_cnt[threadIdx.x][0] = 0;
}

// every (n / count)-th pair: the sample whose exact statistics centre the first pass
__global__ void k_sample_pairs(const double2 *__restrict__ a, const double2 *__restrict__ b, unsigned stride,
                               unsigned count, double2 *__restrict__ sa, double2 *__restrict__ sb) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  sa[i] = a[(size_t)i * stride];
  sb[i] = b[(size_t)i * stride];
}


// LISTS (second pass of the refined windows, n > 4M): rx / ry are not the n residuals but the lists
// k_win_rehist made of the points inside the three fine windows (llen[d] entries, at most lcap) --
// every median or MAD candidate is among them once the bins resolved below are checked to lie inside
// those windows -- so the compaction streams a few hundred thousand values instead of n.
// n_local: the residuals THIS launch streams; n: the points the histogram counts.  They differ only in a
// sharded evaluation (api.hip, "sharded evaluation"), where whist holds the sum over all ranks and every
// rank resolves the same bins but appends only its own candidates.
// FINISH: the launch goes on to select the order statistics itself (k_win_finish below) -- the candidates are
// stored write-through for the workgroup that arrives last, a missed window does not end the workgroup early, and
// what the histogram says about the candidate lists is handed back in `sel` (every workgroup derives the same).

// What the histograms resolve to, per dimension: the bins of the median candidates and of the ring (in every thread).
struct WinBins {
  unsigned mlo[2], mhi[2], a0[2], b1[2], i0[2], i1[2];
  bool fail;
};

// The front half of C, by the whole workgroup (five barriers): the histograms of both dimensions -> `cum` (LDS, 2 x
// kWinBins words: points in lower bins) -> the bracket (window_geometry / bracket_search / resolve_window, four waves
// side by side) -> the candidate bins.  REQUIRE_FINE: the median bins must lie in the middle fine window and the two
// arcs of the ring in the outer ones (the callers whose candidates exist only there: lists, buckets).  WANT_SEL: `sel`
// is filled (what the histogram says about the candidate lists).  Workgroup 0 also leaves everything in `st`.
// SUM_RANKS (the pipelined sharded evaluation): the counts are the sum of `nsrc` histograms `stride` words apart -- every
// rank's, pushed into this rank's inbox (system-scope loads).
template <bool REQUIRE_FINE, bool WANT_SEL, bool SUM_RANKS = false>
__device__ __forceinline__ WinBins win_resolve(const uint32_t *__restrict__ whist, unsigned n, const WinParams &P,
                                               WinState *st, const unsigned *__restrict__ llen, unsigned lcap,
                                               uint32_t *cum, WinSel &sel, int nsrc = 1, size_t stride = 0) {
  __shared__ unsigned s_selu[2][4];
  __shared__ double s_seld[2][4];
  __shared__ unsigned s_wtot[2][16];
  __shared__ int s_rng[2][8];
  __shared__ int s_t[2][2];
  __shared__ int s_j[2][2];  // the bins of the two middle ranks (-1: the counts do not add up to n -- a miss either way)
  constexpr int PER = kWinBins / kWinThreads, NW = kWinThreads / 64;
  static_assert(PER * kWinThreads == kWinBins && (PER == 2 || PER == 4), "bins per thread");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned klo = (n - 1) / 2, khi = n / 2;  // src/stats.rs:18-27
  if (tid < 4) s_j[tid >> 1][tid & 1] = -1;
#ifdef ICP_WIN_DEBUG
  __shared__ long long s_rdbg[8];
  if (tid == 0) s_rdbg[0] = wall_clock64();
#define RSTAMP(k) if (tid == 0) s_rdbg[k] = wall_clock64()
#else
#define RSTAMP(k)
#endif
  unsigned v[2][PER], inc[2], tot[2];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    if (SUM_RANKS) {
      static_assert(!SUM_RANKS || PER == 4, "two 64-bit loads per thread, dimension and rank");
#pragma unroll
      for (int i = 0; i < PER; ++i) v[d][i] = 0u;
      for (int q = 0; q < nsrc; ++q) {
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(whist + (size_t)q * stride + d * kWinBins) + 2 * tid;
        const unsigned long long x0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long x1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        v[d][0] += (unsigned)x0;
        v[d][1] += (unsigned)(x0 >> 32);
        v[d][PER - 2] += (unsigned)x1;
        v[d][PER - 1] += (unsigned)(x1 >> 32);
      }
    } else if (PER == 2) {
      const uint2 x = reinterpret_cast<const uint2 *>(whist + d * kWinBins)[tid];
      v[d][0] = x.x;
      v[d][1] = x.y;
    } else {
      const uint4 x = reinterpret_cast<const uint4 *>(whist + d * kWinBins)[tid];
      v[d][0] = x.x;
      v[d][1] = x.y;
      v[d][PER - 2] = x.z;
      v[d][PER - 1] = x.w;
    }
    tot[d] = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) tot[d] += v[d][i];
    unsigned s = tot[d];  // the wave's inclusive scan by DPP (row shifts, then the two row broadcasts): no LDS crossbar
    s = wave_scan_inclusive(s);
    inc[d] = s;
    if (lane == 63) s_wtot[d][wave] = s;
  }
  RSTAMP(1);
  __syncthreads();
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    unsigned wbase = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) wbase += (w < wave) ? s_wtot[d][w] : 0u;
    unsigned run = wbase + inc[d] - tot[d];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      cum[d * kWinBins + PER * tid + i] = run;
      // (the bin that holds a middle rank: exactly one thread sees it -- the last bin whose cumulative count is <= the rank)
      if (run <= klo && klo < run + v[d][i]) s_j[d][0] = PER * tid + i;
      if (run <= khi && khi < run + v[d][i]) s_j[d][1] = PER * tid + i;
      run += v[d][i];
    }
  }
  __syncthreads();
  RSTAMP(2);
  WinGeom geo = {};
  if (wave < 4) {  // waves 0,1: t1 of x,y; waves 2,3: t2 of x,y -- the two halves of the bracket side by side
    const int d = wave & 1, role = wave >> 1;
    geo = window_geometry(cum + d * kWinBins, n, P.d[d], s_j[d][0], s_j[d][1]);
    RSTAMP(3);
    const int t = geo.ok ? bracket_search(cum + d * kWinBins, n, P.d[d], geo, role) : -1;
    if (lane == 0) s_t[role][d] = t;
  }
  RSTAMP(4);
  __syncthreads();
  RSTAMP(5);
  if (wave < 2) {  // one wave per dimension
    const int d = wave;
    WinRanges R = {};
    unsigned med_base = 0, med_cnt = 0, inner = 0, ring_cnt = 0;
    double range[4] = {0., 0., 0., 0.};
    unsigned counted = 0;  // every point is in exactly one bin: anything else means the histogram is not
#pragma unroll             // this evaluation's (defence in depth for the hand-over between streams)
    for (int w = 0; w < NW; ++w) counted += s_wtot[d][w];
    bool ok = counted == n && resolve_window(cum + d * kWinBins, n, P.d[d], geo, s_t[0][d], s_t[1][d], R,
                                             med_base, med_cnt, inner, ring_cnt, range);
    if (REQUIRE_FINE && ok)  // the median bins in the middle window, the two arcs of the ring in the outer ones
      ok = (!llen || llen[d] <= lcap) && R.mlo >= kF1 && R.mhi < kC1 && R.i0 <= R.i1 && R.a0 >= kF0 &&
           R.i0 - 1 < kC0 && R.i1 + 1 >= kF2 && R.b1 <= kWinBins - 2;
    if (lane == 0) {
      s_rng[d][0] = R.mlo;
      s_rng[d][1] = R.mhi;
      s_rng[d][2] = R.a0;
      s_rng[d][3] = R.b1;
      s_rng[d][4] = R.i0;
      s_rng[d][5] = R.i1;
      s_rng[d][6] = ok ? 0 : 1;
      if (WANT_SEL) {
        s_selu[d][0] = med_base;
        s_selu[d][1] = med_cnt;
        s_selu[d][2] = inner;
        s_selu[d][3] = ring_cnt;
#pragma unroll
        for (int k = 0; k < 4; ++k) s_seld[d][k] = range[k];
      }
      if (blockIdx.x == 0) {
        st->med_base[d] = med_base;
        st->med_cnt[d] = med_cnt;
        st->ring_inner[d] = inner;
        st->ring_cnt[d] = ring_cnt;
        st->med_lo[d] = range[0];
        st->med_hi[d] = range[1];
        st->ring_lo[d] = range[2];
        st->ring_hi[d] = range[3];
      }
    }
  }
  RSTAMP(6);
  __syncthreads();
#ifdef ICP_WIN_DEBUG
  if (tid == 0 && gridDim.x <= 2 && blockIdx.x == 0 && (s_rdbg[0] & 0x3c0) == 0)
    printf("[R] loads+scan %lld cum %lld geometry %lld bracket %lld barrier %lld resolve_window %lld (x10ns)\n", s_rdbg[1] - s_rdbg[0],
           s_rdbg[2] - s_rdbg[1], s_rdbg[3] - s_rdbg[2], s_rdbg[4] - s_rdbg[3], s_rdbg[5] - s_rdbg[4], s_rdbg[6] - s_rdbg[5]);
#endif
#undef RSTAMP
  WinBins B;
  B.fail = (s_rng[0][6] | s_rng[1][6]) != 0;
  if (blockIdx.x == 0 && tid == 0) st->fail = B.fail ? 1u : 0u;
  if (WANT_SEL) {
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      sel.med_base[d] = s_selu[d][0];
      sel.med_cnt[d] = s_selu[d][1];
      sel.inner[d] = s_selu[d][2];
      sel.ring_cnt[d] = s_selu[d][3];
#pragma unroll
      for (int k = 0; k < 4; ++k) sel.range[d][k] = s_seld[d][k];
    }
  }
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    B.mlo[d] = (unsigned)s_rng[d][0];
    B.mhi[d] = (unsigned)s_rng[d][1];
    B.a0[d] = (unsigned)s_rng[d][2];
    B.b1[d] = (unsigned)s_rng[d][3];
    B.i0[d] = (unsigned)s_rng[d][4];
    B.i1[d] = (unsigned)s_rng[d][5];
  }
  return B;
}

template <bool LISTS, bool FINISH>
__device__ __forceinline__ bool win_compact_body(const double *__restrict__ rx, const double *__restrict__ ry,
                                                 unsigned n_local, unsigned n, const WinParams &P,
                                                 const uint32_t *__restrict__ whist, WinState *st, double *wmed,
                                                 double *wring, const unsigned *__restrict__ llen, unsigned lcap,
                                                 WinSel &sel) {
  __shared__ uint32_t cum[2 * kWinBins];  // points in lower bins
  auto put = [](double *p, double v) {
    if (FINISH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
  };
  __shared__ double s_med[2][kWinBlkMed], s_ring[2][kWinBlkRing];
  __shared__ unsigned s_cnt[4];
  const int tid = threadIdx.x;
#ifdef ICP_WIN_DEBUG
  long long cst[6];
  cst[0] = wall_clock64();
  cst[1] = cst[0];
#endif
  if (tid < 4) s_cnt[tid] = 0;
  const unsigned G = gridDim.x * kWinThreads;
  const WinBins R = win_resolve<LISTS, FINISH>(whist, n, P, st, llen, lcap, cum, sel);
#ifdef ICP_WIN_DEBUG
  cst[2] = wall_clock64();
#endif
  const bool fail = R.fail;
  if (!FINISH && fail) return true;
  const unsigned(&mlo)[2] = R.mlo, (&mhi)[2] = R.mhi, (&a0)[2] = R.a0, (&b1)[2] = R.b1, (&i0)[2] = R.i0, (&i1)[2] = R.i1;
  const unsigned lim[2] = {LISTS ? llen[0] : n_local, LISTS ? llen[1] : n_local};
  const unsigned nmax = fail ? 0u : (lim[0] > lim[1] ? lim[0] : lim[1]);  // (a missed window: nothing to collect)
  const unsigned base0 = blockIdx.x * kWinThreads + tid;
  for (unsigned base = base0; base < nmax; base += G * kWinBatch) {
    double v[2][kWinBatch];
#pragma unroll
    for (int u = 0; u < kWinBatch; ++u) {
      const unsigned i = base + u * G;
      if (i < lim[0]) v[0][u] = rx[i];
      if (i < lim[1]) v[1][u] = ry[i];
    }
#pragma unroll
    for (int u = 0; u < kWinBatch; ++u) {
      const unsigned i = base + u * G;
      if (i >= nmax) continue;
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        if (LISTS && i >= lim[d]) continue;
        const double r = v[d][u];
        const unsigned j = wbin(r, P.d[d]);
        // staged in LDS; a workgroup that meets more candidates than it can stage (a run of
        // neighbouring points with equal residuals) appends the excess one by one
        if (j >= mlo[d] && j <= mhi[d]) {
          const unsigned pos = atomicAdd(&s_cnt[d], 1u);
          if (pos < (unsigned)kWinBlkMed) {
            s_med[d][pos] = r;
          } else {
            const unsigned g = atomicAdd(&st->list_cnt[d][0], 1u);
            if (g < (unsigned)kWinCapMed) put(&wmed[(size_t)d * kWinCapMed + g], r);
          }
        }
        if (j >= a0[d] && j <= b1[d] && !(j >= i0[d] && j <= i1[d])) {
          const unsigned pos = atomicAdd(&s_cnt[2 + d], 1u);
          if (pos < (unsigned)kWinBlkRing) {
            s_ring[d][pos] = r;
          } else {
            const unsigned g = atomicAdd(&st->list_cnt[2 + d][0], 1u);
            if (g < (unsigned)kWinCapRing) put(&wring[(size_t)d * kWinCapRing + g], r);
          }
        }
      }
    }
  }
  __syncthreads();
#ifdef ICP_WIN_DEBUG
  cst[3] = wall_clock64();
#endif
  // one reservation per workgroup and list (the totals are known in advance, so the dense
  // lists cannot overflow)
  __shared__ unsigned s_base[4];
  if (tid < 4) {
    const unsigned cap = tid < 2 ? kWinBlkMed : kWinBlkRing;
    const unsigned c = s_cnt[tid] < cap ? s_cnt[tid] : cap;
    s_base[tid] = c ? atomicAdd(&st->list_cnt[tid][0], c) : 0u;
  }
  __syncthreads();
  if (tid < 2 * kWinBlkMed) {
    const int d = tid / kWinBlkMed, e = tid % kWinBlkMed;
    const unsigned pos = s_base[d] + e;
    if ((unsigned)e < s_cnt[d] && pos < (unsigned)kWinCapMed) put(&wmed[(size_t)d * kWinCapMed + pos], s_med[d][e]);
  } else if (tid < 2 * kWinBlkMed + 2 * kWinBlkRing) {
    const int q = tid - 2 * kWinBlkMed, d = q / kWinBlkRing, e = q % kWinBlkRing;
    const unsigned pos = s_base[2 + d] + e;
    if ((unsigned)e < s_cnt[2 + d] && pos < (unsigned)kWinCapRing) put(&wring[(size_t)d * kWinCapRing + pos], s_ring[d][e]);
  }
#ifdef ICP_WIN_DEBUG
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  cst[4] = wall_clock64();
  if (tid == 0 && blockIdx.x == 0 && (clock64() & 15) == 0)
    printf("[C blk %d] scan %lld resolve %lld stream %lld append %lld (x10ns)\n", blockIdx.x, cst[1] - cst[0],
           cst[2] - cst[1], cst[3] - cst[2], cst[4] - cst[3]);
#endif
  return fail;
}

template <bool LISTS>
__global__ __launch_bounds__(kWinThreads) void k_win_compact(const double *__restrict__ rx,
                                                             const double *__restrict__ ry, unsigned n_local,
                                                             unsigned n, WinParams P,
                                                             const uint32_t *__restrict__ whist, WinState *st,
                                                             double *wmed, double *wring,
                                                             const unsigned *__restrict__ llen, unsigned lcap) {
  WinSel sel;
  win_compact_body<LISTS, false>(rx, ry, n_local, n, P, whist, st, wmed, wring, llen, lcap, sel);
}


// The run-ahead search (icp_estimate_device): this is the FIRST evaluation of an outer iteration (inner pose =
// identity, prev_error = f64::MAX); if the inner loop applies this update and stops, the next outer pose is
// Exp(delta) * identity * outer -- estimate_transform_loop's and icp_estimate_device's own operations, with the
// functions the host uses (pose.hpp), so the host can check the bits.  The search behind this launch reads it.
// (one thread of wave 0: ordered before the release in publish_folded)
__device__ __forceinline__ void fill_ahead_pose(const double *s_tot, const double (&sig)[2], bool usable, const Pose &outer,
                                                AheadPose *ahead, GnResult *res) {
  AheadPose np;
  np.valid = 0;
  np.pad = 0;
  np.T = outer;
  if (usable) {
    double acc[kNAcc], delta[3];
#pragma unroll
    for (int k = 0; k < kNAcc; ++k) acc[k] = combine_sum(s_tot, k, sig);
    if (solve_update(acc, acc + 9, delta) &&
        !((delta[0] * delta[0] + delta[1] * delta[1]) + delta[2] * delta[2] < ICP_DELTA_NORM_THRESHOLD)) {
      bool in_range;
      const Pose step = transform_new_in_range(delta, &in_range);
      if (in_range) {
        const Pose inner = transform_mul(step, transform_identity());  // src/lib.rs:81
        np.T = transform_mul(inner, outer);                            // src/lib.rs:127, 170
        np.valid = 1;
      }
    }
  }
  *ahead = np;
  res->next_pose = np.T;
  res->next_valid = np.valid;
}

// C + the rest of the evaluation, for sums that were accumulated beside the histograms (k_win_hist_sums): the
// workgroup that arrives last has every candidate of the launch in reach (write-through stores, sc1 loads), ranks
// them, folds the block sums of the earlier launch and releases the result -- two launches per evaluation.
static_assert(kWinThreads == kReduceThreads, "the launches of an evaluation share the reduction tree's geometry");
// LISTS (second pass of the refined windows): rx / ry are the lists k_win_rehist made (llen entries, at most lcap).
template <bool LISTS>
__global__ __launch_bounds__(kWinThreads) void k_win_finish(const double *__restrict__ rx,
                                                            const double *__restrict__ ry, unsigned n, WinParams P,
                                                            uint32_t *whist, WinState *st, double *wmed, double *wring,
                                                            GnScalars *scal, const double *partials, int sum_blocks,
                                                            SelCtl *ctl, GnResult *res, unsigned seq,
                                                            const unsigned *__restrict__ llen, unsigned lcap,
                                                            AheadPose *ahead, Pose outer) {
  constexpr int PM = kWinCapMed / kReduceThreads, PR = kWinCapRing / kReduceThreads;
  WinSel sel;
#ifdef ICP_WIN_DEBUG
  long long fst[10];
  fst[0] = wall_clock64();
#endif
  // The block sums of the launch in front of this one are complete: the first twenty workgroups fold one SUM each
  // (fold_one_load now, fold_one_reduce behind their candidate pass: gn_device.hpp) and leave the totals in the row
  // behind the block sums; the last workgroup loads twenty doubles instead of folding 256 x 20 in its serial tail.
  double *const totals = const_cast<double *>(partials) + (size_t)kTreeMaxBlocks * (kNSum + 1);
  const bool fold_early = gridDim.x >= (unsigned)(kNSum + 1) && sum_blocks <= kReduceMaxBlocks;  // (beyond 2^20 pairs: the general fold)
  const double fold_x = (fold_early && blockIdx.x < (unsigned)(kNSum + 1)) ? fold_one_load(partials, sum_blocks, (int)blockIdx.x) : 0.;
  bool fail = win_compact_body<LISTS, true>(rx, ry, n, n, P, whist, st, wmed, wring, llen, lcap, sel);
  if (fold_early && blockIdx.x < (unsigned)(kNSum + 1)) {
    __shared__ double s_fold4[4];
    const double tot = fold_one_reduce(fold_x, sum_blocks, s_fold4);
    if (threadIdx.x == 0) __hip_atomic_store(&totals[blockIdx.x], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#ifdef ICP_WIN_DEBUG
  fst[1] = wall_clock64();
#endif
  if (!last_block_arrives(&ctl->t[2])) return;
#ifdef ICP_WIN_DEBUG
  fst[2] = wall_clock64();
#endif
  const unsigned tid = threadIdx.x;
  double med[2] = {0., 0.}, sig[2] = {0., 0.};
  __shared__ double s_tot[kNSum + 1];
  const int nan_flag = __hip_atomic_load(&scal->nan_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  {
    // one round trip for everything the rest of the kernel reads: candidates, their counts, the block sums
    double vm[2][PM], vr[2][PR];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
#pragma unroll
      for (int u = 0; u < PM; ++u)  // (only the slots the histogram says are filled: a tenth of the capacity, typically)
        vm[d][u] = (fail || tid + u * kReduceThreads < sel.med_cnt[d])
                       ? __hip_atomic_load(&wmed[(size_t)d * kWinCapMed + tid + u * kReduceThreads], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT)
                       : 0.;
#pragma unroll
      for (int u = 0; u < PR; ++u)
        vr[d][u] = (fail || tid + u * kReduceThreads < sel.ring_cnt[d])
                       ? __hip_atomic_load(&wring[(size_t)d * kWinCapRing + tid + u * kReduceThreads], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT)
                       : 0.;
    }
    unsigned got[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      got[k] = __hip_atomic_load(&st->list_cnt[k][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (fold_early) {  // (the totals the first workgroups left: stored before their tickets)
      if (tid < kNSum + 1) s_tot[tid] = __hip_atomic_load(&totals[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (sum_blocks <= kReduceMaxBlocks) {
      fold_block_sums_256(partials, sum_blocks, s_tot);  // (they do not depend on the statistics selected below)
    } else {
      fold_block_sums_lean(partials, sum_blocks, s_tot);  // (beyond 2^20 pairs; inside this kernel's register budget)
    }
    // the histograms of the next evaluation start from zero (every workgroup has read them); write-through, and
    // drained before the barriers in front of the release below: the host may hand the next evaluation to the
    // handle's other stream as soon as it sees this result
    for (unsigned i = tid; i < 2u * kWinBins; i += kWinThreads)
      __hip_atomic_store(&whist[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (the appended counts are cross-checked against the histogram: a mismatch is a miss)
    fail = fail || got[0] != sel.med_cnt[0] || got[1] != sel.med_cnt[1] || got[2] != sel.ring_cnt[0] ||
           got[3] != sel.ring_cnt[1];
    const unsigned klo = (n - 1) / 2, khi = n / 2;
    if (!fail) {
      unsigned long long key[2][2];
      const double m_lo[2] = {sel.range[0][0], sel.range[1][0]}, m_hi[2] = {sel.range[0][1], sel.range[1][1]};
      const long long mlo[2] = {(long long)klo - sel.med_base[0], (long long)klo - sel.med_base[1]};
      const long long mhi[2] = {(long long)khi - sel.med_base[0], (long long)khi - sel.med_base[1]};
#ifdef ICP_WIN_DEBUG
      fst[3] = wall_clock64();
#endif
      select_n<2, PM>(vm, sel.med_cnt, m_lo, m_hi, mlo, mhi, key, fail);
#ifdef ICP_WIN_DEBUG
      fst[4] = wall_clock64();
#endif
      if (!fail) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          med[d] = middle_of(n, key[d][0], key[d][1]);
#pragma unroll
          for (int u = 0; u < PR; ++u) vr[d][u] = fabs(vr[d][u] - med[d]);  // src/stats.rs:35
        }
        const double r_lo[2] = {sel.range[0][2], sel.range[1][2]}, r_hi[2] = {sel.range[0][3], sel.range[1][3]};
        const long long dlo[2] = {(long long)klo - sel.inner[0], (long long)klo - sel.inner[1]};
        const long long dhi[2] = {(long long)khi - sel.inner[0], (long long)khi - sel.inner[1]};
        select_n<2, PR>(vr, sel.ring_cnt, r_lo, r_hi, dlo, dhi, key, fail);
        if (!fail) {
          sig[0] = ICP_PPF34 * middle_of(n, key[0][0], key[0][1]);  // src/stats.rs:42-46
          sig[1] = ICP_PPF34 * middle_of(n, key[1][0], key[1][1]);
        } else {
          med[0] = med[1] = 0.;
        }
      }
    }
  }
  if (tid == 0 && fail) st->fail = 1u;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (ahead && tid == 0) fill_ahead_pose(s_tot, sig, !fail && !nan_flag, outer, ahead, res);
#ifdef ICP_WIN_DEBUG
  fst[5] = wall_clock64();
#endif
  publish_folded(s_tot, res, seq, sig, med, nan_flag, fail ? 2 : 0);
#ifdef ICP_WIN_DEBUG
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  fst[6] = wall_clock64();
  if (tid == 0 && seq % 16 == 5)
    printf("[F last blk %d of %d] compact %lld ticket %lld loads %lld sel1 %lld sel2 %lld publish %lld (x10ns) cnt %u %u %u %u\n", blockIdx.x,
           gridDim.x, fst[1] - fst[0], fst[2] - fst[1], fst[3] - fst[2], fst[4] - fst[3], fst[5] - fst[4], fst[6] - fst[5],
           sel.med_cnt[0], sel.med_cnt[1], sel.ring_cnt[0], sel.ring_cnt[1]);
#endif
}

// ---- P (round 5): the rest of an evaluation whose candidates are filed in the workgroups' segments ----------
// ONE workgroup: the histograms -> the candidate bins (win_resolve) -> thread (w, d) reads the directory of segment w
// for the (few) candidate bins of dimension d and lists its members of them -- (segment, position) words in LDS, the
// order does not matter to a selection -- -> every thread loads the candidates of its slots straight into registers
// -> the two exact selections -> the fold of the block sums (the tree's second stage, loaded before anything else)
// -> the solve for the run-ahead search -> the result.  No ticket, no second pass: a chain of dependent steps on one
// CU while the rest of the chip works on something else.
struct PickLds {  // (the descriptor lists are dead when the selections start: their LDS is overlaid)
  union {
    struct {
      uint32_t cum[2 * kWinBins];
      uint32_t med[2][kWinCapMed], ring[2][kWinCapRing];
    } g;
    SelectLds<2> sel;
  };
};

struct PickArgs {
  unsigned n;
  WinParams P;
  uint32_t *whist;
  WinState *st;
  const double *seg_all;
  const unsigned short *dir_all;
  int segments;
  GnScalars *scal;
  const double *partials;
  int sum_blocks;
  GnResult *res;
  unsigned seq;
  AheadPose *ahead;
  Pose outer;
};

// (by the workgroup with blockIdx.x == 0 of its launch: win_resolve leaves the state in `st` from there)
__device__ __forceinline__ void win_pick_body(const PickArgs &A, PickLds &L) {
  const unsigned n = A.n;
  const WinParams &P = A.P;
  uint32_t *const whist = A.whist;
  WinState *const st = A.st;
  const double *__restrict__ seg_all = A.seg_all;
  const unsigned short *__restrict__ dir_all = A.dir_all;
  const int segments = A.segments;
  GnScalars *const scal = A.scal;
  const double *const partials = A.partials;
  const int sum_blocks = A.sum_blocks;
  GnResult *const res = A.res;
  const unsigned seq = A.seq;
  AheadPose *const ahead = A.ahead;
  const Pose outer = A.outer;
  constexpr int PM = kWinCapMed / kReduceThreads, PR = kWinCapRing / kReduceThreads;
  static_assert(kReduceMaxBlocks * 2 == kReduceThreads, "one thread per segment and dimension");
  static_assert(kBktStage <= (1 << 12), "descriptor: segment << 12 | position");
  __shared__ double s_tot[kNSum + 1];
  __shared__ unsigned s_cnt[4];
  const unsigned tid = threadIdx.x;
#ifdef ICP_WIN_DEBUG
  long long pst[10];
  pst[0] = wall_clock64();
#define PSTAMP(k) pst[k] = wall_clock64()
#else
#define PSTAMP(k)
#endif
  if (tid < 4) s_cnt[tid] = 0;
  // (the two flags of the launch in front of this one are asked for FIRST: read after the resolve they were a round trip
  // of their own -- 3 us between the resolve's last stamp and the lists, r05_pick_phases.txt -- because the lists'
  // branch needs them at once)
  const int nan_flag = __hip_atomic_load(&scal->nan_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned stage_overflow = __hip_atomic_load(&st->stage_overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  WinSel sel;
  const WinBins R = win_resolve<true, true>(whist, n, P, st, nullptr, 0u, L.g.cum, sel);
  // (the block sums travel while the candidates are listed: the histograms went first)
  double fx[kFoldH];
  fold256_load(partials, sum_blocks, fx);
  PSTAMP(1);
  bool fail = R.fail;
  bool bucket_miss = stage_overflow != 0;
  double med[2] = {0., 0.}, sig[2] = {0., 0.};
  {
    // the runs of candidate bins of this thread's dimension, in directory order: the median's, the ring's two arcs
    const unsigned w = tid & (unsigned)(kReduceMaxBlocks - 1), d = tid / (unsigned)kReduceMaxBlocks;
    const bool usable = !fail && !bucket_miss;  // (uniform)
    // (this thread's dimension by selects: a run-time index into the arrays of R would put them into scratch memory)
    const unsigned mlo_d = d ? R.mlo[1] : R.mlo[0], mhi_d = d ? R.mhi[1] : R.mhi[0], a0_d = d ? R.a0[1] : R.a0[0];
    const unsigned b1_d = d ? R.b1[1] : R.b1[0], i0_d = d ? R.i0[1] : R.i0[0], i1_d = d ? R.i1[1] : R.i1[0];
    const unsigned len_m = usable ? mhi_d - mlo_d + 1u : 0u;
    const unsigned len_a = usable ? i0_d - a0_d : 0u, len_b = usable ? b1_d - i1_d : 0u;
    __syncthreads();  // (s_cnt's zeros)
    if (usable && !bucket_miss && (int)w < segments) {
      const unsigned short *dir = dir_all + (size_t)w * kBktDir;
      const unsigned fm = word_to_fine(d, mlo_d), fa = word_to_fine(d, a0_d), fb = word_to_fine(d, i1_d + 1u);
      // (the members of a run of consecutive bins are contiguous in the segment: the run's first offset .. the offset
      // behind its last bin -- six directory entries per thread, one trip)
      const unsigned short q0 = dir[fm], q1 = dir[fm + len_m], q2 = dir[fa], q3 = dir[fa + len_a], q4 = dir[fb],
                           q5 = dir[fb + len_b];
      const unsigned m0 = q0, m_end = q1, a_beg = q2, a_end = q3, b_beg = q4, b_end = q5;
      const unsigned cm = m_end - m0, cr = (a_end - a_beg) + (b_end - b_beg);
      if (cm) {
        const unsigned pos = atomicAdd(&s_cnt[d], cm);
        for (unsigned k = 0; k < cm; ++k)
          if (pos + k < (unsigned)kWinCapMed) L.g.med[d][pos + k] = (w << 12) | (m0 + k);
      }
      if (cr) {
        unsigned pos = atomicAdd(&s_cnt[2 + d], cr);
        for (unsigned k = a_beg; k < a_end; ++k, ++pos)
          if (pos < (unsigned)kWinCapRing) L.g.ring[d][pos] = (w << 12) | k;
        for (unsigned k = b_beg; k < b_end; ++k, ++pos)
          if (pos < (unsigned)kWinCapRing) L.g.ring[d][pos] = (w << 12) | k;
      }
    }
    __syncthreads();
    PSTAMP(2);
    // (the listed counts are cross-checked against the histogram: a mismatch is a miss)
    if (usable && !bucket_miss)
      bucket_miss = s_cnt[0] != sel.med_cnt[0] || s_cnt[1] != sel.med_cnt[1] || s_cnt[2] != sel.ring_cnt[0] ||
                    s_cnt[3] != sel.ring_cnt[1];
    double vm[2][PM], vr[2][PR];
#pragma unroll
    for (int dd = 0; dd < 2; ++dd) {
#pragma unroll
      for (int u = 0; u < PM; ++u) {
        const unsigned e = tid + u * kReduceThreads;
        vm[dd][u] = 0.;
        if (usable && !bucket_miss && e < sel.med_cnt[dd]) {
          const uint32_t x = L.g.med[dd][e];
          vm[dd][u] = seg_all[(size_t)(x >> 12) * kBktStage + (x & 0xfffu)];
        }
      }
#pragma unroll
      for (int u = 0; u < PR; ++u) {
        const unsigned e = tid + u * kReduceThreads;
        vr[dd][u] = 0.;
        if (usable && !bucket_miss && e < sel.ring_cnt[dd]) {
          const uint32_t x = L.g.ring[dd][e];
          vr[dd][u] = seg_all[(size_t)(x >> 12) * kBktStage + (x & 0xfffu)];
        }
      }
    }
    // the histograms of the next evaluation start from zero (write-through, drained before the release below: the
    // host may hand the next evaluation to the handle's other stream as soon as it sees this result)
    for (unsigned i = tid; i < 2u * kWinBins; i += kReduceThreads)
      __hip_atomic_store(&whist[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0 && stage_overflow) __hip_atomic_store(&st->stage_overflow, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    fold256_reduce(fx, sum_blocks, s_tot);  // (a barrier inside)
    __syncthreads();                        // (the descriptor lists are read: the selections may overlay them)
    PSTAMP(3);
    const unsigned klo = (n - 1) / 2, khi = n / 2;
    if (!fail && !bucket_miss) {
      unsigned long long key[2][2];
      const double m_lo[2] = {sel.range[0][0], sel.range[1][0]}, m_hi[2] = {sel.range[0][1], sel.range[1][1]};
      const long long mlo[2] = {(long long)klo - sel.med_base[0], (long long)klo - sel.med_base[1]};
      const long long mhi[2] = {(long long)khi - sel.med_base[0], (long long)khi - sel.med_base[1]};
      select_n_lds<2, PM>(vm, sel.med_cnt, m_lo, m_hi, mlo, mhi, key, fail, L.sel);
      PSTAMP(4);
      if (!fail) {
#pragma unroll
        for (int dd = 0; dd < 2; ++dd) {
          med[dd] = middle_of(n, key[dd][0], key[dd][1]);
#pragma unroll
          for (int u = 0; u < PR; ++u) vr[dd][u] = fabs(vr[dd][u] - med[dd]);  // src/stats.rs:35
        }
        const double r_lo[2] = {sel.range[0][2], sel.range[1][2]}, r_hi[2] = {sel.range[0][3], sel.range[1][3]};
        const long long dlo[2] = {(long long)klo - sel.inner[0], (long long)klo - sel.inner[1]};
        const long long dhi[2] = {(long long)khi - sel.inner[0], (long long)khi - sel.inner[1]};
        select_n_lds<2, PR>(vr, sel.ring_cnt, r_lo, r_hi, dlo, dhi, key, fail, L.sel);
        PSTAMP(5);
        if (!fail) {
          sig[0] = ICP_PPF34 * middle_of(n, key[0][0], key[0][1]);  // src/stats.rs:42-46
          sig[1] = ICP_PPF34 * middle_of(n, key[1][0], key[1][1]);
        } else {
          med[0] = med[1] = 0.;
        }
      }
    }
  }
  const bool missed = fail || bucket_miss;
  if (tid == 0 && missed) st->fail = 1u;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  PSTAMP(6);
  // overflow 3: the window may have been right, the FILES were not usable (the host steps back to the second pass)
  const int overflow = missed ? (fail ? 2 : 3) : 0;
  if (ahead) {  // (uniform) lane 0 derives the next outer pose while wave 1 stores the result; wave 0 releases both
    if (tid == 0) fill_ahead_pose(s_tot, sig, !missed && !nan_flag, outer, ahead, res);
    publish_values<1>(s_tot, res, sig, med, nan_flag, overflow);
    PSTAMP(7);
    __syncthreads();
    publish_seq(res, seq);
  } else {
    PSTAMP(7);
    publish_folded(s_tot, res, seq, sig, med, nan_flag, overflow);
  }
#ifdef ICP_WIN_DEBUG
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PSTAMP(8);
  if (tid == 0 && seq % 16 == 5)
    printf("[P] resolve %lld list %lld load+fold %lld sel1 %lld sel2 %lld drain %lld ahead %lld publish %lld (x10ns) cnt %u %u %u %u\n",
           pst[1] - pst[0], pst[2] - pst[1], pst[3] - pst[2], pst[4] - pst[3], pst[5] - pst[4], pst[6] - pst[5],
           pst[7] - pst[6], pst[8] - pst[7], sel.med_cnt[0], sel.med_cnt[1], sel.ring_cnt[0], sel.ring_cnt[1]);
#endif
#undef PSTAMP
}

__global__ __launch_bounds__(kReduceThreads) void k_win_pick(PickArgs A) {
  __shared__ PickLds L;
  win_pick_body(A, L);
}

// Two evaluations side by side (icp_estimate_device with a bet in flight: the next outer iteration's first evaluation
// and the deciding evaluation of the current one): ONE launch files the candidates of both (2 B workgroups: two per
// CU, four waves per SIMD instead of two hide each other's latencies), ONE launch of two workgroups finishes both.
// The deciding evaluation then neither shares the CUs with the search nor needs a stream of its own.
__global__ __launch_bounds__(kWinThreads, 4) void k_win_hist_sums_bkt2(HistBktArgs A, HistBktArgs B) {
  __shared__ HistBktLds S;
  const unsigned half = gridDim.x >> 1;
  if (blockIdx.x < half) win_hist_sums_bkt_body(A, blockIdx.x, half, S);
  else win_hist_sums_bkt_body(B, blockIdx.x - half, half, S);
}
// (win_resolve leaves the state in `st` from the workgroup with blockIdx.x == 0: the second evaluation's `st` is not
// maintained -- nothing reads it on this path)
__global__ __launch_bounds__(kReduceThreads) void k_win_pick2(PickArgs A, PickArgs B) {
  __shared__ PickLds L;
  if (blockIdx.x == 0) win_pick_body(A, L);
  else win_pick_body(B, L);
}

// ---- P across ranks (round 6): the finishing workgroup of an evaluation whose points are SHARDED -----------------
// Every rank has run the first launch (k_win_hist_sums_bkt / _bkt2) over ITS tree blocks on ITS pairs: its window
// histogram, its block sums, its filed candidates.  This workgroup -- one per rank and evaluation -- is where the ranks
// meet (gn_loop.hpp: PipeSlot): push histogram + block sums into every inbox, wait for every rank's, resolve the bins
// from the SUMMED counts (the same on every rank), list the rank's own candidates of those bins out of its own
// segments, push them, wait for every rank's, and from there on be k_win_pick: exact selections over ALL candidates,
// the fold of ALL block sums in block order, the solve, the next outer pose for the run-ahead search.  Every rank
// therefore releases the bits one GPU would have.  Two exchanges, no host, no collective library.
// Ranks that share a device (virtual ranks) ride in ONE launch (blockIdx.y = rank): workgroups of one launch are all
// resident and may wait for each other; separate launches on one hardware queue could not.
namespace {
constexpr long long kPipeTimeoutTicks = 300000000;  // wall_clock64 runs at 100 MHz: 3 s (processes sharing a GPU take turns)
__device__ __forceinline__ void pst_u32(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void pst_f64(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ unsigned pld_u32(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double pld_f64(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// wave 0 waits (bounded) until the first `count` (<= 16) flag words carry generation `gen`; *payload_or = the OR of their
// upper halves.  A wait that runs out raises `abort` in every inbox: the peers are waiting for data this rank will not
// send, and leave with it.  Uniform over the workgroup.
__device__ __forceinline__ bool pipe_poll(const unsigned long long *words, int count, unsigned gen, LoopInbox *const *box, int W,
                                          LoopInbox *me, unsigned *payload_or) {
  __shared__ int s_pok;
  __shared__ unsigned s_ppay;
  if (threadIdx.x < 64) {
    const int lane = (int)threadIdx.x;
    unsigned long long seen = 0;
    int ok = 1;
    const long long t0 = wall_clock64();
    for (;;) {
      bool here = true;
      if (lane < count) {
        seen = __hip_atomic_load(&words[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        here = (int)((unsigned)seen - gen) >= 0;
      }
      if (__all(here)) break;
      int stop = 0;
      if (lane == 0) {
        if (pld_u32(&me->abort[0]) != 0u) stop = 1;
        else if (wall_clock64() - t0 > kPipeTimeoutTicks) {
          for (int q = 0; q < W; ++q) pst_u32(&box[q]->abort[0], 1u);
          stop = 1;
        }
      }
      if (__builtin_amdgcn_readfirstlane(stop)) {
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    unsigned pay = lane < count ? (unsigned)(seen >> 32) : 0u;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) pay |= (unsigned)__shfl_xor((int)pay, off);
    if (lane == 0) {
      s_pok = ok;
      s_ppay = pay;
    }
  }
  __syncthreads();
  *payload_or = s_ppay;
  return s_pok != 0;
}
}  // namespace

constexpr int kPipeFuse = 8;  // ranks of ONE device a launch can carry (its arguments must stay below 4 KB)
struct PickRank {  // what differs from rank to rank
  uint32_t *whist;
  WinState *st;
  const double *seg_all;
  const unsigned short *dir_all;
  GnScalars *scal;
  const double *partials;
  GnResult *res;
  AheadPose *ahead;
  int rank, b0, nbl;  // this rank's tree blocks [b0, b0 + nbl): its segments and block sums are rows 0 .. nbl - 1 of its arrays
  unsigned seq;
};
struct PickEval {
  unsigned n;    // points of ALL ranks: the ranks of the order statistics
  unsigned gen;  // generation of this evaluation's exchange, the same on every rank
  int ahead_on, pad;
  WinParams P;
  Pose outer;
  PickRank rk[kPipeFuse];
};
struct PickShardLaunch {
  int world, B, nranks, nevals;
  LoopInbox *inbox[kShardMaxWorld];  // every rank's inbox as mapped on this device
  PickEval ev[2];
};
static_assert(sizeof(PickShardLaunch) <= 4000, "kernel arguments");

__device__ __forceinline__ void win_pick_shard_body(const unsigned n, const unsigned gen, const bool ahead_on, const WinParams &P,
                                                    const Pose &outer, const PickRank &K, LoopInbox *const *box, const int W,
                                                    const int B, PickLds &L) {
  constexpr int PM = kWinCapMed / kReduceThreads, PR = kWinCapRing / kReduceThreads;
  constexpr int HW = 2 * kWinBins / kReduceThreads;
  __shared__ double s_tot[kNSum + 1];
  __shared__ unsigned s_cnt[4];
  __shared__ unsigned s_base[kShardMaxWorld + 1][4];
  __shared__ unsigned s_bad;
  const unsigned tid = threadIdx.x;
  const unsigned buf = gen & (unsigned)(kPipeBufs - 1);
  LoopInbox *const me = box[K.rank];
  PipeSlot *const mine = &me->pipe[buf];
  if (tid < 4) s_cnt[tid] = 0;
  if (tid == 0) s_bad = 0;
  const int nan_own = __hip_atomic_load(&K.scal->nan_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned ovf_own = __hip_atomic_load(&K.st->stage_overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // ---- exchange 1: this rank's counts and block sums into every inbox ------------------------------------------------
  {
    unsigned hv[HW];
#pragma unroll
    for (int k = 0; k < HW; ++k) hv[k] = __hip_atomic_load(&K.whist[tid + k * kReduceThreads], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int q = 0; q < W; ++q) {
      uint32_t *dst = box[q]->pipe[buf].hist[K.rank];
#pragma unroll
      for (int k = 0; k < HW; ++k) pst_u32(&dst[tid + k * kReduceThreads], hv[k]);
    }
    const unsigned nrow = (unsigned)K.nbl * (unsigned)(kNSum + 1);
    for (unsigned j = tid; j < nrow; j += kReduceThreads) {
      const unsigned row = j / (unsigned)(kNSum + 1), k = j % (unsigned)(kNSum + 1);
      const double v = k < (unsigned)kNSum ? __hip_atomic_load(&K.partials[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.;
      for (int q = 0; q < W; ++q) pst_f64(&box[q]->pipe[buf].rows[(unsigned)K.b0 + row][k], v);
    }
    // the histograms of this rank's next evaluation start from zero (its counts are on their way)
#pragma unroll
    for (int k = 0; k < HW; ++k) __hip_atomic_store(&K.whist[tid + k * kReduceThreads], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0 && ovf_own) __hip_atomic_store(&K.st->stage_overflow, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if ((int)tid < W) {
    const unsigned pay = (nan_own ? 1u : 0u) | (ovf_own ? 2u : 0u);
    __hip_atomic_store(&box[tid]->pipe[buf].flag_hist[K.rank], (unsigned long long)gen | ((unsigned long long)pay << 32), __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
  unsigned pay1 = 0;
  bool alive = pipe_poll(mine->flag_hist, W, gen, box, W, me, &pay1);
  const int nan_flag = (pay1 & 1u) ? 1 : 0;
  bool bucket_miss = (pay1 & 2u) != 0u;
  WinSel sel = {};
  WinBins R = {};
  R.fail = true;
  double fx[kFoldH];
#pragma unroll
  for (int k = 0; k < kFoldH; ++k) fx[k] = 0.;
  if (alive) {  // (uniform)
    R = win_resolve<true, true, true>(&mine->hist[0][0], n, P, K.st, nullptr, 0u, L.g.cum, sel, W, (size_t)2 * kWinBins);
    if (B <= kReduceMaxBlocks) fold256_load<__HIP_MEMORY_SCOPE_SYSTEM>(&mine->rows[0][0], B, fx);
  }
  bool fail = R.fail;
  double med[2] = {0., 0.}, sig[2] = {0., 0.};
  {
    // ---- this rank's candidates of the resolved bins, out of its own segments (win_pick_body has the comments) -------
    const unsigned w = tid & (unsigned)(kReduceMaxBlocks - 1), d = tid / (unsigned)kReduceMaxBlocks;
    const bool usable = alive && !fail && !bucket_miss;  // (uniform, and the same on every rank)
    const unsigned mlo_d = d ? R.mlo[1] : R.mlo[0], mhi_d = d ? R.mhi[1] : R.mhi[0], a0_d = d ? R.a0[1] : R.a0[0];
    const unsigned b1_d = d ? R.b1[1] : R.b1[0], i0_d = d ? R.i0[1] : R.i0[0], i1_d = d ? R.i1[1] : R.i1[0];
    const unsigned len_m = usable ? mhi_d - mlo_d + 1u : 0u;
    const unsigned len_a = usable ? i0_d - a0_d : 0u, len_b = usable ? b1_d - i1_d : 0u;
    __syncthreads();  // (s_cnt's zeros)
    if (usable && (int)w < K.nbl) {
      const unsigned short *dir = K.dir_all + (size_t)w * kBktDir;
      const unsigned fm = word_to_fine(d, mlo_d), fa = word_to_fine(d, a0_d), fb = word_to_fine(d, i1_d + 1u);
      const unsigned short q0 = dir[fm], q1 = dir[fm + len_m], q2 = dir[fa], q3 = dir[fa + len_a], q4 = dir[fb],
                           q5 = dir[fb + len_b];
      const unsigned m0 = q0, m_end = q1, a_beg = q2, a_end = q3, b_beg = q4, b_end = q5;
      const unsigned cm = m_end - m0, cr = (a_end - a_beg) + (b_end - b_beg);
      if (cm) {
        const unsigned pos = atomicAdd(&s_cnt[d], cm);
        for (unsigned k = 0; k < cm; ++k)
          if (pos + k < (unsigned)kWinCapMed) L.g.med[d][pos + k] = (w << 12) | (m0 + k);
      }
      if (cr) {
        unsigned pos = atomicAdd(&s_cnt[2 + d], cr);
        for (unsigned k = a_beg; k < a_end; ++k, ++pos)
          if (pos < (unsigned)kWinCapRing) L.g.ring[d][pos] = (w << 12) | k;
        for (unsigned k = b_beg; k < b_end; ++k, ++pos)
          if (pos < (unsigned)kWinCapRing) L.g.ring[d][pos] = (w << 12) | k;
      }
    }
    __syncthreads();
    // ---- exchange 2: their values (and how many) into every inbox -------------------------------------------------------
    const unsigned own[4] = {s_cnt[0], s_cnt[1], s_cnt[2], s_cnt[3]};
    const bool own_fail = usable && (own[0] > (unsigned)kWinCapMed || own[1] > (unsigned)kWinCapMed || own[2] > (unsigned)kWinCapRing ||
                                     own[3] > (unsigned)kWinCapRing);
    if (usable && !own_fail) {
#pragma unroll
      for (int dd = 0; dd < 2; ++dd) {
#pragma unroll
        for (int u = 0; u < PM; ++u) {
          const unsigned e = tid + u * kReduceThreads;
          if (e < own[dd]) {
            const uint32_t x = L.g.med[dd][e];
            const double v = K.seg_all[(size_t)(x >> 12) * kBktStage + (x & 0xfffu)];
            for (int q = 0; q < W; ++q) pst_f64(&box[q]->pipe[buf].cand[K.rank][dd * kWinCapMed + e], v);
          }
        }
#pragma unroll
        for (int u = 0; u < PR; ++u) {
          const unsigned e = tid + u * kReduceThreads;
          if (e < own[2 + dd]) {
            const uint32_t x = L.g.ring[dd][e];
            const double v = K.seg_all[(size_t)(x >> 12) * kBktStage + (x & 0xfffu)];
            for (int q = 0; q < W; ++q) pst_f64(&box[q]->pipe[buf].cand[K.rank][2 * kWinCapMed + dd * kWinCapRing + e], v);
          }
        }
      }
    }
    if (tid < 5u) {
      const unsigned v = tid < 4u ? ((usable && !own_fail) ? own[tid & 3u] : 0u) : (own_fail ? 1u : 0u);
      for (int q = 0; q < W; ++q) pst_u32(&box[q]->pipe[buf].cand_cnt[K.rank][tid], v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if ((int)tid < W)
      __hip_atomic_store(&box[tid]->pipe[buf].flag_cand[K.rank], (unsigned long long)gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    unsigned pay2 = 0;
    if (alive) alive = pipe_poll(mine->flag_cand, W, gen, box, W, me, &pay2);
    // ---- all candidates: where the lists of the ranks start in the concatenation (rank order; any order would do) ------
    if (tid == 0) {
      unsigned base[4] = {0u, 0u, 0u, 0u}, bad = 0u;
      if (alive)
        for (int q = 0; q < W; ++q) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            s_base[q][k] = base[k];
            base[k] += pld_u32(&mine->cand_cnt[q][k]);
          }
          bad |= pld_u32(&mine->cand_cnt[q][4]);
        }
#pragma unroll
      for (int k = 0; k < 4; ++k) s_base[W][k] = base[k];
      s_bad = bad;
    }
    __syncthreads();
    // (the gathered counts are cross-checked against the histogram: a mismatch is a miss)
    if (usable)
      bucket_miss = s_bad != 0u || s_base[W][0] != sel.med_cnt[0] || s_base[W][1] != sel.med_cnt[1] ||
                    s_base[W][2] != sel.ring_cnt[0] || s_base[W][3] != sel.ring_cnt[1];
    const bool take = usable && alive && !bucket_miss;
    auto fetch = [&](int k, unsigned e) -> double {
      int q = 0;
      while (q + 1 < W && s_base[q + 1][k] <= e) ++q;
      const unsigned off = k < 2 ? (unsigned)k * kWinCapMed : 2u * kWinCapMed + (unsigned)(k - 2) * kWinCapRing;
      return pld_f64(&mine->cand[q][off + (e - s_base[q][k])]);
    };
    double vm[2][PM], vr[2][PR];
#pragma unroll
    for (int dd = 0; dd < 2; ++dd) {
#pragma unroll
      for (int u = 0; u < PM; ++u) {
        const unsigned e = tid + u * kReduceThreads;
        vm[dd][u] = (take && e < sel.med_cnt[dd]) ? fetch(dd, e) : 0.;
      }
#pragma unroll
      for (int u = 0; u < PR; ++u) {
        const unsigned e = tid + u * kReduceThreads;
        vr[dd][u] = (take && e < sel.ring_cnt[dd]) ? fetch(2 + dd, e) : 0.;
      }
    }
    if (B <= kReduceMaxBlocks) fold256_reduce(fx, B, s_tot);  // (a barrier inside)
    else fold_block_sums_lean<__HIP_MEMORY_SCOPE_SYSTEM>(&mine->rows[0][0], B, s_tot);  // (beyond 2^20 points: thread t folds rows t, t + 512, ...)
    __syncthreads();               // (the descriptor lists are read: the selections may overlay them)
    const unsigned klo = (n - 1) / 2, khi = n / 2;
    if (take) {
      unsigned long long key[2][2];
      const double m_lo[2] = {sel.range[0][0], sel.range[1][0]}, m_hi[2] = {sel.range[0][1], sel.range[1][1]};
      const long long mlo[2] = {(long long)klo - sel.med_base[0], (long long)klo - sel.med_base[1]};
      const long long mhi[2] = {(long long)khi - sel.med_base[0], (long long)khi - sel.med_base[1]};
      select_n_lds<2, PM>(vm, sel.med_cnt, m_lo, m_hi, mlo, mhi, key, fail, L.sel);
      if (!fail) {
#pragma unroll
        for (int dd = 0; dd < 2; ++dd) {
          med[dd] = middle_of(n, key[dd][0], key[dd][1]);
#pragma unroll
          for (int u = 0; u < PR; ++u) vr[dd][u] = fabs(vr[dd][u] - med[dd]);  // src/stats.rs:35
        }
        const double r_lo[2] = {sel.range[0][2], sel.range[1][2]}, r_hi[2] = {sel.range[0][3], sel.range[1][3]};
        const long long dlo[2] = {(long long)klo - sel.inner[0], (long long)klo - sel.inner[1]};
        const long long dhi[2] = {(long long)khi - sel.inner[0], (long long)khi - sel.inner[1]};
        select_n_lds<2, PR>(vr, sel.ring_cnt, r_lo, r_hi, dlo, dhi, key, fail, L.sel);
        if (!fail) {
          sig[0] = ICP_PPF34 * middle_of(n, key[0][0], key[0][1]);  // src/stats.rs:42-46
          sig[1] = ICP_PPF34 * middle_of(n, key[1][0], key[1][1]);
        } else {
          med[0] = med[1] = 0.;
        }
      }
    }
  }
  const bool missed = fail || bucket_miss;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // overflow 5: a wait for a peer ran out (every rank reports it: the abort word went into every inbox); 2: the window
  // missed; 3: the files were not usable
  const int overflow = !alive ? 5 : (missed ? (fail ? 2 : 3) : 0);
  if (ahead_on && K.ahead) {  // (uniform)
    if (tid == 0) fill_ahead_pose(s_tot, sig, alive && !missed && !nan_flag, outer, K.ahead, K.res);
    publish_values<1>(s_tot, K.res, sig, med, nan_flag, overflow);
    __syncthreads();
    publish_seq(K.res, K.seq);
  } else {
    publish_folded(s_tot, K.res, K.seq, sig, med, nan_flag, overflow);
  }
}

__global__ __launch_bounds__(kReduceThreads) void k_win_pick_shard(PickShardLaunch A) {
  __shared__ PickLds L;
  // (the per-rank and per-evaluation tables of the arguments into LDS with CONSTANT indices: a run-time index into a
  // by-value argument makes the whole of it a private copy in scratch memory -- gn_loop.hip: k_gn_loop_shard)
  __shared__ LoopInbox *s_box[kShardMaxWorld];
  __shared__ PickRank s_rk;
  __shared__ WinParams s_P;
  __shared__ Pose s_outer;
  __shared__ unsigned s_n, s_gen;
  __shared__ int s_ahead;
#pragma unroll
  for (int q = 0; q < kShardMaxWorld; ++q)
    if ((int)threadIdx.x == q) s_box[q] = A.inbox[q];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    if ((int)blockIdx.x != e || threadIdx.x != 0) continue;
    s_P = A.ev[e].P;
    s_outer = A.ev[e].outer;
    s_n = A.ev[e].n;
    s_gen = A.ev[e].gen;
    s_ahead = A.ev[e].ahead_on;
#pragma unroll
    for (int j = 0; j < kPipeFuse; ++j)
      if ((int)blockIdx.y == j) s_rk = A.ev[e].rk[j];
  }
  __syncthreads();
  const PickRank K = s_rk;
  win_pick_shard_body(s_n, s_gen, s_ahead != 0, s_P, s_outer, K, s_box, A.world, A.B, L);
}

#ifdef ICP_EXPERIMENTS  // ---- the four-launch forms of rounds 1-3 (W, C, selection, A): `make experiments` only ----
// The order statistics of one evaluation from its candidate lists, one workgroup per dimension
// (half the registers of doing both at once): the co-resident variant of the pipeline runs this
// as its own tiny launch so that the accumulate kernel stays small.
__global__ __launch_bounds__(kReduceThreads) void k_win_select(unsigned n, WinState *st,
                                                               const double *__restrict__ wmed,
                                                               const double *__restrict__ wring, GnScalars *scal) {
  constexpr int PM = kWinCapMed / kReduceThreads, PR = kWinCapRing / kReduceThreads;
  const unsigned tid = threadIdx.x;
  const int d = blockIdx.x;
  const unsigned klo = (n - 1) / 2, khi = n / 2;
  double med = 0., sig = 0.;
  bool fail = false;
  if (st->fail == 0) {  // (written by k_win_compact; this kernel only ever raises it)
    double vm[1][PM], vr[1][PR];
#pragma unroll
    for (int u = 0; u < PM; ++u) vm[0][u] = wmed[(size_t)d * kWinCapMed + tid + u * kReduceThreads];
#pragma unroll
    for (int u = 0; u < PR; ++u) vr[0][u] = wring[(size_t)d * kWinCapRing + tid + u * kReduceThreads];
    const unsigned em[1] = {st->med_cnt[d]}, er[1] = {st->ring_cnt[d]};
    // (the appended counts are cross-checked against the histogram: a mismatch is a miss)
    fail = st->list_cnt[d][0] != em[0] || st->list_cnt[2 + d][0] != er[0];
    const double m_lo[1] = {st->med_lo[d]}, m_hi[1] = {st->med_hi[d]};
    const double r_lo[1] = {st->ring_lo[d]}, r_hi[1] = {st->ring_hi[d]};
    const long long mlo[1] = {(long long)klo - st->med_base[d]}, mhi[1] = {(long long)khi - st->med_base[d]};
    const long long dlo[1] = {(long long)klo - st->ring_inner[d]}, dhi[1] = {(long long)khi - st->ring_inner[d]};
    unsigned long long key[1][2];
    if (!fail) select_n<1, PM>(vm, em, m_lo, m_hi, mlo, mhi, key, fail);
    if (!fail) {
      med = middle_of(n, key[0][0], key[0][1]);
#pragma unroll
      for (int u = 0; u < PR; ++u) vr[0][u] = fabs(vr[0][u] - med);  // src/stats.rs:35
      select_n<1, PR>(vr, er, r_lo, r_hi, dlo, dhi, key, fail);
      if (!fail) sig = ICP_PPF34 * middle_of(n, key[0][0], key[0][1]);  // src/stats.rs:42-46
    }
  }
  if (tid == 0) {  // the accumulate kernel is the next launch on this stream
    scal->median[d] = med;
    scal->sigma[d] = sig;
    if (fail) atomicOr(&st->fail, 1u);
  }
}

// INLINE_SELECT: every workgroup derives the order statistics itself (lowest latency: the
// evaluation the host is waiting for).  Otherwise k_win_select has left them in `scal`, and this
// kernel needs few enough registers (68) to be placed beside three search waves per SIMD.
// n_total: the points of the whole evaluation (ranks of the order statistics); n: the points THIS launch
// accumulates.  PUBLISH = false (sharded evaluation): the block sums are all this rank contributes -- no
// ticket, no second stage; the statistics it selected go to `scal` for the kernel that folds every rank's
// block sums (k_shard_fold).
template <bool INLINE_SELECT, bool PUBLISH = true>
__global__ __launch_bounds__(kReduceThreads) void k_win_accumulate(
    const double2 *__restrict__ a, const double *__restrict__ rx, const double *__restrict__ ry, unsigned n,
    unsigned n_total, Pose T, const WinState *__restrict__ st, const double *__restrict__ wmed,
    const double *__restrict__ wring, GnScalars *scal, double *partials, uint32_t *whist, SelCtl *ctl, GnResult *res,
    unsigned seq) {
  if (!INLINE_SELECT) {
    const bool failed = st->fail != 0;
    const double med0[2] = {scal->median[0], scal->median[1]};
    const double sig0[2] = {scal->sigma[0], scal->sigma[1]};
    double acc[kNSum];
#pragma unroll
    for (int k = 0; k < kNSum; ++k) acc[k] = 0.;
    if (!failed) accumulate_points<2>(a, rx, ry, n, T, acc);
    block_reduce_store<kNSum, true>(acc, partials + (size_t)blockIdx.x * (kNSum + 1));
    const unsigned G0 = gridDim.x * kReduceThreads;  // (write-through, see below)
    for (unsigned i = blockIdx.x * kReduceThreads + threadIdx.x; i < 2u * kWinBins; i += G0)
      __hip_atomic_store(&whist[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!last_block_arrives(&ctl->t[2])) return;
    publish_result(partials, res, seq, sig0, med0, scal->nan_flag, failed ? 2 : 0);
    return;
  }
  constexpr int PM = kWinCapMed / kReduceThreads, PR = kWinCapRing / kReduceThreads;
  static_assert(kWinCapMed % kReduceThreads == 0 && kWinCapRing % kReduceThreads == 0, "candidates per thread");
  const unsigned tid = threadIdx.x;
#ifdef ICP_WIN_DEBUG
  long long stamp[8];
  int ns_ = 0;
#define STAMP() stamp[ns_++] = wall_clock64()
#else
#define STAMP()
#endif
  STAMP();
  // every global load of the prologue is issued before the first use: one round trip
  double vm[2][PM], vr[2][PR];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
#pragma unroll
    for (int u = 0; u < PM; ++u) vm[d][u] = wmed[(size_t)d * kWinCapMed + tid + u * kReduceThreads];
#pragma unroll
    for (int u = 0; u < PR; ++u) vr[d][u] = wring[(size_t)d * kWinCapRing + tid + u * kReduceThreads];
  }
  const unsigned got[4] = {st->list_cnt[0][0], st->list_cnt[1][0], st->list_cnt[2][0], st->list_cnt[3][0]};
  const unsigned em[2] = {st->med_cnt[0], st->med_cnt[1]}, er[2] = {st->ring_cnt[0], st->ring_cnt[1]};
  const unsigned mbase[2] = {st->med_base[0], st->med_base[1]}, inner[2] = {st->ring_inner[0], st->ring_inner[1]};
  const double m_lo[2] = {st->med_lo[0], st->med_lo[1]}, m_hi[2] = {st->med_hi[0], st->med_hi[1]};
  const double r_lo[2] = {st->ring_lo[0], st->ring_lo[1]}, r_hi[2] = {st->ring_hi[0], st->ring_hi[1]};
  // (the appended counts are cross-checked against the histogram: a mismatch is a miss)
  bool fail = st->fail != 0 || got[0] != em[0] || got[1] != em[1] || got[2] != er[0] || got[3] != er[1];
#ifdef ICP_WIN_DEBUG
  if (blockIdx.x == 0 && tid == 0 && fail)
    printf("[A] fail: st %u got %u %u %u %u want %u %u %u %u\n", st->fail, got[0], got[1], got[2], got[3], em[0], em[1], er[0], er[1]);
#endif
  STAMP();
  const unsigned klo = (n_total - 1) / 2, khi = n_total / 2;
  double med[2] = {0., 0.}, sig[2] = {0., 0.};
  if (!fail) {
    unsigned long long key[2][2];
    const long long mlo[2] = {(long long)klo - mbase[0], (long long)klo - mbase[1]};
    const long long mhi[2] = {(long long)khi - mbase[0], (long long)khi - mbase[1]};
    select_n<2, PM>(vm, em, m_lo, m_hi, mlo, mhi, key, fail);
    STAMP();
    if (!fail) {
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        med[d] = middle_of(n_total, key[d][0], key[d][1]);
#pragma unroll
        for (int u = 0; u < PR; ++u) vr[d][u] = fabs(vr[d][u] - med[d]);  // src/stats.rs:35
      }
      const long long dlo[2] = {(long long)klo - inner[0], (long long)klo - inner[1]};
      const long long dhi[2] = {(long long)khi - inner[0], (long long)khi - inner[1]};
      select_n<2, PR>(vr, er, r_lo, r_hi, dlo, dhi, key, fail);
      STAMP();
      if (!fail) {
        sig[0] = ICP_PPF34 * middle_of(n_total, key[0][0], key[0][1]);  // src/stats.rs:42-46
        sig[1] = ICP_PPF34 * middle_of(n_total, key[1][0], key[1][1]);
      }
    }
  }
  double acc[kNSum];
#pragma unroll
  for (int k = 0; k < kNSum; ++k) acc[k] = 0.;
  if (!fail) accumulate_points<kWinAccBatch>(a, rx, ry, n, T, acc);
  STAMP();
  block_reduce_store<kNSum, true>(acc, partials + (size_t)blockIdx.x * (kNSum + 1));
  // the histograms of the next evaluation start from zero (nobody reads them in this launch).
  // Write-through stores: the next evaluation may run on the handle's other stream, handed over
  // by the host as soon as it sees this kernel's result -- i.e. possibly before this kernel's
  // end-of-kernel write-back, so nothing it must see may linger in an XCD's L2.
  const unsigned G = gridDim.x * kReduceThreads;
  for (unsigned i = blockIdx.x * kReduceThreads + threadIdx.x; i < 2u * kWinBins; i += G)
    __hip_atomic_store(&whist[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  STAMP();
#ifdef ICP_WIN_DEBUG
  if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 200) && seq % 8 == 2)
    printf("[A blk %d] loads %lld sel1 %lld sel2 %lld acc %lld red %lld (x10ns)\n", blockIdx.x, stamp[1] - stamp[0],
           stamp[2] - stamp[1], stamp[3] - stamp[2], stamp[4] - stamp[3], stamp[5] - stamp[4]);
#endif

  if (!PUBLISH) {
    if (blockIdx.x == 0 && tid == 0) {  // every rank selects the same statistics from the same candidates
      scal->median[0] = med[0];
      scal->median[1] = med[1];
      scal->sigma[0] = sig[0];
      scal->sigma[1] = sig[1];
      scal->overflow = fail ? 2 : 0;
    }
    return;
  }
  if (!last_block_arrives(&ctl->t[2])) return;
  publish_result(partials, res, seq, sig, med, scal->nan_flag, fail ? 2 : 0);
}
#endif  // ICP_EXPERIMENTS

// the sharded evaluation (shard.hip) launches this one from another translation unit
template __global__ void k_win_compact<false>(const double *__restrict__, const double *__restrict__, unsigned, unsigned,
                                              WinParams, const uint32_t *__restrict__, WinState *, double *, double *,
                                              const unsigned *__restrict__, unsigned);

// ---- the last stage of a sharded evaluation (shard.hip has the design) ---------------------------------
// One workgroup: the candidates of every rank, read where the exchange left them (thread t takes elements t,
// t + 512, ... of the concatenation in rank order -- any order would do), the exact statistics from them, the
// block sums of every rank placed in block order and folded like one GPU's, the result released to the host.
__global__ __launch_bounds__(kReduceThreads) void k_shard_finish(ShardPtrs srcs, int world, unsigned n_total,
                                                                 int blocks_total, const WinState *st,
                                                                 double *ordered, uint32_t *whist, GnResult *res,
                                                                 unsigned seq, uint32_t *h_counts) {
  constexpr int PM = kWinCapMed / kReduceThreads, PR = kWinCapRing / kReduceThreads;
  constexpr int W = kNSum + 1;
  __shared__ unsigned s_base[kShardMaxWorld + 1][4];
  __shared__ unsigned s_fail;
  const unsigned tid = threadIdx.x;
#ifdef ICP_WIN_DEBUG
  long long sst[8];
  sst[0] = wall_clock64();
#endif
  const uint32_t *status = whist + 2 * kWinBins;
  if (tid < (unsigned)kShardStatusWords) res->status[tid] = status[tid];  // (ahead of the release of seq below)
  if (tid == 0) {
    unsigned base[4] = {0, 0, 0, 0}, fail = 0;
    for (int q = 0; q < world; ++q) {
      const ShardCandHeader *hq = reinterpret_cast<const ShardCandHeader *>(srcs.p[q]);
      fail |= hq->fail;
      for (int k = 0; k < 4; ++k) {
        s_base[q][k] = base[k];
        base[k] += hq->cnt[k];
      }
    }
    for (int k = 0; k < 4; ++k) s_base[world][k] = base[k];
    s_fail = fail;
  }
  // the block sums into block order (rank r owns blocks [B r / world, B (r + 1) / world)), the flags of every rank
  const int rows = (kTreeMaxBlocks + world - 1) / world + 1;
  const size_t part_off = sizeof(ShardCandHeader) + (size_t)(2 * kWinCapMed + 2 * kWinCapRing) * sizeof(double);
  for (int b = tid; b < blocks_total; b += kReduceThreads) {
    // the rank whose range holds b (32-bit arithmetic: blocks_total <= 4096, world <= 16)
    unsigned r = ((unsigned)(b + 1) * (unsigned)world - 1u) / (unsigned)blocks_total;
    while ((unsigned)blocks_total * r / (unsigned)world > (unsigned)b) --r;
    while ((unsigned)blocks_total * (r + 1u) / (unsigned)world <= (unsigned)b) ++r;
    const int b0 = (int)((unsigned)blocks_total * r / (unsigned)world);
    const double *pr = reinterpret_cast<const double *>(srcs.p[r] + part_off) + (size_t)(b - b0) * W;
    double v[W];
#pragma unroll
    for (int k = 0; k < W; ++k) v[k] = pr[k];  // (every load in flight before the first store)
#pragma unroll
    for (int k = 0; k < W; ++k)
      __hip_atomic_store(&ordered[(size_t)b * W + k], v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  int nan_flag = 0;
  for (int r = 0; r < world; ++r)
    nan_flag |= reinterpret_cast<const double *>(srcs.p[r] + part_off)[(size_t)(rows - 1) * W] != 0.;
  __syncthreads();
#ifdef ICP_WIN_DEBUG
  sst[1] = wall_clock64();
#endif
  bool fail = s_fail != 0;
  const unsigned got[4] = {s_base[world][0], s_base[world][1], s_base[world][2], s_base[world][3]};
  // (the appended counts are cross-checked against the histogram: a mismatch is a miss)
  fail = fail || got[0] != st->med_cnt[0] || got[1] != st->med_cnt[1] || got[2] != st->ring_cnt[0] ||
         got[3] != st->ring_cnt[1] || got[0] > (unsigned)kWinCapMed || got[1] > (unsigned)kWinCapMed ||
         got[2] > (unsigned)kWinCapRing || got[3] > (unsigned)kWinCapRing;
  double med[2] = {0., 0.}, sig[2] = {0., 0.};
  if (!fail) {
    // where element e of list k (0, 1: median candidates x, y; 2, 3: ring x, y) of the concatenation lies; every
    // address first, then every load: one round trip instead of twenty
    auto where = [&](int k, unsigned e) -> const double * {
      const double *body0 = reinterpret_cast<const double *>(srcs.p[0] + sizeof(ShardCandHeader));
      if (e >= got[k]) return body0;  // (not a candidate: any readable address)
      int q = 0;
      while (q + 1 < world && s_base[q + 1][k] <= e) ++q;
      const double *body = reinterpret_cast<const double *>(srcs.p[q] + sizeof(ShardCandHeader));
      const size_t off = k < 2 ? (size_t)k * kWinCapMed : (size_t)2 * kWinCapMed + (size_t)(k - 2) * kWinCapRing;
      return body + off + (e - s_base[q][k]);
    };
    const double *pm[2][PM], *pr[2][PR];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
#pragma unroll
      for (int u = 0; u < PM; ++u) pm[d][u] = where(d, tid + u * kReduceThreads);
#pragma unroll
      for (int u = 0; u < PR; ++u) pr[d][u] = where(2 + d, tid + u * kReduceThreads);
    }
    double vm[2][PM], vr[2][PR];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
#pragma unroll
      for (int u = 0; u < PM; ++u) vm[d][u] = *pm[d][u];
#pragma unroll
      for (int u = 0; u < PR; ++u) vr[d][u] = *pr[d][u];
    }
    const unsigned klo = (n_total - 1) / 2, khi = n_total / 2;
    const unsigned em[2] = {got[0], got[1]}, er[2] = {got[2], got[3]};
    const double m_lo[2] = {st->med_lo[0], st->med_lo[1]}, m_hi[2] = {st->med_hi[0], st->med_hi[1]};
    const double r_lo[2] = {st->ring_lo[0], st->ring_lo[1]}, r_hi[2] = {st->ring_hi[0], st->ring_hi[1]};
    const long long mlo[2] = {(long long)klo - st->med_base[0], (long long)klo - st->med_base[1]};
    const long long mhi[2] = {(long long)khi - st->med_base[0], (long long)khi - st->med_base[1]};
    unsigned long long key[2][2];
#ifdef ICP_WIN_DEBUG
    sst[2] = wall_clock64();
#endif
    select_n<2, PM>(vm, em, m_lo, m_hi, mlo, mhi, key, fail);
#ifdef ICP_WIN_DEBUG
    sst[3] = wall_clock64();
#endif
    if (!fail) {
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        med[d] = middle_of(n_total, key[d][0], key[d][1]);
#pragma unroll
        for (int u = 0; u < PR; ++u) vr[d][u] = fabs(vr[d][u] - med[d]);  // src/stats.rs:35
      }
      const long long dlo[2] = {(long long)klo - st->ring_inner[0], (long long)klo - st->ring_inner[1]};
      const long long dhi[2] = {(long long)khi - st->ring_inner[0], (long long)khi - st->ring_inner[1]};
      select_n<2, PR>(vr, er, r_lo, r_hi, dlo, dhi, key, fail);
      if (!fail) {
        sig[0] = ICP_PPF34 * middle_of(n_total, key[0][0], key[0][1]);  // src/stats.rs:42-46
        sig[1] = ICP_PPF34 * middle_of(n_total, key[1][0], key[1][1]);
      } else {
        med[0] = med[1] = 0.;
      }
    }
  }
  // a window that missed: its (global, exact) counts place the next attempt's -- to the host, by the wave that
  // releases the result (pinned memory; only on this path, so no copy per evaluation)
  if (fail && h_counts && tid < 64)
    for (unsigned i = tid; i < 2u * kWinBins; i += 64) h_counts[i] = whist[i];
  __syncthreads();
  // the histograms of the next evaluation start from zero (this rank's copy of the ranks' sum; the status words
  // behind them are rewritten by every hist stage)
  for (unsigned i = tid; i < 2u * kWinBins; i += kReduceThreads)
    __hip_atomic_store(&whist[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef ICP_WIN_DEBUG
  sst[4] = wall_clock64();
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifdef ICP_WIN_DEBUG
  sst[5] = wall_clock64();
#endif
  publish_result(ordered, res, seq, sig, med, nan_flag, fail ? 2 : 0, blocks_total);
#ifdef ICP_WIN_DEBUG
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  sst[6] = wall_clock64();
  if (tid == 0 && seq % 16 == 3)
    printf("[SF] gather %lld fetch %lld sel1 %lld sel2+zero %lld drain %lld publish %lld (x10ns)\n", sst[1] - sst[0], sst[2] - sst[1],
           sst[3] - sst[2], sst[4] - sst[3], sst[5] - sst[4], sst[6] - sst[5]);
#endif
}

// ---- host ---------------------------------------------------------------------------
bool window_usable(const icp_handle *h, size_t n, WinParams *P, int kind, bool any_n, double f_override) {
  static const bool off = exp_env("ICP_GN_NO_WIN") != nullptr;
  const Workspace &w = h->ws;
  // (any_n: the sharded evaluation, which refines a window that missed from that attempt's own counts
  // instead of giving up -- api.hip, shard_finish_common -- and so serves any number of points)
  if (off || n < kWinMinN || (n > kWinMaxN && !any_n)) return false;
  // the evaluation's own kind first (common.hpp, Workspace::win_kind), else the most recent evaluation
  const bool own = Workspace::kind_has_slot(kind) && w.win_kind[kind].valid;
  if (!own && !w.win_valid) return false;
  const double *p_med = own ? w.win_kind[kind].med : w.win_med;
  const double *p_sigma = own ? w.win_kind[kind].sigma : w.win_sigma;
  double f = window_half_width(n, own ? w.win_kind[kind].wide : w.win_wide);
  if (f_override > 0.) f = f_override < 0.2 ? f_override : 0.2;
  return make_window(p_med, p_sigma, f, P);
}

// half-width of the fine windows in sigmas: the prediction may be off by about that much (4 x after a miss)
double window_half_width(size_t n, bool wide) {
  static const double hw_env = exp_env("ICP_WIN_HW") ? atof(exp_env("ICP_WIN_HW")) : 0.;
  // Clouds of up to 2^17 points (512 per workgroup of the tree): even the widest windows leave a handful of candidates in
  // a fine bin and a few hundred members for a workgroup to stage, so narrow windows save nothing there -- and they cost
  // a fresh handle (one per frame in examples/scan3d.rs) two evaluations through the seven-launch pipeline, when the first
  // evaluations of a kind are predicted from another kind's statistics and miss (0.18 ms of a 1.28-ms frame:
  // profiles/r05_frame_trace_fresh.txt).
  if (hw_env <= 0. && n <= ((size_t)1 << 17)) return 0.2;
  const double hw_sigmas = hw_env > 0. ? hw_env : 0.05;
  double f = hw_sigmas * (wide ? 4. : 1.);
  if (n > 1000000) f *= 1e6 / (double)n;  // candidates per fine bin grow with n
  return f > 0.2 ? 0.2 : f;               // the windows must not overlap (MAD = 0.6745 sigma)
}

bool make_window(const double med[2], const double sigma[2], double f, WinParams *P) { return make_window_hd(med, sigma, f, P); }

// ---- filed candidates: host side ------------------------------------------------------
static unsigned tree_blocks(size_t n) {
  int blocks, threads;
  reduce_geometry(n, &blocks, &threads);
  return (unsigned)blocks;
}
// Can a workgroup stage its fine-window members?  Their expected number under a bell is 1.04 (x[1] - x[0]) / sigma of
// its points per dimension; a factor two to spare.  (Wider windows -- after a miss -- and handles whose files were not
// usable recently take the second pass over the points: Workspace::bkt_off.)
// (bkt_fits: one handle, which has segments for kReduceMaxBlocks workgroups -- the whole tree up to 2^20 pairs;
// bkt_fits_rank: a rank of a sharded registration, which files its own share -- at most that many blocks -- of a tree
// that may be larger)
bool bkt_fits(size_t n, const WinParams &P) { return tree_blocks(n) <= (unsigned)kReduceMaxBlocks && bkt_fits_rank(n, P); }
bool bkt_fits_rank(size_t n, const WinParams &P) {
  double frac = 0.;
  for (int d = 0; d < 2; ++d) {
    const WinDim &D = P.d[d];
    const double mad = 0.5 * ((D.x[4] + D.x[5]) - (D.x[2] + D.x[3]));
    frac += 1.04 * (D.x[1] - D.x[0]) / (ICP_PPF34 * mad);
  }
  return 2. * frac * ((double)n / tree_blocks(n)) <= (double)kBktStage;
}
static HistBktArgs bkt_hist_args(GnCtx &c, const double2 *a, const double2 *b, const Pose &T, unsigned n, const WinParams &P) {
  HistBktArgs A;
  A.a = a;
  A.b = b;
  A.T = T;
  A.n = n;
  A.P = P;
  A.whist = c.d_whist;
  A.st = c.d_wstate;
  A.scal = c.d_scal;
  A.partials = c.d_partials;
  A.seg_all = c.d_bkt;
  A.dir_all = c.d_bkt_dir;
  return A;
}
static PickArgs bkt_pick_args(GnCtx &c, unsigned n, const WinParams &P, bool ahead_on, const Pose &ahead_outer,
                              AheadPose *d_ahead = nullptr) {
  PickArgs A;
  A.n = n;
  A.P = P;
  A.whist = c.d_whist;
  A.st = c.d_wstate;
  A.seg_all = c.d_bkt;
  A.dir_all = c.d_bkt_dir;
  A.segments = (int)tree_blocks(n);
  A.scal = c.d_scal;
  A.partials = c.d_partials;
  A.sum_blocks = A.segments;
  A.res = c.h_res;
  A.seq = ++c.seq;
  A.ahead = ahead_on ? d_ahead : nullptr;
  A.outer = ahead_on ? ahead_outer : transform_identity();
  return A;
}

// Two evaluations in two launches on ONE stream (icp_estimate_device, a bet in flight): `first` -- the next outer
// iteration's first evaluation, on pairs a1 / b1 at the identity -- and `second`, the deciding evaluation of the current
// iteration (pairs a2 / b2 at T2).  Both results are released by the second launch (second.bkt_pair_launched says so:
// wgn_step only waits for it).
hipError_t launch_bkt_pair(icp_handle *h, hipStream_t s, GnCtx &first, const double *a1, const double *b1, const WinParams &P1,
                           bool ahead_on, const Pose &ahead_outer, GnCtx &second, const double *a2, const double *b2,
                           const Pose &T2, const WinParams &P2, size_t n_) {
  Workspace &w = h->ws;
  const unsigned n = (unsigned)n_, blocks = tree_blocks(n_);
  if (blocks > (unsigned)kReduceMaxBlocks) return hipErrorInvalidValue;  // (a context has segments for that many workgroups)
  w.bkt_evals += 2;
  hipLaunchKernelGGL(k_win_hist_sums_bkt2, dim3(2 * blocks), dim3(kWinThreads), 0, s,
                     bkt_hist_args(first, (const double2 *)a1, (const double2 *)b1, transform_identity(), n, P1),
                     bkt_hist_args(second, (const double2 *)a2, (const double2 *)b2, T2, n, P2));
  hipLaunchKernelGGL(k_win_pick2, dim3(2), dim3(kReduceThreads), 0, s,
                     bkt_pick_args(first, n, P1, ahead_on, ahead_outer, w.d_ahead),
                     bkt_pick_args(second, n, P2, false, transform_identity()));
  second.bkt_pair_launched = true;
  return hipGetLastError();
}

// ---- the pipelined sharded evaluation: host side (pipe.hip drives it) -----------------------------------------------------
// One or two evaluations (ev[0], ev[1]) on the ranks rk[0 .. nranks) of ONE device: each rank's first launch over its own
// tree blocks on its own stream, then ONE finishing launch for all of them (k_win_pick_shard: a workgroup per rank and
// evaluation) on rk[0]'s stream -- ranks that share a device share that stream (icp_multi), a rank with a device of
// its own is a call of its own.  gen0: generation of ev[0]'s exchange (ev[1]: gen0 + 1), the same on every rank.
hipError_t launch_shard_evals(const ShardPickRank *rk, int nranks, int world, int B, size_t n_total, unsigned gen0,
                              const ShardPickEval *ev, int nevals) {
  if (nranks < 1 || nranks > kPipeFuse || nevals < 1 || nevals > 2 || world < 1 || world > kShardMaxWorld || B < 1 || B > kTreeMaxBlocks)
    return hipErrorInvalidValue;
  for (int j = 0; j < nranks; ++j)  // (a rank's segments, block sums and inbox rows)
    if (rk[j].nbl < 1 || rk[j].nbl > kReduceMaxBlocks || rk[j].b0 < 0 || rk[j].b0 + rk[j].nbl > B || rk[j].rank < 0 || rk[j].rank >= world ||
        !rk[j].h->ws.d_loop_inbox)
      return hipErrorInvalidValue;
  PickShardLaunch L = {};
  L.world = world;
  L.B = B;
  L.nranks = nranks;
  L.nevals = nevals;
  const Workspace &w0 = rk[0].h->ws;
  for (int q = 0; q < kShardMaxWorld; ++q) L.inbox[q] = reinterpret_cast<LoopInbox *>(w0.loop_peers[q < world ? q : 0]);
  for (int j = 0; j < nranks; ++j) {
    icp_handle *h = rk[j].h;
    Workspace &w = h->ws;
    const unsigned nl = (unsigned)rk[j].n_local;
    GnCtx *ctx[2] = {nullptr, nullptr};
    HistBktArgs HA[2];
    for (int e = 0; e < nevals; ++e) {
      ctx[e] = ev[e].alt_ctx ? &w.alt : static_cast<GnCtx *>(&w);
      HA[e] = bkt_hist_args(*ctx[e], (const double2 *)rk[j].a[e], (const double2 *)rk[j].b[e], ev[e].T, nl, ev[e].P);
      PickRank &K = L.ev[e].rk[j];
      K.whist = ctx[e]->d_whist;
      K.st = ctx[e]->d_wstate;
      K.seg_all = ctx[e]->d_bkt;
      K.dir_all = ctx[e]->d_bkt_dir;
      K.scal = ctx[e]->d_scal;
      K.partials = ctx[e]->d_partials;
      K.res = ctx[e]->h_res;
      K.ahead = w.d_ahead;
      K.rank = rk[j].rank;
      K.b0 = rk[j].b0;
      K.nbl = rk[j].nbl;
      K.seq = ++ctx[e]->seq;
    }
    w.bkt_evals += (unsigned)nevals;
    if (nevals == 2)
      hipLaunchKernelGGL(k_win_hist_sums_bkt2, dim3(2u * (unsigned)rk[j].nbl), dim3(kWinThreads), 0, h->stream, HA[0], HA[1]);
    else
      hipLaunchKernelGGL(k_win_hist_sums_bkt, dim3((unsigned)rk[j].nbl), dim3(kWinThreads), 0, h->stream, HA[0]);
  }
  for (int e = 0; e < nevals; ++e) {
    L.ev[e].n = (unsigned)n_total;
    L.ev[e].gen = gen0 + (unsigned)e;
    L.ev[e].ahead_on = ev[e].ahead_on ? 1 : 0;
    L.ev[e].P = ev[e].P;
    L.ev[e].outer = ev[e].ahead_on ? ev[e].outer : transform_identity();
  }
  hipLaunchKernelGGL(k_win_pick_shard, dim3((unsigned)nevals, (unsigned)nranks), dim3(kReduceThreads), 0, rk[0].h->stream, L);
  return hipGetLastError();
}

hipError_t launch_weighted_gn_win(icp_handle *h, const double *d_a, const double *d_b, size_t n_, const Pose &T,
                                  const WinParams &P) {
  Workspace &w = h->ws;
  const unsigned n = (unsigned)n_;
  const unsigned per = kWinThreads * kWinBatch;
  unsigned hb = (n + per - 1) / per;
  if (hb > (unsigned)kWinBlocks) hb = kWinBlocks;
  const double2 *a = (const double2 *)d_a, *b = (const double2 *)d_b;
  hipStream_t s = h->stream;
  // Two launches either way.  Round 5: the first files the candidates, the second is ONE workgroup (k_win_hist_sums_bkt
  // / k_win_pick) wherever bkt_fits says so; windows too wide for that (after a miss) and handles whose files were not
  // usable recently take the second pass over the points (k_win_hist_sums / k_win_finish, round 3).
#ifdef ICP_EXPERIMENTS
  // ICP_WIN_BKT: 0 = never, 1 = wherever it fits (default), 2 = not beside a search, 3 = only there.  ICP_WIN_FUSE_MODE /
  // ICP_WIN_NO_FUSE / ICP_WIN_NO_CORESIDENT: the four-launch forms of rounds 1-3 (experiments build only).
  static const int bkt_mode = exp_env("ICP_WIN_BKT") ? atoi(exp_env("ICP_WIN_BKT")) : 1;
  const bool on_eval_stream = w.spec_stream && s == w.spec_stream;
  const bool beside_search = on_eval_stream && w.search_beside_eval;
  const bool bkt_here = bkt_mode == 1 || (bkt_mode == 2 && !beside_search) || (bkt_mode == 3 && beside_search);
#else
  constexpr bool bkt_here = true;
#endif
  if (w.bkt_off > 0) --w.bkt_off;
  else if (bkt_here && bkt_fits(n_, P)) {
    ++w.bkt_evals;
    const HistBktArgs HA = bkt_hist_args(w, a, b, T, n, P);
    hipLaunchKernelGGL(k_win_hist_sums_bkt, dim3(tree_blocks(n_)), dim3(kWinThreads), 0, s, HA);
    hipLaunchKernelGGL(k_win_pick, dim3(1), dim3(kReduceThreads), 0, s, bkt_pick_args(w, n, P, w.ahead_on, w.ahead_outer, w.d_ahead));
    return hipGetLastError();
  }
  int blocks, threads;
  reduce_geometry(n_, &blocks, &threads);
#ifdef ICP_EXPERIMENTS
  static const bool no_co = exp_env("ICP_WIN_NO_CORESIDENT") != nullptr;
  static const bool no_fuse = exp_env("ICP_WIN_NO_FUSE") != nullptr;
  static const int fuse_mode = exp_env("ICP_WIN_FUSE_MODE") ? atoi(exp_env("ICP_WIN_FUSE_MODE")) : 1;
  if (no_fuse || !(fuse_mode == 1 || (fuse_mode == 2 && !beside_search) || (fuse_mode == 3 && beside_search) ||
                   (fuse_mode == 4 && !on_eval_stream))) {
    hipLaunchKernelGGL(k_win_hist, dim3(hb), dim3(kWinThreads), 0, s, a, b, T, w.d_rx, w.d_ry, n, P, w.d_whist,
                       w.d_wstate, w.d_scal);
    hipLaunchKernelGGL(k_win_compact<false>, dim3(hb), dim3(kWinThreads), 0, s, (const double *)w.d_rx,
                       (const double *)w.d_ry, n, n, P, (const uint32_t *)w.d_whist, w.d_wstate, w.d_wmed,
                       w.d_wring, (const unsigned *)nullptr, 0u);
    if (!no_co) {
      hipLaunchKernelGGL(k_win_select, dim3(2), dim3(kReduceThreads), 0, s, n, w.d_wstate, (const double *)w.d_wmed,
                         (const double *)w.d_wring, w.d_scal);
      hipLaunchKernelGGL(k_win_accumulate<false>, dim3(blocks), dim3(threads), 0, s, a, (const double *)w.d_rx,
                         (const double *)w.d_ry, n, n, T, (const WinState *)w.d_wstate, (const double *)w.d_wmed,
                         (const double *)w.d_wring, w.d_scal, w.d_partials, w.d_whist, w.d_ctl, w.h_res, ++w.seq);
    } else {
      hipLaunchKernelGGL(k_win_accumulate<true>, dim3(blocks), dim3(threads), 0, s, a, (const double *)w.d_rx,
                         (const double *)w.d_ry, n, n, T, (const WinState *)w.d_wstate, (const double *)w.d_wmed,
                         (const double *)w.d_wring, w.d_scal, w.d_partials, w.d_whist, w.d_ctl, w.h_res, ++w.seq);
    }
    return hipGetLastError();
  }
#endif
  hipLaunchKernelGGL(k_win_hist_sums, dim3(blocks), dim3(threads), 0, s, a, b, T, w.d_rx, w.d_ry, n, P, w.d_whist,
                     w.d_wstate, w.d_scal, w.d_partials, -1);
  hipLaunchKernelGGL(k_win_finish<false>, dim3(hb), dim3(kWinThreads), 0, s, (const double *)w.d_rx,
                     (const double *)w.d_ry, n, P, w.d_whist, w.d_wstate, w.d_wmed, w.d_wring, w.d_scal,
                     (const double *)w.d_partials, blocks, w.d_ctl, w.h_res, ++w.seq, (const unsigned *)nullptr, 0u,
                     w.ahead_on ? w.d_ahead : (AheadPose *)nullptr, w.ahead_on ? w.ahead_outer : transform_identity());
  return hipGetLastError();
}

// ---- refined windows: n > kWinMaxN ---------------------------------------------------
// Beyond 4M points a window wide enough to survive a prediction error holds more candidates than
// the lists take, so the radix pipeline used to serve alone: nine launches that stream 192 B per
// point.  Two window passes stream 112: the exact statistics of a 256k-point strided sample centre
// a first histogram pass (fine windows of 0.02 sigma: eight standard errors of the sample
// median); its counts place the median and the MAD to within a fine bin (8e-5 sigma), and a second
// histogram pass over the stored residuals, with windows as narrow as the lists require, is then
// resolved by the usual C and A launches -- which verify everything by exact counts, so a bad
// sample can only cost a repeat with the radix pipeline, never a different result.
bool refine_applies(size_t n) {
  static const bool off = exp_env("ICP_GN_NO_WIN") != nullptr || getenv("ICP_GN_NO_REFINE") != nullptr;
  return !off && n > kWinMaxN && n / kRefineSample >= 1 && n < 0xffffffffull;
}

hipError_t launch_sample_pairs(icp_handle *h, const double *d_a, const double *d_b, size_t n) {
  Workspace &w = h->ws;
  hipError_t e;
  if (!w.d_sa) {
    if ((e = hipMalloc(&w.d_sa, kRefineSample * 2 * sizeof(double))) != hipSuccess) return e;
    if ((e = hipMalloc(&w.d_sb, kRefineSample * 2 * sizeof(double))) != hipSuccess) return e;
    if ((e = hipHostMalloc(&w.h_whist, (size_t)2 * kWinBins * sizeof(uint32_t), hipHostMallocDefault)) != hipSuccess) return e;
    if ((e = hipMalloc(&w.d_rlist, 2 * kRefineListCap * sizeof(double))) != hipSuccess) return e;
    if ((e = hipMalloc(&w.d_rlist_len, 2 * sizeof(unsigned))) != hipSuccess) return e;
  }
  const unsigned stride = (unsigned)(n / kRefineSample);
  hipLaunchKernelGGL(k_sample_pairs, dim3((unsigned)(kRefineSample / 256)), dim3(256), 0, h->stream, (const double2 *)d_a,
                     (const double2 *)d_b, stride, (unsigned)kRefineSample, (double2 *)w.d_sa, (double2 *)w.d_sb);
  return hipGetLastError();
}

// These launches run against HBM bandwidth, not launch latency: four workgroups per CU (one per CU
// keeps 32 KB of 8-byte loads in flight per CU: 3 TB/s measured for the second histogram pass and the
// compaction on 64M pairs; the flush of 4 x 4096 histogram words per CU is noise at this size)
static unsigned win_hist_blocks(unsigned n) {
  const unsigned per = kWinThreads * kWinBatch;
  unsigned hb = (n + per - 1) / per;
  return hb > 4u * (unsigned)kWinBlocks ? 4u * (unsigned)kWinBlocks : hb;
}

hipError_t launch_win_first_pass(icp_handle *h, const double *d_a, const double *d_b, size_t n_, const Pose &T,
                                 const WinParams &P1) {
  Workspace &w = h->ws;
  const unsigned n = (unsigned)n_;
  // residuals + histograms + the block sums of the reduction tree (they do not depend on the statistics: the pairs
  // are streamed ONCE per evaluation; until round 3 an accumulate pass read them again after the second pass)
  int blocks, threads;
  reduce_geometry(n_, &blocks, &threads);
  hipLaunchKernelGGL(k_win_hist_sums_deep, dim3(blocks), dim3(threads), 0, h->stream, (const double2 *)d_a,
                     (const double2 *)d_b, T, w.d_rx, w.d_ry, n, P1, w.d_whist, w.d_wstate, w.d_scal, w.d_partials);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = hipMemcpyAsync(w.h_whist, w.d_whist, (size_t)2 * kWinBins * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream);
  if (e != hipSuccess) return e;
  // the second pass starts from empty histograms and empty lists
  if ((e = hipMemsetAsync(w.d_rlist_len, 0, 2 * sizeof(unsigned), h->stream)) != hipSuccess) return e;
  return hipMemsetAsync(w.d_whist, 0, (size_t)2 * kWinBins * sizeof(uint32_t), h->stream);
}

// host twin of wedge(): value range [lo, hi) of regular bin j
static void bin_range(int j, const WinDim &w, double *lo, double *hi) {
  auto edge = [&](int k) -> double {
    if (k >= kWinBins - 1) return w.x[5];
    if (k >= kF2) return w.x[4] + (double)(k - kF2) / w.sf;
    if (k >= kC1) return w.x[3] + (double)(k - kC1) / w.sc;
    if (k >= kF1) return w.x[2] + (double)(k - kF1) / w.sf;
    if (k >= kC0) return w.x[1] + (double)(k - kC0) / w.sc;
    return w.x[0] + (double)(k - kF0) / w.sf;
  };
  *lo = edge(j);
  *hi = edge(j + 1);
}

// From the first pass' counts: the bin of the median -> its centre; bins taken outwards from there in
// the order of their distance until half of the points are in -> the MAD.  Windows for the second
// pass: as wide as the candidate lists allow (~256 points per fine bin), at least four first-pass
// bins.  Nothing here needs to be exact: the second pass is verified by counts like any window.
bool refine_window(const uint32_t *hist, size_t n, const WinParams &P1, WinParams *P2) {
  double med[2], sigma[2];
  for (int d = 0; d < 2; ++d) {
    const uint32_t *c = hist + (size_t)d * kWinBins;
    const WinDim &w = P1.d[d];
    const size_t half = (n - 1) / 2;
    size_t run = 0;
    int jm = -1;
    for (int j = 0; j < kWinBins; ++j) {
      if (run + c[j] > half) {
        jm = j;
        break;
      }
      run += c[j];
    }
    if (jm < 1 || jm > kWinBins - 2) return false;  // the median is outside the first pass' windows
    double lo, hi;
    bin_range(jm, w, &lo, &hi);
    const double m = 0.5 * (lo + hi);
    // sweep outwards: the next bin on the left or on the right, whichever centre is closer
    size_t in = c[jm];
    int l = jm - 1, r = jm + 1;
    double mad = 0.5 * (hi - lo);
    while (in <= n / 2) {
      double ll = 0., lh = 0., rl = 0., rh = 0.;
      if (l < 1 || r > kWinBins - 2) return false;  // the MAD reaches a catch-all bin
      bin_range(l, w, &ll, &lh);
      bin_range(r, w, &rl, &rh);
      const double dl = m - 0.5 * (ll + lh), dr = 0.5 * (rl + rh) - m;
      if (dl <= dr) {
        in += c[l--];
        mad = dl;
      } else {
        in += c[r++];
        mad = dr;
      }
    }
    med[d] = m;
    sigma[d] = ICP_PPF34 * mad;
    if (!(sigma[d] > 0.)) return false;
  }
  double f = 256. * kWinFine / (2. * 0.4 * (double)n);    // ~256 points per fine bin at the median
  const double f_min = 4. * (2. * 0.02 / kWinFine);       // four fine bins of the first pass
  if (f < f_min) f = f_min;
  if (f > 0.02) f = 0.02;
  return make_window(med, sigma, f, P2);
}

hipError_t launch_win_second_pass(icp_handle *h, const double *d_a, size_t n_, const Pose &T, const WinParams &P2) {
  Workspace &w = h->ws;
  const unsigned n = (unsigned)n_;
  const unsigned hb = win_hist_blocks(n);
  hipStream_t s = h->stream;
  // (a list longer than kRefineListCap -- a distribution far denser at its quartiles than a bell -- is a miss)
  double *lx = w.d_rlist, *ly = w.d_rlist + kRefineListCap;
  hipLaunchKernelGGL(k_win_rehist, dim3(hb), dim3(kWinThreads), 0, s, (const double *)w.d_rx, (const double *)w.d_ry, n,
                     P2, w.d_whist, w.d_wstate, lx, ly, (unsigned)kRefineListCap, w.d_rlist_len);
  const unsigned cb = hb < (unsigned)kWinBlocks ? hb : (unsigned)kWinBlocks;  // a few hundred thousand values: one workgroup per CU
  int blocks, threads;
  reduce_geometry(n_, &blocks, &threads);
  // candidates out of the lists; the last workgroup selects the statistics and folds the first pass' block sums
  hipLaunchKernelGGL(k_win_finish<true>, dim3(cb), dim3(kWinThreads), 0, s, (const double *)lx, (const double *)ly, n, P2,
                     w.d_whist, w.d_wstate, w.d_wmed, w.d_wring, w.d_scal, (const double *)w.d_partials, blocks, w.d_ctl,
                     w.h_res, ++w.seq, (const unsigned *)w.d_rlist_len, (unsigned)kRefineListCap, (AheadPose *)nullptr,
                     transform_identity());
  return hipGetLastError();
}

}  // namespace icp
