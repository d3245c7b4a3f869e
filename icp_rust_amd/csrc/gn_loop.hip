// The whole inner loop of estimate_transform (src/lib.rs:59-84) in ONE launch, for pair sets that fit the register
// files of the chip (n <= 8 x 256 x 512 = 1 048 576).
//
// The pairs (a_i, b_i) of an outer iteration do not change while the inner loop runs -- only the pose does
// (src/lib.rs:66-82) -- and 1M pairs are 32 MB against 128 MB of vector registers.  So the launch has the geometry of
// the reduction tree (reduce_geometry(n) workgroups of 512), every thread LOADS ITS POINTS ONCE
// (thread g of the tree folds points g, g + G, ...: at most eight) and the up-to-200 evaluations of
// weighted_gauss_newton_update (src/lib.rs:218-261) + huber_error (:45-50) run out of registers:
//
//   A  residuals of the thread's points -> window histograms (LDS, flushed with integer atomics) + the thread's 19
//      running sums -> block sum                                                       [grid barrier]
//   B  every workgroup reads the global counts, derives the bins of the order statistics (gn_win_device.hpp: the same
//      bracket as k_win_finish) and appends its candidates -- from registers again         [grid barrier]
//   C  every workgroup loads the candidate lists and the block sums, selects the four exact order statistics
//      (src/stats.rs:11-47), folds the block sums in the tree's order, applies 1 / sigma, solves the 3 x 3 system
//      (src/linalg.rs:3-29), takes the reference's two break tests (src/lib.rs:71-78) and composes the pose (:81) --
//      redundantly and deterministically, so all workgroups hold the same bits and nothing is broadcast.
//
// Two grid barriers per evaluation and no kernel boundary, no host round trip, no re-read of the pairs or of stored
// residuals.  Every sum is folded in the order of the launch-per-stage pipelines (k_win_hist_sums / k_win_finish), the
// order statistics are the exact ones: results are bit-identical to them and to the oracle's tree variant.
//
// What the kernel cannot do it hands back: a window that missed (status 1), a rotation beyond the restated range of
// sin / cos (status 1 as well: the host repeats that evaluation with its own pipelines), a NaN residual (status 3),
// a grid barrier that timed out because the launch was not fully resident (status 5; nothing else may occupy the CUs
// for longer than kLoopTimeoutTicks).  The state handed back is the state BEFORE the evaluation that was not served.
#include "common.hpp"
#include "gn_device.hpp"
#include "gn_win_device.hpp"
#include "gn_loop.hpp"

#include <algorithm>

namespace icp {

namespace {

// A grid barrier of a launch whose workgroups are all resident ends within microseconds.  Where some are not (another
// tenant holds CUs), waiting is only worth while they keep ARRIVING: the wait gives up 2 ms after the last arrival it
// saw (round 4 waited 250 ms flat: a frame that takes a millisecond took a quarter of a second beside a co-tenant).
constexpr long long kLoopTimeoutTicks = 200000;  // wall_clock64 runs at 100 MHz: 2 ms without progress
// ... between ranks: processes that share ONE GPU (tests) are not always scheduled side by side at once -- the queue of
// the second process may wait for a time slice -- so a rank waits longer for its peers than a launch for its own blocks
constexpr long long kShardTimeoutTicks = 300000000;  // 3 s

__device__ __forceinline__ double to_sgpr(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll));
  const int hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ Pose pose_sgpr(const Pose &T) {
  Pose o;
  o.r00 = to_sgpr(T.r00);
  o.r10 = to_sgpr(T.r10);
  o.r01 = to_sgpr(T.r01);
  o.r11 = to_sgpr(T.r11);
  o.tx = to_sgpr(T.tx);
  o.ty = to_sgpr(T.ty);
  return o;
}
__device__ __forceinline__ WinDim windim_sgpr(const WinDim &w) {
  WinDim o;
#pragma unroll
  for (int k = 0; k < 6; ++k) o.x[k] = to_sgpr(w.x[k]);
  o.sf = to_sgpr(w.sf);
  o.sc = to_sgpr(w.sc);
  return o;
}

__device__ __forceinline__ unsigned ld_u32(const unsigned *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_u32(unsigned *p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_f64(double *p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_f64(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ unsigned long long ld_u64(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Grid barriers WITHOUT read-modify-writes.  A counting barrier costs two dependent returning atomics (shard, top), a
// release store and a poll: four trips to the memory side, 2.5 - 3 us measured.  Here every workgroup owns one 64-bit
// FLAG WORD per barrier kind and stores (generation | payload) into it once its data is out (stores drained first);
// wave 0 of every workgroup polls the flag words of all workgroups, four per lane: one store and one poll.  The words
// a lane saw last are handed back -- the second barrier of an evaluation carries each workgroup's candidate counts in
// its payload, so the poll that ends the barrier is also the gather of the counts.
// Returns false when the wait timed out or another workgroup raised `abort` (uniform over the workgroup).
constexpr unsigned long long kFlagGenMask = 0xffffull;
__device__ __forceinline__ bool flag_barrier(unsigned long long *flags, LoopCtl *c, unsigned gen, unsigned long long payload,
                                             unsigned long long *s_seen /* LDS, kReduceMaxBlocks words, or null */) {
  __shared__ int s_ok;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x < 64) {
    const unsigned nb = gridDim.x, lane = threadIdx.x;
    if (lane == 0)
      __hip_atomic_store(&flags[blockIdx.x], (unsigned long long)gen | (payload << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long seen[4] = {0, 0, 0, 0};
    int ok = 1;
    long long t0 = wall_clock64();
    unsigned arrived_before = 0;
    for (;;) {
      bool all = true;
      unsigned arrived = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned b = lane + 64u * j;
        bool here = true;
        if (b < nb) {
          seen[j] = ld_u64(&flags[b]);
          here = (seen[j] & kFlagGenMask) >= (unsigned long long)gen;
          all = all && here;
        }
        arrived += (unsigned)__popcll(__ballot(here));
      }
      if (__all(all)) break;
      if (arrived > arrived_before) {  // (progress: the clock starts again)
        arrived_before = arrived;
        t0 = wall_clock64();
      }
      int stop = 0;
      if (lane == 0) {
        if (ld_u32(&c->abort[0]) != 0u) stop = 1;
        else if (wall_clock64() - t0 > kLoopTimeoutTicks) {
          st_u32(&c->abort[0], 1u);
          stop = 1;
        }
      }
      if (__builtin_amdgcn_readfirstlane(stop)) {
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (s_seen) {
#pragma unroll
      for (int j = 0; j < 4; ++j) s_seen[lane + 64u * j] = (lane + 64u * j < nb) ? seen[j] : 0ull;
    }
    if (lane == 0) s_ok = ok;
  }
  __syncthreads();
  return s_ok != 0;
}

// The last workgroup to leave the launch puts the control block and both histograms back into their all-zero rest
// state (nobody polls or reads them any more).
__device__ __forceinline__ void leave_launch(LoopCtl *c, uint32_t *whist) {
  __shared__ int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    s_last = __hip_atomic_fetch_add(&c->done[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  for (unsigned i = threadIdx.x; i < 4u * kWinBins; i += blockDim.x) st_u32(&whist[i], 0u);
  for (unsigned i = threadIdx.x; i < sizeof(LoopCtl) / sizeof(unsigned); i += blockDim.x)
    st_u32(reinterpret_cast<unsigned *>(c) + i, 0u);
}

struct LoopBins {  // what phase B derives from the global counts (identical in every workgroup)
  WinSel sel;
  unsigned mlo[2], mhi[2], a0[2], b1[2], i0[2], i1[2];
  bool fail;
};

// Phase B, first half: the global histogram (parity buffer `whist`) -> cumulative counts in LDS -> bins of the two
// middle ranks, bracket of the MAD (gn_win_device.hpp), exactly as win_compact_body<false, true> resolves them.
// SHARDED: the counts are the sum of `nsrc` histograms `stride` words apart (every rank's, pushed into this rank's
// inbox: system-scope loads).
template <bool SHARDED = false>
__device__ __forceinline__ void loop_resolve(const uint32_t *whist, unsigned n, const WinParams &P, uint32_t *cum,
                                             LoopBins &R, int nsrc = 1, size_t stride = 0) {
  __shared__ unsigned s_selu[2][4];
  __shared__ double s_seld[2][4];
  __shared__ unsigned s_wtot[2][16];
  __shared__ int s_rng[2][8];
  __shared__ int s_t[2][2];
  __shared__ int s_j[2][2];  // the bins of the two middle ranks, found by the threads that own them (gn_win.hip: win_resolve)
  constexpr int PER = kWinBins / kReduceThreads, NW = kReduceThreads / 64;
  static_assert(PER == 4, "four bins per thread");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned klo = (n - 1) / 2, khi = n / 2;  // src/stats.rs:18-27
  if (tid < 4) s_j[tid >> 1][tid & 1] = -1;
  unsigned v[2][PER], inc[2], tot[2];
  if (!SHARDED) {
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const unsigned long long *q = reinterpret_cast<const unsigned long long *>(whist + d * kWinBins) + 2 * tid;
      const unsigned long long x0 = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long x1 = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v[d][0] = (unsigned)x0;
      v[d][1] = (unsigned)(x0 >> 32);
      v[d][2] = (unsigned)x1;
      v[d][3] = (unsigned)(x1 >> 32);
    }
  } else {
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int i = 0; i < PER; ++i) v[d][i] = 0u;
    for (int sidx = 0; sidx < nsrc; ++sidx) {
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const unsigned long long *q =
            reinterpret_cast<const unsigned long long *>(whist + (size_t)sidx * stride + d * kWinBins) + 2 * tid;
        const unsigned long long x0 = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long x1 = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        v[d][0] += (unsigned)x0;
        v[d][1] += (unsigned)(x0 >> 32);
        v[d][2] += (unsigned)x1;
        v[d][3] += (unsigned)(x1 >> 32);
      }
    }
  }
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    tot[d] = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) tot[d] += v[d][i];
    unsigned s = tot[d];  // the wave's inclusive scan by DPP (row shifts, then the two row broadcasts): no LDS crossbar
    s = wave_scan_inclusive(s);
    inc[d] = s;
    if (lane == 63) s_wtot[d][wave] = s;
  }
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
      if (run <= klo && klo < run + v[d][i]) s_j[d][0] = PER * tid + i;
      if (run <= khi && khi < run + v[d][i]) s_j[d][1] = PER * tid + i;
      run += v[d][i];
    }
  }
  __syncthreads();
  WinGeom geo = {};
  if (wave < 4) {  // waves 0,1: t1 of x,y; waves 2,3: t2 of x,y
    const int d = wave & 1, role = wave >> 1;
    geo = window_geometry(cum + d * kWinBins, n, P.d[d], s_j[d][0], s_j[d][1]);
    const int t = geo.ok ? bracket_search(cum + d * kWinBins, n, P.d[d], geo, role) : -1;
    if (lane == 0) s_t[role][d] = t;
  }
  __syncthreads();
  if (wave < 2) {
    const int d = wave;
    WinRanges W = {};
    unsigned med_base = 0, med_cnt = 0, inner = 0, ring_cnt = 0;
    double range[4] = {0., 0., 0., 0.};
    unsigned counted = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) counted += s_wtot[d][w];
    const bool ok = counted == n && resolve_window(cum + d * kWinBins, n, P.d[d], geo, s_t[0][d], s_t[1][d], W, med_base,
                                                   med_cnt, inner, ring_cnt, range);
    if (lane == 0) {
      s_rng[d][0] = W.mlo;
      s_rng[d][1] = W.mhi;
      s_rng[d][2] = W.a0;
      s_rng[d][3] = W.b1;
      s_rng[d][4] = W.i0;
      s_rng[d][5] = W.i1;
      s_rng[d][6] = ok ? 0 : 1;
      s_selu[d][0] = med_base;
      s_selu[d][1] = med_cnt;
      s_selu[d][2] = inner;
      s_selu[d][3] = ring_cnt;
#pragma unroll
      for (int k = 0; k < 4; ++k) s_seld[d][k] = range[k];
    }
  }
  __syncthreads();
  R.fail = (s_rng[0][6] | s_rng[1][6]) != 0;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    R.sel.med_base[d] = s_selu[d][0];
    R.sel.med_cnt[d] = s_selu[d][1];
    R.sel.inner[d] = s_selu[d][2];
    R.sel.ring_cnt[d] = s_selu[d][3];
#pragma unroll
    for (int k = 0; k < 4; ++k) R.sel.range[d][k] = s_seld[d][k];
    R.mlo[d] = (unsigned)s_rng[d][0];
    R.mhi[d] = (unsigned)s_rng[d][1];
    R.a0[d] = (unsigned)s_rng[d][2];
    R.b1[d] = (unsigned)s_rng[d][3];
    R.i0[d] = (unsigned)s_rng[d][4];
    R.i1[d] = (unsigned)s_rng[d][5];
  }
}

#ifdef ICP_LOOP_PROFILE
#define LOOP_STAMP(slot)                                \
  do {                                                  \
    const long long now_ = wall_clock64();              \
    prof[slot] += now_ - t_last;                        \
    t_last = now_;                                      \
  } while (0)
#else
#define LOOP_STAMP(slot) ((void)0)
#endif

struct LoopLocal {  // per workgroup, LDS: the loop's state, replicated
  Pose Ti;
  double prev_error;
  double med[2], sig[2];
  double f;  // half-width (in sigmas) of the window the launch will centre on this evaluation's statistics
  WinParams P;
  unsigned applied;
  int done, status;
  int retried;  // the evaluation in hand is being repeated with the widest windows
};

// A window that missed: the same evaluation once more around the same centre with the widest windows the bin layout
// allows (0.2 sigma: it tolerates a prediction that is off by that much) -- 40 us inside the launch instead of handing
// the evaluation to the host's pipelines (three launches, two waits, a relaunch).  false: no such window.
__device__ __forceinline__ bool widen_window(WinParams *P) {
  double med[2], sig[2];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const WinDim &w = P->d[d];
    med[d] = 0.5 * (w.x[2] + w.x[3]);
    sig[d] = (0.5 * (w.x[4] + w.x[5]) - med[d]) * ICP_PPF34;
  }
  return make_window_hd(med, sig, 0.2, P);
}

// How wide the next window has to be: the statistics of consecutive evaluations of one inner loop move less and less
// (a converging Gauss-Newton iteration), and a window eight times the last observed move -- never wider than the
// host's default f_max -- holds a few dozen candidates instead of several hundred: both selections then rank their
// lists directly.  A window that turns out too narrow is a miss like any other (the host's pipelines serve that
// evaluation and re-centre).
__device__ __forceinline__ double next_half_width(const double (&med)[2], const double (&sig)[2], const double *pmed,
                                                  const double *psig, bool have_prev, double f_max) {
  if (!have_prev) return f_max;
  double shift = 0.;
#pragma unroll
  for (int d = 0; d < 2; ++d) shift = fmax(shift, (fabs(med[d] - pmed[d]) + fabs(sig[d] - psig[d])) / sig[d]);
#ifndef ICP_LOOP_FWIDTH
#define ICP_LOOP_FWIDTH 8.
#endif
  const double f = ICP_LOOP_FWIDTH * shift;
  return f != f || f > f_max ? f_max : (f < 0.004 ? 0.004 : f);
}

}  // namespace

// K = pairs per thread (at most kLoopMaxK): n <= K * gridDim.x * 512.  Dynamic LDS: K x 512 source points, then K x 512
// matched targets (16 bytes each: a conflict-free ds_read_b128 per lane and point).  The loops over a thread's points
// are ROLLED: unrolled eight-fold the kernel was 81 KB of code, more than the instruction cache two CUs share, and
// every phase of every evaluation waited for its own instructions.
__global__ __launch_bounds__(kReduceThreads) void k_gn_loop(LoopArgs A, unsigned K) {
  extern __shared__ double2 s_pts[];
  // phase A / B: histograms, cumulative counts; phase C: the selection's workspace
  __shared__ __align__(16) unsigned char s_work[sizeof(SelectLds<2>) > 2 * kWinBins * sizeof(uint32_t) ? sizeof(SelectLds<2>)
                                                                                                         : 2 * kWinBins * sizeof(uint32_t)];
  uint32_t *const s_bins = reinterpret_cast<uint32_t *>(s_work);
  SelectLds<2> &s_sel = *reinterpret_cast<SelectLds<2> *>(s_work);
  // Round 5: the points whose residual landed in a FINE bin in phase A, as (point of the thread << 10 | thread << 1 |
  // dimension) -- in the part of the selection's workspace the histograms leave free.  Every candidate of a window that
  // holds lies in a fine bin, so phase B lists its candidates from these ~800 entries instead of binning all 4 096
  // points of the workgroup a second time (6.6 of an evaluation's 43 us: profiles/r04_loop_phases.txt).
  constexpr unsigned kStageCap = (sizeof(s_work) - 2 * kWinBins * sizeof(uint32_t)) / sizeof(unsigned short);
  static_assert(kStageCap >= 4096 && kLoopMaxK <= 8, "staged members: 3 + 9 + 1 bits each");
  unsigned short *const s_mem = reinterpret_cast<unsigned short *>(s_work + 2 * kWinBins * sizeof(uint32_t));
  __shared__ unsigned s_nmem;
  __shared__ double s_med[2][kWinBlkMed], s_ring[2][kWinBlkRing];
  __shared__ unsigned s_cnt[4], s_base[4];
  __shared__ double s_tot[kNSum + 1], s_acc[kNAcc + 3];
  __shared__ double s_wv[kReduceThreads / 64][kNSum];  // the waves' sums of phase A (block_reduce_waves / _finish)
  __shared__ WinRegion s_tab[2][8];  // phase A's bins: wbin_tab (five rows per dimension, rewritten for every evaluation's windows)
  __shared__ WinSel s_ws;
  __shared__ LoopLocal L;
  constexpr int PM = kWinCapMed / kReduceThreads, PR = kWinCapRing / kReduceThreads;
  const unsigned tid = threadIdx.x, n = A.n;
  const unsigned G = gridDim.x * kReduceThreads, first = blockIdx.x * kReduceThreads + tid;
  LoopCtl *const ctl = A.ctl;
  double2 *const s_a = s_pts, *const s_b = s_pts + (size_t)K * kReduceThreads;
  // this thread's points: first, first + G, ... (index order: the first level of the reduction tree)
  const unsigned mine = first < n ? (n - 1u - first) / G + 1u : 0u;

  // ---- the thread's points, once -------------------------------------------------------------------------------
  for (unsigned k = 0; k < mine; ++k) {
    const unsigned i = first + k * G;
    s_a[k * kReduceThreads + tid] = A.a[i];
    s_b[k * kReduceThreads + tid] = A.b[i];
  }
  if (tid < sizeof(WinParams) / sizeof(double))
    reinterpret_cast<double *>(&L.P)[tid] = reinterpret_cast<const double *>(&A.PA)[tid];
  if (tid == 0) {
    L.Ti = A.T0;
    L.prev_error = A.prev_error0;
    L.applied = A.applied0;
    L.done = 0;
    L.status = 0;
    L.retried = 0;
    L.med[0] = L.med[1] = L.sig[0] = L.sig[1] = 0.;
  }
  __syncthreads();
#ifdef ICP_LOOP_PROFILE
  long long prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = wall_clock64();
  const long long t_begin = t_last;
#endif
  unsigned evals = 0;
  unsigned it = A.it0;
  LOOP_STAMP(0);
  bool aborted = false;
  unsigned round = 0;  // evaluation rounds of this launch, repeated ones included: parity of the double buffers, barrier generation
  for (; it < (unsigned)ICP_INNER_MAX_ITER; ++round) {
    const unsigned par = round & 1u;
    uint32_t *const whist = A.whist + (size_t)par * 2 * kWinBins;
    double *const partials = A.partials + (size_t)par * kReduceMaxBlocks * (kNSum + 1);
    double *const totals = A.partials + (size_t)2 * kReduceMaxBlocks * (kNSum + 1) + (size_t)par * (kNSum + 1);
    unsigned *const lcnt = &ctl->list_cnt[par][0][0];
    const Pose T = pose_sgpr(L.Ti);
    WinParams P;
    P.d[0] = windim_sgpr(L.P.d[0]);
    P.d[1] = windim_sgpr(L.P.d[1]);

    // ---- A: residuals -> histograms + running sums ----------------------------------------------------------------
    for (unsigned i = tid; i < 2u * kWinBins; i += kReduceThreads) s_bins[i] = 0;
    if (tid == 0) s_nmem = 0;
    if (tid < 10u) win_region_table(L.P.d[tid / 5u], s_tab[tid / 5u], (int)(tid % 5u));  // (from the LDS copy: a run-time index into registers would go through scratch)
    __syncthreads();
    {
      double acc[kNSum];
#pragma unroll
      for (int k = 0; k < kNSum; ++k) acc[k] = 0.;
      unsigned edge[4] = {0u, 0u, 0u, 0u};  // wave-uniform: points of this WAVE in the catch-all bins {below, above} x {x, y}
      bool saw_nan = false;
      auto stage = [&](unsigned code) {
        const unsigned pos = atomicAdd(&s_nmem, 1u);
        if (pos < kStageCap) s_mem[pos] = (unsigned short)code;
      };
      for (unsigned k = 0; k < mine; ++k) {
        const double2 ak = s_a[k * kReduceThreads + tid], bk = s_b[k * kReduceThreads + tid];
        // residual(), src/lib.rs:34-36
        const double v0 = ((T.r00 * ak.x + T.r01 * ak.y) + T.tx) - bk.x;
        const double v1 = ((T.r10 * ak.x + T.r11 * ak.y) + T.ty) - bk.y;
        saw_nan |= (v0 != v0) | (v1 != v1);
        bool lo0, hi0, f0, lo1, hi1, f1;  // (gn_win_device.hpp: four compares added up + a table row + ONE region_bin)
        const unsigned j0 = wbin_tab(v0, P.d[0], s_tab[0], lo0, hi0, f0), j1 = wbin_tab(v1, P.d[1], s_tab[1], lo1, hi1, f1);
        // (the catch-all bins hold most of a far-off prediction's points: counted by ballot, not by 64 LDS atomics on
        // one word -- and not by shuffles: four dependent six-step reductions were a microsecond per evaluation)
        edge[0] += (unsigned)__popcll(__ballot(lo0));
        edge[1] += (unsigned)__popcll(__ballot(hi0));
        edge[2] += (unsigned)__popcll(__ballot(lo1));
        edge[3] += (unsigned)__popcll(__ballot(hi1));
        if (!lo0 && !hi0) atomicAdd(&s_bins[j0], 1u);
        if (!lo1 && !hi1) atomicAdd(&s_bins[kWinBins + j1], 1u);
        if (f0) stage((k << 10) | (tid << 1));
        if (f1) stage((k << 10) | (tid << 1) | 1u);
        accumulate_pair<true>(ak, v0, v1, T, acc);  // (this thread's points in index order: the tree's first level)
      }
      if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (edge[k]) atomicAdd(&s_bins[(k >> 1) * kWinBins + ((k & 1) ? kWinBins - 1 : 0)], edge[k]);
      }
      if (saw_nan) atomicOr(&ctl->nan_flag[0], 1u);
      block_reduce_waves<kNSum>(acc, s_wv);  // (in front of the barrier: gn_device.hpp)
      __syncthreads();
      LOOP_STAMP(1);
      block_reduce_finish<kNSum, true>(s_wv, partials + (size_t)blockIdx.x * (kNSum + 1));
      for (unsigned i = tid; i < 2u * kWinBins; i += kReduceThreads) {  // dense flush: contiguous words
        const uint32_t c = s_bins[i];
        if (c) atomicAdd(&whist[i], c);
      }
    }
    LOOP_STAMP(2);
    if (!flag_barrier(ctl->flag1, ctl, round + 1u, 0ull, nullptr)) {
      aborted = true;
      break;
    }
    LOOP_STAMP(3);

    // ---- B: bins of the order statistics from the global counts; this workgroup's candidates -----------------------
    LoopBins R;
    loop_resolve(whist, n, L.P, s_bins, R);  // (the window from LDS: the bracket indexes it by wave)
    LOOP_STAMP(4);
    if (tid < 4) s_cnt[tid] = 0;
    // the buffers of the NEXT evaluation back to zero (read last before the previous evaluation's second barrier,
    // written next behind this evaluation's second barrier)
    {
      uint32_t *const other = A.whist + (size_t)(par ^ 1u) * 2 * kWinBins;
      for (unsigned i = first; i < 2u * kWinBins; i += G) st_u32(&other[i], 0u);
      if (blockIdx.x == 0 && tid < 4) st_u32(&ctl->list_cnt[par ^ 1u][tid][0], 0u);
    }
    __syncthreads();
    // (this workgroup's share of the fold below: its load travels while the candidates are listed)
    const bool fold_early = gridDim.x >= (unsigned)(kNSum + 1) && blockIdx.x < (unsigned)(kNSum + 1);
    const double fold_x = fold_early ? fold_one_load(partials, (int)gridDim.x, (int)blockIdx.x) : 0.;
    if (!R.fail) {
      // residual r of dimension d, in bin j: a median candidate, a ring candidate, both or neither
      auto consider = [&](const int d, const double r, const unsigned j) {
        if (j >= R.mlo[d] && j <= R.mhi[d]) {
          const unsigned pos = atomicAdd(&s_cnt[d], 1u);
          if (pos < (unsigned)kWinBlkMed) {
            s_med[d][pos] = r;
          } else {
            const unsigned g = atomicAdd(&lcnt[d * 32], 1u);
            if (g < (unsigned)kWinCapMed) st_f64(&A.wmed[(size_t)d * kWinCapMed + g], r);
          }
        }
        if (j >= R.a0[d] && j <= R.b1[d] && !(j >= R.i0[d] && j <= R.i1[d])) {
          const unsigned pos = atomicAdd(&s_cnt[2 + d], 1u);
          if (pos < (unsigned)kWinBlkRing) {
            s_ring[d][pos] = r;
          } else {
            const unsigned g = atomicAdd(&lcnt[(2 + d) * 32], 1u);
            if (g < (unsigned)kWinCapRing) st_f64(&A.wring[(size_t)d * kWinCapRing + g], r);
          }
        }
      };
      // the staged members hold every candidate when all candidate bins are fine ones and nothing was dropped
      const unsigned nmem = s_nmem;
      bool staged = nmem <= kStageCap;
#pragma unroll
      for (int d = 0; d < 2; ++d)
        staged = staged && R.mlo[d] >= (unsigned)kF1 && R.mhi[d] < (unsigned)kC1 && R.i0[d] <= R.i1[d] && R.a0[d] >= (unsigned)kF0 &&
                 R.i0[d] - 1u < (unsigned)kC0 && R.i1[d] + 1u >= (unsigned)kF2 && R.b1[d] <= (unsigned)(kWinBins - 2);
      if (staged) {  // (uniform)
        for (unsigned e = tid; e < nmem; e += kReduceThreads) {
          const unsigned code = s_mem[e], k = code >> 10, t = (code >> 1) & 511u;
          const double2 ak = s_a[k * kReduceThreads + t], bk = s_b[k * kReduceThreads + t];
          if (code & 1u) {
            const double r = ((T.r10 * ak.x + T.r11 * ak.y) + T.ty) - bk.y;
            consider(1, r, wbin_cold(r, P.d[1]));
          } else {
            const double r = ((T.r00 * ak.x + T.r01 * ak.y) + T.tx) - bk.x;
            consider(0, r, wbin_cold(r, P.d[0]));
          }
        }
      } else {
        for (unsigned k = 0; k < mine; ++k) {
          const double2 ak = s_a[k * kReduceThreads + tid], bk = s_b[k * kReduceThreads + tid];
          const double v[2] = {((T.r00 * ak.x + T.r01 * ak.y) + T.tx) - bk.x, ((T.r10 * ak.x + T.r11 * ak.y) + T.ty) - bk.y};
#pragma unroll
          for (int d = 0; d < 2; ++d) consider(d, v[d], wbin_cold(v[d], P.d[d]));
        }
      }
    }
    if (tid == 0) s_ws = R.sel;  // (phase C reads it back: fewer registers live across the barrier)
    __syncthreads();
    if (tid < 4) {
      const unsigned cap = tid < 2 ? kWinBlkMed : kWinBlkRing;
      const unsigned c = s_cnt[tid] < cap ? s_cnt[tid] : cap;
      s_base[tid] = c ? atomicAdd(&lcnt[tid * 32], c) : 0u;
    }
    __syncthreads();
    if (tid < 2 * kWinBlkMed) {
      const int d = tid / kWinBlkMed, e = tid % kWinBlkMed;
      const unsigned pos = s_base[d] + e;
      if ((unsigned)e < s_cnt[d] && pos < (unsigned)kWinCapMed) st_f64(&A.wmed[(size_t)d * kWinCapMed + pos], s_med[d][e]);
    } else if (tid < 2 * kWinBlkMed + 2 * kWinBlkRing) {
      const int q = tid - 2 * kWinBlkMed, d = q / kWinBlkRing, e = q % kWinBlkRing;
      const unsigned pos = s_base[2 + d] + e;
      if ((unsigned)e < s_cnt[2 + d] && pos < (unsigned)kWinCapRing)
        st_f64(&A.wring[(size_t)d * kWinCapRing + pos], s_ring[d][e]);
    }
    // ONE workgroup folds the block sums (they have been complete since the first barrier) and leaves the twenty
    // totals for everybody: 256 workgroups reading all 256 rows each was 10 MB of same-address traffic per evaluation
    // (round 4, late: one SUM per workgroup -- the first twenty fold one each beside their candidates, where one
    // workgroup folded all twenty and the second barrier waited two microseconds for it)
    if (fold_early) {
      const double tot = fold_one_reduce(fold_x, (int)gridDim.x, s_tot);
      if (tid == 0) st_f64(&totals[blockIdx.x], tot);
    } else if (gridDim.x < (unsigned)(kNSum + 1)) {  // fewer workgroups than sums: several each
      for (unsigned q = blockIdx.x; q < (unsigned)(kNSum + 1); q += gridDim.x) {
        __syncthreads();  // (s_tot[0..3] of the previous sum have been read)
        const double tot = fold_one_sum_256(partials, (int)gridDim.x, (int)q, s_tot);
        if (tid == 0) st_f64(&totals[q], tot);
      }
    }
    const bool b_fail = R.fail;
    LOOP_STAMP(5);
    if (!flag_barrier(ctl->flag2, ctl, round + 1u, 0ull, nullptr)) {
      aborted = true;
      break;
    }
    LOOP_STAMP(6);

    // ---- C: the exact statistics, the folded sums, the update -- the same in every workgroup ------------------------
    bool fail = b_fail;
    double med[2] = {0., 0.}, sig[2] = {0., 0.};
    {
      const unsigned cm[2] = {s_ws.med_cnt[0], s_ws.med_cnt[1]}, cr[2] = {s_ws.ring_cnt[0], s_ws.ring_cnt[1]};
      double vm[2][PR], vr[2][PR];  // (one instantiation of the selection: both lists in the ring's shape)
#pragma unroll
      for (int d = 0; d < 2; ++d) {
#pragma unroll
        for (int u = 0; u < PR; ++u) {
          const unsigned e = tid + u * kReduceThreads;
          vm[d][u] = (u < PM && !fail && e < cm[d]) ? ld_f64(&A.wmed[(size_t)d * kWinCapMed + e]) : 0.;
          vr[d][u] = (!fail && e < cr[d]) ? ld_f64(&A.wring[(size_t)d * kWinCapRing + e]) : 0.;
        }
      }
      unsigned got[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) got[k] = ld_u32(&lcnt[k * 32]);
      const unsigned nan_flag = ld_u32(&ctl->nan_flag[0]);
      if (tid < kNSum + 1) s_tot[tid] = ld_f64(&totals[tid]);
      LOOP_STAMP(7);
      // (the appended counts are cross-checked against the histogram: a mismatch is a miss)
      fail = fail || got[0] != cm[0] || got[1] != cm[1] || got[2] != cr[0] || got[3] != cr[1];
      const unsigned klo = (n - 1) / 2, khi = n / 2;
      if (!fail && !nan_flag) {
        unsigned long long key[2][2];
        const double m_lo[2] = {s_ws.range[0][0], s_ws.range[1][0]}, m_hi[2] = {s_ws.range[0][1], s_ws.range[1][1]};
        const long long mlo[2] = {(long long)klo - s_ws.med_base[0], (long long)klo - s_ws.med_base[1]};
        const long long mhi[2] = {(long long)khi - s_ws.med_base[0], (long long)khi - s_ws.med_base[1]};
        select_n_lds<2, PR>(vm, cm, m_lo, m_hi, mlo, mhi, key, fail, s_sel);
        if (!fail) {
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            med[d] = middle_of(n, key[d][0], key[d][1]);
#pragma unroll
            for (int u = 0; u < PR; ++u) vr[d][u] = fabs(vr[d][u] - med[d]);  // src/stats.rs:35
          }
          const double r_lo[2] = {s_ws.range[0][2], s_ws.range[1][2]}, r_hi[2] = {s_ws.range[0][3], s_ws.range[1][3]};
          const long long dlo[2] = {(long long)klo - s_ws.inner[0], (long long)klo - s_ws.inner[1]};
          const long long dhi[2] = {(long long)khi - s_ws.inner[0], (long long)khi - s_ws.inner[1]};
          select_n_lds<2, PR>(vr, cr, r_lo, r_hi, dlo, dhi, key, fail, s_sel);
          if (!fail) {
            sig[0] = ICP_PPF34 * middle_of(n, key[0][0], key[0][1]);  // src/stats.rs:42-46
            sig[1] = ICP_PPF34 * middle_of(n, key[1][0], key[1][1]);
          }
        }
      }
      __syncthreads();  // (s_tot)
      LOOP_STAMP(8);
      if (tid < kNAcc) s_acc[tid] = combine_sum(s_tot, (int)tid, sig);
      __syncthreads();
      if (tid == 0) {
        double delta[3];
        if (nan_flag) {
          L.done = 1;
          L.status = 3;
        } else if (fail) {
          if (!L.retried && widen_window(&L.P)) {
            L.retried = 1;
            L.done = 3;  // (the same evaluation again, next round)
          } else {
            L.done = 1;
            L.status = 1;
          }
        } else {
          L.retried = 0;
          L.f = next_half_width(med, sig, L.med, L.sig, it > A.it0, A.f_next);
          L.med[0] = med[0];
          L.med[1] = med[1];
          L.sig[0] = sig[0];
          L.sig[1] = sig[1];
          const double err = s_acc[12];
          if (!solve_update(s_acc, s_acc + 9, delta)) {
            L.done = 1;  // None, src/lib.rs:67-69
          } else if ((delta[0] * delta[0] + delta[1] * delta[1]) + delta[2] * delta[2] < ICP_DELTA_NORM_THRESHOLD) {
            L.done = 1;  // src/lib.rs:71-73
          } else if (err > L.prev_error) {
            L.done = 1;  // src/lib.rs:75-78
          } else {
            bool in_range;
            const Pose D = transform_new_in_range(delta, &in_range);
            if (!in_range) {  // the host repeats this evaluation and applies the update with the C library's sin / cos
              L.done = 1;
              L.status = 4;
            } else {
              L.prev_error = err;
              L.Ti = transform_mul(D, L.Ti);  // src/lib.rs:81
              ++L.applied;
              L.done = 2;  // (applied: the next evaluation's window is made below)
            }
          }
        }
        // what the host records as this evaluation's statistics (first / second / most recent of the launch)
        if (blockIdx.x == 0 && !nan_flag && !fail) {
          const unsigned nth = it - A.it0;  // (which evaluation of the loop this launch has reached)
          const int slot = nth < 2u ? (int)nth : 2;
          A.res->med[slot][0] = med[0];
          A.res->med[slot][1] = med[1];
          A.res->sigma[slot][0] = sig[0];
          A.res->sigma[slot][1] = sig[1];
          if (slot < 2) {
            A.res->med[2][0] = med[0];
            A.res->med[2][1] = med[1];
            A.res->sigma[2][0] = sig[0];
            A.res->sigma[2][1] = sig[1];
          }
        }
      }
      __syncthreads();
      if (L.done == 2) {
        // the next evaluation's window: the prediction the host made for the loop's second evaluation, else this
        // evaluation's own statistics
        if (it == A.it0 && A.pb_valid) {  // (the launch's first evaluation was just applied)
          if (tid < sizeof(WinParams) / sizeof(double))
            reinterpret_cast<double *>(&L.P)[tid] = reinterpret_cast<const double *>(&A.PB)[tid];
          if (tid == 0) L.done = 0;
        } else if (tid == 0) {
          if (make_window_hd(L.med, L.sig, L.f, &L.P)) {
            L.done = 0;
          } else {  // (no usable prediction: the host's pipelines serve the next evaluation)
            L.done = 1;
            L.status = 2;
          }
        }
        __syncthreads();
      }
      LOOP_STAMP(9);
    }
    if (L.done == 3) {  // a window missed: the same evaluation once more, widest windows (uniform: every workgroup decided so)
      __syncthreads();
      if (tid == 0) L.done = 0;
      __syncthreads();
      continue;
    }
    if (L.done) {
      // status 0: evaluation `it` ended the loop (it is not counted as applied); 1 / 4: evaluation `it` was not served;
      // 2: evaluation `it` was served and applied, evaluation `it + 1` has no window
      if (L.status == 0 || L.status == 2) ++evals;
      if (L.status == 2) ++it;
      break;
    }
    ++it;
    ++evals;
  }
  if (blockIdx.x == 0 && tid < 64) {
    LoopResult *res = A.res;
    if (tid == 0) {
      res->Ti = L.Ti;
      res->prev_error = L.prev_error;
      res->applied = L.applied;
      res->it = it;
      res->evals = evals;
      res->rounds = round + ((aborted || L.done == 0) ? 0u : 1u);  // (L.done == 0: the loop ran out of iterations at its head)
      res->status = aborted ? 5 : L.status;
      res->finished = (!aborted && L.status == 0) ? 1 : 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) __hip_atomic_store(&res->seq, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
#ifdef ICP_LOOP_PROFILE
  if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2) && (A.seq % 8) == 1)
    printf("[loop blk %d/%d n %u K %u] evals %u; x10ns: load %lld | A %lld flush+reduce %lld bar1 %lld | resolve %lld cand %lld bar2 %lld | "
           "loads %lld select %lld solve %lld | all %lld\n", blockIdx.x, gridDim.x, n, K, evals, prof[0], prof[1], prof[2], prof[3],
           prof[4], prof[5], prof[6], prof[7], prof[8], prof[9], wall_clock64() - t_begin);
#endif
  if (!aborted) leave_launch(ctl, A.whist);
}

// ================================================================================================================
// The same loop over the ranks of a sharded registration (gn_loop.hpp: LoopInbox has the protocol).  Per evaluation:
//   A   as above, on this rank's tree blocks; the block sum goes straight into every rank's inbox (row = global block)
//   L1  local barrier (this rank's workgroups)            -> the rank's histogram is complete
//   X1  workgroup q pushes it into rank q's inbox, then the rank flag
//   W1  wait for every rank's flag in the own inbox      -> the global counts = sum of the pushed histograms
//   B   bins of the order statistics; this workgroup's candidates go into every rank's inbox, then its flag word
//       (generation | counts)
//   W2  wait for the flag words of ALL tree blocks in the own inbox -- the poll gathers the counts
//   C   candidates gathered through the prefix of the counts, block sums folded in global block order, the update:
//       the same bits on every rank, and the bits of one GPU.
// Three waits per evaluation instead of two; the host is not involved.
namespace {

__device__ __forceinline__ unsigned long long ld_sys64(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double ld_sys_f64(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys_f64(double *p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys_u32(unsigned *p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// wave 0 polls `count` (<= 256) flag words until each carries a generation >= gen; the words it saw last go to
// s_seen (LDS, 256 words) when given.  Callers have drained their stores and stand behind a workgroup barrier.
// A rank that gives up raises `abort` in EVERY rank's inbox (peers: null-terminated list of the other ranks' abort words,
// or null): its peers are waiting for data it will never send, and would otherwise sit out their own three seconds --
// with the raise they all leave within a poll, every rank's launch reports the same "gave up" and every host takes the
// same fallback without a collective.
struct AbortPeers {
  unsigned *w[kShardMaxWorld];
  int n;
};
__device__ __forceinline__ bool poll_words(const unsigned long long *words, unsigned count, unsigned gen, unsigned *abort_word,
                                           unsigned long long *s_seen, const AbortPeers *peers = nullptr) {
  __shared__ int s_okw;
  if (threadIdx.x < 64) {
    const unsigned lane = threadIdx.x;
    unsigned long long seen[4] = {0, 0, 0, 0};
    int ok = 1;
    const long long t0 = wall_clock64();
    for (;;) {
      bool all = true;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned b = lane + 64u * j;
        if (b < count) {
          seen[j] = ld_sys64(&words[b]);
          all = all && (int)((unsigned)seen[j] - gen) >= 0;
        }
      }
      if (__all(all)) break;
      int stop = 0;
      if (lane == 0) {
        if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) stop = 1;
        else if (wall_clock64() - t0 > kShardTimeoutTicks) {
          st_sys_u32(abort_word, 1u);
          if (peers)
            for (int q = 0; q < peers->n; ++q) st_sys_u32(peers->w[q], 1u);
          stop = 1;
        }
      }
      if (__builtin_amdgcn_readfirstlane(stop)) {
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (s_seen) {
#pragma unroll
      for (int j = 0; j < 4; ++j) s_seen[lane + 64u * j] = (lane + 64u * j < count) ? seen[j] : 0ull;
    }
    if (lane == 0) s_okw = ok;
  }
  __syncthreads();
  return s_okw != 0;
}

// payload of a candidate flag word (its upper half)
__device__ __forceinline__ unsigned cand_payload(unsigned mx, unsigned my, unsigned rx, unsigned ry, bool ovf, bool nan) {
  return mx | (my << 6) | (rx << 12) | (ry << 19) | ((ovf ? 1u : 0u) << 26) | ((nan ? 1u : 0u) << 27);
}
__device__ __forceinline__ unsigned payload_count(unsigned p, int k) {
  return k == 0 ? (p & 63u) : (k == 1 ? ((p >> 6) & 63u) : (k == 2 ? ((p >> 12) & 127u) : ((p >> 19) & 127u)));
}

}  // namespace

// gridDim.y > 1: several ranks in ONE launch (ranks that share a device -- "virtual ranks": the launches of ranks wait
// for each other, and more streams than hardware queues would queue one behind the other); blockIdx.y picks the rank,
// R has each rank's pairs and result block.
// (Round 5 streamed more than eight pairs per thread from the rank's arrays to reach 2^23 pairs with 256 blocks; since
// round 6 the tree grows beyond 2^20 points -- more blocks, never more than eight pairs per thread -- and such clouds
// are served by the pipelined evaluation, pipe.hip, and the stage calls: this launch keeps to 2^20 pairs in total.)
__global__ __launch_bounds__(kReduceThreads) void k_gn_loop_shard(LoopArgs A, LoopShardArgs S, LoopRankPtrs R_, unsigned K) {
  // The per-rank tables of the arguments, copied into LDS with CONSTANT indices: a run-time index into a by-value
  // argument struct sends the whole of it through scratch memory (848 bytes per lane until round 5), and the exchange
  // phases index the inboxes by rank in their inner loops.
  __shared__ LoopInbox *s_inbox[kShardMaxWorld];
  __shared__ int s_first[kShardMaxWorld + 1];
  __shared__ const double2 *s_ra[kShardMaxWorld], *s_rb[kShardMaxWorld];
  __shared__ LoopResult *s_rres[kShardMaxWorld];
#pragma unroll
  for (int q = 0; q < kShardMaxWorld; ++q)
    if ((int)threadIdx.x == q) {
      s_inbox[q] = S.inbox[q];
      s_first[q] = S.first_block[q];
      s_ra[q] = R_.a[q];
      s_rb[q] = R_.b[q];
      s_rres[q] = R_.res[q];
    }
  if (threadIdx.x == 0) s_first[kShardMaxWorld] = S.first_block[kShardMaxWorld];
  __syncthreads();
  const int rank = S.rank + (int)blockIdx.y;
  const unsigned nbl = (unsigned)(s_first[rank + 1] - s_first[rank]);  // this rank's workgroups
  if (blockIdx.x >= nbl) return;
  // (locals, not fields of the arguments: writing to a by-value argument makes it a private copy)
  const double2 *const Aa = s_ra[rank], *const Ab = s_rb[rank];
  LoopResult *const Ares = s_rres[rank];
  const int rank_b0 = s_first[rank];
  extern __shared__ double2 s_pts[];
  __shared__ __align__(16) unsigned char s_work[sizeof(SelectLds<2>) > 2 * kWinBins * sizeof(uint32_t) ? sizeof(SelectLds<2>)
                                                                                                         : 2 * kWinBins * sizeof(uint32_t)];
  uint32_t *const s_bins = reinterpret_cast<uint32_t *>(s_work);
  SelectLds<2> &s_sel = *reinterpret_cast<SelectLds<2> *>(s_work);
  __shared__ double s_med[2][kWinBlkMed], s_ring[2][kWinBlkRing];
  __shared__ unsigned s_cnt[4];
  __shared__ double s_tot[kNSum + 1], s_acc[kNAcc + 3], s_row[kNSum + 1];
  __shared__ double s_wv[kReduceThreads / 64][kNSum];  // the waves' sums of phase A (block_reduce_waves / _finish)
  __shared__ WinRegion s_tab[2][8];  // phase A's bins: wbin_tab (five rows per dimension, rewritten for every evaluation's windows)
  // (the flag words of the second wait and the prefix of their counts overlay the same workspace: they live between
  // phase B's last look at the cumulative counts and the first selection)
  unsigned long long *const s_seen = reinterpret_cast<unsigned long long *>(s_work), *const s_incl = s_seen + kReduceMaxBlocks;
  // (the members of the fine bins, staged by phase A behind the histograms: k_gn_loop has the explanation)
  constexpr unsigned kStageCap = (sizeof(s_work) - 2 * kWinBins * sizeof(uint32_t)) / sizeof(unsigned short);
  static_assert(kStageCap >= 4096 && kLoopMaxK <= 8, "staged members: 3 + 9 + 1 bits each");
  unsigned short *const s_mem = reinterpret_cast<unsigned short *>(s_work + 2 * kWinBins * sizeof(uint32_t));
  __shared__ unsigned s_nmem;
  __shared__ unsigned long long s_wsum[4];
  __shared__ unsigned s_flags;
  __shared__ WinSel s_ws;
  __shared__ LoopLocal L;
  constexpr int PR = kWinCapRing / kReduceThreads, PM = kWinCapMed / kReduceThreads;
  const unsigned tid = threadIdx.x, n = A.n;
  const int W = S.world;
  const unsigned gb = (unsigned)rank_b0 + blockIdx.x, B = (unsigned)S.blocks_total;
  const unsigned G = B * kReduceThreads, first = gb * kReduceThreads + tid;
  const unsigned row_w = nbl * kReduceThreads, loc0 = blockIdx.x * kReduceThreads + tid;  // local layout (shard.hip)
  LoopInbox *const me = s_inbox[rank];
  __shared__ AbortPeers s_peers;  // (filled behind a run-time index: LDS, not private memory; read behind the barriers below)
  if (tid == 0) {
    int np = 0;
    for (int q = 0; q < W; ++q)
      if (s_inbox[q] != me) s_peers.w[np++] = &s_inbox[q]->abort[0];
    s_peers.n = np;
  }
  double2 *const s_a = s_pts, *const s_b = s_pts + (size_t)K * kReduceThreads;
  const unsigned mine = first < n ? (n - 1u - first) / G + 1u : 0u;
  for (unsigned k = 0; k < mine; ++k) {  // (the rank's arrays in the local layout of shard.hip)
    s_a[k * kReduceThreads + tid] = Aa[(size_t)k * row_w + loc0];
    s_b[k * kReduceThreads + tid] = Ab[(size_t)k * row_w + loc0];
  }
  // point k of thread t of this workgroup
  auto pair_of = [&](unsigned k, unsigned t, double2 &ak, double2 &bk) {
    ak = s_a[k * kReduceThreads + t];
    bk = s_b[k * kReduceThreads + t];
  };
  if (tid < sizeof(WinParams) / sizeof(double))
    reinterpret_cast<double *>(&L.P)[tid] = reinterpret_cast<const double *>(&A.PA)[tid];
  if (tid == 0) {
    L.Ti = A.T0;
    L.prev_error = A.prev_error0;
    L.applied = A.applied0;
    L.done = 0;
    L.status = 0;
    L.retried = 0;
    L.med[0] = L.med[1] = L.sig[0] = L.sig[1] = 0.;
  }
  __syncthreads();
  unsigned evals = 0;
  unsigned it = A.it0;
  bool aborted = false;
  unsigned round = 0;  // evaluation rounds of this launch, repeated ones included
  for (; it < (unsigned)ICP_INNER_MAX_ITER; ++round) {
    const unsigned par = (S.eval_base + round) & 1u;
    const unsigned gen = S.gen_base + round + 1u;
    uint32_t *const whist = me->hist_local[par];
    const Pose T = pose_sgpr(L.Ti);
    WinParams P;
    P.d[0] = windim_sgpr(L.P.d[0]);
    P.d[1] = windim_sgpr(L.P.d[1]);

    // ---- A ----------------------------------------------------------------------------------------------------------
    for (unsigned i = tid; i < 2u * kWinBins; i += kReduceThreads) s_bins[i] = 0;
    if (tid == 0) {
      s_flags = 0u;
      s_nmem = 0u;
    }
    if (tid < 10u) win_region_table(L.P.d[tid / 5u], s_tab[tid / 5u], (int)(tid % 5u));  // (from the LDS copy: a run-time index into registers would go through scratch)
    __syncthreads();
    {
      double acc[kNSum];
#pragma unroll
      for (int k = 0; k < kNSum; ++k) acc[k] = 0.;
      unsigned edge[4] = {0u, 0u, 0u, 0u};
      bool saw_nan = false;
      auto stage = [&](unsigned code) {
        const unsigned pos = atomicAdd(&s_nmem, 1u);
        if (pos < kStageCap) s_mem[pos] = (unsigned short)code;
      };
      double2 nak = {0., 0.}, nbk = {0., 0.};
      if (mine) pair_of(0, tid, nak, nbk);
      for (unsigned k = 0; k < mine; ++k) {
        const double2 ak = nak, bk = nbk;
        if (k + 1 < mine) pair_of(k + 1, tid, nak, nbk);
        const double v0 = ((T.r00 * ak.x + T.r01 * ak.y) + T.tx) - bk.x;  // residual(), src/lib.rs:34-36
        const double v1 = ((T.r10 * ak.x + T.r11 * ak.y) + T.ty) - bk.y;
        saw_nan |= (v0 != v0) | (v1 != v1);
        bool lo0, hi0, f0, lo1, hi1, f1;  // (gn_win_device.hpp: four compares added up + a table row + ONE region_bin)
        const unsigned j0 = wbin_tab(v0, P.d[0], s_tab[0], lo0, hi0, f0), j1 = wbin_tab(v1, P.d[1], s_tab[1], lo1, hi1, f1);
        edge[0] += (unsigned)__popcll(__ballot(lo0));
        edge[1] += (unsigned)__popcll(__ballot(hi0));
        edge[2] += (unsigned)__popcll(__ballot(lo1));
        edge[3] += (unsigned)__popcll(__ballot(hi1));
        if (!lo0 && !hi0) atomicAdd(&s_bins[j0], 1u);
        if (!lo1 && !hi1) atomicAdd(&s_bins[kWinBins + j1], 1u);
        if (f0) stage((k << 10) | (tid << 1));
        if (f1) stage((k << 10) | (tid << 1) | 1u);
        accumulate_pair<true>(ak, v0, v1, T, acc);
      }
      if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (edge[k]) atomicAdd(&s_bins[(k >> 1) * kWinBins + ((k & 1) ? kWinBins - 1 : 0)], edge[k]);
      }
      if (saw_nan) atomicOr(&s_flags, 2u);
      block_reduce_waves<kNSum>(acc, s_wv);  // (in front of the barrier: gn_device.hpp)
      __syncthreads();
      block_reduce_finish<kNSum, false>(s_wv, s_row);
      for (unsigned i = tid; i < 2u * kWinBins; i += kReduceThreads) {
        const uint32_t c = s_bins[i];
        if (c) atomicAdd(&whist[i], c);
      }
      __syncthreads();
      // the block sum into every rank's inbox, row = global block: the fold of phase C reads them in block order
      for (unsigned j = tid; j < (unsigned)W * (kNSum + 1); j += kReduceThreads) {
        const unsigned q = j / (kNSum + 1), k = j % (kNSum + 1);
        st_sys_f64(&s_inbox[q]->partials[par][gb][k], k < (unsigned)kNSum ? s_row[k] : 0.);
      }
    }
    // ---- L1: this rank's workgroups ----------------------------------------------------------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&me->flag_block[gb], (unsigned long long)gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!poll_words(&me->flag_block[rank_b0], nbl, gen, &me->abort[0], nullptr, &s_peers)) {
      aborted = true;
      break;
    }
    // ---- X1: the rank's histogram into every inbox (workgroup q serves rank q) -------------------------------------
    for (int q = (int)blockIdx.x; q < W; q += (int)nbl) {
      uint32_t *dst = s_inbox[q]->hist_from[rank][par];
      for (unsigned i = tid; i < 2u * kWinBins; i += kReduceThreads) st_sys_u32(&dst[i], ld_u32(&whist[i]));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0)
        __hip_atomic_store(&s_inbox[q]->flag_rank[rank], (unsigned long long)gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- W1 ----------------------------------------------------------------------------------------------------------
    if (!poll_words(me->flag_rank, (unsigned)W, gen, &me->abort[0], nullptr, &s_peers)) {
      aborted = true;
      break;
    }

    // ---- B -----------------------------------------------------------------------------------------------------------
    LoopBins R;
    loop_resolve<true>(&me->hist_from[0][par][0], n, L.P, s_bins, R, W, (size_t)2 * 2 * kWinBins);
    if (tid < 4) s_cnt[tid] = 0;
    {
      uint32_t *const other = me->hist_local[par ^ 1u];
      for (unsigned i = loc0; i < 2u * kWinBins; i += row_w) st_u32(&other[i], 0u);
    }
    __syncthreads();
    if (!R.fail) {
      auto consider = [&](const int d, const double r, const unsigned j) {
        if (j >= R.mlo[d] && j <= R.mhi[d]) {
          const unsigned pos = atomicAdd(&s_cnt[d], 1u);
          if (pos < (unsigned)kWinBlkMed) s_med[d][pos] = r;
        }
        if (j >= R.a0[d] && j <= R.b1[d] && !(j >= R.i0[d] && j <= R.i1[d])) {
          const unsigned pos = atomicAdd(&s_cnt[2 + d], 1u);
          if (pos < (unsigned)kWinBlkRing) s_ring[d][pos] = r;
        }
      };
      const unsigned nmem = s_nmem;
      bool staged = nmem <= kStageCap;
#pragma unroll
      for (int d = 0; d < 2; ++d)
        staged = staged && R.mlo[d] >= (unsigned)kF1 && R.mhi[d] < (unsigned)kC1 && R.i0[d] <= R.i1[d] && R.a0[d] >= (unsigned)kF0 &&
                 R.i0[d] - 1u < (unsigned)kC0 && R.i1[d] + 1u >= (unsigned)kF2 && R.b1[d] <= (unsigned)(kWinBins - 2);
      if (staged) {  // (uniform)
        for (unsigned e = tid; e < nmem; e += kReduceThreads) {
          const unsigned code = s_mem[e], k = code >> 10, t = (code >> 1) & 511u;
          double2 ak, bk;
          pair_of(k, t, ak, bk);
          if (code & 1u) {
            const double r = ((T.r10 * ak.x + T.r11 * ak.y) + T.ty) - bk.y;
            consider(1, r, wbin_cold(r, P.d[1]));
          } else {
            const double r = ((T.r00 * ak.x + T.r01 * ak.y) + T.tx) - bk.x;
            consider(0, r, wbin_cold(r, P.d[0]));
          }
        }
      } else {
        for (unsigned k = 0; k < mine; ++k) {
          double2 ak, bk;
          pair_of(k, tid, ak, bk);
          const double v[2] = {((T.r00 * ak.x + T.r01 * ak.y) + T.tx) - bk.x, ((T.r10 * ak.x + T.r11 * ak.y) + T.ty) - bk.y};
#pragma unroll
          for (int d = 0; d < 2; ++d) consider(d, v[d], wbin_cold(v[d], P.d[d]));
        }
      }
    }
    if (tid == 0) s_ws = R.sel;
    __syncthreads();
    {
      // (more candidates in one workgroup than it can stage -- a run of equal residuals among neighbouring points --
      // is reported as a miss: the host's pipelines serve such an evaluation)
      const bool ovf = s_cnt[0] > (unsigned)kWinBlkMed || s_cnt[1] > (unsigned)kWinBlkMed || s_cnt[2] > (unsigned)kWinBlkRing ||
                       s_cnt[3] > (unsigned)kWinBlkRing;
      const unsigned c[4] = {ovf ? 0u : s_cnt[0], ovf ? 0u : s_cnt[1], ovf ? 0u : s_cnt[2], ovf ? 0u : s_cnt[3]};
      for (int q = 0; q < W; ++q) {
        double *dst = s_inbox[q]->cand[gb];
        if (tid < 2 * kWinBlkMed) {
          const int d = tid / kWinBlkMed, e = tid % kWinBlkMed;
          if ((unsigned)e < c[d]) st_sys_f64(&dst[d * kWinBlkMed + e], s_med[d][e]);
        } else if (tid < 2 * kWinBlkMed + 2 * kWinBlkRing) {
          const int x = tid - 2 * kWinBlkMed, d = x / kWinBlkRing, e = x % kWinBlkRing;
          if ((unsigned)e < c[2 + d]) st_sys_f64(&dst[2 * kWinBlkMed + d * kWinBlkRing + e], s_ring[d][e]);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid < (unsigned)W) {
        const unsigned pay = cand_payload(c[0], c[1], c[2], c[3], ovf || R.fail, (s_flags & 2u) != 0u);
        __hip_atomic_store(&s_inbox[tid]->flag_cand[gb], (unsigned long long)gen | ((unsigned long long)pay << 32), __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    // ---- W2 ----------------------------------------------------------------------------------------------------------
    if (!poll_words(me->flag_cand, B, gen, &me->abort[0], s_seen, &s_peers)) {
      aborted = true;
      break;
    }

    // ---- C -----------------------------------------------------------------------------------------------------------
    bool fail = false;
    unsigned nan_flag = 0;
    double med[2] = {0., 0.}, sig[2] = {0., 0.};
    {
      // inclusive prefix of the four counts over the tree blocks, 16 bits each in one word
      unsigned long long v = 0;
      unsigned pay = 0;
      if (tid < kReduceMaxBlocks) {
        pay = (unsigned)(s_seen[tid] >> 32);
        if (tid < B)
          v = (unsigned long long)payload_count(pay, 0) | ((unsigned long long)payload_count(pay, 1) << 16) |
              ((unsigned long long)payload_count(pay, 2) << 32) | ((unsigned long long)payload_count(pay, 3) << 48);
        else
          pay = 0;
      }
      const int lane = tid & 63, wave = tid >> 6;
      unsigned long long inc = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
      }
      if (tid < kReduceMaxBlocks && lane == 63) s_wsum[wave] = inc;
      const unsigned long long bad = __ballot((pay >> 26) & 1u), nanb = __ballot((pay >> 27) & 1u);
      if (lane == 0 && tid < kReduceMaxBlocks && (bad | nanb)) atomicOr(&s_flags, (bad ? 4u : 0u) | (nanb ? 8u : 0u));
      __syncthreads();
      if (tid < kReduceMaxBlocks) {
        unsigned long long base = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) base += (w < wave) ? s_wsum[w] : 0ull;
        s_incl[tid] = base + inc;
      }
      __syncthreads();
      const unsigned long long totals = s_incl[kReduceMaxBlocks - 1];
      const unsigned tot[4] = {(unsigned)(totals & 0xffffu), (unsigned)((totals >> 16) & 0xffffu), (unsigned)((totals >> 32) & 0xffffu),
                               (unsigned)(totals >> 48)};
      fail = (s_flags & 4u) != 0u;
      nan_flag = (s_flags & 8u) ? 1u : 0u;
      const unsigned cm[2] = {s_ws.med_cnt[0], s_ws.med_cnt[1]}, cr[2] = {s_ws.ring_cnt[0], s_ws.ring_cnt[1]};
      // (the gathered counts are cross-checked against the histogram: a mismatch is a miss)
      fail = fail || tot[0] != cm[0] || tot[1] != cm[1] || tot[2] != cr[0] || tot[3] != cr[1];
      // element e of list k lives in the block g whose inclusive count first exceeds e
      auto fetch = [&](int k, unsigned e) -> double {
        unsigned lo = 0, hi = B - 1;
        while (lo < hi) {
          const unsigned mid = (lo + hi) >> 1;
          if ((unsigned)((s_incl[mid] >> (16 * k)) & 0xffffu) > e) hi = mid;
          else lo = mid + 1;
        }
        const unsigned inc_g = (unsigned)((s_incl[lo] >> (16 * k)) & 0xffffu);
        const unsigned cnt_g = payload_count((unsigned)(s_seen[lo] >> 32), k);
        const unsigned off = e - (inc_g - cnt_g);
        const unsigned basek = k < 2 ? (unsigned)k * kWinBlkMed : 2u * kWinBlkMed + (unsigned)(k - 2) * kWinBlkRing;
        return ld_sys_f64(&me->cand[lo][basek + off]);
      };
      double vm[2][PR], vr[2][PR];
#pragma unroll
      for (int d = 0; d < 2; ++d) {
#pragma unroll
        for (int u = 0; u < PR; ++u) {
          const unsigned e = tid + u * kReduceThreads;
          vm[d][u] = (u < PM && !fail && e < cm[d]) ? fetch(d, e) : 0.;
          vr[d][u] = (!fail && e < cr[d]) ? fetch(2 + d, e) : 0.;
        }
      }
      fold_block_sums_256<__HIP_MEMORY_SCOPE_SYSTEM>(&me->partials[par][0][0], (int)B, s_tot);
      __syncthreads();  // (every fetch has read the prefix: the selections may take the workspace)
      const unsigned klo = (n - 1) / 2, khi = n / 2;
      if (!fail && !nan_flag) {
        unsigned long long key[2][2];
        const double m_lo[2] = {s_ws.range[0][0], s_ws.range[1][0]}, m_hi[2] = {s_ws.range[0][1], s_ws.range[1][1]};
        const long long mlo[2] = {(long long)klo - s_ws.med_base[0], (long long)klo - s_ws.med_base[1]};
        const long long mhi[2] = {(long long)khi - s_ws.med_base[0], (long long)khi - s_ws.med_base[1]};
        select_n_lds<2, PR>(vm, cm, m_lo, m_hi, mlo, mhi, key, fail, s_sel);
        if (!fail) {
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            med[d] = middle_of(n, key[d][0], key[d][1]);
#pragma unroll
            for (int u = 0; u < PR; ++u) vr[d][u] = fabs(vr[d][u] - med[d]);  // src/stats.rs:35
          }
          const double r_lo[2] = {s_ws.range[0][2], s_ws.range[1][2]}, r_hi[2] = {s_ws.range[0][3], s_ws.range[1][3]};
          const long long dlo[2] = {(long long)klo - s_ws.inner[0], (long long)klo - s_ws.inner[1]};
          const long long dhi[2] = {(long long)khi - s_ws.inner[0], (long long)khi - s_ws.inner[1]};
          select_n_lds<2, PR>(vr, cr, r_lo, r_hi, dlo, dhi, key, fail, s_sel);
          if (!fail) {
            sig[0] = ICP_PPF34 * middle_of(n, key[0][0], key[0][1]);  // src/stats.rs:42-46
            sig[1] = ICP_PPF34 * middle_of(n, key[1][0], key[1][1]);
          }
        }
      }
      __syncthreads();
      if (tid < kNAcc) s_acc[tid] = combine_sum(s_tot, (int)tid, sig);
      __syncthreads();
      if (tid == 0) {
        double delta[3];
        if (nan_flag) {
          L.done = 1;
          L.status = 3;
        } else if (fail) {
          if (!L.retried && widen_window(&L.P)) {
            L.retried = 1;
            L.done = 3;  // (the same evaluation again, next round)
          } else {
            L.done = 1;
            L.status = 1;
          }
        } else {
          L.retried = 0;
          L.f = next_half_width(med, sig, L.med, L.sig, it > A.it0, A.f_next);
          L.med[0] = med[0];
          L.med[1] = med[1];
          L.sig[0] = sig[0];
          L.sig[1] = sig[1];
          const double err = s_acc[12];
          if (!solve_update(s_acc, s_acc + 9, delta)) {
            L.done = 1;  // None, src/lib.rs:67-69
          } else if ((delta[0] * delta[0] + delta[1] * delta[1]) + delta[2] * delta[2] < ICP_DELTA_NORM_THRESHOLD) {
            L.done = 1;  // src/lib.rs:71-73
          } else if (err > L.prev_error) {
            L.done = 1;  // src/lib.rs:75-78
          } else {
            bool in_range;
            const Pose D = transform_new_in_range(delta, &in_range);
            if (!in_range) {
              L.done = 1;
              L.status = 4;
            } else {
              L.prev_error = err;
              L.Ti = transform_mul(D, L.Ti);  // src/lib.rs:81
              ++L.applied;
              L.done = 2;
            }
          }
        }
        if (blockIdx.x == 0 && !nan_flag && !fail) {
          const unsigned nth = it - A.it0;  // (which evaluation of the loop this launch has reached)
          const int slot = nth < 2u ? (int)nth : 2;
          Ares->med[slot][0] = med[0];
          Ares->med[slot][1] = med[1];
          Ares->sigma[slot][0] = sig[0];
          Ares->sigma[slot][1] = sig[1];
          if (slot < 2) {
            Ares->med[2][0] = med[0];
            Ares->med[2][1] = med[1];
            Ares->sigma[2][0] = sig[0];
            Ares->sigma[2][1] = sig[1];
          }
        }
      }
      __syncthreads();
      if (L.done == 2) {
        if (it == A.it0 && A.pb_valid) {  // (the launch's first evaluation was just applied)
          if (tid < sizeof(WinParams) / sizeof(double))
            reinterpret_cast<double *>(&L.P)[tid] = reinterpret_cast<const double *>(&A.PB)[tid];
          if (tid == 0) L.done = 0;
        } else if (tid == 0) {
          if (make_window_hd(L.med, L.sig, L.f, &L.P)) {
            L.done = 0;
          } else {
            L.done = 1;
            L.status = 2;
          }
        }
        __syncthreads();
      }
    }
    if (L.done == 3) {  // a window missed: the same evaluation once more, widest windows
      __syncthreads();
      if (tid == 0) L.done = 0;
      __syncthreads();
      continue;
    }
    if (L.done) {
      if (L.status == 0 || L.status == 2) ++evals;
      if (L.status == 2) ++it;
      break;
    }
    ++it;
    ++evals;
  }
  if (blockIdx.x == 0 && tid < 64) {
    LoopResult *res = Ares;
    if (tid == 0) {
      res->Ti = L.Ti;
      res->prev_error = L.prev_error;
      res->applied = L.applied;
      res->it = it;
      res->evals = evals;
      res->rounds = round + ((aborted || L.done == 0) ? 0u : 1u);  // (L.done == 0: the loop ran out of iterations at its head)
      res->status = aborted ? 5 : L.status;
      res->finished = (!aborted && L.status == 0) ? 1 : 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) __hip_atomic_store(&res->seq, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (!aborted) {  // the last workgroup of THIS rank: its own histograms back to zero (nobody else touches them)
    __shared__ int s_lastw;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_lastw = __hip_atomic_fetch_add(&me->done[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nbl - 1;
    __syncthreads();
    if (s_lastw) {
      for (unsigned i = tid; i < 4u * kWinBins; i += kReduceThreads) st_u32(&me->hist_local[0][0] + i, 0u);
      if (tid == 0) st_u32(&me->done[0], 0u);
    }
  }
}

// Every workgroup of a launch must be running at once (they wait for each other): the device has to offer that many
// slots for workgroups of this kernel with the most dynamic LDS a launch asks for.  Asked once per process (the current
// device at the first call: the handles of a process live on devices of one kind).
static int loop_slots() {
  static int slots = -1;
  if (slots < 0) {
    slots = 0;
    int dev = 0, per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gn_loop), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kLoopMaxK * kReduceThreads * 2 * (int)sizeof(double2)) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(&k_gn_loop), kReduceThreads,
                                                     (size_t)kLoopMaxK * kReduceThreads * 2 * sizeof(double2)) == hipSuccess)
      slots = per_cu * prop.multiProcessorCount;
    (void)hipGetLastError();
  }
  return slots;
}

bool gn_loop_applies(size_t n) {
  static const bool off = getenv("ICP_NO_GN_LOOP") != nullptr;
  int blocks, threads;
  reduce_geometry(n, &blocks, &threads);
  return !off && n >= (size_t)(1u << 12) && n <= (size_t)kLoopMaxK * (size_t)blocks * (size_t)threads && blocks <= loop_slots();
}

size_t gn_loop_partials_doubles() { return (size_t)2 * kReduceMaxBlocks * (kNSum + 1) + (size_t)2 * (kNSum + 1); }

hipError_t launch_gn_loop(icp_handle *h, const LoopArgs &args) {
  int blocks, threads;
  reduce_geometry(args.n, &blocks, &threads);
  const size_t G = (size_t)blocks * threads;
  const unsigned K = (unsigned)((args.n + G - 1) / G);
  // up to 128 KB of dynamic LDS beside ~31 KB of static: granted once per process
  static int lds_granted = 0;  // 0 not asked yet, 1 yes, -1 refused
  if (lds_granted == 0) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gn_loop), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             kLoopMaxK * kReduceThreads * 2 * (int)sizeof(double2));
    lds_granted = e == hipSuccess ? 1 : -1;
    if (e != hipSuccess) (void)hipGetLastError();
  }
  if (lds_granted < 0 || K > (unsigned)kLoopMaxK) return hipErrorInvalidValue;
  const size_t lds = (size_t)K * kReduceThreads * 2 * sizeof(double2);
  hipLaunchKernelGGL(k_gn_loop, dim3(blocks), dim3(threads), lds, h->stream, args, K);
  return hipGetLastError();
}

bool gn_loop_shard_applies(size_t n_total, int world) {
  static const bool off = getenv("ICP_NO_GN_LOOP") != nullptr;
  int blocks, threads;
  reduce_geometry(n_total, &blocks, &threads);
  // (ranks that share a device need all `blocks` slots on it; a rank with a device of its own needs fewer: the check is
  // the conservative one)
  return !off && world >= 1 && world <= kShardMaxWorld && blocks >= world && n_total >= (size_t)(1u << 12) &&
         n_total <= (size_t)kLoopMaxK * (size_t)blocks * (size_t)threads && blocks <= loop_slots();
}

hipError_t launch_gn_loop_shard(icp_handle *h, const LoopArgs &args, const LoopShardArgs &sh, const LoopRankPtrs &ptrs, int ranks) {
  int blocks, threads;
  reduce_geometry(args.n, &blocks, &threads);
  const size_t G = (size_t)blocks * threads;
  const unsigned K = (unsigned)((args.n + G - 1) / G);
  static int lds_granted = 0;
  if (lds_granted == 0) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gn_loop_shard),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             kLoopMaxK * kReduceThreads * 2 * (int)sizeof(double2));
    lds_granted = e == hipSuccess ? 1 : -1;
    if (e != hipSuccess) (void)hipGetLastError();
  }
  int nb = 0;  // the widest of the ranks this launch carries
  for (int q = sh.rank; q < sh.rank + ranks; ++q) nb = std::max(nb, sh.first_block[q + 1] - sh.first_block[q]);
  if (lds_granted < 0 || K > (unsigned)kLoopMaxK || nb < 1 || ranks < 1 || sh.rank + ranks > sh.world || sh.blocks_total != blocks)
    return hipErrorInvalidValue;
  const size_t lds = (size_t)K * kReduceThreads * 2 * sizeof(double2);
  hipLaunchKernelGGL(k_gn_loop_shard, dim3(nb, ranks), dim3(threads), lds, h->stream, args, sh, ptrs, K);
  return hipGetLastError();
}

// ---- transport probe -------------------------------------------------------------------------------------------
// Does memory written by a peer (another device, another process) become visible to a kernel of THIS device that is
// already running and polling it?  That is what the exchange above relies on, and it depends on how the inboxes were
// allocated and mapped (coarse-grained device memory is coherent between devices at kernel boundaries only).  One
// wave per rank: in round r it stores the token base + r into its slot of every peer's inbox and waits, bounded, for
// the same token from every peer in its own -- each round overwrites the word of the round before, so a mapping
// that serves a stale copy fails at the latest in round two.  Every rank runs it at the same time (the driver's
// barrier in front); *ok = 1 when every token of every round arrived.
struct ProbePtrs {
  LoopInbox *inbox[kShardMaxWorld];
};
__global__ __launch_bounds__(64) void k_loop_probe(ProbePtrs P, int rank, int world, unsigned base, unsigned rounds, unsigned *ok) {
  const int lane = (int)threadIdx.x;
  LoopInbox *const me = P.inbox[rank];
  int good = 1;
  for (unsigned r = 1; r <= rounds && good; ++r) {
    const unsigned long long token = (unsigned long long)base + r;
    if (lane < world) __hip_atomic_store(&P.inbox[lane]->probe[rank], token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const long long t0 = wall_clock64();
    for (;;) {
      const bool have = lane >= world || ld_sys64(&me->probe[lane]) >= token;
      if (__all(have)) break;
      if (wall_clock64() - t0 > kShardTimeoutTicks) {  // (uniform enough: every lane reads the same clock within a poll)
        good = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
    good = __all(good != 0) ? 1 : 0;
  }
  if (lane == 0) *ok = (unsigned)good;
}

hipError_t launch_loop_probe(icp_handle *h, int rank, int world, void *const *inboxes, unsigned base, unsigned rounds,
                             unsigned *d_ok) {
  ProbePtrs P;
  for (int q = 0; q < kShardMaxWorld; ++q) P.inbox[q] = q < world ? reinterpret_cast<LoopInbox *>(inboxes[q]) : nullptr;
  hipLaunchKernelGGL(k_loop_probe, dim3(1), dim3(64), 0, h->stream, P, rank, world, base, rounds, d_ok);
  return hipGetLastError();
}

}  // namespace icp
