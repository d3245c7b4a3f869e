// Dispatch of one inner Gauss-Newton evaluation (src/lib.rs:218-261 + :45-50) to the short
// pipelines, and the single-workgroup kernel for tiny inputs.
//
//   n <= 1024            k_tiny_eval below: ONE workgroup, ONE launch
//   otherwise            gn_pull.hip (7 or 9 launches); icp_estimate's loop prefers gn_win.hip
//                        (3 launches) once it has a prediction -- see api.hip:wgn_step
//
// (The first short pipeline, whose launches ended in a last-workgroup tail, lived here; the
// "pull" variant replaced it: +8 % per step, same bits.  Its lessons are in DESIGN.md.)
#include "common.hpp"
#include "gn_device.hpp"

namespace icp {

// ---------------------------------------------------------------------------------------
// n <= 1024 (the reference's own 2-D scans have ~650 points): the whole evaluation in ONE
// workgroup and ONE launch -- residuals, both medians (bitonic sort of the order-preserving keys:
// order statistics are then plain lookups), both MADs (ranked on the sorted residuals), and the weighted
// normal equations folded in exactly the multi-workgroup tree of reduce_geometry(n)
// (1 or 2 virtual blocks of 512 threads, then the 512-thread second stage), so the bits are
// the same as on the general path.
// Bitonic sort of 2 x 1024 keys, one key of each array per thread.  The 45 compare-exchange stages
// whose partner is in the same wave (j < 64) are register shuffles; only the 10 stages with
// j >= 64 go through LDS, double-buffered so that each costs ONE workgroup barrier.  (The first
// version ran all 55 stages through LDS with a barrier each, twice per evaluation: 48 us per
// evaluation of a 650-point scan, 40 of them barriers.)  On return thread t holds the t-th
// smallest key of each array.
__device__ __forceinline__ void bitonic_sort2_1024(unsigned long long &ka, unsigned long long &kb,
                                                   unsigned long long (*buf)[2][1024]) {
  const unsigned tid = threadIdx.x;
  int cur = 0;
  for (unsigned k = 2; k <= 1024; k <<= 1)
    for (unsigned j = k >> 1; j > 0; j >>= 1) {
      unsigned long long pa, pb;
      if (j >= 64) {
        buf[cur][0][tid] = ka;
        buf[cur][1][tid] = kb;
        __syncthreads();
        pa = buf[cur][0][tid ^ j];
        pb = buf[cur][1][tid ^ j];
        cur ^= 1;  // the next LDS stage writes the other buffer: nobody is still reading it
      } else {
        pa = __shfl_xor(ka, (int)j);
        pb = __shfl_xor(kb, (int)j);
      }
      // ascending block (tid & k) == 0: the lower index keeps the smaller key
      const bool keep_min = ((tid & j) == 0) == ((tid & k) == 0);
      ka = keep_min ? (ka < pa ? ka : pa) : (ka > pa ? ka : pa);
      kb = keep_min ? (kb < pb ? kb : pb) : (kb > pb ? kb : pb);
    }
}

// The two middle order statistics of fl(|r - med|) over the n residuals whose keys are sorted in
// S (src/stats.rs:30-37) WITHOUT sorting again: left of the median the distances fl(med - r) fall
// with the index, right of it fl(r - med) rise (rounding is monotone), so "how many distances are
// < d" and "<= d" are two binary searches on each side.  Every thread ranks its own distance; the
// threads whose rank interval [less, leq) holds a wanted rank publish it (equal values: benign).
__device__ __forceinline__ void mad_ranks(const unsigned long long *S, unsigned n, double med, unsigned lo_rank,
                                          unsigned hi_rank, double *out /* LDS, [2] */) {
  const unsigned tid = threadIdx.x;
  if (tid >= n) return;
  auto dist = [&](unsigned i) { return fabs(k2f(S[i]) - med); };
  // p = first index with r >= med (NaN residuals are reported through nan_flag; the loops are bounded)
  unsigned p;
  {
    unsigned lo = 0, hi = n;
    while (lo < hi) {
      const unsigned mid = (lo + hi) >> 1;
      if (k2f(S[mid]) < med) lo = mid + 1;
      else hi = mid;
    }
    p = lo;
  }
  const double d = dist(tid);
  // left part [0, p): distances non-increasing in i -> {d_i < d} and {d_i <= d} are suffixes
  auto left_first = [&](bool strict) {
    unsigned lo = 0, hi = p;
    while (lo < hi) {
      const unsigned mid = (lo + hi) >> 1;
      const double v = dist(mid);
      if (strict ? (v < d) : (v <= d)) hi = mid;
      else lo = mid + 1;
    }
    return lo;
  };
  // right part [p, n): non-decreasing -> prefixes
  auto right_end = [&](bool strict) {
    unsigned lo = p, hi = n;
    while (lo < hi) {
      const unsigned mid = (lo + hi) >> 1;
      const double v = dist(mid);
      if (strict ? (v < d) : (v <= d)) lo = mid + 1;
      else hi = mid;
    }
    return lo;
  };
  const unsigned less = (p - left_first(true)) + (right_end(true) - p);
  const unsigned leq = (p - left_first(false)) + (right_end(false) - p);
  if (less <= lo_rank && lo_rank < leq) out[0] = d;
  if (less <= hi_rank && hi_rank < leq) out[1] = d;
}

// fold `acc` over a group of 8 waves (512 threads) in the tree of block_reduce_store:
// wave shuffle tree, then a left fold of the wave sums from the group's first wave
template <int N>
__device__ __forceinline__ void group_reduce(double (&acc)[N], double (*sm)[N], int wave) {
  const int lane = threadIdx.x & 63;
  wave_tree<N>(acc);  // see block_reduce_store
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < N; ++k) sm[wave][k] = acc[k];
  }
}

__global__ __launch_bounds__(1024) void k_tiny_eval(const double2 *__restrict__ a, const double2 *__restrict__ b,
                                                    unsigned n, Pose T, int blocks, GnResult *res,
                                                    unsigned seq) {
  __shared__ unsigned long long buf[2][2][1024];  // [buffer][x | y][slot]; buffer 0 ends up holding the sorted keys
  __shared__ double sm[16][kNAcc + 1];
  __shared__ double part[2][kNAcc + 1];
  __shared__ double s_mad[2][2];
  __shared__ int s_nan;
  const unsigned tid = threadIdx.x;
  const int wave = tid >> 6;
#ifdef ICP_TINY_DEBUG
  long long tst[10];
  int tns = 0;
#define TSTAMP() tst[tns++] = wall_clock64()
#else
#define TSTAMP()
#endif
  TSTAMP();
  if (tid == 0) s_nan = 0;
  __syncthreads();
  const bool has = tid < n;
  double2 s = make_double2(0., 0.);
  double r0 = 0., r1 = 0.;
  if (has) {  // residual(), src/lib.rs:34-36
    s = a[tid];
    const double2 d = b[tid];
    r0 = ((T.r00 * s.x + T.r01 * s.y) + T.tx) - d.x;
    r1 = ((T.r10 * s.x + T.r11 * s.y) + T.ty) - d.y;
    if ((r0 != r0) | (r1 != r1)) s_nan = 1;
  }
  TSTAMP();
  const unsigned lo_rank = (n - 1) / 2, hi_rank = n / 2;
  // medians (src/stats.rs:11-28): sort the order-preserving keys, look the two middle ranks up
  unsigned long long ka = has ? f2k(r0) : ~0ull, kb = has ? f2k(r1) : ~0ull;
  bitonic_sort2_1024(ka, kb, buf);
  TSTAMP();
  __syncthreads();  // (the last LDS stage's readers)
  buf[0][0][tid] = ka;
  buf[0][1][tid] = kb;
  __syncthreads();
  double med[2];
  {
    const double xl = k2f(buf[0][0][lo_rank]), xh = k2f(buf[0][0][hi_rank]);
    const double yl = k2f(buf[0][1][lo_rank]), yh = k2f(buf[0][1][hi_rank]);
    med[0] = (n & 1) ? xl : (xl + xh) / 2.;
    med[1] = (n & 1) ? yl : (yl + yh) / 2.;
  }
  TSTAMP();
  // MADs (src/stats.rs:30-47): ranks of the distances to the median, from the sorted residuals
  mad_ranks(buf[0][0], n, med[0], lo_rank, hi_rank, s_mad[0]);  // (both dimensions in lockstep: slower, measured)
  mad_ranks(buf[0][1], n, med[1], lo_rank, hi_rank, s_mad[1]);
  __syncthreads();
  TSTAMP();
  double sig[2];
  sig[0] = ICP_PPF34 * ((n & 1) ? s_mad[0][0] : (s_mad[0][0] + s_mad[0][1]) / 2.);
  sig[1] = ICP_PPF34 * ((n & 1) ? s_mad[1][0] : (s_mad[1][0] + s_mad[1][1]) / 2.);
  // weighted normal equations + Huber error (src/lib.rs:238-255, 45-50), one point per thread
  double acc[kNAcc + 1];
#pragma unroll
  for (int k = 0; k < kNAcc + 1; ++k) acc[k] = 0.;
  if (has) {
    const double g[2] = {1. / sig[0], 1. / sig[1]};
    const double r[2] = {r0, r1};
    const double a0 = -s.y, a1 = s.x;
    const double b0 = T.r00 * a0 + T.r01 * a1;
    const double b1 = T.r10 * a0 + T.r11 * a1;
    const double J[2][3] = {{T.r00, T.r01, b0}, {T.r10, T.r11, b1}};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (sig[j] == 0.) continue;
      const double r_ij = r[j];
      const double w_ij = huber_drho(r_ij * r_ij);
      const double wg = w_ij * g[j];
#pragma unroll
      for (int k = 0; k < 3; ++k) acc[9 + k] = acc[9 + k] + (wg * J[j][k]) * r_ij;
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q) acc[3 * p + q] = acc[3 * p + q] + (wg * J[j][p]) * J[j][q];
    }
    acc[12] = acc[12] + huber_rho(r[0] * r[0] + r[1] * r[1]);
  }
  TSTAMP();
  // stage 1: virtual blocks of 512 threads (8 waves each); a wave without points sums to +0.0
  if ((unsigned)wave * 64u < n) {
    group_reduce<kNAcc + 1>(acc, sm, wave);
  } else if ((tid & 63) == 0) {
#pragma unroll
    for (int k = 0; k < kNAcc + 1; ++k) sm[wave][k] = 0.;
  }
  __syncthreads();
  if (tid < 2 * (kNAcc + 1)) {
    const int vb = tid / (kNAcc + 1), k = tid % (kNAcc + 1);
    double v = sm[8 * vb][k];
    for (int w = 1; w < 8; ++w) v = v + sm[8 * vb + w][k];
    part[vb][k] = v;
  }
  __syncthreads();
  // stage 2: one block of 512 threads over the `blocks` (<= 2) block sums.  Only threads 0 and 1 of
  // that block hold anything: every other operand of its wave tree and of the fold over its wave
  // sums is +0.0, and x + 0.0 is x (a -0.0 becomes +0.0, once and for all).  So lane 0 of wave 0
  // ends with (p0 + 0.0) + (p1 + 0.0) -- lane 1 joins at the last step -- and the fold adds zeros:
  // the same bits as the general path without its 84 shuffles.
  if (tid < kNAcc + 1) {
    const double p0 = (0. + part[0][tid]) + 0.;
    const double p1 = blocks > 1 ? (0. + part[1][tid]) + 0. : 0.;
    res->acc[tid] = tid < kNAcc ? (p0 + p1) + 0. : 0.;
  }
  TSTAMP();
  // everything the host reads is stored by lanes of wave 0, so one wave's fence orders it before
  // the sequence number (a system-scope fence in all sixteen waves cost 2 us)
  if (wave == 0) {
    if (tid == kNAcc + 1) {
      res->sigma[0] = sig[0];
      res->sigma[1] = sig[1];
    }
    if (tid == kNAcc + 2) {
      res->nan_flag = s_nan;
      res->overflow = 0;
    }
    __threadfence_system();
    TSTAMP();
    if (tid == 0) __hip_atomic_store(&res->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
#ifdef ICP_TINY_DEBUG
  TSTAMP();
  if (tid == 0 && (seq % 64) == 5)
    printf("[tiny] load %lld sort %lld lookup %lld mad %lld accumulate %lld reduce %lld fence %lld publish %lld (x10 ns)\n",
           tst[1] - tst[0], tst[2] - tst[1], tst[3] - tst[2], tst[4] - tst[3], tst[5] - tst[4], tst[6] - tst[5],
           tst[7] - tst[6], tst[8] - tst[7]);
#endif
}

hipError_t launch_weighted_gn_fast(icp_handle *h, const double *d_a, const double *d_b, size_t n_, const Pose &T) {
  Workspace &w = h->ws;
  const unsigned n = (unsigned)n_;
  if (n <= 1024u) {
    int blocks, threads;
    reduce_geometry(n_, &blocks, &threads);
    if (threads == 512 && blocks <= 2) {  // the geometry k_tiny_eval reproduces
      hipLaunchKernelGGL(k_tiny_eval, dim3(1), dim3(1024), 0, h->stream, (const double2 *)d_a, (const double2 *)d_b, n,
                         T, blocks, w.h_res, ++w.seq);
      return hipGetLastError();
    }
  }
  return launch_weighted_gn_pull(h, d_a, d_b, n_, T);
}

}  // namespace icp
