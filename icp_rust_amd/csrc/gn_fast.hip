// Dispatch of one inner Gauss-Newton evaluation (src/lib.rs:218-261 + :45-50) to the short
// pipelines, and the single-workgroup kernel for tiny inputs.
//
//   n <= 1024            k_tiny_eval below: ONE workgroup, ONE launch
//   otherwise            gn_pull.hip (7 or 9 launches); icp_estimate's loop prefers gn_win.hip
//                        (3 launches) once it has a prediction -- see api.hip:wgn_step
//
// (The first short pipeline, whose launches ended in a last-workgroup tail, lived here; the
// "pull" variant replaced it: +8 % per step, same bits.  Its lessons are in DESIGN.md.)
#include <cstdio>

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

// src/stats.rs:18-27 on two order-preserving keys
__device__ __forceinline__ double middle_of_host(unsigned n, unsigned long long klo, unsigned long long khi) {
  const double lo = k2f(klo), hi = k2f(khi);
  return (n & 1) ? lo : (lo + hi) / 2.;
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
  __shared__ double sm[16][kNSum + 1];
  __shared__ double part[2][kNSum + 1];
  __shared__ double s_tot[kNSum + 1];
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
  double acc[kNSum + 1];
#pragma unroll
  for (int k = 0; k < kNSum + 1; ++k) acc[k] = 0.;
  if (has) accumulate_pair<false>(s, r0, r1, T, acc);
  TSTAMP();
  // stage 1: virtual blocks of 512 threads (8 waves each); a wave without points sums to +0.0
  if ((unsigned)wave * 64u < n) {
    group_reduce<kNSum + 1>(acc, sm, wave);
  } else if ((tid & 63) == 0) {
#pragma unroll
    for (int k = 0; k < kNSum + 1; ++k) sm[wave][k] = 0.;
  }
  __syncthreads();
  if (tid < 2 * (kNSum + 1)) {
    const int vb = tid / (kNSum + 1), k = tid % (kNSum + 1);
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
  if (tid < kNSum + 1) {
    const double p0 = (0. + part[0][tid]) + 0.;
    const double p1 = blocks > 1 ? (0. + part[1][tid]) + 0. : 0.;
    s_tot[tid] = tid < kNSum ? (p0 + p1) + 0. : 0.;
  }
  __syncthreads();
  if (tid < kNAcc + 1) res->acc[tid] = tid < kNAcc ? combine_sum(s_tot, (int)tid, sig) : 0.;  // g_x S_x + g_y S_y
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

// =======================================================================================
// The WHOLE registration of a small cloud in ONE workgroup and ONE launch (round 2).
//
// Icp{2,3}d::estimate (src/lib.rs:105-130, 148-173) on the reference's own data -- 2-D scans of ~650
// points (scans/2d) -- used to be ~80 launches with a host round trip per inner iteration
// (src/lib.rs:66-82): 2.15 ms per estimate(20), level with one CPU core.  For n <= 1024 source points
// and m <= 2048 targets everything now stays on one CU: the targets live in LDS (exact f64 + an f32
// copy for the screen), every thread owns one source point; per outer iteration an exact
// nearest-neighbour sweep (warm-started from the previous match), then the inner loop -- residuals,
// the four exact order statistics, the weighted normal equations in the tree of reduce_geometry(n),
// and, on thread 0, the 3x3 solve, the two break tests and Transform::new * T (src/lib.rs:71-81) with
// the sin / cos of include/icp_trig.h, which the host and the oracle share -- so the bits are those of
// the host-driven path.
//
// Order statistics without a full sort (the bitonic sort of k_tiny_eval is 12 of its 29 us): 64
// evenly strided sample keys are sorted by one wave (register shuffles); every key finds its bucket
// among the 64 splitters; one wave scans the 65 counts, finds the bucket(s) of the two middle ranks; the
// ~10 keys in them are ranked by counting.  Exact for any input; more than 128 keys in the middle
// buckets (heavy duplicates) -> the sorting path below serves that evaluation.
// =======================================================================================
struct TinySel {
  unsigned spl[2][64];  // sorted splitters: the HIGH words of 64 sampled keys (monotone in the key, and 32-bit compares)
  unsigned long long list[2][128];
  unsigned long long out[2][2];
  unsigned hist[2][68];
  unsigned nlist[2];
  // round 4: the window the NEXT selection of the same kind (slot 0: medians, slot 1: MADs) tries first -- a key
  // range around this selection's middle ranks, per dimension (tiny_select_window below)
  unsigned long long wlo[2][2], whi[2][2];
  unsigned wvalid[2];
  unsigned wpos[2][2];  // where in the window's list the lower middle rank was expected (0xffffffff: unknown)
};
// ranks the window keeps on either side of the two middle ranks: twice the drift the last selection saw (the listed
// keys are ranked against each other: the cost grows with the square of their number)
constexpr unsigned kTinyWinMarginMin = 6, kTinyWinMarginMax = 32, kTinyWinMarginFirst = 16;

// cross-lane sums without the LDS pipe (DPP): over aligned groups of 8 lanes, and the wave's inclusive scan
#define ICP_DPP(v, ctrl, rows) (unsigned)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rows, 0xf, true)
__device__ __forceinline__ unsigned dpp_sum8(unsigned v) {
  v += ICP_DPP(v, 0xB1, 0xf);   // quad_perm [1,0,3,2]
  v += ICP_DPP(v, 0x4E, 0xf);   // quad_perm [2,3,0,1]
  v += ICP_DPP(v, 0x141, 0xf);  // row_half_mirror: lane i of an 8-group reads lane 7-i, in the other quad
  return v;
}

#ifdef ICP_TINY_PROFILE
#define SEL_STAMP(slot)                                             \
  do {                                                              \
    if (sp) {                                                       \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
      sp[slot] += now_ - st_;                                       \
      st_ = now_;                                                   \
    }                                                               \
  } while (0)
#else
#define SEL_STAMP(slot) ((void)0)
#endif

// the two middle order statistics (ranks (n-1)/2 and n/2) of the keys k0 (dimension 0) and k1, one
// key of each per thread (~0 from the threads past n); false: too many equal-ish keys, the caller
// sorts instead. B threads (a multiple of 64), five barriers, no single-wave phase: sample 64 keys per dimension
// and rank the samples (every thread a few compares) -> bucket every key between the sorted
// samples -> every wave scans the 65 bucket counts itself and the keys of the bucket(s) holding
// the two ranks are listed -> the listed keys are ranked against each other, 8 lanes per key.
// kbuf: 2 x 1024 keys of LDS (the sorting path's buffer).
template <unsigned B>
__device__ __forceinline__ bool tiny_select(unsigned long long k0, unsigned long long k1, bool has, unsigned n,
                                            TinySel *S, unsigned long long (*kbuf)[1024],
                                            unsigned long long *sp = nullptr, int wslot = -1) {
#ifdef ICP_TINY_PROFILE
  unsigned long long st_ = __builtin_amdgcn_s_memtime();
#endif
  // (opaque to the optimiser: everything below that depends on n alone -- the sample positions,
  // for one -- would otherwise be hoisted out of the caller's loop and, at 128 registers, spilled)
  asm volatile("" : "+s"(n));
  const unsigned tid = threadIdx.x, lane = tid & 63;
  const unsigned lo_rank = (n - 1) / 2, hi_rank = n / 2;
  const unsigned long long key[2] = {k0, k1};
  kbuf[0][tid] = k0;
  kbuf[1][tid] = k1;
  if (tid < 2 * 68) (&S->hist[0][0])[tid] = 0;
  if (tid < 2) S->nlist[tid] = 0;
  __syncthreads();
  SEL_STAMP(0);
  // sample j is the key of thread floor(j n / ns); slot (d, i, c) compares sample i with samples 8c .. 8c+7
  for (unsigned slot = tid; slot < 1024u; slot += B) {  // (whole 8-lane groups: B is a multiple of 64)
    const unsigned d = slot >> 9, i = (slot >> 3) & 63, c = slot & 7;
    const unsigned ns = n < 64u ? n : 64u;  // (the missing samples sort last)
    const unsigned *kh = reinterpret_cast<const unsigned *>(kbuf[d]);
    // floor(j n / ns) without a division: ns is 64, or n itself (24-bit products: j < 64, n <= 1024)
    auto sample = [&](unsigned j) -> unsigned {
      const unsigned t = n >= 64u ? __umul24(j, n) >> 6 : j;
      return j < ns ? kh[2 * t + 1] : 0xffffffffu;
    };
    const unsigned si = sample(i);
    unsigned part = 0;
#pragma unroll
    for (unsigned u = 0; u < 8; ++u) {
      const unsigned j = 8 * c + u;
      const unsigned sj = sample(j);
      part += (sj < si) | ((sj == si) & (j < i));
    }
    const unsigned rank = dpp_sum8(part);
    if (c == 0) S->spl[d][rank] = si;
  }
  __syncthreads();
  SEL_STAMP(1);
  // bucket = number of splitters below the key's high word (monotone in the key): a branch-free
  // lower bound over the 64 sorted splitters, the two dimensions' LDS reads in flight together
  unsigned bucket[2] = {0, 0};
  {
    const unsigned kd0 = (unsigned)(k0 >> 32), kd1 = (unsigned)(k1 >> 32);
#pragma unroll
    for (unsigned step = 32; step > 0; step >>= 1) {
      const unsigned s0 = S->spl[0][bucket[0] + step - 1], s1 = S->spl[1][bucket[1] + step - 1];
      bucket[0] += s0 < kd0 ? step : 0u;
      bucket[1] += s1 < kd1 ? step : 0u;
    }
    const unsigned s0 = S->spl[0][bucket[0]], s1 = S->spl[1][bucket[1]];  // (positions 0 .. 63)
    bucket[0] += s0 < kd0;
    bucket[1] += s1 < kd1;
    if (has) {
      atomicAdd(&S->hist[0][bucket[0]], 1u);
      atomicAdd(&S->hist[1][bucket[1]], 1u);
    }
  }
  __syncthreads();
  SEL_STAMP(2);
  unsigned below[2], expect[2];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const unsigned c = S->hist[d][lane], c64 = S->hist[d][64];
    const unsigned inc = wave_scan_inclusive(c);
    const unsigned long long m_lo = __ballot(inc > lo_rank), m_hi = __ballot(inc > hi_rank);
    const unsigned b_lo = m_lo ? (unsigned)__ffsll((long long)m_lo) - 1u : 64u;
    const unsigned b_hi = m_hi ? (unsigned)__ffsll((long long)m_hi) - 1u : 64u;
    const unsigned tot63 = (unsigned)__builtin_amdgcn_readlane((int)inc, 63);
    below[d] = b_lo < 64u ? (unsigned)__builtin_amdgcn_readlane((int)(inc - c), (int)b_lo) : tot63;
    const unsigned upto = b_hi < 64u ? (unsigned)__builtin_amdgcn_readlane((int)inc, (int)b_hi) : tot63 + c64;
    expect[d] = upto - below[d];
    if (has && bucket[d] >= b_lo && bucket[d] <= b_hi) {
      const unsigned pos = atomicAdd(&S->nlist[d], 1u);
      if (pos < 128u) S->list[d][pos] = key[d];
    }
    if (wslot >= 0 && tid == 0) {  // the next selection of this kind looks two buckets either side of these first
      const bool usable = n >= 256u && b_lo < 64u && b_hi < 64u;
      S->wlo[wslot][d] = b_lo >= 3u ? (unsigned long long)S->spl[d][b_lo - 3] << 32 : 0ull;
      S->whi[wslot][d] = b_hi + 2u < 64u ? ((unsigned long long)S->spl[d][b_hi + 2] << 32) | 0xffffffffull : 0xfffffffffffffffeull;
      if (d == 0) S->wvalid[wslot] = usable ? 1u : 0u;
      else if (!usable) S->wvalid[wslot] = 0u;
      S->wpos[wslot][d] = 0xffffffffu;
    }
  }
  __syncthreads();
  SEL_STAMP(3);
  const unsigned cnt0 = S->nlist[0], cnt1 = S->nlist[1];
  const bool bad = cnt0 > 128u || cnt0 != expect[0] || cnt1 > 128u || cnt1 != expect[1];  // the same in every thread
  if (!bad) {  // listed key e is ranked by the 8 lanes (e, 0..7), each against every 8th listed key
    const unsigned cmax = cnt0 > cnt1 ? cnt0 : cnt1;
    for (unsigned slot = tid; slot < 8u * cmax; slot += B) {  // (whole 8-lane groups)
      const unsigned e = slot >> 3, c = slot & 7;
      const bool in0 = e < cnt0, in1 = e < cnt1;
      const unsigned long long ke0 = S->list[0][in0 ? e : 0], ke1 = S->list[1][in1 ? e : 0];
      unsigned acc0 = 0, acc1 = 0;  // keys below in the low half, equal keys in the high half (at most 128 each)
      for (unsigned j = c; j < cmax; j += 8) {
        const unsigned long long kj0 = S->list[0][j < cnt0 ? j : 0], kj1 = S->list[1][j < cnt1 ? j : 0];
        if (j < cnt0) acc0 += (unsigned)(kj0 < ke0) + ((unsigned)(kj0 == ke0) << 16);
        if (j < cnt1) acc1 += (unsigned)(kj1 < ke1) + ((unsigned)(kj1 == ke1) << 16);
      }
      acc0 = dpp_sum8(acc0);
      acc1 = dpp_sum8(acc1);
      if (c == 0) {
        if (in0) {
          const unsigned less = acc0 & 0xffffu, eq = acc0 >> 16, r_lo = lo_rank - below[0], r_hi = hi_rank - below[0];
          if (less <= r_lo && r_lo < less + eq) S->out[0][0] = ke0;
          if (less <= r_hi && r_hi < less + eq) S->out[0][1] = ke0;
        }
        if (in1) {
          const unsigned less = acc1 & 0xffffu, eq = acc1 >> 16, r_lo = lo_rank - below[1], r_hi = hi_rank - below[1];
          if (less <= r_lo && r_lo < less + eq) S->out[1][0] = ke1;
          if (less <= r_hi && r_hi < less + eq) S->out[1][1] = ke1;
        }
      }
    }
  }
  __syncthreads();
  SEL_STAMP(4);
  return !bad;
}

// The same two order statistics from a WINDOW (round 4): consecutive evaluations of a registration see almost the
// same residuals, so the keys between the previous selection's neighbours of the middle ranks are listed directly --
// every thread compares its key with the window's ends, the keys below the window are counted (ballot + one LDS
// atomic per wave), those inside are listed -- and ranked exactly as tiny_select ranks its buckets: three barriers
// instead of five, no sampling, no splitter search.  Exact whenever it answers: it answers only if both middle ranks
// fall among the listed keys (count below <= rank < count below + listed, at most 128 listed); otherwise false, with
// the window dropped, and the caller runs tiny_select (which sets a fresh one).  The next window is cut from this
// one's ranking: the keys `margin` ranks below / above the middle ranks (dropped if either side is short).
template <unsigned B>
__device__ __forceinline__ bool tiny_select_window(unsigned long long k0, unsigned long long k1, bool has, unsigned n,
                                                   TinySel *S, int wslot, unsigned long long *sp = nullptr) {
#ifdef ICP_TINY_PROFILE
  unsigned long long st_ = __builtin_amdgcn_s_memtime();
#endif
  asm volatile("" : "+s"(n));
  const unsigned tid = threadIdx.x, lane = tid & 63;
  const unsigned lo_rank = (n - 1) / 2, hi_rank = n / 2;
  const unsigned long long key[2] = {k0, k1};
  if (!S->wvalid[wslot]) return false;  // (uniform: written behind a barrier of the previous selection)
  const unsigned long long wl[2] = {S->wlo[wslot][0], S->wlo[wslot][1]}, wh[2] = {S->whi[wslot][0], S->whi[wslot][1]};
  __syncthreads();  // (everybody has read the window and the previous selection's S->out)
  if (tid < 2) {
    S->hist[tid][0] = 0;
    S->nlist[tid] = 0;
  }
  __syncthreads();
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const bool under = has && key[d] < wl[d];
    const unsigned long long mb = __ballot(under);
    if (lane == 0 && mb) atomicAdd(&S->hist[d][0], (unsigned)__popcll(mb));
    if (has && !under && key[d] <= wh[d]) {
      const unsigned pos = atomicAdd(&S->nlist[d], 1u);
      if (pos < 128u) S->list[d][pos] = key[d];
    }
  }
  __syncthreads();
  SEL_STAMP(5);
  const unsigned below[2] = {S->hist[0][0], S->hist[1][0]};
  const unsigned cnt0 = S->nlist[0], cnt1 = S->nlist[1];
  const bool bad = cnt0 > 128u || cnt1 > 128u || lo_rank < below[0] || hi_rank >= below[0] + cnt0 || lo_rank < below[1] ||
                   hi_rank >= below[1] + cnt1;  // the same in every thread
  if (bad) {
    __syncthreads();  // (every thread has read the counts tiny_select is about to reset)
    if (tid == 0) S->wvalid[wslot] = 0u;
    return false;
  }
  // the next window: the keys `margin` ranks either side of the middle ranks in this list, margin = twice the drift
  // this selection saw (how far the lower middle rank landed from where the window was cut for it) + 4; a side
  // that cannot give half of it drops the window
  bool keep = true;
  unsigned t_lo[2], t_hi[2], margin = kTinyWinMarginFirst;
  {
    unsigned drift = 0;
    bool known = true;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const unsigned r_lo = lo_rank - below[d], exp_pos = S->wpos[wslot][d];
      known = known && exp_pos != 0xffffffffu;
      const unsigned dd = r_lo > exp_pos ? r_lo - exp_pos : exp_pos - r_lo;
      drift = dd > drift ? dd : drift;
    }
    if (known) {
      margin = 2u * drift + 4u;
      margin = margin < kTinyWinMarginMin ? kTinyWinMarginMin : (margin > kTinyWinMarginMax ? kTinyWinMarginMax : margin);
    }
  }
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const unsigned r_lo = lo_rank - below[d], r_hi = hi_rank - below[d], c = d ? cnt1 : cnt0;
    t_lo[d] = r_lo > margin ? r_lo - margin : 0u;
    t_hi[d] = r_hi + margin < c ? r_hi + margin : c - 1u;
    keep = keep && r_lo >= margin / 2 && r_hi + margin / 2 < c;
  }
  __syncthreads();  // (everybody has read wpos)
  if (tid < 2) S->wpos[wslot][tid] = (lo_rank - below[tid]) - t_lo[tid];
  {
    const unsigned cmax = cnt0 > cnt1 ? cnt0 : cnt1;
    for (unsigned slot = tid; slot < 8u * cmax; slot += B) {  // (whole 8-lane groups)
      const unsigned e = slot >> 3, c = slot & 7;
      const bool in0 = e < cnt0, in1 = e < cnt1;
      const unsigned long long ke0 = S->list[0][in0 ? e : 0], ke1 = S->list[1][in1 ? e : 0];
      unsigned acc0 = 0, acc1 = 0;
      for (unsigned j = c; j < cmax; j += 8) {
        const unsigned long long kj0 = S->list[0][j < cnt0 ? j : 0], kj1 = S->list[1][j < cnt1 ? j : 0];
        if (j < cnt0) acc0 += (unsigned)(kj0 < ke0) + ((unsigned)(kj0 == ke0) << 16);
        if (j < cnt1) acc1 += (unsigned)(kj1 < ke1) + ((unsigned)(kj1 == ke1) << 16);
      }
      acc0 = dpp_sum8(acc0);
      acc1 = dpp_sum8(acc1);
      if (c == 0) {
        if (in0) {
          const unsigned less = acc0 & 0xffffu, eq = acc0 >> 16, r_lo = lo_rank - below[0], r_hi = hi_rank - below[0];
          if (less <= r_lo && r_lo < less + eq) S->out[0][0] = ke0;
          if (less <= r_hi && r_hi < less + eq) S->out[0][1] = ke0;
          if (less <= t_lo[0] && t_lo[0] < less + eq) S->wlo[wslot][0] = ke0;
          if (less <= t_hi[0] && t_hi[0] < less + eq) S->whi[wslot][0] = ke0;
        }
        if (in1) {
          const unsigned less = acc1 & 0xffffu, eq = acc1 >> 16, r_lo = lo_rank - below[1], r_hi = hi_rank - below[1];
          if (less <= r_lo && r_lo < less + eq) S->out[1][0] = ke1;
          if (less <= r_hi && r_hi < less + eq) S->out[1][1] = ke1;
          if (less <= t_lo[1] && t_lo[1] < less + eq) S->wlo[wslot][1] = ke1;
          if (less <= t_hi[1] && t_hi[1] < less + eq) S->whi[wslot][1] = ke1;
        }
      }
    }
    if (tid == 0 && !keep) S->wvalid[wslot] = 0u;
  }
  __syncthreads();
  SEL_STAMP(6);
  return true;
}

struct TinyResult {  // pinned host memory
  Pose pose;
  int status;       // 0 ok, 3 NaN residual (ICP_NAN_INPUT), -1 hand the call to the host-driven path
  unsigned evals;   // Gauss-Newton evaluations run, in all
  unsigned sorted;  // ... of which by the sorting path
  unsigned pad;
  unsigned long long t[6];  // ICP_TINY_PROFILE builds: shader cycles in {setup, search, selections, sums, step, all}
  unsigned long long ts[8]; // ... and inside the selections, per phase
};
constexpr unsigned kTinyMaxN = 1024, kTinyMaxM = 2048, kTinyMaxIter = 1024;

template <int DIM, unsigned B>
__global__ __launch_bounds__(B) void k_tiny_estimate(const double *__restrict__ src, unsigned n,
                                                        const double *__restrict__ dst, unsigned m, Pose T0,
                                                        unsigned max_iter, double cx, double cy, double cz, double scale,
                                                        TinyResult *res, uint32_t *inner_out, uint32_t *idx_out) {
  extern __shared__ unsigned char lds_raw[];
  // ---- LDS carve-up ----  (targets are kept SORTED BY x: position j below is not the target's index)
  const unsigned mp = (m + 63u) & ~63u;
  double *tx = reinterpret_cast<double *>(lds_raw);
  double *ty = tx + mp;
  double *tz = ty + mp;  // (DIM == 2: unused, zero length below)
  unsigned char *p = reinterpret_cast<unsigned char *>(tz + (DIM == 3 ? mp : 0));
  float4 *g4 = reinterpret_cast<float4 *>(p);  // {x, y, z relative to the box centre as f32, original index}
  p += sizeof(float4) * (mp + 4);
  unsigned long long(*sbuf)[2][1024] = reinterpret_cast<unsigned long long(*)[2][1024]>(p);  // sorting path
  p += sizeof(unsigned long long) * 2 * 2 * 1024;
  TinySel *S = reinterpret_cast<TinySel *>(p);
  p += (sizeof(TinySel) + 15) & ~size_t(15);
  double(*sm)[kNSum + 1] = reinterpret_cast<double(*)[kNSum + 1]>(p);
  p += sizeof(double) * 16 * (kNSum + 1);
  double(*part)[kNSum + 1] = reinterpret_cast<double(*)[kNSum + 1]>(p);
  p += sizeof(double) * 2 * (kNSum + 1);
  struct Ctl {
    Pose Ti, T;
    double s_mad[2][2];
    int done, nan, bail, fixed;
    unsigned applied, evals, sorted;
  };
  Ctl *C = reinterpret_cast<Ctl *>(p);

  const unsigned tid = threadIdx.x;
  const int wave = tid >> 6;
  const bool has = tid < n;
#ifdef ICP_TINY_PROFILE
  unsigned long long tp[6] = {0, 0, 0, 0, 0, 0}, t_last = __builtin_amdgcn_s_memtime();
  unsigned long long tsel[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_begin = t_last;
#define TINY_STAMP(slot)                                         \
  do {                                                           \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    tp[slot] += now_ - t_last;                                   \
    t_last = now_;                                               \
  } while (0)
#else
#define TINY_STAMP(slot) ((void)0)
#endif
  // Targets sorted by x (once per call): keys = (order-preserving bits of fl32(x - cx), index), bitonic
  // sort of the next power of two in LDS.  A sweep then visits only targets whose x lies within the
  // current best distance of the query's -- a few of them instead of all m (sweep and prune; exact:
  // a target with |dx| > sqrt(best) is strictly farther).
  {
    unsigned long long *keys = &sbuf[0][0][0];  // 4096 slots: room for 2048 keys
    unsigned P = 64;
    while (P < m) P <<= 1;
    for (unsigned k = tid; k < P; k += B) {
      unsigned long long key = ~0ull;
      if (k < m) {
        const unsigned u = __float_as_uint((float)(dst[(size_t)k * DIM] - cx));
        const unsigned o = (u >> 31) ? ~u : (u | 0x80000000u);
        key = ((unsigned long long)o << 32) | k;
      }
      keys[k] = key;
    }
    __syncthreads();
    for (unsigned kk = 2; kk <= P; kk <<= 1)
      for (unsigned j = kk >> 1; j > 0; j >>= 1) {
        for (unsigned t = tid; t < (P >> 1); t += B) {
          const unsigned i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
          const unsigned long long a = keys[i], c = keys[l];
          const bool up = (i & kk) == 0;
          if ((a > c) == up) {
            keys[i] = c;
            keys[l] = a;
          }
        }
        __syncthreads();
      }
    for (unsigned j = tid; j < mp + 4; j += B) {
      if (j < m) {
        const unsigned k = (unsigned)(keys[j] & 0xffffffffull);
        const double x = dst[(size_t)k * DIM], y = dst[(size_t)k * DIM + 1];
        const double z = DIM == 3 ? dst[(size_t)k * DIM + 2] : 0.;
        tx[j] = x;
        ty[j] = y;
        if (DIM == 3) tz[j] = z;
        g4[j] = make_float4((float)(x - cx), (float)(y - cy), DIM == 3 ? (float)(z - cz) : 0.f, __uint_as_float(k));
      } else {  // pads: beyond every bound
        g4[j] = make_float4(__builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf(), __uint_as_float(0xffffffffu));
      }
    }
    __syncthreads();
  }
  double px = 0., py = 0., pz = 0.;
  if (has) {
    px = src[(size_t)tid * DIM];
    py = src[(size_t)tid * DIM + 1];
    if (DIM == 3) pz = src[(size_t)tid * DIM + 2];
  }
  if (tid == 0) {
    C->T = T0;
    C->nan = C->bail = 0;
    C->evals = C->sorted = 0;
    S->wvalid[0] = S->wvalid[1] = 0u;  // (no window yet: the first selections sample)
  }
  if (tid >= B / 64 && tid < 16) {  // the wave sums of the waves a smaller workgroup does not have
#pragma unroll
    for (int q = 0; q < kNSum + 1; ++q) sm[tid][q] = 0.;
  }

  __syncthreads();
  const int blocks = n > 512u ? 2 : 1;  // reduce_geometry(n) for n <= 1024: 512-thread blocks
  const unsigned lo_rank = (n - 1) / 2, hi_rank = n / 2;
  unsigned bi = 0xffffffffu;
  TINY_STAMP(0);
  for (unsigned it = 0; it < max_iter; ++it) {
    const Pose T = C->T;
    // ---- transform + exact nearest neighbour (src/lib.rs:113-124 / 156-167) ----
    double ax = 0., ay = 0., bx = 0., by = 0.;
    if (has) {
      const double qx = (T.r00 * px + T.r01 * py) + T.tx;  // Transform::transform, src/transform.rs:22-24
      const double qy = (T.r10 * px + T.r11 * py) + T.ty;
      const double qz = pz;
      const double ox = qx - cx, oy = qy - cy, oz = DIM == 3 ? qz - cz : 0.;
      const float hx = (float)ox, hy = (float)oy, hz = (float)oz;
      const double ec = (fmax(fmax(fabs(ox), fabs(oy)), fabs(oz)) + 2. * scale) * 1.2e-7 * 1.7320508075688774;
      double best = __builtin_huge_val();
      float thr = __builtin_huge_valf();
      unsigned nb = 0xffffffffu, nbo = 0xffffffffu;  // sorted position / original index of the best so far
      auto exact = [&](unsigned j, unsigned orig) {
        const double dx = qx - tx[j], dy = qy - ty[j];
        double d = dx * dx + dy * dy;
        if (DIM == 3) {
          const double dz = qz - tz[j];
          d = d + dz * dz;
        }
        if (d < best || (d == best && orig < nbo)) {  // ties -> lowest ORIGINAL index
          best = d;
          nb = j;
          nbo = orig;
          const double rr = sqrt(d) + ec;
          thr = (float)(rr * rr * 1.000004) * 1.000001f + 1e-37f;  // rounded up (nn_brute.hip)
        }
      };
      // start: the previous match (warm), else the first target at or right of the query's x
      unsigned start;
      if (bi != 0xffffffffu) {
        start = bi;
        exact(bi, __float_as_uint(g4[bi].w));
      } else {
        unsigned lo = 0, hi = m;
        while (lo < hi) {
          const unsigned mid = (lo + hi) >> 1;
          if (g4[mid].x < hx) lo = mid + 1;
          else hi = mid;
        }
        start = lo < m ? lo : m - 1;
      }
      // outwards in both directions while a target's x alone does not rule it out.  The f32 x
      // difference is within ec of the true one, so (|dx| - ec)^2 > best is what rules out; thr already
      // carries that margin: dx^2 > thr  =>  strictly farther.
      auto visit = [&](const float4 g, unsigned j) {  // (beyond the x bound: s2 > thr as well)
        const float fx = hx - g.x, fy = hy - g.y;
        float s2 = __builtin_fmaf(fy, fy, fx * fx);
        if (DIM == 3) {
          const float fz = hz - g.z;
          s2 = __builtin_fmaf(fz, fz, s2);
        }
        if (!(s2 > thr) && j < m) exact(j, __float_as_uint(g.w));
      };
      // four targets per step (their LDS reads in flight together: one CU has little else to hide the
      // latency with, and a wave is as slow as its lane with the widest window)
      for (unsigned j = start; j < m; j += 4) {  // (g4 carries four +inf pads past mp)
        const float4 g0 = g4[j], g1 = g4[j + 1], g2 = g4[j + 2], g3 = g4[j + 3];
        const float f0 = hx - g0.x;
        if (f0 * f0 > thr) break;  // sorted by x: everything further right is farther still
        visit(g0, j);
        visit(g1, j + 1);
        visit(g2, j + 2);
        visit(g3, j + 3);
      }
      for (unsigned j = start; j > 0;) {
        const unsigned j0 = j - 1, j1 = j > 1 ? j - 2 : 0, j2 = j > 2 ? j - 3 : 0, j3 = j > 3 ? j - 4 : 0;
        const float4 g0 = g4[j0], g1 = g4[j1], g2 = g4[j2], g3 = g4[j3];  // (a repeated target is harmless)
        const float f0 = hx - g0.x;
        if (f0 * f0 > thr) break;
        visit(g0, j0);
        visit(g1, j1);
        visit(g2, j2);
        visit(g3, j3);
        j = j3;
      }
      bi = nb;
      ax = qx;
      ay = qy;
      if (nb != 0xffffffffu) {
        bx = tx[nb];
        by = ty[nb];
      } else {  // no finite distance (NaN query): index 0, as a scan from 0 would
        bx = dst[0];
        by = dst[1];
        nbo = 0;
      }
      if (idx_out && it + 1 == max_iter) idx_out[tid] = nbo;
    }
    // ---- estimate_transform, src/lib.rs:59-84 ----
    if (tid == 0) {
      C->Ti = transform_identity();
      C->done = n < 2u ? 1 : 0;  // check_input_size, src/lib.rs:186-189
      C->applied = 0;
    }
    double prev_error = 1.7976931348623157e308;  // f64::MAX (thread 0 only)
    __syncthreads();
    TINY_STAMP(1);
    for (int k = 0; k < ICP_INNER_MAX_ITER && !C->done; ++k) {
      const Pose Ti = C->Ti;
      double r0 = 0., r1 = 0.;
      if (has) {  // residual(), src/lib.rs:34-36
        r0 = ((Ti.r00 * ax + Ti.r01 * ay) + Ti.tx) - bx;
        r1 = ((Ti.r10 * ax + Ti.r11 * ay) + Ti.ty) - by;
        if ((r0 != r0) | (r1 != r1)) C->nan = 1;
      }
      double med[2], sig[2];
      // medians, then MADs (src/stats.rs:11-47)
#ifdef ICP_TINY_PROFILE
      unsigned long long *selp = tsel;
#else
      unsigned long long *selp = nullptr;
#endif
      const unsigned long long km0 = has ? f2k(r0) : ~0ull, km1 = has ? f2k(r1) : ~0ull;
      bool ok = tiny_select_window<B>(km0, km1, has, n, S, 0, selp) || tiny_select<B>(km0, km1, has, n, S, sbuf[0], selp, 0);
      if (ok) {
        med[0] = middle_of_host(n, S->out[0][0], S->out[0][1]);
        med[1] = middle_of_host(n, S->out[1][0], S->out[1][1]);
        // (S->out is next written behind the first barriers of the next selection)
        const unsigned long long kd0 = has ? f2k(fabs(r0 - med[0])) : ~0ull, kd1 = has ? f2k(fabs(r1 - med[1])) : ~0ull;
        ok = tiny_select_window<B>(kd0, kd1, has, n, S, 1, selp) || tiny_select<B>(kd0, kd1, has, n, S, sbuf[0], selp, 1);
        if (ok) {
          sig[0] = ICP_PPF34 * middle_of_host(n, S->out[0][0], S->out[0][1]);
          sig[1] = ICP_PPF34 * middle_of_host(n, S->out[1][0], S->out[1][1]);
        }
      }
      if (!ok) {  // (uniform: every thread saw the same list counts)
        if constexpr (B == 1024) {  // the sorting path of k_tiny_eval
          unsigned long long ka = has ? f2k(r0) : ~0ull, kb = has ? f2k(r1) : ~0ull;
          __syncthreads();
          bitonic_sort2_1024(ka, kb, sbuf);
          __syncthreads();
          sbuf[0][0][tid] = ka;
          sbuf[0][1][tid] = kb;
          __syncthreads();
          const double xl = k2f(sbuf[0][0][lo_rank]), xh = k2f(sbuf[0][0][hi_rank]);
          const double yl = k2f(sbuf[0][1][lo_rank]), yh = k2f(sbuf[0][1][hi_rank]);
          med[0] = (n & 1) ? xl : (xl + xh) / 2.;
          med[1] = (n & 1) ? yl : (yl + yh) / 2.;
          mad_ranks(sbuf[0][0], n, med[0], lo_rank, hi_rank, C->s_mad[0]);
          mad_ranks(sbuf[0][1], n, med[1], lo_rank, hi_rank, C->s_mad[1]);
          __syncthreads();
          sig[0] = ICP_PPF34 * ((n & 1) ? C->s_mad[0][0] : (C->s_mad[0][0] + C->s_mad[0][1]) / 2.);
          sig[1] = ICP_PPF34 * ((n & 1) ? C->s_mad[1][0] : (C->s_mad[1][0] + C->s_mad[1][1]) / 2.);
          if (tid == 0) ++C->sorted;
        } else {  // the sort is written for 1024 threads: hand the call back, the host-driven path serves
          med[0] = med[1] = sig[0] = sig[1] = 0.;
          if (tid == 0) C->bail = 1;
        }
      }
      TINY_STAMP(2);
      // weighted normal equations + Huber error (src/lib.rs:238-255, 45-50), one point per thread
      // the tree of reduce_geometry(n), exactly as k_tiny_eval folds it -- one dimension's sums at a time (half
      // the registers of all nineteen at once; the wave trees of different sums are independent)
      if ((unsigned)wave * 64u < n) {
        static_assert(kNSum == 19, "9 + 9 + 1");
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          double half[10];
#pragma unroll
          for (int q = 0; q < 10; ++q) half[q] = 0.;
          if (has) {
            accumulate_dim<false>(j, make_double2(ax, ay), j ? r1 : r0, Ti, half);
            if (j == 0) accumulate_rho<false>(r0, r1, &half[9]);
          }
          wave_tree<10>(half);
          if ((tid & 63) == 0) {
#pragma unroll
            for (int q = 0; q < 9; ++q) sm[wave][9 * j + q] = half[q];
            sm[wave][18 + j] = half[9];  // (the error; the pad sums to +0.0)
          }
        }
      } else if ((tid & 63) == 0) {
#pragma unroll
        for (int q = 0; q < kNSum + 1; ++q) sm[wave][q] = 0.;
      }
      __syncthreads();
      if (tid < 2 * (kNSum + 1)) {
        const int vb = tid / (kNSum + 1), q = tid % (kNSum + 1);
        double v = sm[8 * vb][q];
        for (int w = 1; w < 8; ++w) v = v + sm[8 * vb + w][q];
        part[vb][q] = v;
      }
      __syncthreads();
      TINY_STAMP(3);
      // the last level of the tree and g_x S_x + g_y S_y, one thread per entry (rows 0 and 1 of `sm` are free until
      // the next evaluation's wave sums; thread 0 alone with arrays in scratch memory cost 2.7 us per evaluation)
      if (tid < kNSum) {
        const double p0 = (0. + part[0][tid]) + 0.;
        const double p1 = blocks > 1 ? (0. + part[1][tid]) + 0. : 0.;
        sm[0][tid] = (p0 + p1) + 0.;
      }
      __syncthreads();
      if (tid < kNAcc) sm[1][tid] = combine_sum(sm[0], (int)tid, sig);
      __syncthreads();
      if (tid == 0) {
        const double *tot = sm[1];
        ++C->evals;
        double delta[3];
        if (C->nan | C->bail) {
          C->done = 1;
        } else if (!solve_update(tot, tot + 9, delta)) {
          C->done = 1;  // None, src/lib.rs:67-69
        } else if ((delta[0] * delta[0] + delta[1] * delta[1]) + delta[2] * delta[2] < ICP_DELTA_NORM_THRESHOLD) {
          C->done = 1;  // src/lib.rs:71-73
        } else if (tot[12] > prev_error) {
          C->done = 1;  // src/lib.rs:75-78
        } else {
          prev_error = tot[12];
          bool in_range;
          const Pose D = transform_new_in_range(delta, &in_range);
          if (!in_range) {
            C->bail = 1;  // a rotation beyond the restated range of sin / cos: the host-driven path serves
            C->done = 1;
          } else {
            C->Ti = transform_mul(D, Ti);  // src/lib.rs:81
            ++C->applied;
          }
        }
      }
      __syncthreads();
      TINY_STAMP(4);
    }
    if (tid == 0) {
      if (inner_out) inner_out[it] = C->applied;
      C->T = transform_mul(C->Ti, T);  // src/lib.rs:127, 170
      // An outer iteration that leaves the pose as it found it, bit for bit, is a fixed point of the loop: every
      // later iteration repeats it (correspondences and updates are functions of the pose and the two clouds).  Only
      // the last one still runs -- it is the one that reports the correspondences.
      const Pose &Tn = C->T;
      C->fixed = C->applied == 0 && __double_as_longlong(Tn.tx) == __double_as_longlong(T.tx) &&
                 __double_as_longlong(Tn.ty) == __double_as_longlong(T.ty) &&
                 __double_as_longlong(Tn.r00) == __double_as_longlong(T.r00) &&
                 __double_as_longlong(Tn.r01) == __double_as_longlong(T.r01) &&
                 __double_as_longlong(Tn.r10) == __double_as_longlong(T.r10) &&
                 __double_as_longlong(Tn.r11) == __double_as_longlong(T.r11);
      if (C->fixed && it + 2 < max_iter && inner_out)
        for (unsigned k = it + 1; k + 1 < max_iter; ++k) inner_out[k] = 0;
    }
    __syncthreads();
    if (C->nan | C->bail) break;
    if (C->fixed && it + 2 < max_iter) it = max_iter - 2;  // (uniform: the flag is the workgroup's)
  }
  if (tid == 0) {
    res->pose = C->T;
    res->evals = C->evals;
    res->sorted = C->sorted;
    res->status = C->nan ? 3 : (C->bail ? -1 : 0);
#ifdef ICP_TINY_PROFILE
    tp[5] = __builtin_amdgcn_s_memtime() - t_begin;
    for (int q = 0; q < 6; ++q) res->t[q] = tp[q];
    for (int q = 0; q < 8; ++q) res->ts[q] = tsel[q];
#endif
  }
}

static size_t tiny_lds_bytes(int dim, unsigned m) {
  const size_t mp = (m + 63u) & ~63u;
  size_t b = mp * (size_t)dim * sizeof(double) + (mp + 4) * sizeof(float4) + 16;
  b += sizeof(unsigned long long) * 2 * 2 * 1024;
  b += (sizeof(TinySel) + 15) & ~size_t(15);
  b += sizeof(double) * 18 * (kNSum + 1);
  b += 512;  // Ctl
  return b;
}

// Icp{2,3}d::estimate for a small cloud in one launch.  *status: 0 done, 3 NaN, -1 not served (too large,
// no bounding box, disabled, or the kernel handed the call back): the caller runs the general path.
hipError_t launch_tiny_estimate(icp_handle *h, const double *d_src, size_t n, const Pose &T0, size_t max_iter,
                                Pose *out, uint32_t *d_last_idx, uint32_t *inner_iters, int *status) {
  static const bool off = exp_env("ICP_NO_TINY_ESTIMATE") != nullptr;
  *status = -1;
  if (off || !h->single_launch || n < 1 || n > kTinyMaxN || h->m < 1 || h->m > kTinyMaxM || max_iter < 1 || max_iter > kTinyMaxIter ||
      !h->grid.built || h->nn_mode == ICP_NN_GRID)
    return hipSuccess;
  Workspace &w = h->ws;
  hipError_t e;
  // the 160 KB of dynamic LDS are granted once per process; if the runtime refuses, small clouds simply take the
  // general path (*status stays -1) -- they must never launch with an LDS size that was not granted
  static int lds_granted = 0;  // 0 not asked yet, 1 yes, -1 refused
  if (lds_granted < 0) return hipSuccess;
  if (lds_granted == 0) {
    const void *kernels[] = {
        reinterpret_cast<const void *>(&k_tiny_estimate<2, 512>),  reinterpret_cast<const void *>(&k_tiny_estimate<2, 768>),
        reinterpret_cast<const void *>(&k_tiny_estimate<2, 1024>), reinterpret_cast<const void *>(&k_tiny_estimate<3, 512>),
        reinterpret_cast<const void *>(&k_tiny_estimate<3, 768>),  reinterpret_cast<const void *>(&k_tiny_estimate<3, 1024>)};
    lds_granted = 1;
    for (const void *k : kernels)
      if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess) lds_granted = -1;
    if (lds_granted < 0) {
      (void)hipGetLastError();
      return hipSuccess;
    }
  }
  if (!w.h_tiny &&
      (e = hipHostMalloc(&w.h_tiny, sizeof(TinyResult) + kTinyMaxIter * sizeof(uint32_t), hipHostMallocDefault)) != hipSuccess)
    return e;
  TinyResult *res = reinterpret_cast<TinyResult *>(w.h_tiny);
  uint32_t *inner = reinterpret_cast<uint32_t *>(res + 1);
  const GridParams &g = h->grid.p;
  const double cx = 0.5 * (g.lo[0] + g.hi[0]), cy = 0.5 * (g.lo[1] + g.hi[1]), cz = 0.5 * (g.lo[2] + g.hi[2]);
  const size_t lds = tiny_lds_bytes(h->dim, (unsigned)h->m);
  // the smallest workgroup with a thread per source point: fewer waves per barrier, and registers
  // enough (1024 threads leave 128 per thread, and spill)
  static const unsigned forced = exp_env("ICP_TINY_THREADS") ? (unsigned)atoi(exp_env("ICP_TINY_THREADS")) : 0u;
  unsigned threads = n <= 512 ? 512u : (n <= 768 ? 768u : 1024u);
  if ((forced == 768u || forced == 1024u) && forced >= threads) threads = forced;
#define ICP_TINY_LAUNCH(D, BB)                                                                                       \
  hipLaunchKernelGGL((k_tiny_estimate<D, BB>), dim3(1), dim3(BB), lds, h->stream, d_src, (unsigned)n, h->d_dst, \
                     (unsigned)h->m, T0, (unsigned)max_iter, cx, cy, cz, g.scale, res, inner, d_last_idx)
  if (h->dim == 3) {
    if (threads == 512u) ICP_TINY_LAUNCH(3, 512);
    else if (threads == 768u) ICP_TINY_LAUNCH(3, 768);
    else ICP_TINY_LAUNCH(3, 1024);
  } else {
    if (threads == 512u) ICP_TINY_LAUNCH(2, 512);
    else if (threads == 768u) ICP_TINY_LAUNCH(2, 768);
    else ICP_TINY_LAUNCH(2, 1024);
  }
#undef ICP_TINY_LAUNCH
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return e;
  *status = res->status;
  if (res->status == 0) {
    *out = res->pose;
    if (inner_iters)
      for (size_t i = 0; i < max_iter; ++i) inner_iters[i] = inner[i];
    w.tiny_evals += res->evals;
    w.tiny_sorted += res->sorted;
#ifdef ICP_TINY_PROFILE
    if (exp_env("ICP_TINY_PRINT"))
      fprintf(stderr, "[tiny] evals %u sorted %u; cycles: setup %llu search %llu select %llu sums %llu step %llu all %llu; "
              "selection phases (keys, sample ranks, buckets, scan+list, ranks): %llu %llu %llu %llu %llu; window selections "
              "(count + list, ranks): %llu %llu\n",
              res->evals, res->sorted, res->t[0], res->t[1], res->t[2], res->t[3], res->t[4], res->t[5], res->ts[0], res->ts[1],
              res->ts[2], res->ts[3], res->ts[4], res->ts[5], res->ts[6]);
#endif
  }
  return hipSuccess;
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
