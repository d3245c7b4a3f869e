// Device pieces of the window selection shared by gn_win.hip (launch per stage) and gn_loop.hip (the whole inner
// loop in one launch): the piecewise bin function, the bracket that turns counts into candidate bins, and the exact
// selection on candidates held in registers.  Moved out of gn_win.hip unchanged.
#pragma once
#include "common.hpp"
#include "gn_device.hpp"

namespace icp {

constexpr int kWinThreads = 512;  // workgroups small enough to be placed beside the search kernel's waves
constexpr int kWinBatch = 4;
constexpr int kWinAccBatch = 4;  // loads in flight per lane in A (2 saves 16 VGPRs but loses more than the better placement gains)
#ifndef ICP_SELECT_DIRECT
#define ICP_SELECT_DIRECT 256  // select_n: lists up to this long are ranked directly (128 until round 5)
#endif
constexpr int kSubBins = 1024;   // select_n: linear sub-bins over the candidates
constexpr int kSmallCap = 1024;  // select_n: keys ranked by counting, per dimension (a run of equal keys lands here)
constexpr size_t kWinMinN = 1u << 12;
constexpr size_t kWinMaxN = 1u << 22;

// first bin of each region
constexpr int kF0 = 1, kC0 = kF0 + kWinFine, kF1 = kC0 + kWinCoarse, kC1 = kF1 + kWinFine, kF2 = kC1 + kWinCoarse;
static_assert(kF2 + kWinFine == kWinBins - 1, "bin layout");

__device__ __forceinline__ unsigned region_bin(double off, double scale, int first, int count) {
  const unsigned k = (unsigned)(off * scale);  // off >= 0
  return (unsigned)first + (k < (unsigned)count ? k : (unsigned)(count - 1));
}

// Monotone non-decreasing in r: the regions are ordered, and inside a region it is two
// correctly rounded monotone operations, a floor and a clamp.
__device__ __forceinline__ unsigned wbin(double r, const WinDim &w) {
  if (!(r >= w.x[0])) return 0u;  // below (NaN residuals are reported through nan_flag)
  if (r >= w.x[5]) return (unsigned)(kWinBins - 1);
  if (r < w.x[1]) return region_bin(r - w.x[0], w.sf, kF0, kWinFine);
  if (r < w.x[2]) return region_bin(r - w.x[1], w.sc, kC0, kWinCoarse);
  if (r < w.x[3]) return region_bin(r - w.x[2], w.sf, kF1, kWinFine);
  if (r < w.x[4]) return region_bin(r - w.x[3], w.sc, kC1, kWinCoarse);
  return region_bin(r - w.x[4], w.sf, kF2, kWinFine);
}

// The same function (same operations on the same operands, hence the same bins) as selects of the region's
// origin / scale / first bin followed by ONE region_bin: a fifth of the code, for the places that run once per
// launch -- cold code is fetched while everybody waits ("code size is latency"); the
// streaming loops keep the branches above, whose compares read the window from scalar registers (this form
// copies it into 32 vector registers).
__device__ __forceinline__ unsigned wbin_cold(double r, const WinDim &w) {
  const bool g1 = r >= w.x[1], g2 = r >= w.x[2], g3 = r >= w.x[3], g4 = r >= w.x[4];
  const double base = g4 ? w.x[4] : (g3 ? w.x[3] : (g2 ? w.x[2] : (g1 ? w.x[1] : w.x[0])));
  const bool coarse = g1 != g2 || g3 != g4;  // regions 1 and 3
  const int first = g4 ? kF2 : (g3 ? kC1 : (g2 ? kF1 : (g1 ? kC0 : kF0)));
  const bool inside = r >= w.x[0] && !(r >= w.x[5]);
  const double off = inside ? r - base : 0.;
  const unsigned j = region_bin(off, coarse ? w.sc : w.sf, first, coarse ? kWinCoarse : kWinFine);
  if (!(r >= w.x[0])) return 0u;
  return r >= w.x[5] ? (unsigned)(kWinBins - 1) : j;
}

// wbin() for the streaming loops that are bound by instruction issue (k_win_hist_sums_bkt: 428 instructions per pair,
// 62 of them each wbin -- five exec-masked copies of region_bin behind a ladder of compares): the REGION comes from
// four compares added up, its origin / scale / first bin / last offset from a five-row table in LDS, and region_bin
// runs once.  Same operations on the same operands as wbin, hence the same bins (tests/test_gpu_gn_win.py compares
// every evaluation against the oracle's medians, which the bins decide).  `fine`: the bin is one of a fine window's.
struct WinRegion {
  double base, scale;
  unsigned first, last;  // first bin of the region, count - 1
  unsigned pad[2];
};
static_assert(sizeof(WinRegion) == 32, "one ds_read_b128 + one ds_read_b64");
__device__ __forceinline__ void win_region_table(const WinDim &w, WinRegion *tab, int r) {  // row r of 5
  const bool coarse = r & 1;
  WinRegion e;
  e.base = w.x[r];
  e.scale = coarse ? w.sc : w.sf;
  e.first = (unsigned)(r == 0 ? kF0 : r == 1 ? kC0 : r == 2 ? kF1 : r == 3 ? kC1 : kF2);
  e.last = (unsigned)((coarse ? kWinCoarse : kWinFine) - 1);
  e.pad[0] = e.pad[1] = 0u;
  tab[r] = e;
}
__device__ __forceinline__ unsigned wbin_tab(double r, const WinDim &w, const WinRegion *tab, bool &below, bool &above,
                                             bool &fine) {
  below = !(r >= w.x[0]);  // (NaN residuals are reported through nan_flag)
  above = r >= w.x[5];
  const unsigned reg = (unsigned)(r >= w.x[1]) + (unsigned)(r >= w.x[2]) + (unsigned)(r >= w.x[3]) + (unsigned)(r >= w.x[4]);
  const WinRegion e = tab[reg];
  const unsigned k = (unsigned)((r - e.base) * e.scale);
  fine = !(reg & 1u) & !below & !above;
  const unsigned j = e.first + (k < e.last ? k : e.last);
  return below ? 0u : (above ? (unsigned)(kWinBins - 1) : j);
}

// is regular bin j one of the three fine windows' (where every candidate of a window that holds lies)?
__device__ __forceinline__ bool fine_bin(unsigned j) {
  return (j - (unsigned)kF0 < (unsigned)kWinFine) | (j - (unsigned)kF1 < (unsigned)kWinFine) |
         (j - (unsigned)kF2 < (unsigned)kWinFine);
}

// lower edge of regular bin j (1 <= j <= kWinBins-1; the upper edge of j is the lower edge of j+1)
__device__ __forceinline__ double wedge(int j, const WinDim &w) {
  if (j >= kWinBins - 1) return w.x[5];
  const bool g1 = j >= kC0, g2 = j >= kF1, g3 = j >= kC1, g4 = j >= kF2;
  const double base = g4 ? w.x[4] : (g3 ? w.x[3] : (g2 ? w.x[2] : (g1 ? w.x[1] : w.x[0])));
  const int first = g4 ? kF2 : (g3 ? kC1 : (g2 ? kF1 : (g1 ? kC0 : kF0)));
  const bool coarse = g1 != g2 || g3 != g4;
  return base + (double)(j - first) / (coarse ? w.sc : w.sf);
}

// ---- C ------------------------------------------------------------------------------
// largest t in [lo, hi] with pred(t), for a predicate that holds on a prefix of the range
// (lo - 1 if nowhere); the 64 lanes of a wave probe 64 positions per round
template <typename F>
__device__ __forceinline__ int wave_last_true(int lo, int hi, F &&pred) {
  const int lane = threadIdx.x & 63;
  int best = lo - 1;
  while (lo <= hi) {
    const int step = (hi - lo + 64) >> 6;
    const int t = lo + lane * step;
    const bool ok = t <= hi && pred(t);
    const int cnt = __popcll(__ballot(ok));
    if (cnt == 0) break;
    best = lo + (cnt - 1) * step;
    const int nhi = best + step - 1 < hi ? best + step - 1 : hi;
    lo = best + 1;
    hi = nhi;
  }
  return best;
}

struct WinRanges {  // bins, per dimension
  int mlo, mhi;     // median candidates: [mlo, mhi]
  int a0, b1;       // ring: [a0, b1] without [i0, i1]
  int i0, i1;
};

// The bracket.  The median m lies in the bins [jlo, jhi] of the two middle ranks, i.e. in
// [mL, mU] = [lower edge of jlo, upper edge of jhi].  For a radius d
//   inside(d)   = bins entirely within [mU - d, mL + d]: every point has |r - m| <= d;
//   possible(d) = bins meeting [mL - d, mU + d]:        every OTHER point has |r - m| > d.
// With k the wanted ranks of the distances (0-based) and d_t = t fine bins,
// t1 = max{t: #possible(d_t) <= klo} and t2 = min{t: #inside(d_t) > khi}: at most klo distances
// are <= d_t1, so both order statistics are > d_t1, and more than khi distances are <= d_t2, so
// they are <= d_t2.  With a margin q of a quarter fine bin, I = inside(d_t1 - 2q) holds only
// points closer than the order statistics by more than q, and every point outside
// possible(d_t2 + 2q) is farther by more than q; ranking the ring possible(d_t2 + 2q) \ I by
// exact distance behind |I| therefore yields the exact order statistics (fl(|r - m|) is
// monotone in the true distance, and q is ~2^38 times the rounding error of the bin function
// and of the edges -- the host refuses windows whose fine bins are narrower than 1e-10 of the
// coordinates -- so neither rounding nor the evaluation of edges can create ties or misplace
// a bin; the bins probed are additionally shrunk / widened by q).
struct WinGeom {  // what both halves of the bracket search need, per dimension
  int jlo, jhi;
  double mL, mU, fine, q;
  int tmax;
  bool ok;
};

// bins of the two middle ranks and the median's interval
// (jlo_known / jhi_known >= 0: the bins that hold ranks (n - 1) / 2 and n / 2, found by whoever made the cumulative
// counts -- the thread whose bin contains the rank -- instead of by two searches over them)
__device__ __forceinline__ WinGeom window_geometry(const uint32_t *c, unsigned n, const WinDim &w, int jlo_known = -1,
                                                   int jhi_known = -1) {
  auto C = [&](int j) -> unsigned { return j >= kWinBins ? n : c[j]; };  // points in bins < j
  const unsigned klo = (n - 1) / 2, khi = n / 2;                          // src/stats.rs:18-27
  WinGeom g;
  if (jlo_known >= 0 && jhi_known >= 0) {
    g.jlo = jlo_known;
    g.jhi = jhi_known;
  } else {
    g.jlo = wave_last_true(0, kWinBins - 1, [&](int j) { return c[j] <= klo; });
    // khi is klo or klo + 1: almost always the same bin
    g.jhi = (C(g.jlo + 1) > khi) ? g.jlo : wave_last_true(0, kWinBins - 1, [&](int j) { return c[j] <= khi; });
  }
  g.ok = !(g.jlo < 1 || g.jhi > kWinBins - 2);  // else: a middle rank outside the windows
  g.mL = g.ok ? wedge(g.jlo, w) : 0.;
  g.mU = g.ok ? wedge(g.jhi + 1, w) : 0.;
  g.fine = 1. / w.sf;
  g.q = 0.25 * g.fine;
  g.tmax = (int)((w.x[5] - w.x[0]) * w.sf) + 2;
  return g;
}

__device__ __forceinline__ void inside_bins(const WinGeom &g, const WinDim &w, double d, int &s, int &e) {
  s = (int)wbin_cold(g.mU - d + g.q, w) + 1;  // [s, e): regular bins only
  e = (int)wbin_cold(g.mL + d - g.q, w);
}
__device__ __forceinline__ void possible_bins(const WinGeom &g, const WinDim &w, double d, int &s, int &e) {
  s = (int)wbin_cold(g.mL - d - g.q, w);      // [s, e): may include the catch-all bins
  e = (int)wbin_cold(g.mU + d + g.q, w) + 1;
}

// one half of the bracket: role 0 -> t1 = max{t: #possible(d_t) <= klo}, role 1 -> t2 = min{t: #inside(d_t) > khi}
__device__ __forceinline__ int bracket_search(const uint32_t *c, unsigned n, const WinDim &w, const WinGeom &g,
                                              int role) {
  auto C = [&](int j) -> unsigned { return j >= kWinBins ? n : c[j]; };
  const unsigned klo = (n - 1) / 2, khi = n / 2;
  if (role == 0)
    return wave_last_true(0, g.tmax, [&](int t) {
      int s, e;
      possible_bins(g, w, (double)t * g.fine, s, e);
      return C(e) - C(s) <= klo;
    });
  return wave_last_true(0, g.tmax, [&](int t) {
           int s, e;
           inside_bins(g, w, (double)t * g.fine, s, e);
           return (s < e ? C(e) - C(s) : 0u) <= khi;
         }) + 1;
}

__device__ __forceinline__ bool resolve_window(const uint32_t *c, unsigned n, const WinDim &w, const WinGeom &g,
                                               int t1, int t2, WinRanges &R, unsigned &med_base,
                                               unsigned &med_cnt, unsigned &inner, unsigned &ring_cnt,
                                               double (&range)[4]) {
  auto C = [&](int j) -> unsigned { return j >= kWinBins ? n : c[j]; };
  if (!g.ok || t1 < 0 || t2 > g.tmax) return false;
  const int jlo = g.jlo, jhi = g.jhi;
  const double fine = g.fine, q = g.q;
  int is, ie, ps, pe;
  inside_bins(g, w, (double)t1 * fine - 2. * q, is, ie);
  possible_bins(g, w, (double)t2 * fine + 2. * q, ps, pe);
  if (ps < 1 || pe > kWinBins - 1) return false;  // the ring reaches a catch-all bin
  R.mlo = jlo;
  R.mhi = jhi;
  R.a0 = ps;
  R.b1 = pe - 1;
  if (is < ie) {
    R.i0 = is;
    R.i1 = ie - 1;
    inner = C(ie) - C(is);
  } else {  // nothing is surely inside
    R.i0 = 1;
    R.i1 = 0;
    inner = 0u;
  }
  med_base = c[jlo];
  med_cnt = C(jhi + 1) - c[jlo];
  ring_cnt = (C(pe) - C(ps)) - inner;
  range[0] = g.mL;  // every median candidate lies in [mL, mU]
  range[1] = g.mU;
  range[2] = t1 > 0 ? (double)(t1 - 1) * fine : 0.;  // the MAD lies in (t1, t2] fine bins
  range[3] = (double)(t2 + 1) * fine;
  return med_cnt <= (unsigned)kWinCapMed && ring_cnt <= (unsigned)kWinCapRing;
}

// ---- A ------------------------------------------------------------------------------
// exclusive prefix of v over the 512 threads of the workgroup; *total = the sum
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned *total) {
  __shared__ unsigned s_w[kReduceThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned s = wave_scan_inclusive(v);
  if (lane == 63) s_w[wave] = s;
  __syncthreads();
  unsigned before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kReduceThreads / 64; ++w) {
    const unsigned x = s_w[w];
    all += x;
    before += (w < wave) ? x : 0u;
  }
  __syncthreads();
  *total = all;
  return before + s - v;
}

// ND (1 or 2) exact selections at once, by the whole workgroup, on candidates that stay in
// registers: thread t holds candidates t, t + 512, ... of the dense list of dimension d, cnt[d]
// in all.  Wanted: the keys of ranks rlo[d] <= rhi[d] <= rlo[d] + 1 among them.  Linear sub-bins
// over [lo[d], hi[d]] (monotone in the value; values outside clamp to the end bins) locate the
// few keys around the ranks; those go to LDS and are ranked by counting on their
// order-preserving keys.  fail (uniform): a rank outside the list, or more than kSmallCap keys
// in the wanted sub-bins.
template <int ND>
struct SelectLds {  // the LDS a selection works in (a caller that is short of LDS overlays it on something idle)
  unsigned hist[ND][kSubBins];
  unsigned long long small_keys[ND][kSmallCap];
  unsigned long long out[ND][2];
  unsigned nsmall[ND], sb[ND][2], below[ND];
};

template <int ND, int NV, int DIRECT = ICP_SELECT_DIRECT>
__device__ __forceinline__ void select_n_lds(const double (&v)[ND][NV], const unsigned (&cnt)[ND],
                                             const double (&lo)[ND], const double (&hi)[ND],
                                             const long long (&rlo)[ND], const long long (&rhi)[ND],
                                             unsigned long long (&out)[ND][2], bool &fail, SelectLds<ND> &S) {
  static_assert(ND == 1 || ND == 2, "one or two lists");
  static_assert(kSubBins == 2 * kReduceThreads, "geometry");
  auto &s_hist = S.hist;
  auto &s_small = S.small_keys;
  auto &s_nsmall = S.nsmall;
  auto &s_sb = S.sb;
  auto &s_below = S.below;
  auto &s_out = S.out;
  const unsigned tid = threadIdx.x;
#pragma unroll
  for (int d = 0; d < ND; ++d) out[d][0] = out[d][1] = 0;
#pragma unroll
  for (int d = 0; d < ND; ++d)
    if (rlo[d] < 0 || rhi[d] >= (long long)cnt[d] || rhi[d] < rlo[d] || cnt[d] > 0xffffu) fail = true;
  if (fail) return;  // uniform
  // a handful of candidates (clouds of tens of thousands of points: a fine bin holds one or two): rank
  // them against each other directly -- two barriers instead of the sub-bin machinery's seven
  constexpr unsigned kDirect = DIRECT;
  static_assert(DIRECT <= kReduceThreads && DIRECT <= kSmallCap, "one candidate per thread");
  bool direct = true;
#pragma unroll
  for (int d = 0; d < ND; ++d) direct = direct && cnt[d] <= kDirect;
  if (direct) {  // uniform
#pragma unroll
    for (int d = 0; d < ND; ++d)
      if (tid < cnt[d]) s_small[d][tid] = f2k(v[d][0]);
    __syncthreads();
    if (ND == 2 && kDirect <= kReduceThreads / 2) {
      // round 5: ONE candidate of ONE dimension per thread (threads 0 .. 255: x, 256 .. 511: y) -- both lists ranked side
      // by side, each thread one pass over its list with the reads four ahead (the two dimensions one after the other,
      // read by read, was 2 x 128 dependent LDS trips per thread: 3 us of every finishing launch and 3.4 us of every
      // evaluation of the one-launch loop; lists of up to 256 are now cheaper this way than through the sub-bins)
      const unsigned d = tid >> 8, i = tid & 255u;
      const unsigned c = d ? cnt[ND - 1] : cnt[0];
      if (i < c) {
        const unsigned long long *list = s_small[d];
        const unsigned long long ki = list[i];
        const unsigned rl = (unsigned)(d ? rlo[ND - 1] : rlo[0]), rh = (unsigned)(d ? rhi[ND - 1] : rhi[0]);
        unsigned less = 0, eq = 0;
        unsigned j = 0;
        for (; j + 4 <= c; j += 4) {
          const unsigned long long k0 = list[j], k1 = list[j + 1], k2 = list[j + 2], k3 = list[j + 3];
          less += (k0 < ki) + (k1 < ki) + (k2 < ki) + (k3 < ki);
          eq += (k0 == ki) + (k1 == ki) + (k2 == ki) + (k3 == ki);
        }
        for (; j < c; ++j) {
          const unsigned long long kj = list[j];
          less += kj < ki;
          eq += kj == ki;
        }
        if (less <= rl && rl < less + eq) s_out[d][0] = ki;
        if (less <= rh && rh < less + eq) s_out[d][1] = ki;
      }
    } else {
#pragma unroll
      for (int d = 0; d < ND; ++d)
        if (tid < cnt[d]) {
          const unsigned long long ki = s_small[d][tid];
          const unsigned rl = (unsigned)rlo[d], rh = (unsigned)rhi[d];
          unsigned less = 0, eq = 0;
          for (unsigned j = 0; j < cnt[d]; ++j) {
            const unsigned long long kj = s_small[d][j];
            less += kj < ki;
            eq += kj == ki;
          }
          if (less <= rl && rl < less + eq) s_out[d][0] = ki;
          if (less <= rh && rh < less + eq) s_out[d][1] = ki;
        }
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      out[d][0] = s_out[d][0];
      out[d][1] = s_out[d][1];
    }
    __syncthreads();  // the next call reuses the LDS
    return;
  }
  for (unsigned i = tid; i < (unsigned)ND * kSubBins; i += kReduceThreads) (&s_hist[0][0])[i] = 0;
  if (tid < (unsigned)ND) {
    s_nsmall[tid] = 0;
    s_sb[tid][0] = s_sb[tid][1] = 0;
    s_below[tid] = 0;
    s_out[tid][0] = s_out[tid][1] = 0;
  }
  __syncthreads();
  double scale[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) scale[d] = hi[d] > lo[d] ? (double)(kSubBins - 1) / (hi[d] - lo[d]) : 0.;
  auto sub = [&](int d, double x) -> unsigned {
    const double t = (x - lo[d]) * scale[d];  // monotone in x
    const unsigned sb = t > 0. ? (unsigned)t : 0u;
    return sb < (unsigned)kSubBins ? sb : (unsigned)(kSubBins - 1);
  };
  unsigned sbv[ND][NV];
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      sbv[d][u] = sub(d, v[d][u]);
      if (tid + u * kReduceThreads < cnt[d]) atomicAdd(&s_hist[d][sbv[d][u]], 1u);
    }
  __syncthreads();
  // one scan for both dimensions: 16 bits each (counts <= 65535, checked above)
  const unsigned c0 = s_hist[0][2 * tid] | (ND == 2 ? s_hist[ND - 1][2 * tid] << 16 : 0u);
  const unsigned c1 = s_hist[0][2 * tid + 1] | (ND == 2 ? s_hist[ND - 1][2 * tid + 1] << 16 : 0u);
  unsigned total;
  const unsigned e0 = block_excl_scan(c0 + c1, &total), e1 = e0 + c0;
#pragma unroll
  for (int d = 0; d < ND; ++d) {
    const int sh = 16 * d;
    const unsigned E0 = (e0 >> sh) & 0xffffu, C0 = (c0 >> sh) & 0xffffu;
    const unsigned E1 = (e1 >> sh) & 0xffffu, C1 = (c1 >> sh) & 0xffffu;
    const unsigned rl = (unsigned)rlo[d], rh = (unsigned)rhi[d];
    if (E0 <= rl && rl < E0 + C0) {
      s_sb[d][0] = 2 * tid;
      s_below[d] = E0;
    }
    if (E1 <= rl && rl < E1 + C1) {
      s_sb[d][0] = 2 * tid + 1;
      s_below[d] = E1;
    }
    if (E0 <= rh && rh < E0 + C0) s_sb[d][1] = 2 * tid;
    if (E1 <= rh && rh < E1 + C1) s_sb[d][1] = 2 * tid + 1;
  }
  __syncthreads();
#pragma unroll
  for (int d = 0; d < ND; ++d) {
    const unsigned sb_lo = s_sb[d][0], sb_hi = s_sb[d][1];
#pragma unroll
    for (int u = 0; u < NV; ++u)
      if (tid + u * kReduceThreads < cnt[d] && sbv[d][u] >= sb_lo && sbv[d][u] <= sb_hi) {
        const unsigned pos = atomicAdd(&s_nsmall[d], 1u);
        if (pos < (unsigned)kSmallCap) s_small[d][pos] = f2k(v[d][u]);
      }
  }
  __syncthreads();
  if (s_nsmall[0] > (unsigned)kSmallCap || s_nsmall[ND - 1] > (unsigned)kSmallCap) {
    fail = true;  // uniform
    return;
  }
#pragma unroll
  for (int d = 0; d < ND; ++d) {
    const unsigned ns = s_nsmall[d];
    const unsigned rl = (unsigned)rlo[d], rh = (unsigned)rhi[d];
    for (unsigned i = tid; i < ns; i += kReduceThreads) {
      const unsigned long long ki = s_small[d][i];
      unsigned less = s_below[d], eq = 0;
      for (unsigned j = 0; j < ns; ++j) {
        const unsigned long long kj = s_small[d][j];
        less += kj < ki;
        eq += kj == ki;
      }
      if (less <= rl && rl < less + eq) s_out[d][0] = ki;
      if (less <= rh && rh < less + eq) s_out[d][1] = ki;
    }
  }
  __syncthreads();
#pragma unroll
  for (int d = 0; d < ND; ++d) {
    out[d][0] = s_out[d][0];
    out[d][1] = s_out[d][1];
  }
  __syncthreads();  // the next call reuses the LDS
}

template <int ND, int NV>
__device__ __forceinline__ void select_n(const double (&v)[ND][NV], const unsigned (&cnt)[ND],
                                         const double (&lo)[ND], const double (&hi)[ND],
                                         const long long (&rlo)[ND], const long long (&rhi)[ND],
                                         unsigned long long (&out)[ND][2], bool &fail) {
  __shared__ SelectLds<ND> S;
  select_n_lds<ND, NV>(v, cnt, lo, hi, rlo, rhi, out, fail, S);
}

__device__ __forceinline__ double middle_of(unsigned n, unsigned long long klo, unsigned long long khi) {
  const double lo = k2f(klo), hi = k2f(khi);
  return (n & 1) ? lo : (lo + hi) / 2.;  // src/stats.rs:18-27
}

// what the histogram says about the candidate lists (every workgroup derives the same)
struct WinSel {
  unsigned med_base[2], med_cnt[2], inner[2], ring_cnt[2];
  double range[2][4];
};

// the bins of both dimensions for fine windows of half-width f * sigma around med and med -+ MAD
ICP_HD inline bool make_window_hd(const double med[2], const double sigma[2], double f, WinParams *P) {
  for (int d = 0; d < 2; ++d) {
    const double s = sigma[d], m = med[d];
    if (!(s > 0.) || !(s < 1e300) || !(fabs(m) < 1e300)) return false;
    const double mad = s / ICP_PPF34, hw = f * s;
    // the bracket's quarter-bin margins must dwarf the rounding of (r - x) * scale
    if (!(2. * hw / kWinFine > 1e-10 * (fabs(m) + mad + hw))) return false;
    WinDim &D = P->d[d];
    D.x[0] = m - mad - hw;
    D.x[1] = m - mad + hw;
    D.x[2] = m - hw;
    D.x[3] = m + hw;
    D.x[4] = m + mad - hw;
    D.x[5] = m + mad + hw;
    if (!(D.x[1] < D.x[2] && D.x[3] < D.x[4])) return false;
    D.sf = (double)kWinFine / (2. * hw);
    D.sc = (double)kWinCoarse / (D.x[2] - D.x[1]);
  }
  return true;
}

}  // namespace icp
