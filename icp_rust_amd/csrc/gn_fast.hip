// Short pipeline for one inner Gauss-Newton evaluation (src/lib.rs:218-261 + :45-50):
// 7 launches instead of 27, same results bit for bit.
//
//   median of r = T*a - b, per dimension (src/stats.rs:11-28)
//     H(pass 0)  12-bit histogram of the keys' top digit   -> last block descends one digit
//     H(pass 1)  next 12 bits among the survivors          -> last block descends again
//     C          survivors (same top 24 bits, typically ~10^2 of 10^6) are appended to a
//                candidate list; the last block ranks them in LDS -> exact order statistic
//   MAD = median of |r - median| (src/stats.rs:30-37): H, H, C again -> sigma = 1.4826*MAD
//   A          Huber-weighted normal equations + Huber error, fixed reduction tree; the last
//              block folds the block sums and hands 13 doubles to the host
//
// These kernels move 16-48 MB that sits in L2/MALL; they are latency-bound, not
// bandwidth-bound, so each lane issues a batch of independent loads before it touches
// any of them, the serial tail of a launch is spread over waves (one problem per wave), and
// the arrival tickets are sharded (gn_device.hpp).  "Last block" = the workgroup whose ticket
// is the final one; everything handed to it is written by device-scope atomics or sc1
// stores and read with sc1 loads.  Integer histograms and rank counting are exact and
// independent of arrival order; the sums use the same tree as gn.hip.  If a median sits in
// a run of more than kSelCap equal-prefix keys (heavy duplicates), the pipeline raises
// `overflow` and the host repeats the evaluation with the general radix path.
#include "common.hpp"
#include "gn_device.hpp"

namespace icp {

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int kFastThreads = 1024; // 16 waves, one workgroup (64 KB of LDS histograms) per CU
constexpr int kFastBatch = 4;      // independent elements in flight per lane
constexpr int kScanPad = kSelBins + kSelBins / 64;  // +1 word per 64 bins: conflict-free column reads

// MODE 0: r = T*a - b computed here (residual(), src/lib.rs:34-36) and stored as rx|ry
// MODE 1: keys of the stored residuals; MODE 2: keys of |r - median| (src/stats.rs:35)
template <int MODE, typename F>
__device__ __forceinline__ void for_each_value(const double2 *__restrict__ a, const double2 *__restrict__ b,
                                               const Pose &T, double *__restrict__ rx,
                                               double *__restrict__ ry, unsigned n, double med0, double med1,
                                               bool &saw_nan, F &&f) {
  const unsigned G = gridDim.x * blockDim.x;
  for (unsigned base = blockIdx.x * blockDim.x + threadIdx.x; base < n; base += G * kFastBatch) {
    double v0[kFastBatch], v1[kFastBatch];
    double2 s[kFastBatch], d[kFastBatch];
#pragma unroll
    for (int u = 0; u < kFastBatch; ++u) {
      const unsigned i = base + u * G;
      if (i < n) {
        if (MODE == 0) {
          s[u] = a[i];
          d[u] = b[i];
        } else {
          v0[u] = rx[i];
          v1[u] = ry[i];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kFastBatch; ++u) {
      const unsigned i = base + u * G;
      if (i < n) {
        if (MODE == 0) {
          v0[u] = ((T.r00 * s[u].x + T.r01 * s[u].y) + T.tx) - d[u].x;
          v1[u] = ((T.r10 * s[u].x + T.r11 * s[u].y) + T.ty) - d[u].y;
          rx[i] = v0[u];
          ry[i] = v1[u];
          saw_nan |= (v0[u] != v0[u]) | (v1[u] != v1[u]);
        } else if (MODE == 2) {
          v0[u] = fabs(v0[u] - med0);
          v1[u] = fabs(v1[u] - med1);
        }
        f(f2k(v0[u]), f2k(v1[u]));
      }
    }
  }
}

// Tail of a histogram launch, run by the last block (1024 threads).  Measured with in-kernel
// stamps, a one-wave-per-problem tail (64 dependent-ish sc1 loads per lane, then a serial
// 64-bin walk) took 6-16 us -- more than streaming the data.  So: (1) ALL threads fetch the
// live histograms together, 16 coalesced sc1 loads per lane, into a padded LDS image
// (bin + bin/64, so that lane l can sum bins [64 l, 64 l + 64) without bank conflicts);
// (2) wave p locates problem p's rank: column sums -> wave scan -> the owning lane, then the
// 64 lanes look at that lane's 64 bins in parallel (second wave scan) instead of walking them.
__device__ __forceinline__ void scan_descend(uint32_t *lds, const uint32_t *hist, SelState *sel,
                                             GnScalars *scal, int pass, bool check_cap) {
  __shared__ unsigned found_bin[kSelProblems], found_below[kSelProblems], found_cnt[kSelProblems];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the search state is read ONCE into registers (every later use of `sel` would be a dependent
  // global load of a line this kernel is about to rewrite: ~1-2 us each, serially, on one lane)
  SelState st[kSelProblems];
#pragma unroll
  for (int p = 0; p < kSelProblems; ++p) st[p] = sel[p];
  bool live[kSelProblems];
#pragma unroll
  for (int p = 0; p < kSelProblems; ++p) live[p] = st[p].alias < 0;
  unsigned v[(kSelProblems * kSelBins) / kFastThreads];
#pragma unroll
  for (int u = 0; u < (kSelProblems * kSelBins) / kFastThreads; ++u) {
    const int j = tid + kFastThreads * u, p = j / kSelBins;
    v[u] = live[p] ? __hip_atomic_load(hist + j, RLX_AGENT) : 0u;
  }
#pragma unroll
  for (int u = 0; u < (kSelProblems * kSelBins) / kFastThreads; ++u) {
    const int j = tid + kFastThreads * u, p = j / kSelBins, bin = j % kSelBins;
    lds[p * kScanPad + bin + (bin >> 6)] = v[u];
  }
  __syncthreads();
  if (wave < kSelProblems) {
    const int p = wave;
    int src = p;
    unsigned rank = 0;
#pragma unroll
    for (int pp = 0; pp < kSelProblems; ++pp)
      if (pp == p) {
        src = st[pp].alias >= 0 ? st[pp].alias : pp;
        rank = (unsigned)st[pp].rank;
      }
    const uint32_t *img = lds + src * kScanPad;
    unsigned tot = 0;
#pragma unroll 16
    for (int j = 0; j < 64; ++j) tot += img[lane * 65 + j];
    unsigned inc = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned t = __shfl_up(inc, off);
      if (lane >= off) inc += t;
    }
    const unsigned excl = inc - tot;
    const unsigned long long owners = __ballot(excl <= rank && rank < excl + tot);
    if (owners) {
      const int L = __ffsll((long long)owners) - 1;
      const unsigned base = __shfl(excl, L);
      const unsigned c = img[L * 65 + lane];  // bin 64 L + lane
      unsigned inc2 = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(inc2, off);
        if (lane >= off) inc2 += t;
      }
      const unsigned below = base + inc2 - c;
      if (c > 0 && below <= rank && rank < below + c) {
        found_bin[p] = L * 64 + lane;
        found_below[p] = below;
        found_cnt[p] = c;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int shift = pass_shift(pass);
    bool over = false;
#pragma unroll
    for (int p = 0; p < kSelProblems; ++p) {
      st[p].prefix |= (unsigned long long)found_bin[p] << shift;
      st[p].rank -= found_below[p];
      over |= check_cap && found_cnt[p] > (unsigned)kSelCap;
    }
    // an "upper middle" problem keeps sharing its partner's histogram only while both sit in
    // the same bin (aliases are always p -> p - 1)
#pragma unroll
    for (int p = 1; p < kSelProblems; p += 2)
      if (st[p].alias >= 0 && st[p].prefix != st[p - 1].prefix) st[p].alias = -1;
#pragma unroll
    for (int p = 0; p < kSelProblems; ++p) sel[p] = st[p];
    if (over) scal->overflow = 1;
  }
}

#ifdef ICP_NN_STATS
// diagnostic build only: per (MODE, pass) sums of phase times (shader cycles, thread 0 of each
// workgroup): [0] zero LDS, [1] stream + LDS atomics, [2] flush, [3] ticket, [4] tail (last
// workgroup only), [5] workgroups, [6] tails
__device__ unsigned long long g_hist_stamps[6][8];
#define HSTAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define HSTAMP(var) ((void)0)
#endif

template <int MODE>
__global__ __launch_bounds__(kFastThreads) void k_fast_hist(const double2 *__restrict__ a,
                                                            const double2 *__restrict__ b, Pose T,
                                                            double *__restrict__ rx, double *__restrict__ ry,
                                                            unsigned n, int pass, SelState *sel,
                                                            GnScalars *scal, uint32_t *hist, SelCtl *ctl) {
  __shared__ uint32_t lh[kSelProblems * kScanPad];  // histograms at p*kSelBins; padded staging later
  HSTAMP(ts0);
  unsigned long long prefix[kSelProblems];
  bool active[kSelProblems];
#pragma unroll
  for (int p = 0; p < kSelProblems; ++p) {
    prefix[p] = sel[p].prefix;
    active[p] = sel[p].alias < 0;
  }
#pragma unroll
  for (int p = 0; p < kSelProblems; ++p)
    if (active[p])
      for (unsigned i = threadIdx.x; i < kSelBins; i += kFastThreads) lh[p * kSelBins + i] = 0;
  __syncthreads();
  HSTAMP(ts1);

  const int shift = pass_shift(pass);
  const unsigned mask = (1u << pass_bits(pass)) - 1u;
  const int hs = shift + pass_bits(pass);  // bits above the current digit (64 at pass 0)
  double med0 = 0., med1 = 0.;
  if (MODE == 2) {
    med0 = scal->median[0];
    med1 = scal->median[1];
  }
  bool saw_nan = false;
  for_each_value<MODE>(a, b, T, rx, ry, n, med0, med1, saw_nan, [&](unsigned long long k0, unsigned long long k1) {
#pragma unroll
    for (int p = 0; p < kSelProblems; ++p) {
      if (!active[p]) continue;
      const unsigned long long key = (p < 2) ? k0 : k1;
      const bool match = (hs >= 64) || ((key >> hs) == (prefix[p] >> hs));
      const unsigned digit = (unsigned)(key >> shift) & mask;
      if (match) atomicAdd(&lh[p * kSelBins + digit], 1u);
    }
  });
  if (MODE == 0 && saw_nan) atomicOr(&scal->nan_flag, 1);
  __syncthreads();
  HSTAMP(ts2);
#pragma unroll
  for (int p = 0; p < kSelProblems; ++p)
    if (active[p])
      for (unsigned i = threadIdx.x; i < kSelBins; i += kFastThreads) {
        const uint32_t c = lh[p * kSelBins + i];
        if (c) atomicAdd(&hist[p * kSelBins + i], c);
      }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  HSTAMP(ts3);
  const bool last = last_block_arrives(&ctl->t[0]);
  HSTAMP(ts4);
  if (last) scan_descend(lh, hist, sel, scal, pass, /*check_cap=*/pass == 1);
#ifdef ICP_NN_STATS
  if (last) __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long ts5 = __builtin_amdgcn_s_memtime();
    unsigned long long *g = g_hist_stamps[(MODE == 0 ? 0 : (MODE == 1 ? 1 : 2 + (pass != 0)))];
    atomicAdd(&g[0], ts1 - ts0);
    atomicAdd(&g[1], ts2 - ts1);
    atomicAdd(&g[2], ts3 - ts2);
    atomicAdd(&g[3], ts4 - ts3);
    atomicAdd(&g[5], 1ull);
    if (last) {
      atomicAdd(&g[4], ts5 - ts4);
      atomicAdd(&g[6], 1ull);
    }
  }
#endif
}

// Append the keys that share the first `prefix_bits` bits with a problem's prefix to its
// candidate list; in the last block wave p ranks problem p's list and the block produces
// median (stage 0) or sigma (stage 1) and re-arms the search state for the next stage.
template <int MODE>
__global__ __launch_bounds__(kFastThreads) void k_fast_compact(const double2 *__restrict__ a,
                                                               const double2 *__restrict__ b, Pose T,
                                                               double *__restrict__ rx,
                                                               double *__restrict__ ry, unsigned n,
                                                               int prefix_bits, int stage, SelState *sel,
                                                               GnScalars *scal, unsigned long long *cand,
                                                               SelCtl *ctl) {
  __shared__ unsigned long long keys[1][kSelCap];
  __shared__ unsigned long long result[kSelProblems];
  __shared__ int s_over;
  unsigned long long prefix[kSelProblems];
  bool active[kSelProblems];
#pragma unroll
  for (int p = 0; p < kSelProblems; ++p) {
    prefix[p] = sel[p].prefix;
    active[p] = sel[p].alias < 0;
  }
  double med0 = 0., med1 = 0.;
  if (MODE == 2) {
    med0 = scal->median[0];
    med1 = scal->median[1];
  }
  const int hs = 64 - prefix_bits;
  bool saw_nan = false;
  for_each_value<MODE>(a, b, T, rx, ry, n, med0, med1, saw_nan, [&](unsigned long long k0, unsigned long long k1) {
#pragma unroll
    for (int p = 0; p < kSelProblems; ++p) {
      if (!active[p]) continue;
      const unsigned long long key = (p < 2) ? k0 : k1;
      const bool match = (hs >= 64) || ((key >> hs) == (prefix[p] >> hs));
      if (match) {
        const unsigned pos = atomicAdd(&ctl->cand_cnt[p], 1u);
        if (pos < (unsigned)kSelCap) __hip_atomic_store(&cand[p * kSelCap + pos], key, RLX_AGENT);
      }
    }
  });
  if (MODE == 0 && saw_nan) atomicOr(&scal->nan_flag, 1);

  if (!last_block_arrives(&ctl->t[1])) return;
  // Rank the candidate lists with the whole workgroup: list `l` (a problem that owns a
  // histogram/candidate buffer) is copied to LDS, thread i counts how many keys are smaller
  // than / equal to key i, and every problem that reads this list (itself, plus an aliased
  // "upper middle" problem) takes the key whose rank interval contains its rank.  O(c^2 / 1024)
  // per list; c <= 1024.
  const int tid = threadIdx.x;
  if (tid == 0) s_over = 0;
  if (tid < kSelProblems) result[tid] = 0;
  SelState st[kSelProblems];  // read once (see scan_descend)
#pragma unroll
  for (int p = 0; p < kSelProblems; ++p) st[p] = sel[p];
  __syncthreads();
#pragma unroll
  for (int l = 0; l < kSelProblems; ++l) {
    if (st[l].alias >= 0) continue;  // reads another problem's list
    unsigned c = __hip_atomic_load(&ctl->cand_cnt[l], RLX_AGENT);
    if (c > (unsigned)kSelCap) {
      if (tid == 0) s_over = 1;
      c = kSelCap;
    }
    for (unsigned i = tid; i < c; i += kFastThreads) keys[0][i] = __hip_atomic_load(&cand[l * kSelCap + i], RLX_AGENT);
    __syncthreads();
    for (unsigned i = tid; i < c; i += kFastThreads) {
      const unsigned long long ki = keys[0][i];
      unsigned less = 0, eq = 0;
      for (unsigned j = 0; j < c; ++j) {
        const unsigned long long kj = keys[0][j];
        less += kj < ki;
        eq += kj == ki;
      }
#pragma unroll
      for (int p = l; p < kSelProblems; ++p) {
        if (p != l && st[p].alias != l) continue;
        const unsigned long long rank = st[p].rank;
        if ((unsigned long long)less <= rank && rank < (unsigned long long)less + eq) result[p] = ki;
      }
    }
    __syncthreads();
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int j = 0; j < 2; ++j) {
      const double lo = k2f(result[2 * j]), hi = k2f(result[2 * j + 1]);
      const double med = (n & 1) ? lo : (lo + hi) / 2.;  // src/stats.rs:18-27
      if (stage == 0) scal->median[j] = med;
      else scal->sigma[j] = ICP_PPF34 * med;             // src/stats.rs:42-46
    }
    for (int p = 0; p < kSelProblems; ++p) {
      const bool hi = p & 1;
      sel[p].prefix = 0;
      sel[p].rank = hi ? (n / 2) : ((n - 1) / 2);
      sel[p].alias = hi ? p - 1 : -1;
      __hip_atomic_store(&ctl->cand_cnt[p], 0u, RLX_AGENT);
    }
    if (s_over) scal->overflow = 1;
  }
}

// src/lib.rs:238-255 (+ :45-50), block sums, and -- in the last block -- the second stage of
// the fixed reduction tree (identical to k_final_reduce in gn.hip).  Every block also clears
// its slice of the histograms for the next evaluation.
__global__ __launch_bounds__(kReduceThreads) void k_fast_accumulate(const double2 *__restrict__ a,
                                                                    const double *__restrict__ rx,
                                                                    const double *__restrict__ ry,
                                                                    unsigned n, Pose T, GnScalars *scal,
                                                                    double *partials, uint32_t *hist,
                                                                    SelCtl *ctl, GnResult *res, unsigned seq) {
  HSTAMP(ta0);
  const double sig[2] = {scal->sigma[0], scal->sigma[1]};
  double g[2];
  g[0] = 1. / sig[0];
  g[1] = 1. / sig[1];
  const double k2 = ICP_HUBER_K * ICP_HUBER_K;
  double acc[kNAcc];
#pragma unroll
  for (int k = 0; k < kNAcc; ++k) acc[k] = 0.;
  const unsigned G = gridDim.x * kReduceThreads;
  for (unsigned base = blockIdx.x * kReduceThreads + threadIdx.x; base < n; base += G * kFastBatch) {
    double2 s[kFastBatch];
    double r0[kFastBatch], r1[kFastBatch];
#pragma unroll
    for (int u = 0; u < kFastBatch; ++u) {
      const unsigned i = base + u * G;
      if (i < n) {
        s[u] = a[i];
        r0[u] = rx[i];
        r1[u] = ry[i];
      }
    }
#pragma unroll
    for (int u = 0; u < kFastBatch; ++u) {
      const unsigned i = base + u * G;
      if (i >= n) continue;
      const double r[2] = {r0[u], r1[u]};
      const double a0 = -s[u].y, a1 = s[u].x;  // jacobian(), src/lib.rs:176-184
      const double b0 = T.r00 * a0 + T.r01 * a1;
      const double b1 = T.r10 * a0 + T.r11 * a1;
      const double J[2][3] = {{T.r00, T.r01, b0}, {T.r10, T.r11, b1}};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (sig[j] == 0.) continue;  // src/lib.rs:243-245
        const double r_ij = r[j];
        const double e = r_ij * r_ij;
        double w_ij = 1.;  // huber::drho, src/huber.rs:17-26; sqrt+divide only where a lane needs it
        if (__ballot(e > k2)) w_ij = huber_drho(e);
        const double wg = w_ij * g[j];
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[9 + k] = acc[9 + k] + (wg * J[j][k]) * r_ij;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int q = 0; q < 3; ++q) acc[3 * p + q] = acc[3 * p + q] + (wg * J[j][p]) * J[j][q];
      }
      const double e2 = r[0] * r[0] + r[1] * r[1];
      double rho = e2;  // huber::rho, src/huber.rs:6-15
      if (__ballot(e2 > k2)) rho = huber_rho(e2);
      acc[12] = acc[12] + rho;
    }
  }
  HSTAMP(ta1);
  block_reduce_store<kNAcc, true>(acc, partials + (size_t)blockIdx.x * (kNAcc + 1));
  HSTAMP(ta2);
  for (unsigned i = blockIdx.x * kReduceThreads + threadIdx.x;
       i < (unsigned)(kSelRoles * kSelProblems * kSelBins); i += G)
    hist[i] = 0;

  const bool last_blk = last_block_arrives(&ctl->t[2]);
#ifdef ICP_NN_STATS
  if (threadIdx.x == 0) {
    const unsigned long long ta3 = __builtin_amdgcn_s_memtime();
    unsigned long long *gs = g_hist_stamps[4];
    atomicAdd(&gs[0], ta1 - ta0);
    atomicAdd(&gs[1], ta2 - ta1);
    atomicAdd(&gs[2], ta3 - ta2);
    atomicAdd(&gs[5], 1ull);
  }
  const unsigned long long ta4 = __builtin_amdgcn_s_memtime();
#endif
  if (!last_blk) return;
  const int nan_flag = scal->nan_flag, overflow = scal->overflow;  // issued now, used at the very end
  double tot[kNAcc + 1];
#pragma unroll
  for (int k = 0; k < kNAcc + 1; ++k) tot[k] = 0.;
  const int blocks = gridDim.x;
  for (int i = threadIdx.x; i < blocks; i += kReduceThreads) {
    double v[kNAcc];  // all 13 loads in flight before the first add (a load-add-load-add chain
                      // would pay the sc1 latency 13 times over)
#pragma unroll
    for (int k = 0; k < kNAcc; ++k) v[k] = __hip_atomic_load(&partials[(size_t)i * (kNAcc + 1) + k], RLX_AGENT);
#pragma unroll
    for (int k = 0; k < kNAcc; ++k) tot[k] = tot[k] + v[k];
  }
  block_reduce_store<kNAcc + 1>(tot, res->acc);
  // Publish to the host: the result lives in coherent pinned memory; every writing lane makes
  // its stores visible at system scope before lane 0 releases the sequence number the host
  // is polling (saves the kernel-completion -> stream-sync wake-up on every inner iteration).
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    res->sigma[0] = sig[0];
    res->sigma[1] = sig[1];
    res->nan_flag = nan_flag;
    res->overflow = overflow;
    scal->overflow = 0;
    __threadfence_system();
    __hip_atomic_store(&res->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}


// ---------------------------------------------------------------------------------------
// n <= 1024 (the reference's own 2-D scans have ~650 points): the whole evaluation in ONE
// workgroup and ONE launch -- residuals, both medians, both MADs (bitonic sort of the
// order-preserving keys in LDS: order statistics are then plain lookups), and the weighted
// normal equations folded in exactly the multi-workgroup tree of reduce_geometry(n)
// (1 or 2 virtual blocks of 512 threads, then the 512-thread second stage), so the bits are
// the same as on the general path.
__device__ __forceinline__ void bitonic_sort2_1024(unsigned long long *A, unsigned long long *B) {
  const unsigned tid = threadIdx.x;
  for (unsigned k = 2; k <= 1024; k <<= 1)
    for (unsigned j = k >> 1; j > 0; j >>= 1) {
      const unsigned ixj = tid ^ j;
      if (ixj > tid) {
        const bool asc = (tid & k) == 0;
        const unsigned long long a0 = A[tid], a1 = A[ixj];
        if ((a0 > a1) == asc) {
          A[tid] = a1;
          A[ixj] = a0;
        }
        const unsigned long long b0 = B[tid], b1 = B[ixj];
        if ((b0 > b1) == asc) {
          B[tid] = b1;
          B[ixj] = b0;
        }
      }
      __syncthreads();
    }
}

// fold `acc` over a group of 8 waves (512 threads) in the tree of block_reduce_store:
// wave shuffle tree, then a left fold of the wave sums from the group's first wave
template <int N>
__device__ __forceinline__ void group_reduce(double (&acc)[N], double (*sm)[N], int wave) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {  // step-major: see block_reduce_store
    double t[N];
#pragma unroll
    for (int k = 0; k < N; ++k) t[k] = __shfl_down(acc[k], off);
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k] = acc[k] + t[k];
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < N; ++k) sm[wave][k] = acc[k];
  }
}

__global__ __launch_bounds__(1024) void k_tiny_eval(const double2 *__restrict__ a, const double2 *__restrict__ b,
                                                    unsigned n, Pose T, int blocks, GnResult *res,
                                                    unsigned seq) {
  __shared__ unsigned long long KA[1024], KB[1024];
  __shared__ double sm[16][kNAcc + 1];
  __shared__ double part[2][kNAcc + 1];
  __shared__ int s_nan;
  const unsigned tid = threadIdx.x;
  const int wave = tid >> 6;
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
  const unsigned lo_rank = (n - 1) / 2, hi_rank = n / 2;
  // medians (src/stats.rs:11-28)
  KA[tid] = has ? f2k(r0) : ~0ull;
  KB[tid] = has ? f2k(r1) : ~0ull;
  __syncthreads();
  bitonic_sort2_1024(KA, KB);
  double med[2];
  {
    const double xl = k2f(KA[lo_rank]), xh = k2f(KA[hi_rank]), yl = k2f(KB[lo_rank]), yh = k2f(KB[hi_rank]);
    med[0] = (n & 1) ? xl : (xl + xh) / 2.;
    med[1] = (n & 1) ? yl : (yl + yh) / 2.;
  }
  __syncthreads();
  // MADs (src/stats.rs:30-47)
  KA[tid] = has ? f2k(fabs(r0 - med[0])) : ~0ull;
  KB[tid] = has ? f2k(fabs(r1 - med[1])) : ~0ull;
  __syncthreads();
  bitonic_sort2_1024(KA, KB);
  double sig[2];
  {
    const double xl = k2f(KA[lo_rank]), xh = k2f(KA[hi_rank]), yl = k2f(KB[lo_rank]), yh = k2f(KB[hi_rank]);
    sig[0] = ICP_PPF34 * ((n & 1) ? xl : (xl + xh) / 2.);
    sig[1] = ICP_PPF34 * ((n & 1) ? yl : (yl + yh) / 2.);
  }
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
  // stage 1: virtual blocks of 512 threads (8 waves each)
  group_reduce<kNAcc + 1>(acc, sm, wave);
  __syncthreads();
  if (tid < 2 * (kNAcc + 1)) {
    const int vb = tid / (kNAcc + 1), k = tid % (kNAcc + 1);
    double v = sm[8 * vb][k];
    for (int w = 1; w < 8; ++w) v = v + sm[8 * vb + w][k];
    part[vb][k] = v;
  }
  __syncthreads();
  // stage 2: one block of 512 threads over the `blocks` block sums
  double tot[kNAcc + 1];
#pragma unroll
  for (int k = 0; k < kNAcc + 1; ++k) tot[k] = 0.;
  if (tid < (unsigned)blocks)
#pragma unroll
    for (int k = 0; k < kNAcc; ++k) tot[k] = tot[k] + part[tid][k];
  if (tid < 512) group_reduce<kNAcc + 1>(tot, sm, wave);
  __syncthreads();
  if (tid < kNAcc + 1) {
    double v = sm[0][tid];
    for (int w = 1; w < 8; ++w) v = v + sm[w][tid];
    res->acc[tid] = v;
  }
  __threadfence_system();
  __syncthreads();
  if (tid == 0) {
    res->sigma[0] = sig[0];
    res->sigma[1] = sig[1];
    res->nan_flag = s_nan;
    res->overflow = 0;
    __threadfence_system();
    __hip_atomic_store(&res->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static unsigned fast_blocks(unsigned n) {
  const unsigned per = kFastThreads * kFastBatch;
  unsigned b = (n + per - 1) / per;
  if (b < 1) b = 1;
  if (b > 256) b = 256;  // one per CU: every extra workgroup costs a histogram flush of global atomics
  return b;
}

hipError_t launch_weighted_gn_fast(icp_handle *h, const double *d_a, const double *d_b, size_t n_, const Pose &T) {
  Workspace &w = h->ws;
  const unsigned n = (unsigned)n_;
  const unsigned hb = fast_blocks(n);
  const double2 *a = (const double2 *)d_a, *b = (const double2 *)d_b;
  hipStream_t s = h->stream;
  if (n <= 1024u) {
    int blocks, threads;
    reduce_geometry(n_, &blocks, &threads);
    if (threads == 512 && blocks <= 2) {  // the geometry k_tiny_eval reproduces
      hipLaunchKernelGGL(k_tiny_eval, dim3(1), dim3(1024), 0, s, a, b, n, T, blocks, w.h_res, ++w.seq);
      return hipGetLastError();
    }
  }
  static const bool push = getenv("ICP_GN_PUSH") != nullptr;
  if (!push) return launch_weighted_gn_pull(h, d_a, d_b, n_, T);  // same launches, no serial tails
  // second-digit passes flush dense histograms (8192 global atomics per workgroup): half the
  // workgroups, twice the elements per lane
  static const unsigned hb1_div = getenv("ICP_HB1_DIV") ? (unsigned)atoi(getenv("ICP_HB1_DIV")) : 2u;
  const unsigned hb1 = hb / hb1_div > 0 ? hb / hb1_div : 1;
  const size_t role = (size_t)kSelProblems * kSelBins;
#define HIST(MODE, PASS, ROLE)                                                                            \
  hipLaunchKernelGGL(k_fast_hist<MODE>, dim3((PASS) == 1 ? hb1 : hb), dim3(kFastThreads), 0, s, a, b, T,  \
                     w.d_rx, w.d_ry, n,                                                                    \
                     PASS, w.d_sel, w.d_scal, w.d_hist + (ROLE) * role, w.d_ctl)
#define COMPACT(MODE, BITS, STAGE)                                                                        \
  hipLaunchKernelGGL(k_fast_compact<MODE>, dim3(hb), dim3(kFastThreads), 0, s, a, b, T, w.d_rx, w.d_ry, n, \
                     BITS, STAGE, w.d_sel, w.d_scal, w.d_cand, w.d_ctl)
  if (n > (unsigned)kSelCap) {
    HIST(0, 0, 0);
    HIST(1, 1, 1);
    COMPACT(1, 24, 0);
    HIST(2, 0, 2);
    HIST(2, 1, 3);
    COMPACT(2, 24, 1);
  } else {  // every element is a candidate: one launch per stage
    COMPACT(0, 0, 0);
    COMPACT(2, 0, 1);
  }
#undef HIST
#undef COMPACT
  int blocks, threads;
  reduce_geometry(n_, &blocks, &threads);
  hipLaunchKernelGGL(k_fast_accumulate, dim3(blocks), dim3(threads), 0, s, a, w.d_rx, w.d_ry, n, T, w.d_scal,
                     w.d_partials, w.d_hist, w.d_ctl, w.h_res, ++w.seq);
  return hipGetLastError();
}

}  // namespace icp

#ifdef ICP_NN_STATS
extern "C" int icp_debug_hist_stamps(unsigned long long out[48], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(icp::g_hist_stamps), 48 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    const unsigned long long z[48] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(icp::g_hist_stamps), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif
