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
// "Last block" = the workgroup whose arrival ticket is the final one; it sees the other
// workgroups' atomics/stores through the agent-scope release/acquire of
// last_block_arrives() (gn_device.hpp).  Integer histograms and rank counting are exact
// and independent of arrival order; the sums use the same tree as gn.hip.  If a median
// sits in a run of more than kSelCap equal-prefix keys (heavy duplicates), the pipeline
// raises `overflow` and the host repeats the evaluation with the general radix path.
#include "common.hpp"
#include "gn_device.hpp"

namespace icp {

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

template <int MODE>
__device__ __forceinline__ void load_values(unsigned i, const double2 *__restrict__ a,
                                            const double2 *__restrict__ b, const Pose &T,
                                            double *__restrict__ rx, double *__restrict__ ry, double med0,
                                            double med1, double &v0, double &v1, bool &saw_nan) {
  if (MODE == 0) {  // residual(), src/lib.rs:34-36, stored for the later passes
    const double2 s = a[i], d = b[i];
    v0 = ((T.r00 * s.x + T.r01 * s.y) + T.tx) - d.x;
    v1 = ((T.r10 * s.x + T.r11 * s.y) + T.ty) - d.y;
    rx[i] = v0;
    ry[i] = v1;
    saw_nan |= (v0 != v0) | (v1 != v1);
  } else {
    v0 = rx[i];
    v1 = ry[i];
    if (MODE == 2) {  // src/stats.rs:35
      v0 = fabs(v0 - med0);
      v1 = fabs(v1 - med1);
    }
  }
}

// Executed by the last block (256 threads): for every problem find the digit bin that
// holds its rank and descend into it.
__device__ void scan_descend(const uint32_t *hist, SelState *sel, GnScalars *scal, int pass, bool check_cap) {
  constexpr int PER = kSelBins / 256;
  __shared__ unsigned wave_sum[4];
  __shared__ unsigned found_bin[kSelProblems], found_below[kSelProblems], found_cnt[kSelProblems];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int p = 0; p < kSelProblems; ++p) {
    const int src = sel[p].alias >= 0 ? sel[p].alias : p;
    const unsigned long long rank = sel[p].rank;
    const uint32_t *hp = hist + src * kSelBins;
    unsigned loc[PER], tot = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      loc[j] = __hip_atomic_load(hp + tid * PER + j, RLX_AGENT);
      tot += loc[j];
    }
    unsigned inc = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned v = __shfl_up(inc, off);
      if (lane >= off) inc += v;
    }
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < wave; ++w) base += wave_sum[w];
    const unsigned excl = base + inc - tot;
    if ((unsigned long long)excl <= rank && rank < (unsigned long long)excl + tot) {
      unsigned below = excl;
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        if (rank < (unsigned long long)below + loc[j]) {
          found_bin[p] = tid * PER + j;
          found_below[p] = below;
          found_cnt[p] = loc[j];
          break;
        }
        below += loc[j];
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    const int shift = pass_shift(pass);
    bool over = false;
    for (int p = 0; p < kSelProblems; ++p) {
      sel[p].prefix |= (unsigned long long)found_bin[p] << shift;
      sel[p].rank -= found_below[p];
      over |= check_cap && found_cnt[p] > (unsigned)kSelCap;
    }
    for (int p = 0; p < kSelProblems; ++p)
      if (sel[p].alias >= 0 && sel[p].prefix != sel[sel[p].alias].prefix) sel[p].alias = -1;
    if (over) scal->overflow = 1;
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_fast_hist(const double2 *__restrict__ a, const double2 *__restrict__ b,
                                                   Pose T, double *__restrict__ rx, double *__restrict__ ry,
                                                   unsigned n, int pass, SelState *sel, GnScalars *scal,
                                                   uint32_t *hist, SelCtl *ctl) {
  __shared__ uint32_t lh[kSelProblems][kSelBins];
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
      for (unsigned i = threadIdx.x; i < kSelBins; i += 256) lh[p][i] = 0;
  __syncthreads();

  const int shift = pass_shift(pass);
  const unsigned mask = (1u << pass_bits(pass)) - 1u;
  const int hs = shift + pass_bits(pass);
  double med0 = 0., med1 = 0.;
  if (MODE == 2) {
    med0 = scal->median[0];
    med1 = scal->median[1];
  }
  bool saw_nan = false;
  const unsigned G = gridDim.x * 256;
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += G) {
    double v0, v1;
    load_values<MODE>(i, a, b, T, rx, ry, med0, med1, v0, v1, saw_nan);
    const unsigned long long k0 = f2k(v0), k1 = f2k(v1);
#pragma unroll
    for (int p = 0; p < kSelProblems; ++p) {
      if (!active[p]) continue;
      const unsigned long long key = (p < 2) ? k0 : k1;
      const bool match = (hs >= 64) || ((key >> hs) == (prefix[p] >> hs));
      if (match) atomicAdd(&lh[p][(unsigned)(key >> shift) & mask], 1u);
    }
  }
  if (MODE == 0 && saw_nan) atomicOr(&scal->nan_flag, 1);
  __syncthreads();
#pragma unroll
  for (int p = 0; p < kSelProblems; ++p)
    if (active[p])
      for (unsigned i = threadIdx.x; i < kSelBins; i += 256) {
        const uint32_t c = lh[p][i];
        if (c) atomicAdd(&hist[p * kSelBins + i], c);
      }
  if (last_block_arrives(&ctl->ticket[0])) scan_descend(hist, sel, scal, pass, /*check_cap=*/pass == 1);
}

// Append the keys that share the first `prefix_bits` bits with a problem's prefix to its
// candidate list; the last block then ranks each list and produces median (stage 0) or
// sigma (stage 1), and re-arms the search state for the next stage.
template <int MODE>
__global__ __launch_bounds__(256) void k_fast_compact(const double2 *__restrict__ a,
                                                      const double2 *__restrict__ b, Pose T,
                                                      double *__restrict__ rx, double *__restrict__ ry,
                                                      unsigned n, int prefix_bits, int stage, SelState *sel,
                                                      GnScalars *scal, unsigned long long *cand, SelCtl *ctl) {
  __shared__ unsigned long long keys[kSelCap];
  __shared__ unsigned long long result[kSelProblems];
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
  const unsigned G = gridDim.x * 256;
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += G) {
    double v0, v1;
    load_values<MODE>(i, a, b, T, rx, ry, med0, med1, v0, v1, saw_nan);
    const unsigned long long k0 = f2k(v0), k1 = f2k(v1);
#pragma unroll
    for (int p = 0; p < kSelProblems; ++p) {
      if (!active[p]) continue;
      const unsigned long long key = (p < 2) ? k0 : k1;
      const bool match = (hs >= 64) || ((key >> hs) == (prefix[p] >> hs));
      if (match) {
        const unsigned pos = atomicAdd(&ctl->cand_cnt[p], 1u);
        if (pos < (unsigned)kSelCap) cand[p * kSelCap + pos] = key;
      }
    }
  }
  if (MODE == 0 && saw_nan) atomicOr(&scal->nan_flag, 1);

  if (!last_block_arrives(&ctl->ticket[1])) return;
  const int tid = threadIdx.x;
  bool over = false;
  for (int p = 0; p < kSelProblems; ++p) {
    const int src = sel[p].alias >= 0 ? sel[p].alias : p;
    unsigned c = __hip_atomic_load(&ctl->cand_cnt[src], RLX_AGENT);
    if (c > (unsigned)kSelCap) {
      over = true;
      c = kSelCap;
    }
    const unsigned long long rank = sel[p].rank;
    __syncthreads();
    for (unsigned i = tid; i < c; i += 256) keys[i] = cand[src * kSelCap + i];
    if (tid == 0) result[p] = 0;
    __syncthreads();
    for (unsigned i = tid; i < c; i += 256) {
      const unsigned long long ki = keys[i];
      unsigned less = 0, eq = 0;
      for (unsigned j = 0; j < c; ++j) {
        const unsigned long long kj = keys[j];
        less += kj < ki;
        eq += kj == ki;
      }
      if ((unsigned long long)less <= rank && rank < (unsigned long long)less + eq) result[p] = ki;
    }
  }
  __syncthreads();
  if (tid == 0) {
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
    if (over) scal->overflow = 1;
  }
}

// src/lib.rs:238-255 (+ :45-50), block sums, and -- in the last block -- the second stage of
// the fixed reduction tree (identical to k_final_reduce in gn.hip).  Every block also clears
// its slice of the histograms for the next evaluation.
__global__ __launch_bounds__(256) void k_fast_accumulate(const double2 *__restrict__ a,
                                                         const double *__restrict__ rx,
                                                         const double *__restrict__ ry, unsigned n, Pose T,
                                                         GnScalars *scal, double *partials, uint32_t *hist,
                                                         SelCtl *ctl, GnResult *res) {
  const double sig[2] = {scal->sigma[0], scal->sigma[1]};
  double g[2];
  g[0] = 1. / sig[0];
  g[1] = 1. / sig[1];
  double acc[kNAcc];
#pragma unroll
  for (int k = 0; k < kNAcc; ++k) acc[k] = 0.;
  const unsigned G = gridDim.x * 256;
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += G) {
    const double2 s = a[i];
    const double r[2] = {rx[i], ry[i]};
    const double a0 = -s.y, a1 = s.x;  // jacobian(), src/lib.rs:176-184
    const double b0 = T.r00 * a0 + T.r01 * a1;
    const double b1 = T.r10 * a0 + T.r11 * a1;
    const double J[2][3] = {{T.r00, T.r01, b0}, {T.r10, T.r11, b1}};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (sig[j] == 0.) continue;  // src/lib.rs:243-245
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
  block_reduce_store<kNAcc>(acc, partials + (size_t)blockIdx.x * (kNAcc + 1));
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < (unsigned)(kSelRoles * kSelProblems * kSelBins); i += G)
    hist[i] = 0;

  if (!last_block_arrives(&ctl->ticket[2])) return;
  double tot[kNAcc + 1];
#pragma unroll
  for (int k = 0; k < kNAcc + 1; ++k) tot[k] = 0.;
  const int blocks = gridDim.x;
  for (int i = threadIdx.x; i < blocks; i += 256)
#pragma unroll
    for (int k = 0; k < kNAcc; ++k) tot[k] = tot[k] + partials[(size_t)i * (kNAcc + 1) + k];
  block_reduce_store<kNAcc + 1>(tot, res->acc);
  if (threadIdx.x == 0) {
    res->sigma[0] = sig[0];
    res->sigma[1] = sig[1];
    res->nan_flag = scal->nan_flag;
    res->overflow = scal->overflow;
    scal->overflow = 0;
  }
}

static unsigned fast_blocks(unsigned n) {
  unsigned b = (n + 256 * 8 - 1) / (256 * 8);
  if (b < 1) b = 1;
  if (b > 512) b = 512;
  return b;
}

hipError_t launch_weighted_gn_fast(icp_handle *h, const double *d_a, const double *d_b, size_t n_, const Pose &T) {
  Workspace &w = h->ws;
  const unsigned n = (unsigned)n_;
  const unsigned hb = fast_blocks(n);
  const double2 *a = (const double2 *)d_a, *b = (const double2 *)d_b;
  hipStream_t s = h->stream;
  const size_t role = (size_t)kSelProblems * kSelBins;
#define HIST(MODE, PASS, ROLE)                                                                          \
  hipLaunchKernelGGL(k_fast_hist<MODE>, dim3(hb), dim3(256), 0, s, a, b, T, w.d_rx, w.d_ry, n, PASS,    \
                     w.d_sel, w.d_scal, w.d_hist + (ROLE) * role, w.d_ctl)
#define COMPACT(MODE, BITS, STAGE)                                                                      \
  hipLaunchKernelGGL(k_fast_compact<MODE>, dim3(hb), dim3(256), 0, s, a, b, T, w.d_rx, w.d_ry, n, BITS,  \
                     STAGE, w.d_sel, w.d_scal, w.d_cand, w.d_ctl)
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
                     w.d_partials, w.d_hist, w.d_ctl, w.h_res);
  return hipGetLastError();
}

}  // namespace icp
