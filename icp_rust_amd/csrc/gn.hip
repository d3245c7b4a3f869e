// Device side of one inner Gauss-Newton iteration of icp::estimate_transform
// (src/lib.rs:59-84): residuals, the four exact medians behind calc_stddevs
// (src/stats.rs:11-60, called at src/lib.rs:236), the Huber-weighted normal equations
// (src/lib.rs:238-255) and the Huber error (src/lib.rs:45-50), all on data that stays in
// HBM/MALL.  The host keeps the 3x3 solve, the break tests and Transform::new.
//
// Exact medians on a GPU: an MSD radix select over order-preserving 64-bit keys
// (12-bit digits, LDS histograms flushed with integer atomics => exact and independent
// of arrival order).  The two middle order statistics of an even-length input are
// searched together (problem "lo" and problem "hi" share a histogram while they share a
// prefix).  Sums use a fixed tree (DESIGN.md "GN reduction order"): no float atomics,
// run-to-run bit-identical.
#include "common.hpp"
#include "gn_device.hpp"

namespace icp {

// ------------------------------------------------------------- selection ---------
__global__ void k_sel_init(SelState *sel, GnScalars *scal, uint32_t *hist, SelCtl *ctl, unsigned n) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  for (unsigned i = t; i < kSelRoles * kSelProblems * kSelBins; i += gridDim.x * blockDim.x) hist[i] = 0;
  if (t < sizeof(SelCtl) / sizeof(unsigned)) reinterpret_cast<unsigned *>(ctl)[t] = 0;
  if (t < kSelProblems) {
    const bool hi = t & 1;
    sel[t].prefix = 0;
    sel[t].rank = hi ? (n / 2) : ((n - 1) / 2);  // src/stats.rs:18-27
    sel[t].alias = hi ? (int)(t - 1) : -1;
    sel[t].pad = 0;
  }
  if (t == 0) {
    scal->nan_flag = 0;
    scal->overflow = 0;
    scal->median[0] = scal->median[1] = 0.;
    scal->sigma[0] = scal->sigma[1] = 0.;
  }
}

// MODE 0: stage "median", pass 0: r = T*a - b is computed here (src/lib.rs:230-234 ->
//         :34-36) and stored as SoA rx|ry for the later passes.
// MODE 1: stage "median", pass >= 1 (keys of rx, ry).
// MODE 2: stage "MAD": keys of |r - median| (src/stats.rs:35).
template <int MODE>
__global__ __launch_bounds__(256) void k_sel_hist(const double2 *__restrict__ a,
                                                  const double2 *__restrict__ b, Pose T,
                                                  double *__restrict__ rx, double *__restrict__ ry,
                                                  unsigned n, int pass,
                                                  const SelState *__restrict__ sel,
                                                  GnScalars *__restrict__ scal,
                                                  uint32_t *__restrict__ hist) {
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
  const int hs = shift + pass_bits(pass);  // bits above the current digit (64 at pass 0)
  double med0 = 0., med1 = 0.;
  if (MODE == 2) {
    med0 = scal->median[0];
    med1 = scal->median[1];
  }
  bool saw_nan = false;
  const unsigned G = gridDim.x * 256;
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += G) {
    double v0, v1;
    if (MODE == 0) {
      const double2 s = a[i], d = b[i];
      v0 = ((T.r00 * s.x + T.r01 * s.y) + T.tx) - d.x;
      v1 = ((T.r10 * s.x + T.r11 * s.y) + T.ty) - d.y;
      rx[i] = v0;
      ry[i] = v1;
      saw_nan |= (v0 != v0) | (v1 != v1);
    } else {
      v0 = rx[i];
      v1 = ry[i];
      if (MODE == 2) {
        v0 = fabs(v0 - med0);
        v1 = fabs(v1 - med1);
      }
    }
    const unsigned long long k0 = f2k(v0), k1 = f2k(v1);
#pragma unroll
    for (int p = 0; p < kSelProblems; ++p) {
      if (!active[p]) continue;
      const unsigned long long key = (p < 2) ? k0 : k1;
      const bool match = (MODE == 0) || (hs >= 64) || ((key >> hs) == (prefix[p] >> hs));
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
}

// One block: for every problem find the digit bin that holds its rank, descend, and on
// the last pass turn the keys into median / sigma (src/stats.rs:11-47).  Re-arms the
// search state for the next stage and clears the histograms.
__global__ __launch_bounds__(256) void k_sel_scan(uint32_t *__restrict__ hist, SelState *sel,
                                                  GnScalars *scal, int stage, int pass, unsigned n) {
  constexpr int PER = kSelBins / 256;  // bins per thread
  __shared__ unsigned wave_sum[4];
  __shared__ unsigned found_bin[kSelProblems];
  __shared__ unsigned found_below[kSelProblems];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nbins = 1 << pass_bits(pass);

  for (int p = 0; p < kSelProblems; ++p) {
    const int src = sel[p].alias >= 0 ? sel[p].alias : p;
    const unsigned long long rank = sel[p].rank;
    const uint32_t *hp = hist + src * kSelBins;
    unsigned loc[PER];
    unsigned tot = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int bin = tid * PER + j;
      loc[j] = bin < nbins ? hp[bin] : 0u;
      tot += loc[j];
    }
    // exclusive scan of `tot` over the 256 threads
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
    unsigned excl = base + inc - tot;
    if ((unsigned long long)excl <= rank && rank < (unsigned long long)excl + tot) {
      unsigned below = excl;
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        if (rank < (unsigned long long)below + loc[j]) {
          found_bin[p] = tid * PER + j;
          found_below[p] = below;
          break;
        }
        below += loc[j];
      }
    }
    __syncthreads();
  }

  if (tid == 0) {
    const int shift = pass_shift(pass);
    for (int p = 0; p < kSelProblems; ++p) {
      sel[p].prefix |= (unsigned long long)found_bin[p] << shift;
      sel[p].rank -= found_below[p];
    }
    for (int p = 0; p < kSelProblems; ++p)
      if (sel[p].alias >= 0 && sel[p].prefix != sel[sel[p].alias].prefix) sel[p].alias = -1;
    if (pass == kSelPasses - 1) {
      for (int j = 0; j < 2; ++j) {
        const double lo = k2f(sel[2 * j].prefix), hi = k2f(sel[2 * j + 1].prefix);
        const double med = (n & 1) ? lo : (lo + hi) / 2.;  // src/stats.rs:18-27
        if (stage == 0) scal->median[j] = med;
        else scal->sigma[j] = ICP_PPF34 * med;             // src/stats.rs:42-46
      }
      for (int p = 0; p < kSelProblems; ++p) {
        const bool hi = p & 1;
        sel[p].prefix = 0;
        sel[p].rank = hi ? (n / 2) : ((n - 1) / 2);
        sel[p].alias = hi ? p - 1 : -1;
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < kSelProblems * kSelBins; i += 256) hist[i] = 0;
}

// src/lib.rs:238-255 (+ :45-50 fused: same T, same residuals)
__global__ __launch_bounds__(kReduceThreads) void k_wgn_accumulate(const double2 *__restrict__ a,
                                                        const double *__restrict__ rx,
                                                        const double *__restrict__ ry, unsigned n,
                                                        Pose T, const GnScalars *__restrict__ scal,
                                                        double *__restrict__ partials) {
  double acc[kNSum];
#pragma unroll
  for (int k = 0; k < kNSum; ++k) acc[k] = 0.;
  const unsigned G = gridDim.x * kReduceThreads;
  for (unsigned i = blockIdx.x * kReduceThreads + threadIdx.x; i < n; i += G) accumulate_pair<false>(a[i], rx[i], ry[i], T, acc);
  block_reduce_store<kNSum>(acc, partials + (size_t)blockIdx.x * (kNSum + 1));
}

// gauss_newton_update (src/lib.rs:191-216), error (:38-43), huber_error (:45-50)
__global__ __launch_bounds__(kReduceThreads) void k_plain_accumulate(const double2 *__restrict__ a,
                                                          const double2 *__restrict__ b, unsigned n,
                                                          Pose T, double *__restrict__ partials) {
  double acc[kNAcc + 1];
#pragma unroll
  for (int k = 0; k < kNAcc + 1; ++k) acc[k] = 0.;
  const unsigned G = gridDim.x * kReduceThreads;
  for (unsigned i = blockIdx.x * kReduceThreads + threadIdx.x; i < n; i += G) {
    const double2 s = a[i], d = b[i];
    const double r0 = ((T.r00 * s.x + T.r01 * s.y) + T.tx) - d.x;
    const double r1 = ((T.r10 * s.x + T.r11 * s.y) + T.ty) - d.y;
    const double a0 = -s.y, a1 = s.x;
    const double b0 = T.r00 * a0 + T.r01 * a1;
    const double b1 = T.r10 * a0 + T.r11 * a1;
    const double J0[3] = {T.r00, T.r01, b0}, J1[3] = {T.r10, T.r11, b1};
#pragma unroll
    for (int k = 0; k < 3; ++k) acc[9 + k] = acc[9 + k] + (J0[k] * r0 + J1[k] * r1);
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[3 * p + q] = acc[3 * p + q] + (J0[p] * J0[q] + J1[p] * J1[q]);
    const double e = r0 * r0 + r1 * r1;
    acc[12] = acc[12] + huber_rho(e);
    acc[13] = acc[13] + e;
  }
  block_reduce_store<kNAcc + 1>(acc, partials + (size_t)blockIdx.x * (kNAcc + 1));
}

// second stage: one block over the block sums; hands the result to the host
__global__ __launch_bounds__(kReduceThreads) void k_final_reduce(const double *__restrict__ partials, int blocks,
                                                      int nacc, const GnScalars *__restrict__ scal,
                                                      GnResult *__restrict__ res) {
  double acc[kNAcc + 1];
#pragma unroll
  for (int k = 0; k < kNAcc + 1; ++k) acc[k] = 0.;
  for (int i = threadIdx.x; i < blocks; i += kReduceThreads)
#pragma unroll
    for (int k = 0; k < kNAcc + 1; ++k)
      if (k < nacc) acc[k] = acc[k] + partials[(size_t)i * (kNAcc + 1) + k];
  block_reduce_store<kNAcc + 1>(acc, res->acc);
  if (threadIdx.x == 0) {
    res->sigma[0] = scal->sigma[0];
    res->sigma[1] = scal->sigma[1];
    res->nan_flag = scal->nan_flag;
    res->overflow = 0;
  }
}

// second stage of a WEIGHTED evaluation: fold the per-dimension sums, then jtj = g_x S_x + g_y S_y (combine_sum)
__global__ __launch_bounds__(kReduceThreads) void k_final_reduce_weighted(const double *__restrict__ partials, int blocks,
                                                                          const GnScalars *__restrict__ scal,
                                                                          GnResult *__restrict__ res) {
  __shared__ double s_tot[kNSum + 1];
  double acc[kNSum + 1];
#pragma unroll
  for (int k = 0; k < kNSum + 1; ++k) acc[k] = 0.;
  for (int i = threadIdx.x; i < blocks; i += kReduceThreads)
#pragma unroll
    for (int k = 0; k < kNSum; ++k) acc[k] = acc[k] + partials[(size_t)i * (kNSum + 1) + k];
  block_reduce_store<kNSum + 1>(acc, s_tot);
  __syncthreads();
  const double sig[2] = {scal->sigma[0], scal->sigma[1]};
  if (threadIdx.x < kNAcc + 1) res->acc[threadIdx.x] = threadIdx.x < kNAcc ? combine_sum(s_tot, (int)threadIdx.x, sig) : 0.;
  if (threadIdx.x == 0) {
    res->sigma[0] = sig[0];
    res->sigma[1] = sig[1];
    res->nan_flag = scal->nan_flag;
    res->overflow = 0;
  }
}

// ---------------------------------------------------------------- launchers ------
static unsigned hist_blocks(unsigned n) {
  unsigned b = (n + 256 * 8 - 1) / (256 * 8);
  if (b < 1) b = 1;
  if (b > 512) b = 512;
  return b;
}

hipError_t launch_sel_init(icp_handle *h, size_t n) {
  Workspace &w = h->ws;
  hipLaunchKernelGGL(k_sel_init, dim3(64), dim3(256), 0, h->stream, w.d_sel, w.d_scal, w.d_hist, w.d_ctl,
                     (unsigned)n);
  return hipGetLastError();
}

hipError_t launch_stddevs(icp_handle *h, const double *d_a, const double *d_b, size_t n_, const Pose &T) {
  Workspace &w = h->ws;
  const unsigned n = (unsigned)n_;
  const unsigned hb = hist_blocks(n);
  const double2 *a = (const double2 *)d_a, *b = (const double2 *)d_b;
  hipStream_t s = h->stream;
  for (int stage = 0; stage < 2; ++stage) {
    for (int pass = 0; pass < kSelPasses; ++pass) {
      if (stage == 0 && pass == 0)
        hipLaunchKernelGGL(k_sel_hist<0>, dim3(hb), dim3(256), 0, s, a, b, T, w.d_rx, w.d_ry, n, pass,
                           w.d_sel, w.d_scal, w.d_hist);
      else if (stage == 0)
        hipLaunchKernelGGL(k_sel_hist<1>, dim3(hb), dim3(256), 0, s, a, b, T, w.d_rx, w.d_ry, n, pass,
                           w.d_sel, w.d_scal, w.d_hist);
      else
        hipLaunchKernelGGL(k_sel_hist<2>, dim3(hb), dim3(256), 0, s, a, b, T, w.d_rx, w.d_ry, n, pass,
                           w.d_sel, w.d_scal, w.d_hist);
      hipLaunchKernelGGL(k_sel_scan, dim3(1), dim3(256), 0, s, w.d_hist, w.d_sel, w.d_scal, stage, pass, n);
    }
  }
  return hipGetLastError();
}

hipError_t launch_weighted_gn(icp_handle *h, const double *d_a, const double *d_b, size_t n_, const Pose &T) {
  Workspace &w = h->ws;
  const unsigned n = (unsigned)n_;
  hipError_t e = launch_stddevs(h, d_a, d_b, n_, T);
  if (e != hipSuccess) return e;
  int blocks, threads;
  reduce_geometry(n_, &blocks, &threads);
  hipLaunchKernelGGL(k_wgn_accumulate, dim3(blocks), dim3(threads), 0, h->stream, (const double2 *)d_a,
                     w.d_rx, w.d_ry, n, T, w.d_scal, w.d_partials);
  hipLaunchKernelGGL(k_final_reduce_weighted, dim3(1), dim3(kReduceThreads), 0, h->stream, w.d_partials, blocks, w.d_scal,
                     w.h_res);
  return hipGetLastError();
}

hipError_t launch_plain_gn(icp_handle *h, const double *d_a, const double *d_b, size_t n_, const Pose &T) {
  Workspace &w = h->ws;
  const unsigned n = (unsigned)n_;
  int blocks, threads;
  reduce_geometry(n_, &blocks, &threads);
  hipLaunchKernelGGL(k_plain_accumulate, dim3(blocks), dim3(threads), 0, h->stream, (const double2 *)d_a,
                     (const double2 *)d_b, n, T, w.d_partials);
  hipLaunchKernelGGL(k_final_reduce, dim3(1), dim3(kReduceThreads), 0, h->stream, w.d_partials, blocks, kNAcc + 1,
                     w.d_scal, w.h_res);
  return hipGetLastError();
}

}  // namespace icp
