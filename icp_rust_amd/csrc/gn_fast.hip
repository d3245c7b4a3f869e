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
