// Device helpers shared by the Gauss-Newton kernels (gn.hip, gn_fast.hip).
#pragma once
#include "common.hpp"

namespace icp {

// ------------------------------------------------------------------ keys ---------
__device__ __forceinline__ unsigned long long f2k(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double k2f(unsigned long long k) {
  const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)u);
}

__device__ __forceinline__ int pass_shift(int pass) { return pass < 5 ? 52 - 12 * pass : 0; }
__device__ __forceinline__ int pass_bits(int pass) { return pass < 5 ? 12 : 4; }

// huber::rho / huber::drho on the squared error (src/huber.rs:6-26), k = HUBER_K
__device__ __forceinline__ double huber_rho(double e) {
  const double k = ICP_HUBER_K;
  const double k2 = k * k;
  return (e <= k2) ? e : (2. * k * __dsqrt_rn(e) - k2);
}
__device__ __forceinline__ double huber_drho(double e) {
  const double k = ICP_HUBER_K;
  const double k2 = k * k;
  return (e <= k2) ? 1. : (k / __dsqrt_rn(e));
}

// ------------------------------------------------------------- reductions --------
// Fixed association order (mirrored by the oracle's *_tree variant): a wave folds with
// v[l] += v[l+off], off = 32..1; thread 0 left-folds the wave sums from wave 0.
template <int N>
__device__ __forceinline__ void block_reduce_store(double (&acc)[N], double *__restrict__ out) {
  __shared__ double sm[4][N];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_down(v, off);
    if (lane == 0) sm[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < N) {
    const int k = threadIdx.x;
    double s = sm[0][k];
    for (int w = 1; w < 4; ++w) s = s + sm[w][k];
    out[k] = s;
  }
}


// The last workgroup to arrive gets `true` (CDNA4: per-CU L1s are never refreshed and the
// per-XCD L2s are not coherent, so the hand-off follows the agent-scope release/acquire
// recipe: every wave drains its stores/atomics, workgroup barrier, one lane releases and
// takes a ticket, the last arriver acquires before anyone in it loads).  The ticket word is
// reset by the last arriver, so it is zero again for the next launch.
__device__ __forceinline__ bool last_block_arrives(unsigned *ticket) {
  __shared__ int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == gridDim.x - 1);
    if (last) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    s_last = last;
  }
  __syncthreads();
  return s_last != 0;
}

}  // namespace icp
