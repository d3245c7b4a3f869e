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
template <int N, bool SC1 = false>
__device__ __forceinline__ void block_reduce_store(double (&acc)[N], double *__restrict__ out) {
  __shared__ double sm[16][N];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // step-major: the N shuffles of one step are independent and pipeline through the LDS
  // crossbar; accumulator-major code runs N dependent 6-step chains one after the other
  // (measured: 14k cycles for N = 13).  Same association order per accumulator either way.
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
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
  __syncthreads();
  if (threadIdx.x < N) {
    const int k = threadIdx.x;
    double s = sm[0][k];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) s = s + sm[w][k];
    if (SC1) __hip_atomic_store(out + k, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else out[k] = s;
  }
}


// The last workgroup to arrive gets `true`.  CDNA4 hand-off rules (per-CU L1s are never
// refreshed, per-XCD L2s are not coherent): every byte handed to the last workgroup is
// written by a device-scope atomic or an `sc1` (write-through) store and read back with
// `sc1` loads (__hip_atomic_load/_store, relaxed, agent scope), so no L2 write-back or
// invalidate is needed -- a release fence here costs 10-20 us per launch because the L2 is
// full of freshly written residuals.  What IS needed: every wave drains its own stores and
// atomics (s_waitcnt vmcnt(0)) before the workgroup barrier behind which one lane takes the
// ticket; the last arriver's other waves load only after the second barrier.  The ticket is
// reset by the last arriver, so it is zero again for the next launch.
__device__ __forceinline__ bool last_block_arrives(TicketSet *t) {
  __shared__ int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned nb = gridDim.x, sh = blockIdx.x & 15u;
    const unsigned in_shard = (nb - sh + 15u) >> 4;  // workgroups with this shard id
    int last = 0;
    if (__hip_atomic_fetch_add(&t->shard[sh][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_shard - 1) {
      __hip_atomic_store(&t->shard[sh][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned nshards = nb < 16u ? nb : 16u;
      if (__hip_atomic_fetch_add(&t->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nshards - 1) {
        __hip_atomic_store(&t->top[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = 1;
      }
    }
    s_last = last;
  }
  __syncthreads();
  return s_last != 0;
}

}  // namespace icp
