// Device helpers shared by the Gauss-Newton kernels (gn.hip, gn_fast.hip, gn_pull.hip, gn_win.hip).
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

// The wave's inclusive prefix sum without the LDS pipe: DPP row shifts inside rows of 16, then the two row broadcasts
// (every lane of the wave active; lane 63 ends up with the wave's total).
__device__ __forceinline__ unsigned wave_scan_inclusive(unsigned v) {
#define ICP_SCAN_DPP(x, ctrl, rows) (unsigned)__builtin_amdgcn_update_dpp(0, (int)(x), ctrl, rows, 0xf, true)
  v += ICP_SCAN_DPP(v, 0x111, 0xf);  // row_shr:1 (zero fill at the row's start)
  v += ICP_SCAN_DPP(v, 0x112, 0xf);  // row_shr:2
  v += ICP_SCAN_DPP(v, 0x114, 0xf);  // row_shr:4
  v += ICP_SCAN_DPP(v, 0x118, 0xf);  // row_shr:8  -> inclusive within each row of 16
  v += ICP_SCAN_DPP(v, 0x142, 0xa);  // row_bcast:15 into rows 1 and 3
  v += ICP_SCAN_DPP(v, 0x143, 0xc);  // row_bcast:31 into rows 2 and 3
#undef ICP_SCAN_DPP
  return v;
}

// ------------------------------------------------------------- reductions --------
// v[l + OFF] for the wave tree below.  Offsets below 16 stay inside a row of 16 lanes for every lane
// whose result is still needed (after the step with offset OFF only lanes < OFF matter, and they read
// lanes < 2 OFF <= 16): a DPP row shift, one v_mov per dword, instead of a trip through the LDS
// crossbar (ds_bpermute).  Lanes whose source falls outside their row get zeros, and are never read
// again.
template <int OFF>
__device__ __forceinline__ double tree_down(double v) {
  if constexpr (OFF >= 16) {
#ifdef ICP_TREE_BPERMUTE
    return __shfl_down(v, OFF);
#else
    // gfx950: v_permlane32_swap exchanges the upper 32 lanes of its first operand with the lower 32 of its second,
    // v_permlane16_swap the odd rows of 16 with the even ones -- with both operands the same value, the SECOND result
    // holds lane l + OFF in every lane l < OFF of each 2 OFF lanes (all the tree reads): two VALU instructions per
    // double instead of two trips through the LDS crossbar
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)(b & 0xffffffffll), hi = (unsigned)((unsigned long long)b >> 32);
    if constexpr (OFF == 32) {
      const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
      const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
      return __longlong_as_double((long long)(((unsigned long long)rh[1] << 32) | rl[1]));
    } else {
      const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
      const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
      return __longlong_as_double((long long)(((unsigned long long)rh[1] << 32) | rl[1]));
    }
#endif
  } else {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), 0x100 + OFF, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x100 + OFF, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
  }
}
template <int N, int OFF>
__device__ __forceinline__ void tree_step(double (&acc)[N]) {
  double t[N];
#pragma unroll
  for (int k = 0; k < N; ++k) t[k] = tree_down<OFF>(acc[k]);
#pragma unroll
  for (int k = 0; k < N; ++k) acc[k] = acc[k] + t[k];
}
// the six steps, step-major (the N shuffles of one step are independent and pipeline)
template <int N>
__device__ __forceinline__ void wave_tree(double (&acc)[N]) {
  tree_step<N, 32>(acc);
  tree_step<N, 16>(acc);
  tree_step<N, 8>(acc);
  tree_step<N, 4>(acc);
  tree_step<N, 2>(acc);
  tree_step<N, 1>(acc);
}

// Fixed association order (mirrored by the oracle's *_tree variant): a wave folds with
// v[l] += v[l+off], off = 32..1; thread 0 left-folds the wave sums from wave 0.
// ... in two halves, for kernels that have a workgroup barrier of their own between the pairs and what follows (a
// histogram to complete): the wave's tree and its sum into LDS go IN FRONT of that barrier -- a wave that is through
// with its pairs folds while the slower ones finish -- and the fold of the wave sums behind it.  Same operations in
// the same order as block_reduce_store.
template <int N>
__device__ __forceinline__ void block_reduce_waves(double (&acc)[N], double (*sm)[N]) {
  wave_tree<N>(acc);
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < N; ++k) sm[threadIdx.x >> 6][k] = acc[k];
  }
}
template <int N, bool SC1 = false>
__device__ __forceinline__ void block_reduce_finish(const double (*sm)[N], double *__restrict__ out) {
  if (threadIdx.x < N) {
    const int k = threadIdx.x;
    double s = sm[0][k];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) s = s + sm[w][k];
    if (SC1) __hip_atomic_store(out + k, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else out[k] = s;
  }
}
template <int N, bool SC1 = false>
__device__ __forceinline__ void block_reduce_store(double (&acc)[N], double *__restrict__ out) {
  __shared__ double sm[16][N];
  // step-major: the N shuffles of one step are independent and pipeline through the LDS
  // crossbar; accumulator-major code runs N dependent 6-step chains one after the other
  // (measured: 14k cycles for N = 13).  Same association order per accumulator either way.
  block_reduce_waves<N>(acc, sm);
  __syncthreads();
  block_reduce_finish<N, SC1>(sm, out);
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
    if (nb <= 32u) {  // few enough arrivals for one word: one round trip instead of two
      if (__hip_atomic_fetch_add(&t->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nb - 1) {
        __hip_atomic_store(&t->top[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = 1;
      }
    } else if (__hip_atomic_fetch_add(&t->shard[sh][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_shard - 1) {
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

// One point's terms of the weighted normal equations (src/lib.rs:241-254) and of the Huber error (:45-50), added
// into the kNSum running sums (common.hpp: per dimension, without 1 / sigma, upper triangle).  UNIFORM: called with
// every lane of the wave active -- the square root / division of the Huber weight then runs only where some lane
// needs it.
// the nine sums of dimension j (upper triangle of w J^T J, then w J^T r), r_j the residual in that dimension
template <bool UNIFORM>
__device__ __forceinline__ void accumulate_dim(int j, const double2 &s, double r_j, const Pose &T, double *S) {
  const double k2 = ICP_HUBER_K * ICP_HUBER_K;
  const double a0 = -s.y, a1 = s.x;  // jacobian(), src/lib.rs:176-184
  const double J0 = j ? T.r10 : T.r00, J1 = j ? T.r11 : T.r01;
  const double J2 = J0 * a0 + J1 * a1;
  const double e = r_j * r_j;
  double w = 1.;  // huber::drho, src/huber.rs:17-26
  if (UNIFORM) {
    if (__ballot(e > k2)) w = huber_drho(e);
  } else {
    w = huber_drho(e);
  }
  const double t0 = w * J0, t1 = w * J1, t2 = w * J2;
  S[0] = S[0] + t0 * J0;
  S[1] = S[1] + t0 * J1;
  S[2] = S[2] + t0 * J2;
  S[3] = S[3] + t1 * J1;
  S[4] = S[4] + t1 * J2;
  S[5] = S[5] + t2 * J2;
  S[6] = S[6] + t0 * r_j;
  S[7] = S[7] + t1 * r_j;
  S[8] = S[8] + t2 * r_j;
}

// huber::rho of the squared residual norm (src/huber.rs:6-15, src/lib.rs:45-50)
template <bool UNIFORM>
__device__ __forceinline__ void accumulate_rho(double r0, double r1, double *err) {
  const double k2 = ICP_HUBER_K * ICP_HUBER_K;
  const double e2 = r0 * r0 + r1 * r1;
  double rho = e2;
  if (UNIFORM) {
    if (__ballot(e2 > k2)) rho = huber_rho(e2);
  } else {
    rho = huber_rho(e2);
  }
  *err = *err + rho;
}

// the nine sums of dimension j with the Huber weight w handed in (accumulate_dim's arithmetic, operation by operation)
__device__ __forceinline__ void accumulate_dim_w(int j, const double2 &s, double r_j, double w, const Pose &T, double *S) {
  const double a0 = -s.y, a1 = s.x;  // jacobian(), src/lib.rs:176-184
  const double J0 = j ? T.r10 : T.r00, J1 = j ? T.r11 : T.r01;
  const double J2 = J0 * a0 + J1 * a1;
  const double t0 = w * J0, t1 = w * J1, t2 = w * J2;
  S[0] = S[0] + t0 * J0;
  S[1] = S[1] + t0 * J1;
  S[2] = S[2] + t0 * J2;
  S[3] = S[3] + t1 * J1;
  S[4] = S[4] + t1 * J2;
  S[5] = S[5] + t2 * J2;
  S[6] = S[6] + t0 * r_j;
  S[7] = S[7] + t1 * r_j;
  S[8] = S[8] + t2 * r_j;
}

template <bool UNIFORM>
__device__ __forceinline__ void accumulate_pair(const double2 &s, double r0, double r1, const Pose &T, double *acc) {
  if constexpr (UNIFORM) {
    // ONE test for the wave instead of three (x, y, the norm): the square roots and divisions of huber::drho / rho run
    // only where some lane of the wave has a squared residual above k^2 -- the streaming launches are bound by
    // instruction issue (profiles/r05_p1_instruction_issue.txt).  A lane at or below k^2 gets w = 1 and rho = e either
    // way, so the sums are the ones accumulate_dim / accumulate_rho give.
    const double k2 = ICP_HUBER_K * ICP_HUBER_K;
    const double e0 = r0 * r0, e1 = r1 * r1, e2 = e0 + e1;
    double w0 = 1., w1 = 1., rho = e2;  // huber::drho, src/huber.rs:17-26; huber::rho, :6-15
    if (__ballot((e0 > k2) | (e1 > k2) | (e2 > k2))) {
      w0 = huber_drho(e0);
      w1 = huber_drho(e1);
      rho = huber_rho(e2);
    }
    accumulate_dim_w(0, s, r0, w0, T, acc);
    accumulate_dim_w(1, s, r1, w1, T, acc + 9);
    acc[18] = acc[18] + rho;
  } else {
    accumulate_dim<UNIFORM>(0, s, r0, T, acc);
    accumulate_dim<UNIFORM>(1, s, r1, T, acc + 9);
    accumulate_rho<UNIFORM>(r0, r1, acc + 18);
  }
}

// Entry k of what the host solves with -- jtj[9] (k = 3 p + q), jtr[3] (k = 9 ..), the Huber error (k = 12) -- from
// the folded sums: g_x S_x + g_y S_y, a dimension whose sigma is 0 left out (src/lib.rs:243-245).  Host and device,
// one definition (the oracle restates it).
__host__ __device__ __forceinline__ double combine_sum(const double *S, int k, const double *sig) {
  if (k >= 12) return S[18];
  int u;
  if (k >= 9) {
    u = 6 + (k - 9);
  } else {
    const int p = k / 3, q = k % 3;
    const int lo = p < q ? p : q, hi = p < q ? q : p;
    u = lo == 0 ? hi : (lo == 1 ? 2 + hi : 5);  // 00 01 02 11 12 22 -> 0 .. 5
  }
  double v = 0.;
  if (sig[0] != 0.) v = v + (1. / sig[0]) * S[u];
  if (sig[1] != 0.) v = v + (1. / sig[1]) * S[9 + u];
  return v;
}

// src/lib.rs:238-255 (+ :45-50) over this thread's points g, g + G, ... in index order (the
// first level of the fixed reduction tree); residuals come from the arrays the first launch wrote
template <int kAccBatch = 4>
__device__ __forceinline__ void accumulate_points(const double2 *__restrict__ a, const double *__restrict__ rx,
                                                  const double *__restrict__ ry, unsigned n, const Pose &T,
                                                  double (&acc)[kNSum]) {
  const unsigned G = gridDim.x * kReduceThreads;
  for (unsigned base = blockIdx.x * kReduceThreads + threadIdx.x; base < n; base += G * kAccBatch) {
    double2 s[kAccBatch];
    double r0[kAccBatch], r1[kAccBatch];
#pragma unroll
    for (int u = 0; u < kAccBatch; ++u) {
      const unsigned i = base + u * G;
      if (i < n) {
        s[u] = a[i];
        r0[u] = rx[i];
        r1[u] = ry[i];
      }
    }
#pragma unroll
    for (int u = 0; u < kAccBatch; ++u) {
      const unsigned i = base + u * G;
      if (i >= n) continue;
      accumulate_pair<true>(s[u], r0[u], r1[u], T, acc);
    }
  }
}

// Executed by the last workgroup: fold the block sums (block order, same tree) into s_tot[kNSum + 1] (LDS; valid
// after the next barrier).  The sums do not depend on the order statistics, so a kernel that still has to select
// those folds first -- the loads of the block sums then share a round trip with the loads of its candidates.
template <int SCOPE = __HIP_MEMORY_SCOPE_AGENT>
__device__ __forceinline__ void fold_block_sums(const double *partials, int blocks, double *s_tot) {
  double tot[kNSum + 1];
#pragma unroll
  for (int k = 0; k < kNSum + 1; ++k) tot[k] = 0.;
  for (int i = threadIdx.x; i < blocks; i += kReduceThreads) {
    double v[kNSum];
#pragma unroll
    for (int k = 0; k < kNSum; ++k)
      v[k] = __hip_atomic_load(&partials[(size_t)i * (kNSum + 1) + k], __ATOMIC_RELAXED, SCOPE);
#pragma unroll
    for (int k = 0; k < kNSum; ++k) tot[k] = tot[k] + v[k];
  }
  block_reduce_store<kNSum + 1>(tot, s_tot);  // (stored by lanes of wave 0)
}

// The same fold, bit for bit, one sum at a time: two registers instead of forty, twenty rounds of two barriers -- for
// kernels that must stay inside a register budget (tests/test_registers.py) and only meet more than 256 block sums on
// clouds beyond 2^20 points.  Every thread of a 512-thread workgroup calls it; s_tot is valid after it returns.
template <int SCOPE = __HIP_MEMORY_SCOPE_AGENT>
__device__ __forceinline__ void fold_block_sums_lean(const double *partials, int blocks, double *s_tot) {
  __shared__ double sm[kReduceThreads / 64];
  const int t = threadIdx.x;
  for (int q = 0; q < kNSum; ++q) {
    double v[1] = {0.};
    for (int i = t; i < blocks; i += kReduceThreads)
      v[0] = v[0] + __hip_atomic_load(&partials[(size_t)i * (kNSum + 1) + q], __ATOMIC_RELAXED, SCOPE);
    wave_tree<1>(v);
    if ((t & 63) == 0) sm[t >> 6] = v[0];
    __syncthreads();
    if (t == 0) {
      double s = sm[0];
      for (int w = 1; w < kReduceThreads / 64; ++w) s = s + sm[w];
      s_tot[q] = s;
    }
    __syncthreads();
  }
  if (t == 0) s_tot[kNSum] = 0.;
  __syncthreads();
}

// The same fold (bit for bit) for at most 256 block sums -- what reduce_geometry yields -- in half the registers:
// thread t takes sums [10 h, 10 h + 10) of row t % 256, h = t / 256, so waves 0-3 and 4-7 each see the rows in the
// lanes fold_block_sums has them in; the wave sums are left-folded per half, plus the one `+ 0.` that stands for
// the four all-zero waves of the general form (it only matters for a sum that is -0.).
// (in two halves, so that the loads can be in flight across other work: fold256_load early, fold256_reduce later)
constexpr int kFoldH = (kNSum + 1) / 2;
template <int SCOPE = __HIP_MEMORY_SCOPE_AGENT>
__device__ __forceinline__ void fold256_load(const double *partials, int blocks, double (&x)[kFoldH]) {
  static_assert(kReduceThreads == 512 && (kNSum + 1) == 20 && kReduceMaxBlocks <= 256, "two halves of ten sums, one row per thread");
  const int t = threadIdx.x, row = t & 255, half = t >> 8;
#pragma unroll
  for (int k = 0; k < kFoldH; ++k)
    x[k] = (row < blocks && half * kFoldH + k < kNSum)
               ? __hip_atomic_load(&partials[(size_t)row * (kNSum + 1) + half * kFoldH + k], __ATOMIC_RELAXED, SCOPE)
               : 0.;
}
__device__ __forceinline__ void fold256_reduce(const double (&x)[kFoldH], int blocks, double *s_tot) {
  constexpr int H = kFoldH;
  __shared__ double sm[8][H];
  const int t = threadIdx.x, row = t & 255, lane = t & 63, wave = t >> 6;
  double v[H];
#pragma unroll
  for (int k = 0; k < H; ++k) v[k] = 0.;
  if (row < blocks) {
#pragma unroll
    for (int k = 0; k < H; ++k) v[k] = v[k] + x[k];
  }
  wave_tree<H>(v);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < H; ++k) sm[wave][k] = v[k];
  }
  __syncthreads();
  if (t < kNSum + 1) {
    const int h = t / H, k = t % H;
    double s = sm[4 * h][k];
    for (int w = 1; w < 4; ++w) s = s + sm[4 * h + w][k];
    s_tot[t] = s + 0.;
  }
}
template <int SCOPE = __HIP_MEMORY_SCOPE_AGENT>
__device__ __forceinline__ void fold_block_sums_256(const double *partials, int blocks, double *s_tot) {
  double x[kFoldH];
  fold256_load<SCOPE>(partials, blocks, x);
  fold256_reduce(x, blocks, s_tot);
}

// ONE of those sums, folded exactly as fold_block_sums_256 folds it (row r in lane r % 64 of wave r / 64, the wave
// tree, the left fold of the four wave sums, the `+ 0.`), by the first 256 threads of a workgroup: twenty workgroups
// fold one sum each beside their other work instead of one workgroup folding all twenty while 255 wait for it.
// sm4: four doubles of LDS.  Every thread of the workgroup must call it (one barrier); the value is valid in all.
// (in two halves so that the load can be in flight across other work: fold_one_load early, fold_one_reduce later)
template <int SCOPE = __HIP_MEMORY_SCOPE_AGENT>
__device__ __forceinline__ double fold_one_load(const double *partials, int blocks, int q) {
  const int t = threadIdx.x;
  return (t < 256 && t < blocks && q < kNSum) ? __hip_atomic_load(&partials[(size_t)t * (kNSum + 1) + q], __ATOMIC_RELAXED, SCOPE)
                                               : 0.;
}
__device__ __forceinline__ double fold_one_reduce(double x, int blocks, double *sm4) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  double v[1] = {0.};
  if (t < 256 && t < blocks) v[0] = v[0] + x;
  wave_tree<1>(v);
  if (t < 256 && lane == 0) sm4[wave] = v[0];
  __syncthreads();
  double s = sm4[0];
  for (int w = 1; w < 4; ++w) s = s + sm4[w];
  return s + 0.;
}
template <int SCOPE = __HIP_MEMORY_SCOPE_AGENT>
__device__ __forceinline__ double fold_one_sum_256(const double *partials, int blocks, int q, double *sm4) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  double v[1] = {0.};
  if (t < 256) {
    const double x = (t < blocks && q < kNSum) ? __hip_atomic_load(&partials[(size_t)t * (kNSum + 1) + q], __ATOMIC_RELAXED, SCOPE) : 0.;
    if (t < blocks) v[0] = v[0] + x;
  }
  wave_tree<1>(v);
  if (t < 256 && lane == 0) sm4[wave] = v[0];
  __syncthreads();
  double s = sm4[0];
  for (int w = 1; w < 4; ++w) s = s + sm4[w];
  return s + 0.;
}

// ... and release the result to the host, which polls `seq` in pinned memory.
__device__ __forceinline__ void publish_folded(const double *s_tot, GnResult *res, unsigned seq, const double (&sig)[2],
                                               const double (&med)[2], int nan_flag, int overflow) {
  // everything the host reads is stored by lanes of wave 0, and the sequence number by its lane 0 with release
  // semantics at system scope: the write-back and the wait that implement the release are wave-wide, so they
  // order every lane's stores before it (an extra __threadfence_system() in front cost ~1 us, measured)
  if (threadIdx.x < 64) {
    if (threadIdx.x < kNAcc + 1)
      res->acc[threadIdx.x] = threadIdx.x < kNAcc ? combine_sum(s_tot, (int)threadIdx.x, sig) : 0.;
    if (threadIdx.x == kNAcc + 1) {
      res->sigma[0] = sig[0];
      res->sigma[1] = sig[1];
      res->median[0] = med[0];
      res->median[1] = med[1];
    }
    if (threadIdx.x == kNAcc + 2) {
      res->nan_flag = nan_flag;
      res->overflow = overflow;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) __hip_atomic_store(&res->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// The same result, stored by the lanes of wave `W` WITHOUT the sequence number: for a finishing launch whose lane 0 is
// busy deriving the next outer pose meanwhile (gn_win.hip: fill_ahead_pose -- a 3 x 3 solve, sin, cos and two pose
// products, as long as these stores to host memory take).  The caller drains (done here), meets at a workgroup barrier
// and releases the sequence number from wave 0 (publish_seq).
template <int W>
__device__ __forceinline__ void publish_values(const double *s_tot, GnResult *res, const double (&sig)[2],
                                               const double (&med)[2], int nan_flag, int overflow) {
  const int t = (int)threadIdx.x - 64 * W;
  if (t >= 0 && t < 64) {
    if (t < kNAcc + 1) res->acc[t] = t < kNAcc ? combine_sum(s_tot, t, sig) : 0.;
    if (t == kNAcc + 1) {
      res->sigma[0] = sig[0];
      res->sigma[1] = sig[1];
      res->median[0] = med[0];
      res->median[1] = med[1];
    }
    if (t == kNAcc + 2) {
      res->nan_flag = nan_flag;
      res->overflow = overflow;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}
__device__ __forceinline__ void publish_seq(GnResult *res, unsigned seq) {
  if (threadIdx.x < 64) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) __hip_atomic_store(&res->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__device__ __forceinline__ void publish_result(const double *partials, GnResult *res, unsigned seq,
                                               const double (&sig)[2], const double (&med)[2], int nan_flag,
                                               int overflow, int blocks_override = 0) {
  __shared__ double s_tot[kNSum + 1];
  // (a sharded evaluation folds the block sums of ALL ranks from a one-workgroup launch)
  fold_block_sums(partials, blocks_override > 0 ? blocks_override : (int)gridDim.x, s_tot);
  __syncthreads();
  publish_folded(s_tot, res, seq, sig, med, nan_flag, overflow);
}

}  // namespace icp
