// "Pull" variant of the short pipeline for one inner Gauss-Newton evaluation
// (src/lib.rs:218-261 + :45-50): seven launches,
//     H(median, digit 0)  H(median, digit 1)  C(median)  H(MAD, digit 0)  H(MAD, digit 1)  C(MAD)  A
// and nobody waits for a last workgroup.  In the first ("push") version every selection launch ended
// with a serial tail (arrival ticket -> one workgroup scans the histograms / ranks the
// candidates: 3.5-8 us per launch by in-kernel stamps).  Here the NEXT launch resolves the
// previous launch's output in its prologue, redundantly in every workgroup (a 2 x 4096-bin
// histogram or a <= 1024-key candidate list is a few us of L2-resident reads, all workgroups
// in parallel): the kernel boundary is the only synchronisation, and workgroup 0 records the
// resolved state for the launch after next.  Only A, which must produce ONE sum, keeps a
// last-workgroup tail.  Results are bit-identical to gn.hip: integer
// histograms, exact rank counting, the same reduction tree.
#include "common.hpp"
#include "gn_device.hpp"

namespace icp {

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int kPullThreads = 1024;
constexpr int kPullBatch = 4;
constexpr int kPullPad = kSelBins + kSelBins / 64;  // bin + bin/64: conflict-free column sums

__device__ __forceinline__ void init_state(SelState (&st)[kSelProblems], unsigned n) {
#pragma unroll
  for (int p = 0; p < kSelProblems; ++p) {
    const bool hi = p & 1;
    st[p].prefix = 0;
    st[p].rank = hi ? (n / 2) : ((n - 1) / 2);  // src/stats.rs:18-27
    st[p].alias = hi ? p - 1 : -1;
    st[p].pad = 0;
  }
}

// Descend one radix digit for all problems from the histograms another launch produced.
// Executed by every workgroup (1024 threads); `lds` holds kSelProblems * kPullPad words.
__device__ __forceinline__ void resolve_hist(uint32_t *lds, const uint32_t *__restrict__ hist,
                                             SelState (&st)[kSelProblems], int pass, bool check_cap,
                                             bool &over) {
  __shared__ unsigned found_bin[kSelProblems], found_below[kSelProblems], found_cnt[kSelProblems];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  bool live[kSelProblems];
#pragma unroll
  for (int p = 0; p < kSelProblems; ++p) live[p] = st[p].alias < 0;
  constexpr int PER = (kSelProblems * kSelBins) / kPullThreads;
  unsigned v[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int j = tid + kPullThreads * u, p = j / kSelBins;
    v[u] = live[p] ? hist[j] : 0u;
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int j = tid + kPullThreads * u, p = j / kSelBins, bin = j % kSelBins;
    lds[p * kPullPad + bin + (bin >> 6)] = v[u];
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
    const uint32_t *img = lds + src * kPullPad;
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
      const unsigned c = img[L * 65 + lane];
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
  const int shift = pass_shift(pass);
#pragma unroll
  for (int p = 0; p < kSelProblems; ++p) {
    st[p].prefix |= (unsigned long long)found_bin[p] << shift;
    st[p].rank -= found_below[p];
    over |= check_cap && found_cnt[p] > (unsigned)kSelCap;
  }
#pragma unroll
  for (int p = 1; p < kSelProblems; p += 2)
    if (st[p].alias >= 0 && st[p].prefix != st[p - 1].prefix) st[p].alias = -1;
  __syncthreads();  // the caller reuses `lds`
}

// The order statistics themselves, from the candidate lists another launch appended.
// Executed by every workgroup (any blockDim); out[p] = key of problem p.
__device__ __forceinline__ void resolve_cand(unsigned long long *keys, const unsigned long long *__restrict__ cand,
                                             const unsigned *__restrict__ cand_cnt,
                                             const SelState (&st)[kSelProblems],
                                             unsigned long long (&out)[kSelProblems], bool &over) {
  __shared__ unsigned long long result[kSelProblems];
  const unsigned tid = threadIdx.x, nt = blockDim.x;
  if (tid < kSelProblems) result[tid] = 0;
  __syncthreads();
#pragma unroll
  for (int l = 0; l < kSelProblems; ++l) {
    if (st[l].alias >= 0) continue;  // reads its partner's list
    unsigned c = cand_cnt[l];
    if (c > (unsigned)kSelCap) {
      over = true;
      c = kSelCap;
    }
    for (unsigned i = tid; i < c; i += nt) keys[i] = cand[l * kSelCap + i];
    __syncthreads();
    for (unsigned i = tid; i < c; i += nt) {
      const unsigned long long ki = keys[i];
      unsigned less = 0, eq = 0;
      for (unsigned j = 0; j < c; ++j) {
        const unsigned long long kj = keys[j];
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
#pragma unroll
  for (int p = 0; p < kSelProblems; ++p) out[p] = result[p];
  __syncthreads();
}

__device__ __forceinline__ double middle(unsigned n, unsigned long long klo, unsigned long long khi) {
  const double lo = k2f(klo), hi = k2f(khi);
  return (n & 1) ? lo : (lo + hi) / 2.;  // src/stats.rs:18-27
}

// the element stream shared by H and C: MODE 0 computes and stores the residuals
// (residual(), src/lib.rs:34-36), MODE 1 re-reads them, MODE 2 keys |r - median| (stats.rs:35)
template <int MODE, typename F>
__device__ __forceinline__ void stream_keys(const double2 *__restrict__ a, const double2 *__restrict__ b,
                                            const Pose &T, double *__restrict__ rx, double *__restrict__ ry,
                                            unsigned n, double med0, double med1, bool &saw_nan, F &&f) {
  const unsigned G = gridDim.x * blockDim.x;
  for (unsigned base = blockIdx.x * blockDim.x + threadIdx.x; base < n; base += G * kPullBatch) {
    double v0[kPullBatch], v1[kPullBatch];
    double2 s[kPullBatch], d[kPullBatch];
#pragma unroll
    for (int u = 0; u < kPullBatch; ++u) {
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
    for (int u = 0; u < kPullBatch; ++u) {
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

// H: histogram of digit `pass` of the keys that match the prefix resolved so far.
//   MODE 0: median stage, digit 0.   MODE 1: median stage, digit 1 (resolves digit 0 first).
//   MODE 2: MAD stage; pass 0 first turns the median candidates into the median, pass 1
//           resolves digit 0 of the MAD search.
template <int MODE>
__global__ __launch_bounds__(kPullThreads) void k_pull_hist(const double2 *__restrict__ a,
                                                            const double2 *__restrict__ b, Pose T,
                                                            double *__restrict__ rx, double *__restrict__ ry,
                                                            unsigned n, int pass, const SelState *sel_in,
                                                            SelState *sel_out, GnScalars *scal,
                                                            const uint32_t *__restrict__ hist_prev,
                                                            uint32_t *hist_out,
                                                            const unsigned long long *__restrict__ cand_prev,
                                                            SelCtl *ctl) {
  __shared__ uint32_t lh[kSelProblems * kPullPad];
  SelState st[kSelProblems];
  bool over = false;
  double med0 = 0., med1 = 0.;
  if (MODE == 2 && pass == 0) {
    // the median: rank the candidates the previous launch collected (state of the median search in `sel`)
#pragma unroll
    for (int p = 0; p < kSelProblems; ++p) st[p] = sel_in[p];
    unsigned long long key[kSelProblems];
    resolve_cand(reinterpret_cast<unsigned long long *>(lh), cand_prev, ctl->cand_cnt_pull[0], st, key, over);
    med0 = middle(n, key[0], key[1]);
    med1 = middle(n, key[2], key[3]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      scal->median[0] = med0;
      scal->median[1] = med1;
    }
    init_state(st, n);
  } else {
    if (MODE == 2) {
      med0 = scal->median[0];
      med1 = scal->median[1];
    }
    if (pass >= 2) {  // third digit (large n): continue from the state the previous launch recorded
#pragma unroll
      for (int p = 0; p < kSelProblems; ++p) st[p] = sel_in[p];
    } else {
      init_state(st, n);
    }
    if (pass >= 1) {
      resolve_hist(lh, hist_prev, st, pass - 1, false, over);
      if (blockIdx.x == 0 && threadIdx.x == 0)
#pragma unroll
        for (int p = 0; p < kSelProblems; ++p) sel_out[p] = st[p];
    } else if (blockIdx.x == 0 && threadIdx.x < kSelProblems) {
      ctl->cand_cnt_pull[1][threadIdx.x] = 0;  // MAD candidates of the previous evaluation: A has read them
    }
  }
  if (over && blockIdx.x == 0 && threadIdx.x == 0) scal->overflow = 1;

  bool active[kSelProblems];
  unsigned long long prefix[kSelProblems];
#pragma unroll
  for (int p = 0; p < kSelProblems; ++p) {
    active[p] = st[p].alias < 0;
    prefix[p] = st[p].prefix;
  }
#pragma unroll
  for (int p = 0; p < kSelProblems; ++p)
    if (active[p])
      for (unsigned i = threadIdx.x; i < kSelBins; i += kPullThreads) lh[p * kSelBins + i] = 0;
  __syncthreads();

  const int shift = pass_shift(pass);
  const unsigned mask = (1u << pass_bits(pass)) - 1u;
  const int hs = shift + pass_bits(pass);  // bits above the current digit (64 at pass 0)
  bool saw_nan = false;
  stream_keys<MODE>(a, b, T, rx, ry, n, med0, med1, saw_nan, [&](unsigned long long k0, unsigned long long k1) {
#pragma unroll
    for (int p = 0; p < kSelProblems; ++p) {
      if (!active[p]) continue;
      const unsigned long long key = (p < 2) ? k0 : k1;
      const bool match = (hs >= 64) || ((key >> hs) == (prefix[p] >> hs));
      if (match) atomicAdd(&lh[p * kSelBins + ((unsigned)(key >> shift) & mask)], 1u);
    }
  });
  if (MODE == 0 && saw_nan) atomicOr(&scal->nan_flag, 1);
  __syncthreads();
#pragma unroll
  for (int p = 0; p < kSelProblems; ++p)
    if (active[p])
      for (unsigned i = threadIdx.x; i < kSelBins; i += kPullThreads) {
        const uint32_t c = lh[p * kSelBins + i];
        if (c) atomicAdd(&hist_out[p * kSelBins + i], c);
      }
}

// C: resolve the last digit (the 2nd, or the 3rd for large n), then append the keys sharing the
// 24- resp. 36-bit prefix to the candidate lists
template <int MODE>
__global__ __launch_bounds__(kPullThreads) void k_pull_compact(const double2 *__restrict__ a,
                                                               const double2 *__restrict__ b, Pose T,
                                                               double *__restrict__ rx, double *__restrict__ ry,
                                                               unsigned n, int stage, int digits,
                                                               const SelState *sel_in, SelState *sel_out,
                                                               GnScalars *scal,
                                                               const uint32_t *__restrict__ hist_prev,
                                                               unsigned long long *cand, SelCtl *ctl) {
  __shared__ uint32_t lh[kSelProblems * kPullPad];
  SelState st[kSelProblems];
  bool over = false;
  double med0 = 0., med1 = 0.;
  if (MODE == 2) {
    med0 = scal->median[0];
    med1 = scal->median[1];
  }
#pragma unroll
  for (int p = 0; p < kSelProblems; ++p) st[p] = sel_in[p];
  resolve_hist(lh, hist_prev, st, digits - 1, true, over);
  // the resolved state goes to the OTHER half of the state buffer: workgroups of this launch
  // that start later must still read the input state
  if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int p = 0; p < kSelProblems; ++p) sel_out[p] = st[p];
    if (over) scal->overflow = 1;
  }
  bool active[kSelProblems];
  unsigned long long prefix[kSelProblems];
#pragma unroll
  for (int p = 0; p < kSelProblems; ++p) {
    active[p] = st[p].alias < 0;
    prefix[p] = st[p].prefix;
  }
  bool saw_nan = false;
  const int cshift = 64 - 12 * digits;  // keys sharing the digits resolved so far
  unsigned *cnt = ctl->cand_cnt_pull[stage];
  stream_keys<MODE>(a, b, T, rx, ry, n, med0, med1, saw_nan, [&](unsigned long long k0, unsigned long long k1) {
#pragma unroll
    for (int p = 0; p < kSelProblems; ++p) {
      if (!active[p]) continue;
      const unsigned long long key = (p < 2) ? k0 : k1;
      if ((key >> cshift) == (prefix[p] >> cshift)) {
        const unsigned pos = atomicAdd(&cnt[p], 1u);
        if (pos < (unsigned)kSelCap) cand[p * kSelCap + pos] = key;
      }
    }
  });
}

// A: sigma from the MAD candidates, then src/lib.rs:238-255 (+ :45-50) in the fixed tree; the
// last workgroup folds the block sums and publishes to the host.
__global__ __launch_bounds__(kReduceThreads) void k_pull_accumulate(const double2 *__restrict__ a,
                                                                    const double *__restrict__ rx,
                                                                    const double *__restrict__ ry,
                                                                    unsigned n, Pose T, const SelState *sel,
                                                                    GnScalars *scal,
                                                                    const unsigned long long *__restrict__ cand,
                                                                    double *partials, uint32_t *hist,
                                                                    SelCtl *ctl, GnResult *res, unsigned seq) {
  __shared__ unsigned long long keys[kSelCap];
  SelState st[kSelProblems];
#pragma unroll
  for (int p = 0; p < kSelProblems; ++p) st[p] = sel[p];
  bool over = false;
  unsigned long long key[kSelProblems];
  resolve_cand(keys, cand, ctl->cand_cnt_pull[1], st, key, over);
  const double sig[2] = {ICP_PPF34 * middle(n, key[0], key[1]),  // src/stats.rs:42-46
                         ICP_PPF34 * middle(n, key[2], key[3])};
  double acc[kNSum];
#pragma unroll
  for (int k = 0; k < kNSum; ++k) acc[k] = 0.;
  accumulate_points(a, rx, ry, n, T, acc);
  block_reduce_store<kNSum, true>(acc, partials + (size_t)blockIdx.x * (kNSum + 1));
  // clear what the next evaluation accumulates into (nobody reads these in this launch)
  const unsigned G = gridDim.x * kReduceThreads;
  // (write-through: the next evaluation may run on the handle's other stream before this
  // kernel's end-of-kernel write-back, see gn_win.hip)
  for (unsigned i = blockIdx.x * kReduceThreads + threadIdx.x;
       i < (unsigned)(kSelRoles * kSelProblems * kSelBins); i += G)
    __hip_atomic_store(&hist[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (blockIdx.x == 0 && threadIdx.x < kSelProblems)
    __hip_atomic_store(&ctl->cand_cnt_pull[0][threadIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  if (!last_block_arrives(&ctl->t[2])) return;
  const int nan_flag = scal->nan_flag, overflow = scal->overflow | (over ? 1 : 0);
  const double med[2] = {scal->median[0], scal->median[1]};
  if (threadIdx.x == 0) __hip_atomic_store(&scal->overflow, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  publish_result(partials, res, seq, sig, med, nan_flag, overflow);
}

hipError_t launch_weighted_gn_pull(icp_handle *h, const double *d_a, const double *d_b, size_t n_, const Pose &T) {
  Workspace &w = h->ws;
  const unsigned n = (unsigned)n_;
  const unsigned per = kPullThreads * kPullBatch;
  unsigned hb = (n + per - 1) / per;
  if (hb > 256) hb = 256;  // one workgroup per CU
  const unsigned hb1 = hb / 2 > 0 ? hb / 2 : 1;  // later digits flush dense histograms: fewer, fatter workgroups
  // two 12-bit digits leave ~1e-4 n keys per prefix; beyond a few million points a third digit
  // keeps the candidate lists short (n <= 2^32)
  static const size_t three_from = exp_env("ICP_PULL_3DIGITS_FROM") ? (size_t)atoll(exp_env("ICP_PULL_3DIGITS_FROM")) : ((size_t)4 << 20);
  const int digits = n_ > three_from ? 3 : 2;
  const double2 *a = (const double2 *)d_a, *b = (const double2 *)d_b;
  hipStream_t s = h->stream;
  const size_t role = (size_t)kSelProblems * kSelBins;
  uint32_t *H = w.d_hist;
  unsigned long long *C0 = w.d_cand, *C1 = w.d_cand + (size_t)kSelProblems * kSelCap;
  const dim3 bt(kPullThreads);
  SelState *S[2] = {w.d_sel, w.d_sel + kSelProblems};  // ping-pong: a launch never rewrites what it reads
  const unsigned long long *no_cand = nullptr;
  const uint32_t *no_hist = nullptr;
  int cur = 1;  // S[cur] = what the next launch reads, S[cur ^ 1] = what it records
  // ---- median stage
  hipLaunchKernelGGL(k_pull_hist<0>, dim3(hb), bt, 0, s, a, b, T, w.d_rx, w.d_ry, n, 0, (const SelState *)S[cur],
                     S[cur ^ 1], w.d_scal, no_hist, H + 0 * role, no_cand, w.d_ctl);
  hipLaunchKernelGGL(k_pull_hist<1>, dim3(hb1), bt, 0, s, a, b, T, w.d_rx, w.d_ry, n, 1, (const SelState *)S[cur],
                     S[cur ^ 1], w.d_scal, (const uint32_t *)(H + 0 * role), H + 1 * role, no_cand, w.d_ctl);
  cur ^= 1;
  if (digits == 3) {
    hipLaunchKernelGGL(k_pull_hist<1>, dim3(hb1), bt, 0, s, a, b, T, w.d_rx, w.d_ry, n, 2, (const SelState *)S[cur],
                       S[cur ^ 1], w.d_scal, (const uint32_t *)(H + 1 * role), H + 2 * role, no_cand, w.d_ctl);
    cur ^= 1;
  }
  hipLaunchKernelGGL(k_pull_compact<1>, dim3(hb), bt, 0, s, a, b, T, w.d_rx, w.d_ry, n, 0, digits,
                     (const SelState *)S[cur], S[cur ^ 1], w.d_scal, (const uint32_t *)(H + (digits - 1) * role), C0,
                     w.d_ctl);
  cur ^= 1;
  // ---- MAD stage (its first launch turns the median candidates into the median)
  hipLaunchKernelGGL(k_pull_hist<2>, dim3(hb), bt, 0, s, a, b, T, w.d_rx, w.d_ry, n, 0, (const SelState *)S[cur],
                     S[cur ^ 1], w.d_scal, no_hist, H + 3 * role, (const unsigned long long *)C0, w.d_ctl);
  hipLaunchKernelGGL(k_pull_hist<2>, dim3(hb1), bt, 0, s, a, b, T, w.d_rx, w.d_ry, n, 1, (const SelState *)S[cur],
                     S[cur ^ 1], w.d_scal, (const uint32_t *)(H + 3 * role), H + 4 * role, no_cand, w.d_ctl);
  cur ^= 1;
  if (digits == 3) {
    hipLaunchKernelGGL(k_pull_hist<2>, dim3(hb1), bt, 0, s, a, b, T, w.d_rx, w.d_ry, n, 2, (const SelState *)S[cur],
                       S[cur ^ 1], w.d_scal, (const uint32_t *)(H + 4 * role), H + 5 * role, no_cand, w.d_ctl);
    cur ^= 1;
  }
  hipLaunchKernelGGL(k_pull_compact<2>, dim3(hb), bt, 0, s, a, b, T, w.d_rx, w.d_ry, n, 1, digits,
                     (const SelState *)S[cur], S[cur ^ 1], w.d_scal, (const uint32_t *)(H + (3 + digits - 1) * role), C1,
                     w.d_ctl);
  cur ^= 1;
  int blocks, threads;
  reduce_geometry(n_, &blocks, &threads);
  hipLaunchKernelGGL(k_pull_accumulate, dim3(blocks), dim3(threads), 0, s, a, w.d_rx, w.d_ry, n, T,
                     (const SelState *)S[cur], w.d_scal, (const unsigned long long *)C1, w.d_partials, w.d_hist, w.d_ctl,
                     w.h_res, ++w.seq);
  return hipGetLastError();
}

}  // namespace icp
