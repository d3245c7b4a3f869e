// Stable sort of the source points by target-grid cell: the once-per-estimate-call plumbing behind the cell-sorted source
// snapshot (nn_grid.hip: prepare_queries).  Sorting the cell keys with the values 0 .. n-1 by a STABLE sort orders the
// points by (cell, original index): a pure function of the inputs, which is what lets the snapshot order double as the
// order in which the Gauss-Newton sums are folded (icp_last_fold_order).
//
// Round 6: hand-written (rounds 1-5 called rocprim's onesweep: three 8-bit passes of 28.5 us, four small launches and
// six fills per call for the benchmark's 21-bit keys -- the only library kernels on the path).  LSD radix sort with
// digits of up to ELEVEN bits: two passes for 21-bit keys, two launches per pass, no fill, no atomics on global memory.
//   tile      16 384 consecutive items = one workgroup of 16 waves, 1 024 consecutive items per wave
//   hist      every wave counts the digits of its items (LDS, 16-bit counters packed in pairs); the tile's counts go
//             to cnt[tile][bin] (a row per tile: coalesced for the writer and for every reader below)
//   items     between two passes (key, index) pairs travel together: one 8-byte store per item; the last pass stores the
//             indices alone (the sorted keys are cell_of[perm[k]]: nothing on the path reads them)
//   scatter   the position of an item = (items of lower bins, all tiles) + (items of its bin in earlier tiles)
//                                     + (items of its bin in earlier waves of its tile) + (rank among its wave's items of
//             that bin, in item order).  The first two come straight out of cnt: a workgroup adds up the rows of the
//             tiles (all of them: totals; those before its own: prefix) and scans the 2 048 bin totals in LDS -- no scan
//             launch; beyond 128 tiles (2M items) one small launch first adds the rows up in chunks of 64, so that a
//             workgroup reads at most T / 64 + 63 rows; the third is a prefix over the tile's 16 waves in
//             LDS; the fourth is taken step by step -- 64 items at a time, in order -- from a match of equal digits
//             across the wave (one ballot per digit bit) and a running counter per bin.  Every term counts items that
//             come EARLIER in the input, so equal keys keep their order: stable, deterministic, whatever the data
//             (a cloud crowded into one cell costs what any other cloud costs).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

namespace icp {

namespace {

constexpr int kRsThreads = 1024, kRsWaves = kRsThreads / 64;
#ifndef ICP_RS_PER_WAVE
#define ICP_RS_PER_WAVE 1024
#endif
constexpr int kRsPerWave = ICP_RS_PER_WAVE, kRsSteps = kRsPerWave / 64;
constexpr unsigned kRsTile = (unsigned)kRsWaves * kRsPerWave;
constexpr int kRsMaxBits = 11, kRsMaxBins = 1 << kRsMaxBits;
constexpr unsigned kRsDirectTiles = 128;  // up to here a scatter workgroup adds up the tiles' rows itself
constexpr unsigned kRsChunk = 64;         // beyond: rows added up in chunks of this many tiles first (k_rs_chunks)
constexpr unsigned kRsMaxTiles = kRsChunk * kRsDirectTiles;
static_assert(kRsMaxBins / 2 == kRsThreads, "one thread per pair of bins");

struct RsLds {
  uint32_t cw[kRsWaves][kRsMaxBins / 2];  // per wave and bin: count, then running offset inside the tile's bin group; two bins per word
  uint32_t gbase[kRsMaxBins];             // where the tile's items of a bin start in the output
  uint32_t wsum[kRsWaves];
};

// the waves' digit counts of tile `tile` (every wave its 1 024 items); the lane's sixteen keys stay in registers (all
// loads in flight at once: a wave per SIMD has nothing else to hide their latency behind)
// PACKED: the items are (key, index) pairs, as the pass in front left them; else keys alone, the index = the position
template <bool PACKED>
__device__ __forceinline__ void rs_count(const void *__restrict__ items, unsigned n, unsigned tile, unsigned shift, unsigned dmask,
                                         RsLds &S, uint32_t (&kreg)[kRsSteps], uint32_t (&vreg)[kRsSteps]) {
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const unsigned base = tile * kRsTile + wave * kRsPerWave + lane;
#pragma unroll
  for (int st = 0; st < kRsSteps; ++st) {
    const unsigned e = base + 64u * st;
    if (PACKED) {
      const uint2 kv = e < n ? reinterpret_cast<const uint2 *>(items)[e] : make_uint2(0u, 0u);
      kreg[st] = kv.x;
      vreg[st] = kv.y;
    } else {
      kreg[st] = e < n ? reinterpret_cast<const uint32_t *>(items)[e] : 0u;
      vreg[st] = e;
    }
  }
  for (unsigned i = tid; i < (unsigned)kRsWaves * (kRsMaxBins / 2); i += kRsThreads) (&S.cw[0][0])[i] = 0u;
  __syncthreads();
#pragma unroll
  for (int st = 0; st < kRsSteps; ++st) {
    const unsigned e = base + 64u * st;
    if (e < n) {
      const unsigned d = (kreg[st] >> shift) & dmask;
      atomicAdd(&S.cw[wave][d >> 1], 1u << (16u * (d & 1u)));  // (at most 1 024 per half: no carry)
    }
  }
  __syncthreads();
}

template <bool PACKED>
__global__ __launch_bounds__(kRsThreads) void k_rs_hist(const void *__restrict__ items, unsigned n, unsigned shift, unsigned bits,
                                                        uint32_t *__restrict__ cnt, unsigned T) {
  extern __shared__ unsigned char rs_raw[];
  RsLds &S = *reinterpret_cast<RsLds *>(rs_raw);
  const unsigned bins = 1u << bits;
  uint32_t kreg[kRsSteps], vreg[kRsSteps];
  rs_count<PACKED>(items, n, blockIdx.x, shift, bins - 1u, S, kreg, vreg);
  const unsigned j = threadIdx.x;  // bins 2 j, 2 j + 1
  if (2u * j < bins) {
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int w = 0; w < kRsWaves; ++w) {
      const uint32_t x = S.cw[w][j];
      lo += x & 0xffffu;
      hi += x >> 16;
    }
    reinterpret_cast<uint2 *>(cnt + (size_t)blockIdx.x * kRsMaxBins)[j] = make_uint2(lo, hi);
  }
}

// rows r0 .. r1 - 1 of `rows` (kRsMaxBins words each) added up for this thread's pair of bins; those below `split`
// also into (g0, g1); eight loads in flight
__device__ __forceinline__ void rs_add_rows(const uint32_t *__restrict__ rows, unsigned r0, unsigned r1, unsigned split, unsigned j,
                                            unsigned &t0, unsigned &t1, unsigned &g0, unsigned &g1) {
  for (unsigned rb = r0; rb < r1; rb += 8u) {
    uint2 v[8];
#pragma unroll
    for (unsigned u = 0; u < 8u; ++u)
      v[u] = rb + u < r1 ? reinterpret_cast<const uint2 *>(rows + (size_t)(rb + u) * kRsMaxBins)[j] : make_uint2(0u, 0u);
#pragma unroll
    for (unsigned u = 0; u < 8u; ++u) {
      t0 += v[u].x;
      t1 += v[u].y;
      if (rb + u < split) {
        g0 += v[u].x;
        g1 += v[u].y;
      }
    }
  }
}

// more than kRsDirectTiles tiles: the rows of each chunk of kRsChunk tiles added up, ctot[chunk][bin]
__global__ __launch_bounds__(kRsThreads) void k_rs_chunks(const uint32_t *__restrict__ cnt, unsigned T, unsigned bits,
                                                          uint32_t *__restrict__ ctot) {
  const unsigned j = threadIdx.x, c = blockIdx.x, r0 = c * kRsChunk, r1 = r0 + kRsChunk < T ? r0 + kRsChunk : T;
  unsigned t0 = 0, t1 = 0, g0 = 0, g1 = 0;
  if (2u * j < (1u << bits)) rs_add_rows(cnt, r0, r1, 0u, j, t0, t1, g0, g1);
  reinterpret_cast<uint2 *>(ctot + (size_t)c * kRsMaxBins)[j] = make_uint2(t0, t1);
}

// CHUNKED: more than kRsDirectTiles tiles -- ctot holds the rows of cnt added up per chunk of kRsChunk tiles
// IN_PACKED / OUT_PACKED: (key, index) pairs between the passes -- one 8-byte store per item instead of two of four;
// the LAST pass stores the indices alone (the sorted keys are cell_of[perm[k]]: nobody on the path reads them)
template <bool CHUNKED, bool IN_PACKED, bool OUT_PACKED>
__global__ __launch_bounds__(kRsThreads) void k_rs_scatter(const void *__restrict__ items, unsigned n, unsigned shift, unsigned bits,
                                                           const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ ctot,
                                                           unsigned T, void *__restrict__ out) {
  extern __shared__ unsigned char rs_raw[];
  RsLds &S = *reinterpret_cast<RsLds *>(rs_raw);
  const unsigned bins = 1u << bits, dmask = bins - 1u, tile = blockIdx.x;
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  uint32_t kreg[kRsSteps], vreg[kRsSteps];
  rs_count<IN_PACKED>(items, n, tile, shift, dmask, S, kreg, vreg);
  {
    // the tile's bins 2 j, 2 j + 1: counts per wave -> exclusive prefix over the waves; where the tile's items of these bins
    // start in the output
    const unsigned j = tid;
    unsigned g0 = 0, g1 = 0, t0 = 0, t1 = 0;
    if (2u * j < bins) {
      unsigned run0 = 0, run1 = 0;
#pragma unroll
      for (int w = 0; w < kRsWaves; ++w) {
        const uint32_t x = S.cw[w][j];
        S.cw[w][j] = run0 | (run1 << 16);
        run0 += x & 0xffffu;
        run1 += x >> 16;
      }
      // items of these bins in all tiles (t), in the tiles before this one (g): the rows of cnt, added up
      if (CHUNKED) {
        const unsigned c = tile / kRsChunk, nc = (T + kRsChunk - 1) / kRsChunk;
        rs_add_rows(ctot, 0u, nc, c, j, t0, t1, g0, g1);                      // whole chunks
        unsigned x0 = 0, x1 = 0;
        rs_add_rows(cnt, c * kRsChunk, tile, tile, j, x0, x1, g0, g1);        // the tiles of this chunk before this one
      } else {
        rs_add_rows(cnt, 0u, T, tile, j, t0, t1, g0, g1);
      }
    }
    {  // exclusive scan of the bin totals over the workgroup (two per thread)
      const unsigned pair = t0 + t1;
      unsigned inc = pair;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(inc, off);
        if ((int)lane >= off) inc += t;
      }
      if (lane == 63u) S.wsum[wave] = inc;
      __syncthreads();
      unsigned wb = 0;
#pragma unroll
      for (int w = 0; w < kRsWaves; ++w) wb += (w < (int)wave) ? S.wsum[w] : 0u;
      const unsigned before = wb + inc - pair;
      g0 += before;
      g1 += before + t0;
    }
    if (2u * j < bins) {
      S.gbase[2u * j] = g0;
      S.gbase[2u * j + 1u] = g1;
    }
  }
  __syncthreads();
  // the wave's items, 64 at a time, in order
  const unsigned base = tile * kRsTile + wave * kRsPerWave + lane;
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int st = 0; st < kRsSteps; ++st) {
    const unsigned e = base + 64u * st;
    const bool valid = e < n;
    const uint32_t key = kreg[st], val = vreg[st];
    const unsigned d = (key >> shift) & dmask;
    unsigned long long same = __ballot(valid);  // the lanes of this step with the same digit
    for (unsigned bit = 0; bit < bits; ++bit) {
      const bool one = (d >> bit) & 1u;
      const unsigned long long b = __ballot(one);
      same &= one ? b : ~b;
    }
    const unsigned long long below = same & lt;
    const uint32_t word = S.cw[wave][d >> 1];
    const unsigned pos = S.gbase[d] + ((word >> (16u * (d & 1u))) & 0xffffu) + (unsigned)__popcll(below);
    if (valid) {
      if (OUT_PACKED) reinterpret_cast<uint2 *>(out)[pos] = make_uint2(key, val);
      else reinterpret_cast<uint32_t *>(out)[pos] = val;
      if (below == 0ull) atomicAdd(&S.cw[wave][d >> 1], (unsigned)__popcll(same) << (16u * (d & 1u)));  // (one lane per digit)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the next step reads the counters this one advanced: same wave, in order)
  }
}

}  // namespace

// stable LSD radix sort of the indices 0 .. n-1 by key: perm_out[k] = the index of the k-th item in ascending (key, index)
// order.  `bits`: the keys' significant bits.  tmp: grow-only scratch of the handle (two arrays of pairs, the counts).
namespace {
template <bool CH, bool IP, bool OP>
void rs_launch_scatter(unsigned T, hipStream_t s, const void *in, unsigned n, unsigned shift, unsigned d, const uint32_t *cnt,
                       const uint32_t *ctot, void *out) {
  hipLaunchKernelGGL((k_rs_scatter<CH, IP, OP>), dim3(T), dim3(kRsThreads), sizeof(RsLds), s, in, n, shift, d, cnt, ctot, T, out);
}
template <bool CH, bool IP, bool OP>
bool rs_grant() {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rs_scatter<CH, IP, OP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)sizeof(RsLds)) == hipSuccess;
}
}  // namespace

hipError_t stable_sort_cells(const uint32_t *keys_in, uint32_t *perm_out, unsigned n, unsigned bits, void *&tmp, size_t &cap_tmp,
                             hipStream_t s) {
  if (n == 0) return hipSuccess;
  if (bits < 1) bits = 1;
  if (bits > 32) bits = 32;
  const unsigned passes = (bits + kRsMaxBits - 1) / kRsMaxBits;
  const unsigned T = (n + kRsTile - 1) / kRsTile;
  if (T > kRsMaxTiles) return hipErrorInvalidValue;  // (134M items: a third level of row sums would be next)
  const size_t cnt_words = (size_t)kRsMaxBins * T;
  const unsigned nc = (T + kRsChunk - 1) / kRsChunk;
  const size_t need = ((size_t)4 * n + cnt_words + (size_t)kRsMaxBins * nc + 64) * sizeof(uint32_t);
  hipError_t e;
  if (need > cap_tmp || !tmp) {
    if (tmp) {
      if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;  // an earlier sort may still use it
      (void)hipFree(tmp);
      tmp = nullptr;
      cap_tmp = 0;
    }
    const size_t want = need + need / 8 + 256;
    if ((e = hipMalloc(&tmp, want)) != hipSuccess) return e;
    cap_tmp = want;
  }
  static signed char lds_granted[64] = {};  // 72 KB of LDS per workgroup: granted once per device and process
  int dev = 0;
  (void)hipGetDevice(&dev);
  signed char &granted = lds_granted[(unsigned)dev % 64u];
  if (granted == 0) {
    bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rs_hist<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RsLds)) == hipSuccess;
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rs_hist<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RsLds)) == hipSuccess;
    ok = ok && rs_grant<false, false, false>() && rs_grant<false, false, true>() && rs_grant<false, true, false>() && rs_grant<false, true, true>();
    ok = ok && rs_grant<true, false, false>() && rs_grant<true, false, true>() && rs_grant<true, true, false>() && rs_grant<true, true, true>();
    granted = ok ? 1 : -1;
    if (!ok) (void)hipGetLastError();
  }
  if (granted < 0) return hipErrorInvalidValue;
  uint2 *pairs_a = reinterpret_cast<uint2 *>(tmp), *pairs_b = pairs_a + n;
  uint32_t *cnt = reinterpret_cast<uint32_t *>(pairs_b + n), *ctot = cnt + cnt_words;
  // pass p reads what pass p - 1 wrote (pairs, alternating between the two scratch arrays); the first reads the keys,
  // the last writes the indices
  const void *in = keys_in;
  unsigned shift = 0;
  for (unsigned p = 0; p < passes; ++p) {
    const unsigned left = bits - shift, d = (left + (passes - p) - 1) / (passes - p);  // the remaining bits, evenly
    const bool first = p == 0, last = p + 1 == passes, chunked = T > kRsDirectTiles;
    void *out = last ? static_cast<void *>(perm_out) : static_cast<void *>((p & 1u) ? pairs_b : pairs_a);
    if (first) hipLaunchKernelGGL(k_rs_hist<false>, dim3(T), dim3(kRsThreads), sizeof(RsLds), s, in, n, shift, d, cnt, T);
    else hipLaunchKernelGGL(k_rs_hist<true>, dim3(T), dim3(kRsThreads), sizeof(RsLds), s, in, n, shift, d, cnt, T);
    if (chunked) hipLaunchKernelGGL(k_rs_chunks, dim3(nc), dim3(kRsThreads), 0, s, (const uint32_t *)cnt, T, d, ctot);
    const uint32_t *cc = cnt, *ct = chunked ? ctot : nullptr;
    if (chunked) {
      if (first && last) rs_launch_scatter<true, false, false>(T, s, in, n, shift, d, cc, ct, out);
      else if (first) rs_launch_scatter<true, false, true>(T, s, in, n, shift, d, cc, ct, out);
      else if (last) rs_launch_scatter<true, true, false>(T, s, in, n, shift, d, cc, ct, out);
      else rs_launch_scatter<true, true, true>(T, s, in, n, shift, d, cc, ct, out);
    } else {
      if (first && last) rs_launch_scatter<false, false, false>(T, s, in, n, shift, d, cc, ct, out);
      else if (first) rs_launch_scatter<false, false, true>(T, s, in, n, shift, d, cc, ct, out);
      else if (last) rs_launch_scatter<false, true, false>(T, s, in, n, shift, d, cc, ct, out);
      else rs_launch_scatter<false, true, true>(T, s, in, n, shift, d, cc, ct, out);
    }
    in = out;
    shift += d;
  }
  return hipGetLastError();
}

}  // namespace icp

// Observability (include/icp_mi355x_debug.h): the sort alone on host arrays -- keys_out / perm_out = the keys in
// ascending order and, for equal keys, ascending original index (tests/test_gpu_sort.py compares with numpy's stable sort)
extern "C" int icp_debug_sort_cells(const uint32_t *keys, size_t n, unsigned bits, uint32_t *keys_out, uint32_t *perm_out) {
  if ((n > 0 && (!keys || !keys_out || !perm_out)) || n >= 0xffff0000ull) return 4;  // ICP_BAD_ARGUMENT
  if (n == 0) return 0;
  uint32_t *d_in = nullptr, *d_p = nullptr;
  void *tmp = nullptr;
  size_t cap = 0;
  hipError_t e = hipMalloc(&d_in, n * 4);
  if (e == hipSuccess) e = hipMalloc(&d_p, n * 4);
  if (e == hipSuccess) e = hipMemcpy(d_in, keys, n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = icp::stable_sort_cells(d_in, d_p, (unsigned)n, bits, tmp, cap, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(perm_out, d_p, n * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess)
    for (size_t k = 0; k < n; ++k) keys_out[k] = perm_out[k] < n ? keys[perm_out[k]] : 0xffffffffu;  // (the sort leaves the indices)
  (void)hipFree(d_in);
  (void)hipFree(d_p);
  (void)hipFree(tmp);
  return e == hipSuccess ? 0 : 6;  // ICP_OK / ICP_HIP_ERROR
}
