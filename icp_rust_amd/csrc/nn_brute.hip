// Exact brute-force nearest neighbour for gfx950, fused with the pose transform.
//
// Replaces, for one outer ICP iteration, the reference's
//   src.map(transform / transform_xy)            src/lib.rs:113-116, 156-159 (-> :52-57)
//   .map(|sp| dst[kdtree.search(&sp).0.unwrap()]) src/lib.rs:118-124, 161-167
//   get_xy(..)                                    src/lib.rs:86-89
// The kd-tree (un-vendored crate `nearest_neighbor`) is replaced by an exhaustive scan:
// exact NN is exact NN.  Contract: d^2 = ((dx*dx + dy*dy) + dz*dz) in f64
// without FMA contraction, ties -> lowest target index (targets are scanned in ascending
// index order with a strict `<`).
//
// Mapping to the hardware: the path is FP64-VALU bound (8 f64 flops + compare/select per
// pair, N*M pairs; 52 MB of compulsory HBM traffic at 1M x 1M), so the kernel is built
// to keep the vector ALUs busy and everything else off the critical path:
//   * every lane owns R query points in registers (R x {x,y,z,best,idx});
//   * targets stream through LDS as SoA tiles, filled with coalesced 8-B loads and read
//     back with wave-uniform addresses (LDS broadcast: one ds_read feeds 64 x R pairs);
//   * the target array is padded with +inf to a whole number of tiles, so the inner loop
//     has no bounds test;
//   * small source clouds split the target range over blockIdx.y so that the grid still
//     covers the 1024 SIMDs; the per-chunk minima are merged in index order.
#include "common.hpp"

namespace icp {

constexpr int kNnThreads = 256;
constexpr int kNnTile = 1024;  // targets per LDS tile

// FULL: every chunk is whole tiles (compile-time trip count, unrolled); otherwise chunks are whole
// 64-target granules of a tile (small clouds)
template <int DIM, int R, bool XFORM, bool FULL = true>
__global__ __launch_bounds__(kNnThreads) void k_nn_brute(
    const double *__restrict__ src, unsigned n, const double *__restrict__ tx,
    const double *__restrict__ ty, const double *__restrict__ tz, unsigned m_pad, unsigned chunk,
    Pose T, double *__restrict__ part_d, uint32_t *__restrict__ part_i) {
  __shared__ double sx[kNnTile];
  __shared__ double sy[kNnTile];
  __shared__ double sz[DIM == 3 ? kNnTile : 1];

  const unsigned q0 = blockIdx.x * (kNnThreads * R) + threadIdx.x;
  double qx[R], qy[R], qz[R], best[R];
  unsigned bi[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const unsigned q = q0 + r * kNnThreads;
    double x = 0., y = 0., z = 0.;
    if (q < n) {
      x = src[(size_t)q * DIM + 0];
      y = src[(size_t)q * DIM + 1];
      if (DIM == 3) z = src[(size_t)q * DIM + 2];
    }
    if (XFORM) {  // Transform::transform, src/transform.rs:22-24 (z untouched, lib.rs:52-57)
      const double nx = (T.r00 * x + T.r01 * y) + T.tx;
      const double ny = (T.r10 * x + T.r11 * y) + T.ty;
      x = nx;
      y = ny;
    }
    qx[r] = x;
    qy[r] = y;
    qz[r] = z;
    best[r] = __builtin_huge_val();
    bi[r] = 0xffffffffu;
  }

  const unsigned t_begin = blockIdx.y * chunk;
  const unsigned t_end = min(m_pad, t_begin + chunk);
  for (unsigned t0 = t_begin; t0 < t_end; t0 += kNnTile) {
    const unsigned len = FULL ? (unsigned)kNnTile : min((unsigned)kNnTile, t_end - t0);  // a multiple of 64
    __syncthreads();
    for (unsigned k = threadIdx.x; k < len; k += kNnThreads) {
      sx[k] = tx[t0 + k];
      sy[k] = ty[t0 + k];
      if (DIM == 3) sz[k] = tz[t0 + k];
    }
    __syncthreads();
auto pair_step = [&](unsigned k) {
      const double px = sx[k], py = sy[k];
      const double pz = (DIM == 3) ? sz[k] : 0.;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const double dx = qx[r] - px;
        const double dy = qy[r] - py;
        double d = dx * dx + dy * dy;
        if (DIM == 3) {
          const double dz = qz[r] - pz;
          d = d + dz * dz;
        }
        if (d < best[r]) {
          best[r] = d;
          bi[r] = t0 + k;
        }
      }
    };
    if (FULL) {
#pragma unroll 4
      for (unsigned k = 0; k < (unsigned)kNnTile; ++k) pair_step(k);
    } else {
      for (unsigned k = 0; k < len; ++k) pair_step(k);
    }
  }

#pragma unroll
  for (int r = 0; r < R; ++r) {
    const unsigned q = q0 + r * kNnThreads;
    if (q < n) {
      part_i[(size_t)blockIdx.y * n + q] = bi[r];
      if (gridDim.y > 1) part_d[(size_t)blockIdx.y * n + q] = best[r];
    }
  }
}

// The same sweep with an f32 screen in front of the exact f64 test.  Targets are also kept as
// fl32(p - org) (org = lower corner of the target bounding box); with qf = fl32(q - org) every
// component of df = qf - pf is within 2^-23 (E + |q - org|) of the true q - p, so the true
// distance is >= |df| - ec (ec = sqrt(3) times that bound) and |df|^2 >= s32 (1 - 1e-6) for the
// f32-evaluated sum.  A pair is skipped iff that lower bound exceeds sqrt(best) -- it can
// neither win nor tie -- and every other pair is evaluated exactly with the contract's f64
// formula, in the same ascending target order with the same strict `<`.  Same indices, bit
// for bit; ~8 f32 instead of ~12 f64-rate instructions for all but ~ln(M) pairs per query.
template <int DIM, int R, bool XFORM, bool FULL = true>
__global__ __launch_bounds__(kNnThreads) void k_nn_brute_scr(
    const double *__restrict__ src, unsigned n, const double *__restrict__ tx,
    const double *__restrict__ ty, const double *__restrict__ tz, const float *__restrict__ fx,
    const float *__restrict__ fy, const float *__restrict__ fz, unsigned m_pad, unsigned chunk, Pose T,
    double ox, double oy, double oz, double scale, double *__restrict__ part_d,
    uint32_t *__restrict__ part_i) {
  __shared__ double sx[kNnTile];
  __shared__ double sy[kNnTile];
  __shared__ double sz[DIM == 3 ? kNnTile : 1];
  __shared__ float gx[kNnTile];
  __shared__ float gy[kNnTile];
  __shared__ float gz[DIM == 3 ? kNnTile : 1];

  const unsigned q0 = blockIdx.x * (kNnThreads * R) + threadIdx.x;
  double qx[R], qy[R], qz[R], best[R], ec[R];
  float hx[R], hy[R], hz[R], thr[R];
  unsigned bi[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const unsigned q = q0 + r * kNnThreads;
    double x = 0., y = 0., z = 0.;
    if (q < n) {
      x = src[(size_t)q * DIM + 0];
      y = src[(size_t)q * DIM + 1];
      if (DIM == 3) z = src[(size_t)q * DIM + 2];
    }
    if (XFORM) {  // Transform::transform, src/transform.rs:22-24 (z untouched, lib.rs:52-57)
      const double nx = (T.r00 * x + T.r01 * y) + T.tx;
      const double ny = (T.r10 * x + T.r11 * y) + T.ty;
      x = nx;
      y = ny;
    }
    qx[r] = x;
    qy[r] = y;
    qz[r] = z;
    const double ax = x - ox, ay = y - oy, az = DIM == 3 ? z - oz : 0.;
    hx[r] = (float)ax;
    hy[r] = (float)ay;
    hz[r] = (float)az;
    ec[r] = (fmax(fmax(fabs(ax), fabs(ay)), fabs(az)) + 2. * scale) * 1.2e-7 * 1.7320508075688774;
    best[r] = __builtin_huge_val();
    thr[r] = __builtin_huge_valf();
    bi[r] = 0xffffffffu;
  }

  const unsigned t_begin = blockIdx.y * chunk;
  const unsigned t_end = min(m_pad, t_begin + chunk);
  for (unsigned t0 = t_begin; t0 < t_end; t0 += kNnTile) {
    const unsigned len = FULL ? (unsigned)kNnTile : min((unsigned)kNnTile, t_end - t0);  // a multiple of 64
    __syncthreads();
    for (unsigned k = threadIdx.x; k < len; k += kNnThreads) {
      sx[k] = tx[t0 + k];
      sy[k] = ty[t0 + k];
      gx[k] = fx[t0 + k];
      gy[k] = fy[t0 + k];
      if (DIM == 3) {
        sz[k] = tz[t0 + k];
        gz[k] = fz[t0 + k];
      }
    }
    __syncthreads();
auto pair_step = [&](unsigned k) {
      const float px = gx[k], py = gy[k];
      const float pz = (DIM == 3) ? gz[k] : 0.f;
      float s[R];
      unsigned long long any = 0;  // per-r compares go straight to scalar masks, OR-ed on the SALU
      if constexpr (R >= 2) {
        // two queries per instruction (v_pk_add/mul/fma_f32: the packed rate is what the FP32 vector
        // peak is quoted on) and fused multiply-adds.  The contract's "no FMA" is about the exact f64
        // evaluation below; the screen only needs s <= |df|^2 (1 + 1e-6), and fewer roundings keep
        // that with room to spare.
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 px2 = {px, px}, py2 = {py, py}, pz2 = {pz, pz};
#pragma unroll
        for (int r = 0; r < R; r += 2) {
          const f2 qx2 = {hx[r], hx[r + 1]}, qy2 = {hy[r], hy[r + 1]};
          const f2 dx = qx2 - px2, dy = qy2 - py2;
          f2 v = __builtin_elementwise_fma(dy, dy, dx * dx);
          if (DIM == 3) {
            const f2 qz2 = {hz[r], hz[r + 1]};
            const f2 dz = qz2 - pz2;
            v = __builtin_elementwise_fma(dz, dz, v);
          }
          s[r] = v.x;
          s[r + 1] = v.y;
          any |= __ballot(!(v.x > thr[r]));
          any |= __ballot(!(v.y > thr[r + 1]));
        }
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float dx = hx[r] - px;
          const float dy = hy[r] - py;
          float v = dx * dx + dy * dy;
          if (DIM == 3) {
            const float dz = hz[r] - pz;
            v = v + dz * dz;
          }
          s[r] = v;
          any |= __ballot(!(v > thr[r]));
        }
      }
      if (any) {  // wave-uniform: some lane has a pair that could win or tie
        const double ex = sx[k], ey = sy[k];
        const double ez = (DIM == 3) ? sz[k] : 0.;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (s[r] > thr[r]) continue;
          const double dx = qx[r] - ex;
          const double dy = qy[r] - ey;
          double d = dx * dx + dy * dy;
          if (DIM == 3) {
            const double dz = qz[r] - ez;
            d = d + dz * dz;
          }
          if (d < best[r]) {
            best[r] = d;
            bi[r] = t0 + k;
            const double rr = sqrt(d) + ec[r];
            thr[r] = (float)(rr * rr * 1.000004) * 1.000001f + 1e-37f;  // rounded up
          }
        }
      }
    };
    if (FULL) {
#pragma unroll 4
      for (unsigned k = 0; k < (unsigned)kNnTile; ++k) pair_step(k);
    } else {
      for (unsigned k = 0; k < len; ++k) pair_step(k);
    }
  }

#pragma unroll
  for (int r = 0; r < R; ++r) {
    const unsigned q = q0 + r * kNnThreads;
    if (q < n) {
      part_i[(size_t)blockIdx.y * n + q] = bi[r];
      if (gridDim.y > 1) part_d[(size_t)blockIdx.y * n + q] = best[r];
    }
  }
}

// The sweep with the screen as a DOT PRODUCT (round 2).  The screen above spends 3 subtractions, a
// multiplication and two FMAs on a pair; |q - p|^2 = |q|^2 + (|p|^2 - 2 q.p) needs three FMAs: the
// bracket is accumulated from the target's precomputed |p|^2 with the query's precomputed -2q, and
// |q|^2 moves into the threshold.  7 -> 4 vector operations per pair (the compare included), which is
// what this kernel is bound by (10^12 pairs against 52 MB).  The price is
// cancellation: the bracket is evaluated in f32 on magnitudes up to (|q| + |p|)^2, so it differs from
// the exact |qf - pf|^2 - |qf|^2 by up to 6 * 2^-24 (|q| + |p|)^2 (three FMA roundings on partial sums
// bounded by |p|^2 + 2|q||p|, the rounding of the stored |p|^2, of |q|^2 and of the threshold
// arithmetic).  That bound -- `margin`, with coordinates taken relative to the CENTRE of the target
// bounding box to keep it small -- is added to the threshold, so the screen lets more pairs through to
// the exact f64 test (at 1M points in an 80 m box: pairs within ~0.15 m of a query instead of ~0.07 m),
// never fewer: same indices, bit for bit.  A tile holds {x, y, z, |p|^2} per target: one 16-byte LDS
// broadcast read per target.
template <int DIM, int R, bool XFORM, bool FULL = true>
__global__ __launch_bounds__(kNnThreads) void k_nn_brute_dot(
    const double *__restrict__ src, unsigned n, const double *__restrict__ tx, const double *__restrict__ ty,
    const double *__restrict__ tz, const float *__restrict__ fx, const float *__restrict__ fy,
    const float *__restrict__ fz, unsigned m_pad, unsigned chunk, Pose T, double ox, double oy, double oz, double scale,
    float pmax, double *__restrict__ part_d, uint32_t *__restrict__ part_i) {
  static_assert(R >= 2 && R % 2 == 0, "queries are processed two per packed instruction");
  typedef float f2 __attribute__((ext_vector_type(2)));
  __shared__ double sx[kNnTile];
  __shared__ double sy[kNnTile];
  __shared__ double sz[DIM == 3 ? kNnTile : 1];
  __shared__ float4 g4[kNnTile];

  const unsigned q0 = blockIdx.x * (kNnThreads * R) + threadIdx.x;
  // Registers decide how many waves hide the LDS latency of the hot loop (the first version kept the
  // queries' f64 coordinates resident: 174 VGPRs, two waves per SIMD, and every target step waited out
  // its LDS read -- profiles/r02_brute_pmc.txt).  The exact coordinates are needed only in the rare
  // exact test: they are re-read from `src` there (same loads, same arithmetic, same bits).
  auto query = [&](int r, double &x, double &y, double &z) {
    const unsigned q = q0 + r * kNnThreads;
    x = y = z = 0.;
    if (q < n) {
      x = src[(size_t)q * DIM + 0];
      y = src[(size_t)q * DIM + 1];
      if (DIM == 3) z = src[(size_t)q * DIM + 2];
    }
    if (XFORM) {  // Transform::transform, src/transform.rs:22-24 (z untouched, lib.rs:52-57)
      const double nx = (T.r00 * x + T.r01 * y) + T.tx;
      const double ny = (T.r10 * x + T.r11 * y) + T.ty;
      x = nx;
      y = ny;
    }
  };
  double best[R];
  f2 mx[R / 2], my[R / 2], mz[R / 2];  // -2 (q - org), two queries per register pair
  float thr[R], mq[R];                 // mq = margin - |q - org|^2
  float ecf = 0.f;                     // (one bound for the lane's queries: the largest)
  unsigned bi[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    double x, y, z;
    query(r, x, y, z);
    const double ax = x - ox, ay = y - oy, az = DIM == 3 ? z - oz : 0.;
    const float hx = (float)ax, hy = (float)ay, hz = (float)az;
    if (r & 1) {
      mx[r / 2].y = -2.f * hx;
      my[r / 2].y = -2.f * hy;
      mz[r / 2].y = -2.f * hz;
    } else {
      mx[r / 2].x = -2.f * hx;
      my[r / 2].x = -2.f * hy;
      mz[r / 2].x = -2.f * hz;
    }
    const float qq = (float)(((double)hx * hx + (double)hy * hy) + (double)hz * hz);  // exact in f64, rounded once
    const float qn = __builtin_sqrtf(qq) * 1.000001f + pmax;
    mq[r] = 5.5e-7f * qn * qn - qq;  // see the header comment
    // rounded up: the bound on |(qf - pf) - (q - p)|, as in the screen above
    ecf = fmaxf(ecf, (float)((fmax(fmax(fabs(ax), fabs(ay)), fabs(az)) + 2. * scale) * 1.2e-7 * 1.7320508075688774) * 1.000001f);
    best[r] = __builtin_huge_val();
    thr[r] = __builtin_huge_valf();
    bi[r] = 0xffffffffu;
  }

  const unsigned t_begin = blockIdx.y * chunk;
  const unsigned t_end = min(m_pad, t_begin + chunk);
  for (unsigned t0 = t_begin; t0 < t_end; t0 += kNnTile) {
    const unsigned len = FULL ? (unsigned)kNnTile : min((unsigned)kNnTile, t_end - t0);  // a multiple of 64
    __syncthreads();
    for (unsigned k = threadIdx.x; k < len; k += kNnThreads) {
      sx[k] = tx[t0 + k];
      sy[k] = ty[t0 + k];
      const float px = fx[t0 + k], py = fy[t0 + k];
      float pz = 0.f;
      if (DIM == 3) {
        sz[k] = tz[t0 + k];
        pz = fz[t0 + k];
      }
      g4[k] = make_float4(px, py, pz, __builtin_fmaf(pz, pz, __builtin_fmaf(py, py, px * px)));
    }
    __syncthreads();
    auto pair_step = [&](unsigned k, const float4 g) {
      const f2 px2 = {g.x, g.x}, py2 = {g.y, g.y}, pz2 = {g.z, g.z}, pp2 = {g.w, g.w};
      float s[R];
      unsigned long long any = 0;
#pragma unroll
      for (int r = 0; r < R; r += 2) {
        f2 v = __builtin_elementwise_fma(mx[r / 2], px2, pp2);
        v = __builtin_elementwise_fma(my[r / 2], py2, v);
        if (DIM == 3) v = __builtin_elementwise_fma(mz[r / 2], pz2, v);
        s[r] = v.x;
        s[r + 1] = v.y;
        any |= __ballot(!(v.x > thr[r]));
        any |= __ballot(!(v.y > thr[r + 1]));
      }
      if (any) {  // wave-uniform: some lane has a pair that could win or tie
        const double ex = sx[k], ey = sy[k];
        const double ez = (DIM == 3) ? sz[k] : 0.;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (s[r] > thr[r]) continue;
          double x, y, z;
          query(r, x, y, z);
          const double dx = x - ex;
          const double dy = y - ey;
          double d = dx * dx + dy * dy;
          if (DIM == 3) {
            const double dz = z - ez;
            d = d + dz * dz;
          }
          if (d < best[r]) {
            best[r] = d;
            bi[r] = t0 + k;
            const double rr = sqrt(d) + (double)ecf;
            const float tq = (float)(rr * rr * 1.000004) * 1.000001f + 1e-37f;  // rounded up, as in the screen above
            thr[r] = tq + mq[r];
          }
        }
      }
    };
    // the LDS read of the next target is issued before the current one is processed: with few waves per
    // SIMD nothing else hides its latency
    if (FULL) {
      float4 nxt = g4[0];
#pragma unroll 4
      for (unsigned k = 0; k < (unsigned)kNnTile; ++k) {
        const float4 g = nxt;
        nxt = g4[(k + 1) & (kNnTile - 1)];
        pair_step(k, g);
      }
    } else {
      for (unsigned k = 0; k < len; ++k) pair_step(k, g4[k]);
    }
  }

#pragma unroll
  for (int r = 0; r < R; ++r) {
    const unsigned q = q0 + r * kNnThreads;
    if (q < n) {
      part_i[(size_t)blockIdx.y * n + q] = bi[r];
      if (gridDim.y > 1) part_d[(size_t)blockIdx.y * n + q] = best[r];
    }
  }
}

// Merge the per-chunk minima (ascending chunk order + strict `<` keeps the lowest
// index on ties), then emit idx and the matched xy pairs a = xy(T.src), b = xy(dst[idx]).
template <int DIM, bool XFORM>
__global__ __launch_bounds__(256) void k_nn_finalize(
    const double *__restrict__ src, unsigned n, Pose T, const double *__restrict__ part_d,
    const uint32_t *__restrict__ part_i, unsigned chunks, const double *__restrict__ tx,
    const double *__restrict__ ty, uint32_t *__restrict__ idx, double2 *__restrict__ a,
    double2 *__restrict__ b) {
  const unsigned q = blockIdx.x * 256 + threadIdx.x;
  if (q >= n) return;
  unsigned bi = part_i[q];
  if (chunks > 1) {
    double best = part_d[q];
    for (unsigned c = 1; c < chunks; ++c) {
      const double d = part_d[(size_t)c * n + q];
      if (d < best) {
        best = d;
        bi = part_i[(size_t)c * n + q];
      }
    }
  }
  if (bi == 0xffffffffu) bi = 0;  // no finite distance at all: index 0, as a scan from 0 would
  if (idx) idx[q] = bi;
  if (a) {
    double x = src[(size_t)q * DIM + 0];
    double y = src[(size_t)q * DIM + 1];
    if (XFORM) {
      const double nx = (T.r00 * x + T.r01 * y) + T.tx;
      const double ny = (T.r10 * x + T.r11 * y) + T.ty;
      x = nx;
      y = ny;
    }
    a[q] = make_double2(x, y);
  }
  if (b) b[q] = make_double2(tx[bi], ty[bi]);
}

// Tiny clouds (the reference's 2-D scans: ~650 x ~650 points): the tiled sweep + the merge of its
// partial minima are two launches of ~17 + 5 us, all of it latency.  Here 16 lanes share a query --
// lane `sub` scans targets sub, sub + 16, ... straight from the AoS cloud (it sits in L1 after the
// first query of the workgroup) with the contract's formula -- and the lanes merge their minima by
// (d^2, index): one launch, same results.
template <int DIM, bool XFORM>
__global__ __launch_bounds__(256) void k_nn_tiny(const double *__restrict__ src, unsigned n, Pose T,
                                                 const double *__restrict__ dst, unsigned m,
                                                 uint32_t *__restrict__ idx, double2 *__restrict__ a,
                                                 double2 *__restrict__ b) {
  const unsigned t = blockIdx.x * 256 + threadIdx.x;
  const unsigned q = t >> 4, sub = t & 15u;
  if (q >= n) return;  // whole groups leave together
  double qx = src[(size_t)q * DIM + 0], qy = src[(size_t)q * DIM + 1];
  const double qz = DIM == 3 ? src[(size_t)q * DIM + 2] : 0.;
  if (XFORM) {  // Transform::transform, src/transform.rs:22-24
    const double nx = (T.r00 * qx + T.r01 * qy) + T.tx;
    const double ny = (T.r10 * qx + T.r11 * qy) + T.ty;
    qx = nx;
    qy = ny;
  }
  double best = __builtin_huge_val();
  uint32_t bi = 0xffffffffu;
  for (unsigned j = sub; j < m; j += 16) {
    const double dx = qx - dst[(size_t)j * DIM + 0], dy = qy - dst[(size_t)j * DIM + 1];
    double dd = dx * dx + dy * dy;
    if (DIM == 3) {
      const double dz = qz - dst[(size_t)j * DIM + 2];
      dd = dd + dz * dz;
    }
    if (dd < best) {  // ascending j: the lowest index of a tie stays
      best = dd;
      bi = j;
    }
  }
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) {
    const double ob = __shfl_xor(best, off);
    const uint32_t obi = (uint32_t)__shfl_xor((int)bi, off);
    if (ob < best || (ob == best && obi < bi)) {
      best = ob;
      bi = obi;
    }
  }
  if (sub != 0) return;
  if (bi == 0xffffffffu) bi = 0;  // no finite distance at all: index 0, as a scan from 0 would
  if (idx) idx[q] = bi;
  if (a) a[q] = make_double2(qx, qy);
  if (b) b[q] = make_double2(dst[(size_t)bi * DIM + 0], dst[(size_t)bi * DIM + 1]);
}

// pairs from given indices: a = xy(T.src) (Transform::transform, src/transform.rs:22-24),
// b = xy(dst[idx]) (get_xy, src/lib.rs:86-89)
template <int DIM>
__global__ __launch_bounds__(256) void k_materialize(const double *__restrict__ src, unsigned n, Pose T,
                                                     const uint32_t *__restrict__ idx,
                                                     const double *__restrict__ dst, double2 *__restrict__ a,
                                                     double2 *__restrict__ b) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double x = src[(size_t)i * DIM + 0], y = src[(size_t)i * DIM + 1];
  const uint32_t j = idx[i];
  a[i] = make_double2((T.r00 * x + T.r01 * y) + T.tx, (T.r10 * x + T.r11 * y) + T.ty);
  b[i] = make_double2(dst[(size_t)j * DIM + 0], dst[(size_t)j * DIM + 1]);
}

hipError_t launch_materialize(icp_handle *h, const double *d_src, size_t n, const Pose &T, const uint32_t *d_idx,
                              double *d_a, double *d_b) {
  if (n == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (h->dim == 3)
    hipLaunchKernelGGL(k_materialize<3>, dim3(blocks), dim3(256), 0, h->stream, d_src, (unsigned)n, T, d_idx,
                       h->d_dst, (double2 *)d_a, (double2 *)d_b);
  else
    hipLaunchKernelGGL(k_materialize<2>, dim3(blocks), dim3(256), 0, h->stream, d_src, (unsigned)n, T, d_idx,
                       h->d_dst, (double2 *)d_a, (double2 *)d_b);
  return hipGetLastError();
}

// AoS (as handed over by the host) -> padded SoA x|y|z used by the scan
__global__ void k_build_soa(const double *__restrict__ dst, unsigned m, unsigned m_pad, int dim,
                            double *__restrict__ soa) {
  const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m_pad) return;
  const double inf = __builtin_huge_val();
  for (int d = 0; d < 3; ++d) {
    double v = inf;
    if (j < m) v = (d < dim) ? dst[(size_t)j * dim + d] : 0.;
    soa[(size_t)d * m_pad + j] = v;
  }
}

// fl32(p - org) per axis, padded with +inf like the f64 tiles
__global__ void k_build_soa_f32(const double *__restrict__ dst, unsigned m, unsigned m_pad, int dim, double ox,
                                double oy, double oz, float *__restrict__ soa) {
  const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m_pad) return;
  const double org[3] = {ox, oy, oz};
  for (int d = 0; d < 3; ++d) {
    float v = __builtin_huge_valf();
    if (j < m) v = (d < dim) ? (float)(dst[(size_t)j * dim + d] - org[d]) : 0.f;
    soa[(size_t)d * m_pad + j] = v;
  }
}

// needs the target bounding box (build_grid); without one the plain f64 sweep serves
hipError_t build_target_screen(icp_handle *h) {
  h->screen_valid = false;  // a pooled handle may still hold the previous cloud's screen
  if (!h->grid.built || h->m_pad == 0) return hipSuccess;
  hipError_t e = reserve(h->d_dst_f32, h->cap_f32, 3 * h->m_pad);
  if (e != hipSuccess) return e;
  const GridParams &g = h->grid.p;
  // relative to the CENTRE of the bounding box: halves the magnitudes the dot-product screen cancels on
  hipLaunchKernelGGL(k_build_soa_f32, dim3((unsigned)((h->m_pad + 255) / 256)), dim3(256), 0, h->stream, h->d_dst,
                     (unsigned)h->m, (unsigned)h->m_pad, h->dim, 0.5 * (g.lo[0] + g.hi[0]), 0.5 * (g.lo[1] + g.hi[1]),
                     0.5 * (g.lo[2] + g.hi[2]), h->d_dst_f32);
  h->screen_valid = true;
  return hipGetLastError();
}

hipError_t build_target_soa(icp_handle *h) {
  const size_t m_pad = ((h->m + kNnTile - 1) / kNnTile) * kNnTile;
  h->m_pad = m_pad;
  h->brute_valid = h->screen_valid = false;
  if (m_pad == 0) {
    h->brute_valid = true;  // nothing to describe
    return hipSuccess;
  }
  hipError_t e = reserve(h->d_dst_soa, h->cap_soa, 3 * m_pad);
  if (e != hipSuccess) return e;
  h->brute_valid = true;
  const unsigned blocks = (unsigned)((m_pad + 255) / 256);
  hipLaunchKernelGGL(k_build_soa, dim3(blocks), dim3(256), 0, h->stream, h->d_dst, (unsigned)h->m,
                     (unsigned)m_pad, h->dim, h->d_dst_soa);
  return hipGetLastError();
}

static int env_int(const char *name, int dflt) {
  const char *s = exp_env(name);
  return s ? atoi(s) : dflt;
}

template <int DIM, int R, bool XFORM>
static void launch_one(icp_handle *h, const double *d_src, unsigned n, const Pose &T, unsigned qblocks,
                       unsigned chunks, unsigned chunk) {
  const bool full = chunk % kNnTile == 0;
  const double *tx = h->d_dst_soa, *ty = tx + h->m_pad, *tz = ty + h->m_pad;
  static const bool no_screen = exp_env("ICP_NN_NO_SCREEN") != nullptr;
  if (h->screen_valid && !no_screen) {
    const float *fx = h->d_dst_f32, *fy = fx + h->m_pad, *fz = fy + h->m_pad;
    const GridParams &g = h->grid.p;
    const double cx = 0.5 * (g.lo[0] + g.hi[0]), cy = 0.5 * (g.lo[1] + g.hi[1]), cz = 0.5 * (g.lo[2] + g.hi[2]);
    // the dot-product screen (two queries per packed instruction) unless its cancellation margin would
    // swamp the distances it has to tell apart; ICP_NN_OLD_SCREEN: the difference-based screen, for A/B
    static const bool old_screen = exp_env("ICP_NN_OLD_SCREEN") != nullptr;
    const double ex = g.hi[0] - g.lo[0], ey = g.hi[1] - g.lo[1], ez = g.hi[2] - g.lo[2];
    const double half_diag = 0.5 * sqrt((ex * ex + ey * ey) + ez * ez) * 1.000001;
    if constexpr (R >= 2) {
      if (!old_screen && half_diag < 1e15 && half_diag > 0.) {
        if (full)
          hipLaunchKernelGGL((k_nn_brute_dot<DIM, R, XFORM, true>), dim3(qblocks, chunks), dim3(kNnThreads), 0, h->stream,
                             d_src, n, tx, ty, tz, fx, fy, fz, (unsigned)h->m_pad, chunk, T, cx, cy, cz, g.scale,
                             (float)half_diag, h->ws.d_part_d, h->ws.d_part_i);
        else
          hipLaunchKernelGGL((k_nn_brute_dot<DIM, R, XFORM, false>), dim3(qblocks, chunks), dim3(kNnThreads), 0, h->stream,
                             d_src, n, tx, ty, tz, fx, fy, fz, (unsigned)h->m_pad, chunk, T, cx, cy, cz, g.scale,
                             (float)half_diag, h->ws.d_part_d, h->ws.d_part_i);
        return;
      }
    }
    if (full)
      hipLaunchKernelGGL((k_nn_brute_scr<DIM, R, XFORM, true>), dim3(qblocks, chunks), dim3(kNnThreads), 0, h->stream,
                         d_src, n, tx, ty, tz, fx, fy, fz, (unsigned)h->m_pad, chunk, T, cx, cy, cz,
                         g.scale, h->ws.d_part_d, h->ws.d_part_i);
    else
      hipLaunchKernelGGL((k_nn_brute_scr<DIM, R, XFORM, false>), dim3(qblocks, chunks), dim3(kNnThreads), 0, h->stream,
                         d_src, n, tx, ty, tz, fx, fy, fz, (unsigned)h->m_pad, chunk, T, cx, cy, cz,
                         g.scale, h->ws.d_part_d, h->ws.d_part_i);
    return;
  }
  if (full)
    hipLaunchKernelGGL((k_nn_brute<DIM, R, XFORM, true>), dim3(qblocks, chunks), dim3(kNnThreads), 0, h->stream,
                       d_src, n, tx, ty, tz, (unsigned)h->m_pad, chunk, T, h->ws.d_part_d, h->ws.d_part_i);
  else
    hipLaunchKernelGGL((k_nn_brute<DIM, R, XFORM, false>), dim3(qblocks, chunks), dim3(kNnThreads), 0, h->stream,
                       d_src, n, tx, ty, tz, (unsigned)h->m_pad, chunk, T, h->ws.d_part_d, h->ws.d_part_i);
}

template <int DIM, bool XFORM>
static void launch_r(icp_handle *h, int R, const double *d_src, unsigned n, const Pose &T, unsigned qblocks,
                     unsigned chunks, unsigned chunk) {
  switch (R) {
    case 1: launch_one<DIM, 1, XFORM>(h, d_src, n, T, qblocks, chunks, chunk); break;
    case 2: launch_one<DIM, 2, XFORM>(h, d_src, n, T, qblocks, chunks, chunk); break;
    case 4: launch_one<DIM, 4, XFORM>(h, d_src, n, T, qblocks, chunks, chunk); break;
    default: launch_one<DIM, 8, XFORM>(h, d_src, n, T, qblocks, chunks, chunk); break;
  }
}

hipError_t launch_nn_brute(icp_handle *h, const double *d_src, size_t n_, const Pose *Tp, double *d_a,
                           double *d_b, uint32_t *d_idx) {
  if (n_ == 0) return hipSuccess;
  const unsigned n = (unsigned)n_;
  const bool xform = Tp != nullptr;
  const Pose T = xform ? *Tp : transform_identity();
  if (!h->brute_valid) {
    // a grown map (icp_append_targets) rebuilds only the grid; the sweep's structures follow on
    // first use.  Synchronised: later launches may come from another stream.
    hipError_t e = build_target_soa(h);
    if (e == hipSuccess) e = build_target_screen(h);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return e;
  }

  // queries per lane: enough waves to cover 1024 SIMDs several times over, then as
  // many registers-resident queries as that allows (fewer LDS reads per pair)
  static const int forced_r = env_int("ICP_NN_R", 0);
  int R = 1;
  if (n >= 900000u) R = 8;  // (with the packed screen 8 queries per lane win from ~1M: 143 vs 147 ms)
  else if (n >= 2u * 256u * 1024u) R = 4;
  else if (n >= 256u * 1024u) R = 2;
  if (forced_r == 1 || forced_r == 2 || forced_r == 4 || forced_r == 8) R = forced_r;
  const unsigned qblocks = (n + kNnThreads * R - 1) / (kNnThreads * R);
  // split the targets when the query blocks alone cannot fill the chip: chunks of whole 64-target
  // granules (a tile is 16 of them), so that even a 650-point scan spreads over ~11 workgroups per
  // query block instead of one lane looping over a whole padded tile
  static const int forced_chunks = env_int("ICP_NN_CHUNKS", 0);
  const unsigned granules = (unsigned)((h->m + 63) / 64);  // beyond them the SoA holds only +inf padding
  unsigned chunks = 1;
  const unsigned want_blocks = 2048;
  if (qblocks < want_blocks) chunks = (want_blocks + qblocks - 1) / qblocks;
  if (forced_chunks > 0) chunks = (unsigned)forced_chunks;
  if (chunks > granules) chunks = granules;
  if (chunks > 64) chunks = 64;
  if (chunks < 1) chunks = 1;
  unsigned granules_per_chunk = (granules + chunks - 1) / chunks;
  if (granules_per_chunk >= 16) granules_per_chunk = (granules_per_chunk + 15) / 16 * 16;  // whole tiles: unrolled kernel
  chunks = (granules + granules_per_chunk - 1) / granules_per_chunk;
  const unsigned chunk = granules_per_chunk * 64;

  static const bool no_tiny = exp_env("ICP_NN_NO_TINY") != nullptr;
  const bool tiny = !no_tiny && n <= 2048u && h->m <= 2048;
  // partial buffers
  const size_t need = tiny ? 0 : (size_t)chunks * n;
  if (need > h->ws.cap_part) {
    if (h->ws.d_part_d) (void)hipFree(h->ws.d_part_d);
    if (h->ws.d_part_i) (void)hipFree(h->ws.d_part_i);
    h->ws.d_part_d = nullptr;
    h->ws.d_part_i = nullptr;
    h->ws.cap_part = 0;
    hipError_t e = hipMalloc(&h->ws.d_part_d, need * sizeof(double));
    if (e != hipSuccess) return e;
    e = hipMalloc(&h->ws.d_part_i, need * sizeof(uint32_t));
    if (e != hipSuccess) return e;
    h->ws.cap_part = need;
  }

  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (h->profile > 0 && (h->prof_seen++ % (unsigned)h->profile) == 0) {
    if (!h->prof_free.empty()) {
      ev0 = h->prof_free.back().first;
      ev1 = h->prof_free.back().second;
      h->prof_free.pop_back();
      (void)hipEventRecord(ev0, h->stream);
    } else if (hipEventCreate(&ev0) == hipSuccess && hipEventCreate(&ev1) == hipSuccess)
      (void)hipEventRecord(ev0, h->stream);
  }
  if (tiny) {
    const unsigned tb = (n * 16u + 255u) / 256u;
#define TINY(DIM, XF)                                                                                          \
  hipLaunchKernelGGL((k_nn_tiny<DIM, XF>), dim3(tb), dim3(256), 0, h->stream, d_src, n, T, h->d_dst, (unsigned)h->m, \
                     d_idx, (double2 *)d_a, (double2 *)d_b)
    if (h->dim == 3) {
      if (xform) TINY(3, true); else TINY(3, false);
    } else {
      if (xform) TINY(2, true); else TINY(2, false);
    }
#undef TINY
  } else if (h->dim == 3) {
    if (xform) launch_r<3, true>(h, R, d_src, n, T, qblocks, chunks, chunk);
    else       launch_r<3, false>(h, R, d_src, n, T, qblocks, chunks, chunk);
  } else {
    if (xform) launch_r<2, true>(h, R, d_src, n, T, qblocks, chunks, chunk);
    else       launch_r<2, false>(h, R, d_src, n, T, qblocks, chunks, chunk);
  }
  hipError_t e = hipGetLastError();
  if (ev0 && ev1) {
    (void)hipEventRecord(ev1, h->stream);
    h->prof_events.emplace_back(ev0, ev1);
  }
  if (e != hipSuccess || tiny) return e;

  const double *tx = h->d_dst_soa, *ty = tx + h->m_pad;
  const unsigned fblocks = (n + 255) / 256;
#define FIN(DIM, XF)                                                                             \
  hipLaunchKernelGGL((k_nn_finalize<DIM, XF>), dim3(fblocks), dim3(256), 0, h->stream, d_src, n, T, \
                     h->ws.d_part_d, h->ws.d_part_i, chunks, tx, ty, d_idx, (double2 *)d_a,       \
                     (double2 *)d_b)
  if (h->dim == 3) {
    if (xform) FIN(3, true); else FIN(3, false);
  } else {
    if (xform) FIN(2, true); else FIN(2, false);
  }
#undef FIN
  return hipGetLastError();
}

}  // namespace icp
