// EXTENSION beyond the reference (BASELINE.json configs[4]): point-to-plane
// residuals.  tier4/icp_rust is point-to-point only -- there is no normal, no plane, no neighbourhood
// anywhere in src/ (src/lib.rs:218-261) -- so NOTHING here has a reference counterpart and no parity
// claim is made.  What is kept from the reference is everything around the residual: the SE(2) pose on
// the xy-plane with z carried through (src/lib.rs:52-57), the exact nearest neighbour in 3-D
// (:161-167), the Huber / MAD weighting with the same constants and the same inner loop with its two
// break tests (:59-84, :218-261, src/huber.rs, src/stats.rs), the 3x3 adjugate solve and Transform::new.
// Definition (an independent CPU statement of it is the checker of tests/test_p2plane.py):
//   normal of a target q   = unit eigenvector of the smallest eigenvalue of the covariance of the k
//                            targets nearest to q (q itself included; ties by lowest index), cyclic
//                            Jacobi in f64, sign: first non-zero of (n_z, n_y, n_x) positive;
//   residual of a pair     = n_q . (T p - q), T p = (R p_xy + t, p_z): ONE scalar per pair;
//   weights                = sigma = 1.4826 MAD(r), g = 1 / sigma, w = drho(r^2, 1.345), skipped when
//                            sigma == 0, exactly as the reference treats each of its two rows;
//   Jacobian row           = n_xy^T [R | R (-a_y, a_x)^T] with the reference's jacobian() (:176-184);
//   Huber error            = sum rho(r^2).
#include "common.hpp"
#include "gn_device.hpp"

namespace icp {
hipError_t launch_sel_init(icp_handle *h, size_t n);
hipError_t launch_stddevs(icp_handle *h, const double *d_a, const double *d_b, size_t n, const Pose &T);
__global__ void k_final_reduce(const double *__restrict__ partials, int blocks, int nacc,
                               const GnScalars *__restrict__ scal, GnResult *__restrict__ res);

constexpr int kNormalKMax = 16;

// cyclic Jacobi on a symmetric 3x3 (a: upper triangle used, row-major full), eigenvectors in the
// columns of v.  Fixed operation order (the CPU restatement runs the same sequence).
__device__ inline void jacobi3(double a[3][3], double v[3][3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) v[i][j] = i == j ? 1. : 0.;
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = (a[0][1] * a[0][1] + a[0][2] * a[0][2]) + a[1][2] * a[1][2];
    if (off == 0.) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (a[p][q] == 0.) continue;
        const double theta = (a[q][q] - a[p][p]) / (2. * a[p][q]);
        const double t = (theta >= 0. ? 1. : -1.) / (fabs(theta) + sqrt(theta * theta + 1.));
        const double c = 1. / sqrt(t * t + 1.), s = t * c;
        for (int k = 0; k < 3; ++k) {  // A <- A J (columns p, q)
          const double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {  // A <- J^T A (rows p, q)
          const double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = v[k][p], vkq = v[k][q];
          v[k][p] = c * vkp - s * vkq;
          v[k][q] = s * vkp + c * vkq;
        }
      }
  }
}

// One lane per target: its k nearest targets through the grid (Chebyshev rings of cells around its
// own cell until the k-th distance is covered by the visited block), then the covariance's smallest
// eigenvector.  The running k-best lists live in LDS (dynamic indexing).
__global__ __launch_bounds__(64) void k_target_normals(const double *__restrict__ dst, unsigned m, GridParams g,
                                                       const uint32_t *__restrict__ start,
                                                       const GridPoint *__restrict__ pts, int kk,
                                                       double *__restrict__ normals, unsigned first) {
  __shared__ double s_d[64][kNormalKMax];
  __shared__ uint32_t s_i[64][kNormalKMax];
  const unsigned i = first + blockIdx.x * 64 + threadIdx.x;  // (targets [first, m): all of them, or the appended ones)
  if (i >= m) return;
  double *bd = s_d[threadIdx.x];
  uint32_t *bi = s_i[threadIdx.x];
  const double p[3] = {dst[(size_t)i * 3], dst[(size_t)i * 3 + 1], dst[(size_t)i * 3 + 2]};
  int c[3];
  for (int d = 0; d < 3; ++d) {
    double t = floor((p[d] - g.lo[d]) * g.inv_h[d]);
    t = fmin(fmax(t, 0.), (double)(g.n[d] - 1));
    c[d] = (int)t;
  }
  const int k = kk < (int)m ? kk : (int)m;
  int cnt = 0;
  const int rmax = max(max(g.n[0], g.n[1]), g.n[2]);
  for (int r = 1; r <= rmax; ++r) {
    cnt = 0;
    const int x0 = max(c[0] - r * g.fx, 0), x1 = min(c[0] + r * g.fx, g.n[0] - 1);
    const int y0 = max(c[1] - r, 0), y1 = min(c[1] + r, g.n[1] - 1);
    const int z0 = max(c[2] - r, 0), z1 = min(c[2] + r, g.n[2] - 1);
    for (int iz = z0; iz <= z1; ++iz)
      for (int iy = y0; iy <= y1; ++iy) {
        const uint32_t row = ((uint32_t)iz * g.n[1] + iy) * g.n[0];
        const uint32_t s = start[row + x0], e = start[row + x1 + 1];
        for (uint32_t j = s; j < e; ++j) {
          const uint32_t ti = pts[j].idx;
          const double dx = p[0] - dst[(size_t)ti * 3], dy = p[1] - dst[(size_t)ti * 3 + 1],
                       dz = p[2] - dst[(size_t)ti * 3 + 2];
          const double dd = (dx * dx + dy * dy) + dz * dz;
          // insertion by (d^2, index) into the k best
          if (cnt == k && !(dd < bd[k - 1] || (dd == bd[k - 1] && ti < bi[k - 1]))) continue;
          int pos = cnt < k ? cnt : k - 1;
          while (pos > 0 && (dd < bd[pos - 1] || (dd == bd[pos - 1] && ti < bi[pos - 1]))) {
            bd[pos] = bd[pos - 1];
            bi[pos] = bi[pos - 1];
            --pos;
          }
          bd[pos] = dd;
          bi[pos] = ti;
          if (cnt < k) ++cnt;
        }
      }
    // is everything outside the visited block strictly farther than the k-th best?
    double cover = __builtin_huge_val();
    for (int d = 0; d < 3; ++d) {
      const int w = d == 0 ? r * g.fx : r;
      if (c[d] - w > 0) cover = fmin(cover, p[d] - (g.lo[d] + (c[d] - w) * g.h[d]));
      if (c[d] + w < g.n[d] - 1) cover = fmin(cover, (g.lo[d] + (c[d] + w + 1) * g.h[d]) - p[d]);
    }
    if (cover == __builtin_huge_val()) break;  // the whole grid
    cover -= 1e-9 * (g.scale + fabs(p[0]) + fabs(p[1]) + fabs(p[2]));
    if (cnt == k && cover > 0. && bd[k - 1] < cover * cover) break;
  }
  double nrm[3] = {0., 0., 0.};
  if (cnt >= 3) {
    double mean[3] = {0., 0., 0.};
    for (int j = 0; j < cnt; ++j)
      for (int d = 0; d < 3; ++d) mean[d] = mean[d] + dst[(size_t)bi[j] * 3 + d];
    for (int d = 0; d < 3; ++d) mean[d] = mean[d] / (double)cnt;
    double a[3][3] = {{0., 0., 0.}, {0., 0., 0.}, {0., 0., 0.}};
    for (int j = 0; j < cnt; ++j) {
      double e[3];
      for (int d = 0; d < 3; ++d) e[d] = dst[(size_t)bi[j] * 3 + d] - mean[d];
      for (int r = 0; r < 3; ++r)
        for (int s = 0; s < 3; ++s) a[r][s] = a[r][s] + e[r] * e[s];
    }
    double v[3][3];
    jacobi3(a, v);
    int col = 0;  // smallest eigenvalue; ties -> lowest column
    if (a[1][1] < a[col][col]) col = 1;
    if (a[2][2] < a[col][col]) col = 2;
    double n0 = v[0][col], n1 = v[1][col], n2 = v[2][col];
    const double len = sqrt((n0 * n0 + n1 * n1) + n2 * n2);
    if (len > 0.) {
      n0 = n0 / len;
      n1 = n1 / len;
      n2 = n2 / len;
      const double lead = n2 != 0. ? n2 : (n1 != 0. ? n1 : n0);
      if (lead < 0.) {
        n0 = -n0;
        n1 = -n1;
        n2 = -n2;
      }
      nrm[0] = n0;
      nrm[1] = n1;
      nrm[2] = n2;
    }
  }
  normals[(size_t)i * 3] = nrm[0];
  normals[(size_t)i * 3 + 1] = nrm[1];
  normals[(size_t)i * 3 + 2] = nrm[2];
}

// per pair: everything the inner loop needs that does not change with the inner pose
struct PlanePair {
  double ax, ay;        // xy(T_outer p)
  double qx, qy, dz;    // matched target xy, p_z - q_z
  double nx, ny, nz;    // its normal
};

__global__ void k_p2pl_gather(const double *__restrict__ src, unsigned n, Pose T, const uint32_t *__restrict__ idx,
                              const double *__restrict__ dst, const double *__restrict__ normals,
                              PlanePair *__restrict__ out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = src[(size_t)i * 3], y = src[(size_t)i * 3 + 1], z = src[(size_t)i * 3 + 2];
  const uint32_t j = idx[i];
  PlanePair o;
  o.ax = (T.r00 * x + T.r01 * y) + T.tx;  // Transform::transform, src/transform.rs:22-24
  o.ay = (T.r10 * x + T.r11 * y) + T.ty;
  o.qx = dst[(size_t)j * 3];
  o.qy = dst[(size_t)j * 3 + 1];
  o.dz = z - dst[(size_t)j * 3 + 2];
  o.nx = normals[(size_t)j * 3];
  o.ny = normals[(size_t)j * 3 + 1];
  o.nz = normals[(size_t)j * 3 + 2];
  out[i] = o;
}

__device__ __forceinline__ double plane_residual(const PlanePair &p, const Pose &T) {
  const double rx = ((T.r00 * p.ax + T.r01 * p.ay) + T.tx) - p.qx;
  const double ry = ((T.r10 * p.ax + T.r11 * p.ay) + T.ty) - p.qy;
  return (p.nx * rx + p.ny * ry) + p.nz * p.dz;
}

// residuals as the pairs ((r, 0), (0, 0)): launch_stddevs under the identity pose then selects the
// exact median and MAD of r (dimension 0); dimension 1 is all zeros
__global__ void k_p2pl_residual(const PlanePair *__restrict__ pp, unsigned n, Pose T, double2 *__restrict__ fa,
                                double2 *__restrict__ fb) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fa[i] = make_double2(plane_residual(pp[i], T), 0.);
  fb[i] = make_double2(0., 0.);
}

__global__ __launch_bounds__(kReduceThreads) void k_p2pl_accumulate(const PlanePair *__restrict__ pp, unsigned n, Pose T,
                                                                    const GnScalars *__restrict__ scal,
                                                                    double *__restrict__ partials) {
  double acc[kNAcc];
#pragma unroll
  for (int k = 0; k < kNAcc; ++k) acc[k] = 0.;
  const double sigma = scal->sigma[0];
  const double gw = 1. / sigma;
  const unsigned G = gridDim.x * kReduceThreads;
  for (unsigned i = blockIdx.x * kReduceThreads + threadIdx.x; i < n; i += G) {
    const PlanePair p = pp[i];
    const double r = plane_residual(p, T);
    const double e = r * r;
    if (sigma != 0.) {  // src/lib.rs:243-245
      const double a0 = -p.ay, a1 = p.ax;  // jacobian(), src/lib.rs:176-184
      const double b0 = T.r00 * a0 + T.r01 * a1;
      const double b1 = T.r10 * a0 + T.r11 * a1;
      const double J[3] = {p.nx * T.r00 + p.ny * T.r10, p.nx * T.r01 + p.ny * T.r11, p.nx * b0 + p.ny * b1};
      const double wg = huber_drho(e) * gw;
#pragma unroll
      for (int k = 0; k < 3; ++k) acc[9 + k] = acc[9 + k] + (wg * J[k]) * r;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[3 * a + b] = acc[3 * a + b] + (wg * J[a]) * J[b];
    }
    acc[12] = acc[12] + huber_rho(e);
  }
  block_reduce_store<kNAcc>(acc, partials + (size_t)blockIdx.x * (kNAcc + 1));
}

hipError_t launch_target_normals(icp_handle *h, int k, double *d_normals, size_t first) {
  const unsigned m = (unsigned)h->m;
  if (first >= h->m) return hipSuccess;
  hipLaunchKernelGGL(k_target_normals, dim3((m - (unsigned)first + 63) / 64), dim3(64), 0, h->stream, h->d_dst, m, h->grid.p,
                     (const uint32_t *)h->grid.d_start, (const GridPoint *)h->grid.d_pts, k, d_normals, (unsigned)first);
  return hipGetLastError();
}

hipError_t launch_p2pl_gather(icp_handle *h, const double *d_src, size_t n, const Pose &T, const uint32_t *d_idx,
                              const double *d_normals, void *d_pairs) {
  hipLaunchKernelGGL(k_p2pl_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, d_src, (unsigned)n, T,
                     d_idx, h->d_dst, d_normals, (PlanePair *)d_pairs);
  return hipGetLastError();
}

// one evaluation: result in h->ws.h_res after the stream is synchronised (acc[0..8] jtj, [9..11] jtr, [12] error)
hipError_t launch_p2pl_eval(icp_handle *h, const void *d_pairs, size_t n_, const Pose &T, double *d_fa, double *d_fb) {
  Workspace &w = h->ws;
  const unsigned n = (unsigned)n_;
  hipError_t e;
  hipLaunchKernelGGL(k_p2pl_residual, dim3((n + 255) / 256), dim3(256), 0, h->stream, (const PlanePair *)d_pairs, n, T,
                     (double2 *)d_fa, (double2 *)d_fb);
  if ((e = launch_sel_init(h, n_)) != hipSuccess) return e;
  if ((e = launch_stddevs(h, d_fa, d_fb, n_, transform_identity())) != hipSuccess) return e;
  int blocks, threads;
  reduce_geometry(n_, &blocks, &threads);
  hipLaunchKernelGGL(k_p2pl_accumulate, dim3(blocks), dim3(threads), 0, h->stream, (const PlanePair *)d_pairs, n, T,
                     (const GnScalars *)w.d_scal, w.d_partials);
  hipLaunchKernelGGL(k_final_reduce, dim3(1), dim3(kReduceThreads), 0, h->stream, (const double *)w.d_partials, blocks,
                     kNAcc, (const GnScalars *)w.d_scal, w.h_res);
  w.gn_dirty = true;  // the radix path leaves its selection state behind
  return hipGetLastError();
}

size_t p2pl_pair_bytes() { return sizeof(PlanePair); }

}  // namespace icp
