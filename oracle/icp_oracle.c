/*
 * icp_oracle.c -- CPU restatement of tier4/icp_rust's ICP path.  TEST INFRASTRUCTURE ONLY
 * (see icp_oracle.h for the rules and the pinning status).
 *
 * Every function follows the reference's operation order; the cited lines are under
 * /root/reference.  Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).
 * nalgebra 0.32.3 evaluates the small fixed-size products used here column by column:
 *   (M * v)_i = (M_i0 * v_0 + M_i1 * v_1) + M_i2 * v_2, and scalar * matrix elementwise,
 * which is what the helpers below spell out.
 */
#define _GNU_SOURCE
#include "icp_oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

/* f64::sin / f64::cos of the reference's no_std build: num-traits' `libm` feature (Cargo.toml:17-20)
 * -> the Rust libm crate, a port of musl's kernels.  The crate is not under /root/reference; its
 * published algorithm is restated ONCE, in include/icp_trig.h, and shared with the library's host and
 * device code so that all three agree to the bit (tests/test_trig.py pins it against the C library:
 * <= 1 ulp).  Outside the restated range (|x| >= 2^20 pi/2) the C library serves. */
#include "../include/icp_trig.h"
static double ref_sin(double x) {
  int ok;
  double v = icp_sin(x, &ok);
  return ok ? v : sin(x);
}
static double ref_cos(double x) {
  int ok;
  double v = icp_cos(x, &ok);
  return ok ? v : cos(x);
}

/* ------------------------------------------------------------------ so2 / se2 ---- */

double orc_sin(double x) { return ref_sin(x); }
double orc_cos(double x) { return ref_cos(x); }

/* so2.rs:23-31 `exp`; so2.rs:8-17 `new_rotation2` builds the same matrix. */
void orc_so2_exp(double theta, double m[4]) {
  double c = ref_cos(theta), s = ref_sin(theta);
  m[0] = c;  /* (0,0) */
  m[1] = s;  /* (1,0) */
  m[2] = -s; /* (0,1) */
  m[3] = c;  /* (1,1) */
}

/* so2.rs:19-21 */
double orc_so2_log(const double m[4]) { return atan2(m[1], m[0]); }

/* se2.rs:21-41 */
void orc_se2_calc_rt(const double param[3], orc_pose *out) {
  double theta = param[2];
  double rot[4];
  orc_so2_exp(theta, rot); /* so2::new_rotation2(theta) */
  double c = ref_cos(theta), s = ref_sin(theta);
  double vx = param[0], vy = param[1];
  out->r00 = rot[0];
  out->r10 = rot[1];
  out->r01 = rot[2];
  out->r11 = rot[3];
  if (theta == 0.) {
    out->tx = vx;
    out->ty = vy;
  } else {
    out->tx = (s * vx - (1. - c) * vy) / theta;
    out->ty = ((1. - c) * vx + s * vy) / theta;
  }
}

/* se2.rs:43-52 (row-major 3x3 out, as the Matrix3::new literal reads) */
void orc_se2_exp(const double param[3], double m[9]) {
  orc_pose p;
  orc_se2_calc_rt(param, &p);
  m[0] = p.r00; m[1] = p.r01; m[2] = p.tx;
  m[3] = p.r10; m[4] = p.r11; m[5] = p.ty;
  m[6] = 0.;    m[7] = 0.;    m[8] = 1.;
}

/* se2.rs:11-19 */
void orc_se2_get_rt(const double m[9], double rot[4], double t[2]) {
  rot[0] = m[0]; rot[1] = m[1];
  rot[2] = m[3]; rot[3] = m[4];
  t[0] = m[2];
  t[1] = m[5];
}

/* se2.rs:54-77 */
void orc_se2_log(const double m[9], double param[3]) {
  double rot[4], t[2];
  orc_se2_get_rt(m, rot, t);
  double rot_cm[4] = {rot[0], rot[2], rot[1], rot[3]};
  double theta = orc_so2_log(rot_cm);
  double v00, v01, v10, v11; /* v_inv, row-major */
  if (theta == 0.) {
    v00 = 1.; v01 = 0.; v10 = 0.; v11 = 1.;
  } else if (theta == M_PI) {
    v00 = 0.; v01 = 0.5 * theta; v10 = -0.5 * theta; v11 = 0.;
  } else {
    double k = ref_sin(theta) / (1. - ref_cos(theta));
    double h = 0.5 * theta; /* `0.5 * theta * m`: (0.5*theta) then scalar * matrix */
    v00 = h * k; v01 = h * 1.; v10 = h * -1.; v11 = h * k;
  }
  param[0] = v00 * t[0] + v01 * t[1];
  param[1] = v10 * t[0] + v11 * t[1];
  param[2] = theta;
}

/* ------------------------------------------------------------------ Transform ---- */

void orc_transform_new(const double param[3], orc_pose *out) { orc_se2_calc_rt(param, out); }

void orc_transform_identity(orc_pose *out) {
  out->r00 = 1.; out->r10 = 0.; out->r01 = 0.; out->r11 = 1.;
  out->tx = 0.;  out->ty = 0.;
}

/* transform.rs:22-24: self.rot * landmark + self.t */
void orc_transform_apply(const orc_pose *T, const double p[2], double out[2]) {
  double x = p[0], y = p[1];
  out[0] = (T->r00 * x + T->r01 * y) + T->tx;
  out[1] = (T->r10 * x + T->r11 * y) + T->ty;
}

/* transform.rs:26-32: Rotation2::inverse is the transpose */
void orc_transform_inverse(const orc_pose *T, orc_pose *out) {
  orc_pose r;
  r.r00 = T->r00; r.r01 = T->r10;
  r.r10 = T->r01; r.r11 = T->r11;
  double ix = r.r00 * T->tx + r.r01 * T->ty;
  double iy = r.r10 * T->tx + r.r11 * T->ty;
  r.tx = -ix;
  r.ty = -iy;
  *out = r;
}

/* transform.rs:42-51: rot = l.rot * r.rot; t = l.rot * r.t + l.t */
void orc_transform_mul(const orc_pose *l, const orc_pose *r, orc_pose *out) {
  orc_pose o;
  o.r00 = l->r00 * r->r00 + l->r01 * r->r10;
  o.r10 = l->r10 * r->r00 + l->r11 * r->r10;
  o.r01 = l->r00 * r->r01 + l->r01 * r->r11;
  o.r11 = l->r10 * r->r01 + l->r11 * r->r11;
  o.tx = (l->r00 * r->tx + l->r01 * r->ty) + l->tx;
  o.ty = (l->r10 * r->tx + l->r11 * r->ty) + l->ty;
  *out = o;
}

/* lib.rs:52-57 */
void orc_transform_xy(const orc_pose *T, const double p[3], double out[3]) {
  double xy[2] = {p[0], p[1]}, d[2];
  orc_transform_apply(T, xy, d);
  out[0] = d[0];
  out[1] = d[1];
  out[2] = p[2];
}

/* ---------------------------------------------------------------------- norm ---- */

/* norm.rs:8-17: per column, res += col.dot(col) */
double orc_norm_squared(const double *m, size_t nrows, size_t ncols) {
  double res = 0.;
  for (size_t c = 0; c < ncols; ++c) {
    const double *col = m + c * nrows;
    double d = 0.;
    for (size_t r = 0; r < nrows; ++r) d = (r == 0) ? col[0] * col[0] : d + col[r] * col[r];
    res += d;
  }
  return res;
}

/* norm.rs:19-21 */
double orc_norm(const double *m, size_t nrows, size_t ncols) {
  return sqrt(orc_norm_squared(m, nrows, ncols));
}

/* --------------------------------------------------------------------- huber ---- */

/* huber.rs:6-15 */
double orc_huber_rho(double e, double k) {
  double k_squared = k * k;
  if (e <= k_squared) return e;
  return 2. * k * sqrt(e) - k_squared;
}

/* huber.rs:17-26 */
double orc_huber_drho(double e, double k) {
  double k_squared = k * k;
  if (e <= k_squared) return 1.;
  return k / sqrt(e);
}

/* -------------------------------------------------------------------- linalg ---- */

/* linalg.rs:3-29; row-major in/out (matrix[(r,c)] = m[3*r+c]) */
int orc_inverse3x3(const double m[9], double out[9]) {
  double m00 = m[0], m01 = m[1], m02 = m[2];
  double m10 = m[3], m11 = m[4], m12 = m[5];
  double m20 = m[6], m21 = m[7], m22 = m[8];
  double det = m00 * (m22 * m11 - m21 * m12) - m10 * (m22 * m01 - m21 * m02) +
               m20 * (m12 * m01 - m11 * m02);
  if (det == 0.) return ORC_NONE;
  double a[9] = {
      m22 * m11 - m21 * m12,    -(m22 * m01 - m21 * m02), m12 * m01 - m11 * m02,
      -(m22 * m10 - m20 * m12), m22 * m00 - m20 * m02,    -(m12 * m00 - m10 * m02),
      m21 * m10 - m20 * m11,    -(m21 * m00 - m20 * m01), m11 * m00 - m10 * m01,
  };
  for (int i = 0; i < 9; ++i) out[i] = a[i] / det;
  return ORC_OK;
}

/* --------------------------------------------------------------------- stats ---- */

/* select_nth_unstable_by (stats.rs:19,23,24) places an exact order statistic at its index; which
 * selection algorithm finds it cannot change THAT value.  Quickselect, median-of-3.
 *
 * What the oracle (and therefore the GPU contract) pins for even n is the mathematically exact
 * median: the mean of the lower and the upper middle order statistic.  The reference reads
 * input[n/2 - 1] AFTER a second select_nth_unstable_by(n/2) on the whole vector (stats.rs:23-26);
 * that second call only guarantees index n/2 and is free to permute the lower half, so whether
 * input[n/2 - 1] still holds the lower middle order statistic depends on the std implementation of
 * the toolchain the crate is built with.  For the slice lengths std finishes by insertion sort
 * (all 24 reference tests: n <= 19) it provably does; beyond that this is PARITY UNPINNED until an
 * even-n known answer (n > 50) from the real crate is available -- no Rust toolchain exists here.
 * This restatement selects both ranks exactly (the second select below leaves v[n/2 - 1] the
 * maximum of the lower half only because the first one already partitioned around it; see
 * orc_median). */
static void select_nth(double *v, size_t n, size_t kk) {
  ptrdiff_t lo = 0, hi = (ptrdiff_t)n - 1, k = (ptrdiff_t)kk;
  while (lo < hi) {
    ptrdiff_t mid = lo + (hi - lo) / 2;
    double a = v[lo], b = v[mid], c = v[hi], pivot;
    if (a < b) pivot = (b < c) ? b : (a < c ? c : a);
    else       pivot = (a < c) ? a : (b < c ? c : b);
    ptrdiff_t i = lo, j = hi;
    do {
      while (v[i] < pivot) ++i;
      while (v[j] > pivot) --j;
      if (i <= j) {
        double t = v[i]; v[i] = v[j]; v[j] = t;
        ++i;
        --j;
      }
    } while (i <= j);
    if (k <= j) hi = j;
    else if (k >= i) lo = i;
    else return;
  }
}

/* stats.rs:11-28 */
int orc_median(double *v, size_t n, double *out) {
  if (n == 0) return ORC_NONE;
  for (size_t i = 0; i < n; ++i)
    if (v[i] != v[i]) return ORC_NAN; /* partial_cmp().unwrap() panics, stats.rs:12 */
  if (n % 2 == 1) {
    select_nth(v, n, n / 2);
    *out = v[n / 2];
    return ORC_OK;
  }
  /* upper middle first, then the lower middle INSIDE the lower half: v[n/2 - 1] is then the exact
   * lower middle order statistic whatever the partitioning scheme does (see the note above) */
  select_nth(v, n, n / 2);
  select_nth(v, n / 2, n / 2 - 1);
  double b = v[n / 2 - 1];
  double c = v[n / 2];
  *out = (b + c) / 2.;
  return ORC_OK;
}

/* stats.rs:30-37 */
int orc_mad(double *v, size_t n, double *out) {
  double m;
  int rc = orc_median(v, n, &m);
  if (rc != ORC_OK) return rc;
  double *a = (double *)malloc(n * sizeof(double));
  for (size_t i = 0; i < n; ++i) a[i] = fabs(v[i] - m);
  rc = orc_median(a, n, out);
  free(a);
  return rc;
}

/* stats.rs:39-47 */
int orc_standard_deviation(double *v, size_t n, double *out) {
  double m;
  int rc = orc_mad(v, n, &m);
  if (rc != ORC_OK) return rc;
  *out = ORC_PPF34 * m;
  return ORC_OK;
}

/* stats.rs:49-60 */
int orc_calc_stddevs(const double *r, size_t n, size_t dim, double *out) {
  double *col = (double *)malloc((n ? n : 1) * sizeof(double));
  for (size_t j = 0; j < dim; ++j) {
    for (size_t i = 0; i < n; ++i) col[i] = r[i * dim + j];
    double s;
    int rc = orc_standard_deviation(col, n, &s);
    if (rc != ORC_OK) {
      free(col);
      return rc;
    }
    out[j] = s;
  }
  free(col);
  return ORC_OK;
}

/* ----------------------------------------------------------------- estimator ---- */

/* lib.rs:34-36 */
void orc_residual(const orc_pose *T, const double s[2], const double d[2], double out[2]) {
  double p[2];
  orc_transform_apply(T, s, p);
  out[0] = p[0] - d[0];
  out[1] = p[1] - d[1];
}

/* lib.rs:38-43 */
double orc_error(const orc_pose *T, const double *a, const double *b, size_t n) {
  double sum = 0.;
  for (size_t i = 0; i < n; ++i) {
    double r[2];
    orc_residual(T, a + 2 * i, b + 2 * i, r);
    sum = sum + (r[0] * r[0] + r[1] * r[1]);
  }
  return sum;
}

/* lib.rs:45-50 */
double orc_huber_error(const orc_pose *T, const double *a, const double *b, size_t n) {
  double sum = 0.;
  for (size_t i = 0; i < n; ++i) {
    double r[2];
    orc_residual(T, a + 2 * i, b + 2 * i, r);
    sum = sum + orc_huber_rho(r[0] * r[0] + r[1] * r[1], ORC_HUBER_K);
  }
  return sum;
}

/* lib.rs:176-184: rows of [R | R*(-y, x)^T]; J[j][k] */
static void jacobian(const orc_pose *T, const double s[2], double J[2][3]) {
  double a0 = -s[1], a1 = s[0];
  double b0 = T->r00 * a0 + T->r01 * a1;
  double b1 = T->r10 * a0 + T->r11 * a1;
  J[0][0] = T->r00; J[0][1] = T->r01; J[0][2] = b0;
  J[1][0] = T->r10; J[1][1] = T->r11; J[1][2] = b1;
}

/* lib.rs:186-189: len > 0 && len >= input[0].len() (= 2) */
static int check_input_size(size_t n) { return n > 0 && n >= 2; }

/* lib.rs:212-215 / 257-260: -jtj_inv * jtr */
static int solve_update(const double jtj[9], const double jtr[3], double delta[3]) {
  double inv[9];
  if (orc_inverse3x3(jtj, inv) != ORC_OK) return ORC_NONE;
  for (int i = 0; i < 3; ++i)
    delta[i] = ((-inv[3 * i + 0]) * jtr[0] + (-inv[3 * i + 1]) * jtr[1]) + (-inv[3 * i + 2]) * jtr[2];
  return ORC_OK;
}

/* lib.rs:191-216 */
int orc_gauss_newton_update(const orc_pose *T, const double *a, const double *b, size_t n,
                            double delta[3]) {
  if (!check_input_size(n)) return ORC_NONE;
  double jtr[3] = {0., 0., 0.}, jtj[9] = {0.};
  for (size_t i = 0; i < n; ++i) {
    double J[2][3], r[2];
    jacobian(T, a + 2 * i, J);
    orc_residual(T, a + 2 * i, b + 2 * i, r);
    /* j.transpose() * r and j.transpose() * j: sums over the two rows, row 0 first */
    for (int k = 0; k < 3; ++k) jtr[k] = jtr[k] + (J[0][k] * r[0] + J[1][k] * r[1]);
    for (int p = 0; p < 3; ++p)
      for (int q = 0; q < 3; ++q)
        jtj[3 * p + q] = jtj[3 * p + q] + (J[0][p] * J[0][q] + J[1][p] * J[1][q]);
  }
  return solve_update(jtj, jtr, delta);
}

/* one point's contribution, lib.rs:241-254, added into the running accumulators */
static inline void wgn_accumulate_point(const orc_pose *T, const double s[2], const double r[2],
                                        const double stddevs[2], double jtr[3], double jtj[9]) {
  double J[2][3];
  jacobian(T, s, J);
  for (int j = 0; j < 2; ++j) {
    if (stddevs[j] == 0.) continue;
    double g = 1. / stddevs[j];
    double r_ij = r[j];
    double w_ij = orc_huber_drho(r_ij * r_ij, ORC_HUBER_K);
    double wg = w_ij * g;
    for (int k = 0; k < 3; ++k) jtr[k] = jtr[k] + (wg * J[j][k]) * r_ij;
    for (int p = 0; p < 3; ++p)
      for (int q = 0; q < 3; ++q) jtj[3 * p + q] = jtj[3 * p + q] + (wg * J[j][p]) * J[j][q];
  }
}

/* lib.rs:218-261 */
int orc_weighted_gauss_newton_update(const orc_pose *T, const double *a, const double *b,
                                     size_t n, double delta[3]) {
  if (!check_input_size(n)) return ORC_NONE;
  double *res = (double *)malloc(n * 2 * sizeof(double));
  for (size_t i = 0; i < n; ++i) orc_residual(T, a + 2 * i, b + 2 * i, res + 2 * i);
  double stddevs[2];
  int rc = orc_calc_stddevs(res, n, 2, stddevs);
  if (rc != ORC_OK) {
    free(res);
    return rc;
  }
  double jtr[3] = {0., 0., 0.}, jtj[9] = {0.};
  for (size_t i = 0; i < n; ++i) wgn_accumulate_point(T, a + 2 * i, res + 2 * i, stddevs, jtr, jtj);
  free(res);
  return solve_update(jtj, jtr, delta);
}

/* ---- device-reduction-order variant (test tool; see header) ---------------------
 * The device sums N per-point terms with: global thread g = block*threads + thread folds
 * points g, g+G, g+2G, ... (G = blocks*threads) left to right from 0; a 64-lane wave
 * combines with v[l] += v[l+off] for off = 32,16,8,4,2,1; thread 0 left-folds the wave
 * sums of its block starting from wave 0's; a second stage applies the same scheme with
 * one block of `threads` threads over the `blocks` block sums.
 * What is summed (round 3; icp_rust_amd/csrc/common.hpp: kNSum): PER DIMENSION j, WITHOUT the factor
 * g_j = 1 / sigma_j, and only the upper triangle of J^T W J --
 *   S_j[u(p,q)] = sum_i (w_ij J_ij[p]) J_ij[q]   (p <= q; u = 0..5 for 00 01 02 11 12 22)
 *   S_j[6 + k]  = sum_i (w_ij J_ij[k]) r_ij
 * plus the Huber error: 19 sums; jtj[p][q] = g_x S_x[u] + g_y S_y[u] (mirrored), jtr likewise, are formed
 * from the folded totals (tree_combine).  The reference multiplies every term by w g first and evaluates
 * all nine products (lib.rs:246-254; orc_weighted_gauss_newton_update above restates THAT): the two differ
 * by rounding only. */
#define NACC 19
static void tree_block_reduce(double (*v)[NACC], int threads, double out[NACC]) {
  /* v: per-thread accumulators of one block */
  int waves = (threads + 63) / 64;
  for (int w = 0; w < waves; ++w) {
    double(*lane)[NACC] = v + 64 * w;
    for (int off = 32; off >= 1; off >>= 1)
      for (int l = 0; l < off; ++l)
        for (int k = 0; k < NACC; ++k) lane[l][k] = lane[l][k] + lane[l + off][k];
  }
  for (int k = 0; k < NACC; ++k) {
    double s = v[0][k];
    for (int w = 1; w < waves; ++w) s = s + v[64 * w][k];
    out[k] = s;
  }
}

/* one point's terms, added into the 19 running sums */
static inline void tree_accumulate_point(const orc_pose *T, const double s[2], const double r[2], double acc[NACC]) {
  double J[2][3];
  jacobian(T, s, J);
  for (int j = 0; j < 2; ++j) {
    double r_ij = r[j];
    double w_ij = orc_huber_drho(r_ij * r_ij, ORC_HUBER_K);
    double *S = acc + 9 * j;
    double t0 = w_ij * J[j][0], t1 = w_ij * J[j][1], t2 = w_ij * J[j][2];
    S[0] = S[0] + t0 * J[j][0];
    S[1] = S[1] + t0 * J[j][1];
    S[2] = S[2] + t0 * J[j][2];
    S[3] = S[3] + t1 * J[j][1];
    S[4] = S[4] + t1 * J[j][2];
    S[5] = S[5] + t2 * J[j][2];
    S[6] = S[6] + t0 * r_ij;
    S[7] = S[7] + t1 * r_ij;
    S[8] = S[8] + t2 * r_ij;
  }
  acc[18] = acc[18] + orc_huber_rho(r[0] * r[0] + r[1] * r[1], ORC_HUBER_K);
}

/* jtj[9], jtr[3] from the folded sums: g_x S_x + g_y S_y, a dimension whose sigma is 0 left out (lib.rs:243-245) */
static void tree_combine(const double S[NACC], const double stddevs[2], double jtj[9], double jtr[3]) {
  static const int upper[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
  for (int k = 0; k < 12; ++k) {
    int u = k >= 9 ? 6 + (k - 9) : upper[k / 3][k % 3];
    double v = 0.;
    if (stddevs[0] != 0.) v = v + (1. / stddevs[0]) * S[u];
    if (stddevs[1] != 0.) v = v + (1. / stddevs[1]) * S[9 + u];
    if (k >= 9) jtr[k - 9] = v;
    else jtj[k] = v;
  }
}

int orc_weighted_gauss_newton_update_tree(const orc_pose *T, const double *a, const double *b,
                                          size_t n, int blocks, int threads, double delta[3],
                                          double *huber_err) {
  if (!check_input_size(n)) return ORC_NONE;
  if (threads % 64 != 0 || blocks < 1) return -1;
  double *res = (double *)malloc(n * 2 * sizeof(double));
  for (size_t i = 0; i < n; ++i) orc_residual(T, a + 2 * i, b + 2 * i, res + 2 * i);
  double stddevs[2];
  int rc = orc_calc_stddevs(res, n, 2, stddevs);
  if (rc != ORC_OK) {
    free(res);
    return rc;
  }
  size_t G = (size_t)blocks * (size_t)threads;
  double(*thr)[NACC] = (double(*)[NACC])malloc((size_t)threads * sizeof(*thr));
  int stage2_threads = threads;
  size_t part_n = (size_t)blocks;
  double(*part)[NACC] = (double(*)[NACC])malloc(part_n * sizeof(*part));
  for (int blk = 0; blk < blocks; ++blk) {
    for (int t = 0; t < threads; ++t) {
      double acc[NACC];
      for (int k = 0; k < NACC; ++k) acc[k] = 0.;
      for (size_t i = (size_t)blk * threads + t; i < n; i += G) tree_accumulate_point(T, a + 2 * i, res + 2 * i, acc);
      memcpy(thr[t], acc, sizeof(acc));
    }
    tree_block_reduce(thr, threads, part[blk]);
  }
  /* stage 2: one block over the block sums */
  double total[NACC];
  for (int t = 0; t < stage2_threads; ++t) {
    for (int k = 0; k < NACC; ++k) thr[t][k] = 0.;
    for (size_t i = t; i < part_n; i += stage2_threads)
      for (int k = 0; k < NACC; ++k) thr[t][k] = thr[t][k] + part[i][k];
  }
  tree_block_reduce(thr, stage2_threads, total);
  free(thr);
  free(part);
  free(res);
  if (huber_err) *huber_err = total[18];
  double jtj[9], jtr[3];
  tree_combine(total, stddevs, jtj, jtr);
  return solve_update(jtj, jtr, delta);
}

/* The two halves of orc_weighted_gauss_newton_update_tree, for checking a SHARDED evaluation
 * (tests/test_dist_gloo.py): a rank that owns `blocks_local` of the tree's blocks holds exactly the
 * points they fold, compacted in fold order (chunk `it` of the local arrays is the part of global row
 * `it` its blocks cover), so running stage 1 over the local arrays with G = blocks_local * threads
 * reproduces those blocks' sums; stage 2 folds the block sums of all ranks in block order and applies
 * 1 / stddevs (the global statistics every rank selected). */
int orc_wgn_tree_partials(const orc_pose *T, const double *a, const double *b, size_t n_local, int blocks_local,
                          int threads, double *out /* blocks_local x 19 */) {
  if (threads % 64 != 0 || blocks_local < 0) return -1;
  size_t G = (size_t)blocks_local * (size_t)threads;
  double(*thr)[NACC] = (double(*)[NACC])malloc((size_t)threads * sizeof(*thr));
  for (int blk = 0; blk < blocks_local; ++blk) {
    for (int t = 0; t < threads; ++t) {
      double acc[NACC];
      for (int k = 0; k < NACC; ++k) acc[k] = 0.;
      for (size_t i = (size_t)blk * threads + t; i < n_local; i += G) {
        double r[2];
        orc_residual(T, a + 2 * i, b + 2 * i, r);
        tree_accumulate_point(T, a + 2 * i, r, acc);
      }
      memcpy(thr[t], acc, sizeof(acc));
    }
    tree_block_reduce(thr, threads, out + (size_t)blk * NACC);
  }
  free(thr);
  return ORC_OK;
}

int orc_wgn_tree_fold(const double *partials /* blocks x 19, block order */, int blocks, int threads,
                      const double stddevs[2], double delta[3], double *huber_err) {
  if (threads % 64 != 0 || blocks < 1) return -1;
  double(*thr)[NACC] = (double(*)[NACC])malloc((size_t)threads * sizeof(*thr));
  double total[NACC];
  for (int t = 0; t < threads; ++t) {
    for (int k = 0; k < NACC; ++k) thr[t][k] = 0.;
    for (int i = t; i < blocks; i += threads)
      for (int k = 0; k < NACC; ++k) thr[t][k] = thr[t][k] + partials[(size_t)i * NACC + k];
  }
  tree_block_reduce(thr, threads, total);
  free(thr);
  if (huber_err) *huber_err = total[18];
  double jtj[9], jtr[3];
  tree_combine(total, stddevs, jtj, jtr);
  return solve_update(jtj, jtr, delta);
}

/* lib.rs:59-84, generic over the summation order */
static int estimate_transform_impl(const double *a, const double *b, size_t n,
                                   const orc_icp_opts *opts, orc_pose *out) {
  double prev_error = 1.7976931348623157e308; /* f64::MAX */
  orc_pose T;
  orc_transform_identity(&T);
  int applied = 0;
  for (int it = 0; it < ORC_INNER_MAX_ITER; ++it) {
    double delta[3], err = 0.;
    int rc;
    int tree = opts && opts->sum_mode == 1;
    if (tree)
      rc = orc_weighted_gauss_newton_update_tree(&T, a, b, n, opts->reduce_blocks,
                                                 opts->reduce_threads, delta, &err);
    else
      rc = orc_weighted_gauss_newton_update(&T, a, b, n, delta);
    if (rc == ORC_NAN) {
      *out = T;
      return -ORC_NAN;
    }
    if (rc != ORC_OK) break;
    if ((delta[0] * delta[0] + delta[1] * delta[1]) + delta[2] * delta[2] < ORC_DELTA_NORM_THRESHOLD)
      break;
    if (!tree) err = orc_huber_error(&T, a, b, n);
    if (err > prev_error) break;
    prev_error = err;
    orc_pose D, Tn;
    orc_transform_new(delta, &D);
    orc_transform_mul(&D, &T, &Tn);
    T = Tn;
    ++applied;
  }
  *out = T;
  return applied;
}

int orc_estimate_transform(const double *a, const double *b, size_t n, orc_pose *out) {
  return estimate_transform_impl(a, b, n, NULL, out);
}

/* ------------------------------------------------------------------ exact NN ---- */

static inline double dist2(const double *p, const double *q, int dim) {
  double dx = q[0] - p[0], dy = q[1] - p[1];
  double d = dx * dx + dy * dy;
  if (dim == 3) {
    double dz = q[2] - p[2];
    d = d + dz * dz;
  }
  return d;
}

static int g_threads; /* orc_set_threads, below: splits independent queries over host cores */

int orc_nn_brute(const double *dst, size_t m, int dim, const double *q, size_t n, uint32_t *idx) {
  if (m == 0) return ORC_EMPTY_DST;
#pragma omp parallel for schedule(dynamic, 16) num_threads(g_threads) if (g_threads > 1)
  for (size_t i = 0; i < n; ++i) {
    const double *qi = q + (size_t)dim * i;
    double best = dist2(dst, qi, dim);
    uint32_t bi = 0;
    for (size_t j = 1; j < m; ++j) {
      double d = dist2(dst + (size_t)dim * j, qi, dim);
      if (d < best) { /* strict: the lowest index wins ties */
        best = d;
        bi = (uint32_t)j;
      }
    }
    idx[i] = bi;
  }
  return ORC_OK;
}

/* Exact kd-tree, leaf size 1 (KdTree::new(dst, 1), lib.rs:99,141): a left-balanced
 * median-split tree over a permutation of the points; node = median of its range on the
 * axis of largest extent.  Ties are explored (prune only on strictly larger plane
 * distance) and resolved to the lowest original index, so results equal orc_nn_brute. */
struct orc_kdtree {
  int dim;
  size_t m;
  double *pts;    /* permuted coordinates, dim per node */
  uint32_t *orig; /* original index per node */
  uint8_t *axis;  /* split axis per node */
};

static void kd_select(double *pts, uint32_t *orig, int dim, size_t lo_, size_t hi_, size_t k_, int ax) {
  /* quickselect on pts[.][ax] over [lo, hi], permuting points and indices together */
  ptrdiff_t lo = (ptrdiff_t)lo_, hi = (ptrdiff_t)hi_, k = (ptrdiff_t)k_;
  while (lo < hi) {
    ptrdiff_t mid = lo + (hi - lo) / 2;
    double a = pts[lo * dim + ax], b = pts[mid * dim + ax], c = pts[hi * dim + ax], pivot;
    if (a < b) pivot = (b < c) ? b : (a < c ? c : a);
    else       pivot = (a < c) ? a : (b < c ? c : b);
    ptrdiff_t i = lo, j = hi;
    do {
      while (pts[i * dim + ax] < pivot) ++i;
      while (pts[j * dim + ax] > pivot) --j;
      if (i <= j) {
        for (int d = 0; d < dim; ++d) {
          double t = pts[i * dim + d]; pts[i * dim + d] = pts[j * dim + d]; pts[j * dim + d] = t;
        }
        uint32_t t = orig[i]; orig[i] = orig[j]; orig[j] = t;
        ++i;
        --j;
      }
    } while (i <= j);
    if (k <= j) hi = j;
    else if (k >= i) lo = i;
    else return;
  }
}

static void kd_build(orc_kdtree *t, size_t lo, size_t hi) { /* [lo, hi) */
  if (hi - lo <= 1) {
    if (hi > lo) t->axis[lo] = 0;
    return;
  }
  int dim = t->dim, best_ax = 0;
  double best_ext = -1.;
  for (int d = 0; d < dim; ++d) {
    double mn = t->pts[lo * dim + d], mx = mn;
    for (size_t i = lo + 1; i < hi; ++i) {
      double v = t->pts[i * dim + d];
      if (v < mn) mn = v;
      if (v > mx) mx = v;
    }
    if (mx - mn > best_ext) { best_ext = mx - mn; best_ax = d; }
  }
  size_t mid = lo + (hi - lo) / 2;
  kd_select(t->pts, t->orig, dim, lo, hi - 1, mid, best_ax);
  t->axis[mid] = (uint8_t)best_ax;
  kd_build(t, lo, mid);
  kd_build(t, mid + 1, hi);
}

orc_kdtree *orc_kdtree_build(const double *dst, size_t m, int dim) {
  orc_kdtree *t = (orc_kdtree *)calloc(1, sizeof(*t));
  t->dim = dim;
  t->m = m;
  t->pts = (double *)malloc((m ? m : 1) * dim * sizeof(double));
  t->orig = (uint32_t *)malloc((m ? m : 1) * sizeof(uint32_t));
  t->axis = (uint8_t *)malloc(m ? m : 1);
  memcpy(t->pts, dst, m * dim * sizeof(double));
  for (size_t i = 0; i < m; ++i) t->orig[i] = (uint32_t)i;
  kd_build(t, 0, m);
  return t;
}

void orc_kdtree_free(orc_kdtree *t) {
  if (!t) return;
  free(t->pts);
  free(t->orig);
  free(t->axis);
  free(t);
}

static void kd_search(const orc_kdtree *t, size_t lo, size_t hi, const double *q, double *best,
                      uint32_t *bi) {
  while (hi > lo) {
    size_t mid = lo + (hi - lo) / 2;
    const double *p = t->pts + mid * t->dim;
    double d = dist2(p, q, t->dim);
    uint32_t o = t->orig[mid];
    if (d < *best || (d == *best && o < *bi)) {
      *best = d;
      *bi = o;
    }
    if (hi - lo == 1) return;
    int ax = t->axis[mid];
    double diff = q[ax] - p[ax];
    size_t nlo, nhi, flo, fhi;
    if (diff < 0.) { nlo = lo; nhi = mid; flo = mid + 1; fhi = hi; }
    else           { nlo = mid + 1; nhi = hi; flo = lo; fhi = mid; }
    kd_search(t, nlo, nhi, q, best, bi);
    if (diff * diff > *best) return; /* strictly farther: cannot win or tie */
    lo = flo;
    hi = fhi;
  }
}

/* The reference is single-threaded (no_std, no rayon); threads > 1 only splits the
 * independent queries of one search over host cores for bench.py's "all cores" baseline
 * (SURVEY.md 8(d)).  Results do not depend on it. */
static int g_threads = 1;
void orc_set_threads(int threads) { g_threads = threads > 1 ? threads : 1; }

int orc_kdtree_search(const orc_kdtree *t, const double *q, size_t n, uint32_t *idx) {
  if (t->m == 0) return ORC_EMPTY_DST;
#pragma omp parallel for schedule(dynamic, 4096) num_threads(g_threads) if (g_threads > 1)
  for (size_t i = 0; i < n; ++i) {
    double best = INFINITY;
    uint32_t bi = 0xffffffffu;
    kd_search(t, 0, t->m, q + (size_t)t->dim * i, &best, &bi);
    idx[i] = bi;
  }
  return ORC_OK;
}

/* ---------------------------------------------------------------- ICP driver ---- */

/* lib.rs:105-130 (dim 2) and lib.rs:148-173 (dim 3) */
int orc_icp_estimate_tree(const orc_kdtree *tree, const double *dst, size_t m, const double *src,
                          size_t n, const orc_pose *init, size_t max_iter,
                          const orc_icp_opts *opts, orc_pose *out, uint32_t *last_idx,
                          uint32_t *inner_iters) {
  int dim = tree ? tree->dim : 0;
  if (!tree) return -1;
  orc_pose T = *init;
  size_t nn = n ? n : 1;
  double *st = (double *)malloc(nn * dim * sizeof(double)); /* src_tranformed */
  double *a = (double *)malloc(nn * 2 * sizeof(double));
  double *b = (double *)malloc(nn * 2 * sizeof(double));
  uint32_t *idx = (uint32_t *)malloc(nn * sizeof(uint32_t));
  int rc = ORC_OK;
  for (size_t it = 0; it < max_iter; ++it) {
    for (size_t i = 0; i < n; ++i) {
      if (dim == 3) orc_transform_xy(&T, src + 3 * i, st + 3 * i);
      else          orc_transform_apply(&T, src + 2 * i, st + 2 * i);
    }
    if (n > 0) {
      if (m == 0) { rc = ORC_EMPTY_DST; break; } /* index.unwrap() panics, lib.rs:122,165 */
      if (opts && opts->use_kdtree) rc = orc_kdtree_search(tree, st, n, idx);
      else                          rc = orc_nn_brute(dst, m, dim, st, n, idx);
      if (rc != ORC_OK) break;
    }
    for (size_t i = 0; i < n; ++i) { /* get_xy, lib.rs:86-89 */
      a[2 * i] = st[dim * i];
      a[2 * i + 1] = st[dim * i + 1];
      b[2 * i] = dst[(size_t)dim * idx[i]];
      b[2 * i + 1] = dst[(size_t)dim * idx[i] + 1];
    }
    orc_pose dT, Tn;
    int applied = estimate_transform_impl(a, b, n, opts, &dT);
    if (applied < 0) { rc = ORC_NAN; break; }
    if (inner_iters) inner_iters[it] = (uint32_t)applied;
    orc_transform_mul(&dT, &T, &Tn);
    T = Tn;
  }
  if (last_idx && rc == ORC_OK && max_iter > 0) memcpy(last_idx, idx, n * sizeof(uint32_t));
  *out = T;
  free(st);
  free(a);
  free(b);
  free(idx);
  return rc;
}

int orc_icp_estimate(int dim, const double *dst, size_t m, const double *src, size_t n,
                     const orc_pose *init, size_t max_iter, const orc_icp_opts *opts,
                     orc_pose *out, uint32_t *last_idx, uint32_t *inner_iters) {
  if (dim != 2 && dim != 3) return -1;
  orc_kdtree *t;
  if (opts && opts->use_kdtree) {
    t = orc_kdtree_build(dst, m, dim);
  } else { /* brute force needs no tree; keep a stub for dim */
    t = (orc_kdtree *)calloc(1, sizeof(*t));
    t->dim = dim;
    t->m = m;
  }
  int rc = orc_icp_estimate_tree(t, dst, m, src, n, init, max_iter, opts, out, last_idx, inner_iters);
  orc_kdtree_free(t);
  return rc;
}

/* ======================================================================================
 * EXTENSION CHECKER (no reference counterpart): point-to-plane residuals.
 * tier4/icp_rust has no normals and no plane residual anywhere in src/, so there is nothing to
 * restate from the reference here and NO parity claim is attached to these functions: they are an
 * independent CPU statement of the definition the library documents (include/icp_mi355x.h section
 * 7, icp_rust_amd/csrc/p2plane.hip) -- brute-force k nearest neighbours instead of the device's grid
 * walk, the reference's own left-fold sums instead of the device's tree -- used by
 * tests/test_p2plane.py for self-consistency.  Everything around the residual follows the cited
 * reference lines (inner loop lib.rs:59-84, weights :236-255, huber.rs, stats.rs, linalg.rs).
 * ==================================================================================== */

static void p2pl_jacobi3(double a[3][3], double v[3][3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) v[i][j] = i == j ? 1. : 0.;
  for (int sweep = 0; sweep < 12; ++sweep) {
    double off = (a[0][1] * a[0][1] + a[0][2] * a[0][2]) + a[1][2] * a[1][2];
    if (off == 0.) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (a[p][q] == 0.) continue;
        double theta = (a[q][q] - a[p][p]) / (2. * a[p][q]);
        double t = (theta >= 0. ? 1. : -1.) / (fabs(theta) + sqrt(theta * theta + 1.));
        double c = 1. / sqrt(t * t + 1.), s = t * c;
        for (int k = 0; k < 3; ++k) {
          double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          double vkp = v[k][p], vkq = v[k][q];
          v[k][p] = c * vkp - s * vkq;
          v[k][q] = s * vkp + c * vkq;
        }
      }
  }
}

/* unit normals (m x 3) from the k nearest targets (itself included; ties by lowest index) */
int orc_p2pl_normals(const double *dst, size_t m, int k_, double *normals) {
  return orc_p2pl_normals_range(dst, m, 0, k_, normals);
}

/* normals of the targets [first, m) only, from their k nearest among ALL m targets; normals_out holds m x 3
 * values and rows [0, first) are left alone (the growing map's "normals at insertion time") */
int orc_p2pl_normals_range(const double *dst, size_t m, size_t first, int k_, double *normals) {
  if (k_ < 3 || k_ > 16 || first > m) return -1;
  int k = (size_t)k_ < m ? k_ : (int)m;
#pragma omp parallel for schedule(dynamic, 64) num_threads(g_threads) if (g_threads > 1)
  for (size_t i = first; i < m; ++i) {
    double bd[16];
    uint32_t bi[16];
    int cnt = 0;
    const double *p = dst + 3 * i;
    for (size_t j = 0; j < m; ++j) {
      double dd = dist2(dst + 3 * j, p, 3);
      uint32_t tj = (uint32_t)j;
      if (cnt == k && !(dd < bd[k - 1] || (dd == bd[k - 1] && tj < bi[k - 1]))) continue;
      int pos = cnt < k ? cnt : k - 1;
      while (pos > 0 && (dd < bd[pos - 1] || (dd == bd[pos - 1] && tj < bi[pos - 1]))) {
        bd[pos] = bd[pos - 1];
        bi[pos] = bi[pos - 1];
        --pos;
      }
      bd[pos] = dd;
      bi[pos] = tj;
      if (cnt < k) ++cnt;
    }
    double nrm[3] = {0., 0., 0.};
    if (cnt >= 3) {
      double mean[3] = {0., 0., 0.};
      for (int j = 0; j < cnt; ++j)
        for (int d = 0; d < 3; ++d) mean[d] = mean[d] + dst[3 * (size_t)bi[j] + d];
      for (int d = 0; d < 3; ++d) mean[d] = mean[d] / (double)cnt;
      double a[3][3] = {{0., 0., 0.}, {0., 0., 0.}, {0., 0., 0.}}, v[3][3];
      for (int j = 0; j < cnt; ++j) {
        double e[3];
        for (int d = 0; d < 3; ++d) e[d] = dst[3 * (size_t)bi[j] + d] - mean[d];
        for (int r = 0; r < 3; ++r)
          for (int s = 0; s < 3; ++s) a[r][s] = a[r][s] + e[r] * e[s];
      }
      p2pl_jacobi3(a, v);
      int col = 0;
      if (a[1][1] < a[col][col]) col = 1;
      if (a[2][2] < a[col][col]) col = 2;
      double n0 = v[0][col], n1 = v[1][col], n2 = v[2][col];
      double len = sqrt((n0 * n0 + n1 * n1) + n2 * n2);
      if (len > 0.) {
        n0 = n0 / len;
        n1 = n1 / len;
        n2 = n2 / len;
        double lead = n2 != 0. ? n2 : (n1 != 0. ? n1 : n0);
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
    normals[3 * i] = nrm[0];
    normals[3 * i + 1] = nrm[1];
    normals[3 * i + 2] = nrm[2];
  }
  return ORC_OK;
}

/* Icp3d::estimate (lib.rs:148-173) with the scalar residual n_q . (T p - q); sums as left folds */
int orc_p2pl_estimate(const orc_kdtree *tree, const double *dst, size_t m, const double *normals,
                      const double *src, size_t n, const orc_pose *init, size_t max_iter, orc_pose *out,
                      uint32_t *last_idx, uint32_t *inner_iters) {
  if (!tree || tree->dim != 3) return -1;
  orc_pose T = *init;
  size_t nn = n ? n : 1;
  double *st = (double *)malloc(nn * 3 * sizeof(double));
  double *r = (double *)malloc(nn * sizeof(double));
  uint32_t *idx = (uint32_t *)malloc(nn * sizeof(uint32_t));
  int rc = ORC_OK;
  for (size_t it = 0; it < max_iter && rc == ORC_OK; ++it) {
    for (size_t i = 0; i < n; ++i) orc_transform_xy(&T, src + 3 * i, st + 3 * i);
    if (n > 0) {
      if (m == 0) { rc = ORC_EMPTY_DST; break; }
      rc = orc_kdtree_search(tree, st, n, idx);
      if (rc != ORC_OK) break;
    }
    orc_pose Ti;
    orc_transform_identity(&Ti);
    int applied = 0;
    double prev_error = 1.7976931348623157e308;
    for (int k = 0; k < ORC_INNER_MAX_ITER && n >= 2; ++k) {
      for (size_t i = 0; i < n; ++i) {
        const double *q = dst + 3 * (size_t)idx[i], *nq = normals + 3 * (size_t)idx[i];
        double a[2] = {st[3 * i], st[3 * i + 1]}, ta[2];
        orc_transform_apply(&Ti, a, ta);
        r[i] = (nq[0] * (ta[0] - q[0]) + nq[1] * (ta[1] - q[1])) + nq[2] * (st[3 * i + 2] - q[2]);
      }
      double sigma;
      {
        double *tmp = (double *)malloc(nn * sizeof(double));
        memcpy(tmp, r, n * sizeof(double));
        int src_ = orc_standard_deviation(tmp, n, &sigma);
        free(tmp);
        if (src_ == ORC_NAN) { rc = ORC_NAN; break; }
      }
      double jtr[3] = {0., 0., 0.}, jtj[9] = {0.}, err = 0.;
      for (size_t i = 0; i < n; ++i) {
        const double *nq = normals + 3 * (size_t)idx[i];
        double e = r[i] * r[i];
        if (sigma != 0.) {
          double s2[2] = {st[3 * i], st[3 * i + 1]}, J2[2][3];
          jacobian(&Ti, s2, J2);
          double J[3];
          for (int c = 0; c < 3; ++c) J[c] = nq[0] * J2[0][c] + nq[1] * J2[1][c];
          double wg = orc_huber_drho(e, ORC_HUBER_K) * (1. / sigma);
          for (int c = 0; c < 3; ++c) jtr[c] = jtr[c] + (wg * J[c]) * r[i];
          for (int p = 0; p < 3; ++p)
            for (int q2 = 0; q2 < 3; ++q2) jtj[3 * p + q2] = jtj[3 * p + q2] + (wg * J[p]) * J[q2];
        }
        err = err + orc_huber_rho(e, ORC_HUBER_K);
      }
      double delta[3];
      if (solve_update(jtj, jtr, delta) != ORC_OK) break;
      if ((delta[0] * delta[0] + delta[1] * delta[1]) + delta[2] * delta[2] < ORC_DELTA_NORM_THRESHOLD) break;
      if (err > prev_error) break;
      prev_error = err;
      orc_pose D, Tn;
      orc_transform_new(delta, &D);
      orc_transform_mul(&D, &Ti, &Tn);
      Ti = Tn;
      ++applied;
    }
    if (rc != ORC_OK) break;
    if (inner_iters) inner_iters[it] = (uint32_t)applied;
    orc_pose Tn;
    orc_transform_mul(&Ti, &T, &Tn);
    T = Tn;
  }
  if (last_idx && rc == ORC_OK && max_iter > 0) memcpy(last_idx, idx, n * sizeof(uint32_t));
  *out = T;
  free(st);
  free(r);
  free(idx);
  return rc;
}
