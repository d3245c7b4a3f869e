// Host-side pose algebra of the ICP path: Transform, se2, so2, inverse3x3, norm.
//
// These run on the host exactly as in the reference (a handful of flops per inner
// iteration); the device kernels only ever see the six numbers of a pose.  Operation
// order follows the reference so that results agree to the last bit with a
// same-order evaluation (citations: file:line under /root/reference).
#pragma once
#include <cmath>
#include <cstddef>

#include "../../include/icp_mi355x.h"
#include "../../include/icp_trig.h"

#if defined(__HIPCC__)
#define ICP_HD __host__ __device__
#else
#define ICP_HD
#endif

namespace icp {

using Pose = icp_pose;

// f64::sin / f64::cos as the reference's no_std build evaluates them (num-traits -> libm crate, a
// port of musl's kernels: include/icp_trig.h); the same definition runs on the device and in the
// oracle.  Beyond 2^20 pi/2 (Payne-Hanek, not restated) the C library serves.
inline double ref_sin(double x) {
  int ok;
  const double v = icp_sin(x, &ok);
  return ok ? v : std::sin(x);
}
inline double ref_cos(double x) {
  int ok;
  const double v = icp_cos(x, &ok);
  return ok ? v : std::cos(x);
}

// se2::calc_rt (src/se2.rs:21-41) from given cos / sin of theta -- the part host and device share
ICP_HD inline icp_pose calc_rt_cs(const double p[3], double c, double s) {
  const double theta = p[2];
  const double vx = p[0], vy = p[1];
  icp_pose o;
  o.r00 = c;
  o.r10 = s;
  o.r01 = -s;
  o.r11 = c;
  if (theta == 0.) {
    o.tx = vx;
    o.ty = vy;
  } else {
    o.tx = (s * vx - (1. - c) * vy) / theta;
    o.ty = ((1. - c) * vx + s * vy) / theta;
  }
  return o;
}

// Transform::new where only the restated range of sin / cos is available (device code): *ok = false
// when |theta| >= 2^20 pi/2 -- the caller hands the update to the host
ICP_HD inline icp_pose transform_new_in_range(const double p[3], bool *ok) {
  int ok_s, ok_c;
  const double s = icp_sin(p[2], &ok_s), c = icp_cos(p[2], &ok_c);
  *ok = ok_s && ok_c;
  return calc_rt_cs(p, c, s);
}

// so2::exp / so2::new_rotation2 (src/so2.rs:8-31); column-major 2x2
inline void so2_exp(double theta, double m[4]) {
  const double c = ref_cos(theta), s = ref_sin(theta);
  m[0] = c;
  m[1] = s;
  m[2] = -s;
  m[3] = c;
}

// so2::log (src/so2.rs:19-21)
inline double so2_log(const double m[4]) { return std::atan2(m[1], m[0]); }

// se2::calc_rt (src/se2.rs:21-41) == Transform::new (src/transform.rs:13-16)
inline Pose transform_new(const double p[3]) { return calc_rt_cs(p, ref_cos(p[2]), ref_sin(p[2])); }

// Transform::identity (src/transform.rs:34-39)
ICP_HD inline Pose transform_identity() { return Pose{1., 0., 0., 1., 0., 0.}; }

// Transform::transform (src/transform.rs:22-24): rot * p + t
ICP_HD inline void transform_apply(const Pose &T, const double p[2], double out[2]) {
  const double x = p[0], y = p[1];
  out[0] = (T.r00 * x + T.r01 * y) + T.tx;
  out[1] = (T.r10 * x + T.r11 * y) + T.ty;
}

// Transform::inverse (src/transform.rs:26-32)
ICP_HD inline Pose transform_inverse(const Pose &T) {
  Pose o;
  o.r00 = T.r00;
  o.r01 = T.r10;
  o.r10 = T.r01;
  o.r11 = T.r11;
  const double ix = o.r00 * T.tx + o.r01 * T.ty;
  const double iy = o.r10 * T.tx + o.r11 * T.ty;
  o.tx = -ix;
  o.ty = -iy;
  return o;
}

// impl Mul for Transform (src/transform.rs:42-51)
ICP_HD inline Pose transform_mul(const Pose &l, const Pose &r) {
  Pose o;
  o.r00 = l.r00 * r.r00 + l.r01 * r.r10;
  o.r10 = l.r10 * r.r00 + l.r11 * r.r10;
  o.r01 = l.r00 * r.r01 + l.r01 * r.r11;
  o.r11 = l.r10 * r.r01 + l.r11 * r.r11;
  o.tx = (l.r00 * r.tx + l.r01 * r.ty) + l.tx;
  o.ty = (l.r10 * r.tx + l.r11 * r.ty) + l.ty;
  return o;
}

// se2::exp (src/se2.rs:43-52), row-major 3x3
inline void se2_exp(const double p[3], double m[9]) {
  const Pose T = transform_new(p);
  m[0] = T.r00; m[1] = T.r01; m[2] = T.tx;
  m[3] = T.r10; m[4] = T.r11; m[5] = T.ty;
  m[6] = 0.;    m[7] = 0.;    m[8] = 1.;
}

// se2::get_rt (src/se2.rs:11-19)
inline void se2_get_rt(const double m[9], double rot_rowmajor[4], double t[2]) {
  rot_rowmajor[0] = m[0]; rot_rowmajor[1] = m[1];
  rot_rowmajor[2] = m[3]; rot_rowmajor[3] = m[4];
  t[0] = m[2];
  t[1] = m[5];
}

// se2::log (src/se2.rs:54-77)
inline void se2_log(const double m[9], double p[3]) {
  double rot[4], t[2];
  se2_get_rt(m, rot, t);
  const double cm[4] = {rot[0], rot[2], rot[1], rot[3]};
  const double theta = so2_log(cm);
  double v00, v01, v10, v11;
  if (theta == 0.) {
    v00 = 1.; v01 = 0.; v10 = 0.; v11 = 1.;
  } else if (theta == M_PI) {
    v00 = 0.; v01 = 0.5 * theta; v10 = -0.5 * theta; v11 = 0.;
  } else {
    const double k = ref_sin(theta) / (1. - ref_cos(theta));
    const double h = 0.5 * theta;
    v00 = h * k; v01 = h * 1.; v10 = h * -1.; v11 = h * k;
  }
  p[0] = v00 * t[0] + v01 * t[1];
  p[1] = v10 * t[0] + v11 * t[1];
  p[2] = theta;
}

// linalg::inverse3x3 (src/linalg.rs:3-29); row-major; false iff det == 0
ICP_HD inline bool inverse3x3(const double m[9], double out[9]) {
  const double m00 = m[0], m01 = m[1], m02 = m[2];
  const double m10 = m[3], m11 = m[4], m12 = m[5];
  const double m20 = m[6], m21 = m[7], m22 = m[8];
  const double det = m00 * (m22 * m11 - m21 * m12) - m10 * (m22 * m01 - m21 * m02) +
                     m20 * (m12 * m01 - m11 * m02);
  if (det == 0.) return false;
  const double a[9] = {
      m22 * m11 - m21 * m12,    -(m22 * m01 - m21 * m02), m12 * m01 - m11 * m02,
      -(m22 * m10 - m20 * m12), m22 * m00 - m20 * m02,    -(m12 * m00 - m10 * m02),
      m21 * m10 - m20 * m11,    -(m21 * m00 - m20 * m01), m11 * m00 - m10 * m01,
  };
  for (int i = 0; i < 9; ++i) out[i] = a[i] / det;
  return true;
}

// `-jtj_inv * jtr` (src/lib.rs:212-215, 257-260); false where the reference returns None
ICP_HD inline bool solve_update(const double jtj[9], const double jtr[3], double delta[3]) {
  double inv[9];
  if (!inverse3x3(jtj, inv)) return false;
  for (int i = 0; i < 3; ++i)
    delta[i] = ((-inv[3 * i + 0]) * jtr[0] + (-inv[3 * i + 1]) * jtr[1]) + (-inv[3 * i + 2]) * jtr[2];
  return true;
}

// norm (src/norm.rs:8-21); column-major
inline double norm(const double *m, size_t nrows, size_t ncols) {
  double res = 0.;
  for (size_t c = 0; c < ncols; ++c) {
    const double *col = m + c * nrows;
    double d = 0.;
    for (size_t r = 0; r < nrows; ++r) d = (r == 0) ? col[0] * col[0] : d + col[r] * col[r];
    res += d;
  }
  return std::sqrt(res);
}

}  // namespace icp
