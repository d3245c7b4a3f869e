/*
 * icp_trig.h -- the sine / cosine that Transform::new evaluates (src/so2.rs:8-17, src/se2.rs:26-27).
 *
 * The reference is a `no_std` crate: its `f64::cos` / `f64::sin` resolve through
 * `num_traits::real::Real` with the `libm` feature (Cargo.toml:17-20), i.e. to the Rust `libm`
 * crate, which is a port of musl's (FreeBSD msun's) fdlibm kernels.  The source of that crate is
 * not under /root/reference; what is restated here is the published algorithm it ports:
 *     sin.c / cos.c          argument classes and quadrant selection
 *     __rem_pio2.c           Cody-Waite reduction by pi/2 in up to three rounds (|x| < 2^20 pi/2)
 *     __sin.c / __cos.c      the degree-13 / degree-14 minimax kernels on [-pi/4, pi/4]
 * operation by operation, in IEEE double arithmetic without FMA contraction (every translation unit
 * that includes this file is compiled with -ffp-contract=off).  ONE definition serves the host
 * (pose.hpp), the device (the Gauss-Newton kernels apply updates on the GPU) and the CPU oracle, so
 * the three agree to the last bit; tests/test_trig.py checks it against the C library (<= 1 ulp,
 * millions of arguments incl. every branch boundary) and against exactly known values.
 *
 * NOT restated: the Payne-Hanek path for |x| >= 2^20 pi/2 ~ 1.6e6 rad (__rem_pio2_large.c, 690
 * table words).  icp_trig_in_range() tells the callers: host and oracle use the C library there,
 * the device hands the update back to the host.  An se(2) update of that size does not occur
 * (the inner loop applies |delta| ~ 1e-3 .. 1e-1).
 */
#ifndef ICP_TRIG_H
#define ICP_TRIG_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define ICP_TRIG_FN __host__ __device__ static inline
#else
#define ICP_TRIG_FN static inline
#endif

ICP_TRIG_FN uint64_t icp_trig_bits(double x) {
  uint64_t u;
  memcpy(&u, &x, sizeof u);
  return u;
}
ICP_TRIG_FN double icp_trig_from_bits(uint64_t u) {
  double x;
  memcpy(&x, &u, sizeof x);
  return x;
}

/* __sin.c: sin(x + y) for |x| <= pi/4, y the tail of x (iy == 0: y is known to be 0) */
ICP_TRIG_FN double icp_k_sin(double x, double y, int iy) {
  const double S1 = icp_trig_from_bits(0xBFC5555555555549ull); /* -1.66666666666666324348e-01 */
  const double S2 = icp_trig_from_bits(0x3F8111111110F8A6ull); /*  8.33333333332248946124e-03 */
  const double S3 = icp_trig_from_bits(0xBF2A01A019C161D5ull); /* -1.98412698298579493134e-04 */
  const double S4 = icp_trig_from_bits(0x3EC71DE357B1FE7Dull); /*  2.75573137070700676789e-06 */
  const double S5 = icp_trig_from_bits(0xBE5AE5E68A2B9CEBull); /* -2.50507602534068634195e-08 */
  const double S6 = icp_trig_from_bits(0x3DE5D93A5ACFD57Cull); /*  1.58969099521155010221e-10 */
  const double z = x * x;
  const double w = z * z;
  const double r = S2 + z * (S3 + z * S4) + z * w * (S5 + z * S6);
  const double v = z * x;
  if (iy == 0) return x + v * (S1 + z * r);
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

/* __cos.c: cos(x + y) for |x| <= pi/4 */
ICP_TRIG_FN double icp_k_cos(double x, double y) {
  const double C1 = icp_trig_from_bits(0x3FA555555555554Cull); /*  4.16666666666666019037e-02 */
  const double C2 = icp_trig_from_bits(0xBF56C16C16C15177ull); /* -1.38888888888741095749e-03 */
  const double C3 = icp_trig_from_bits(0x3EFA01A019CB1590ull); /*  2.48015872894767294178e-05 */
  const double C4 = icp_trig_from_bits(0xBE927E4F809C52ADull); /* -2.75573143513906633035e-07 */
  const double C5 = icp_trig_from_bits(0x3E21EE9EBDB4B1C4ull); /*  2.08757232129817482790e-09 */
  const double C6 = icp_trig_from_bits(0xBDA8FAE9BE8838D4ull); /* -1.13596475577881948265e-11 */
  const double z = x * x;
  double w = z * z;
  const double r = z * (C1 + z * (C2 + z * C3)) + w * w * (C4 + z * (C5 + z * C6));
  const double hz = 0.5 * z;
  w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + (z * r - x * y));
}

/* |x| < 2^20 * pi/2 (and finite): the reduction below covers it */
ICP_TRIG_FN int icp_trig_in_range(double x) {
  const uint32_t ix = (uint32_t)(icp_trig_bits(x) >> 32) & 0x7fffffffu;
  return ix < 0x413921fbu;
}

/* __rem_pio2.c: n and y[0] + y[1] with x = n * pi/2 + y, |y| <= pi/4; caller guarantees
 * icp_trig_in_range(x) and |x| > pi/4 */
ICP_TRIG_FN int icp_rem_pio2(double x, double y[2]) {
  const double toint = 6755399441055744.0;                           /* 1.5 / DBL_EPSILON */
  const double pio4 = icp_trig_from_bits(0x3FE921FB54442D18ull);    /* 0x1.921fb54442d18p-1 */
  const double invpio2 = icp_trig_from_bits(0x3FE45F306DC9C883ull); /* 6.36619772367581382433e-01 */
  const double pio2_1 = icp_trig_from_bits(0x3FF921FB54400000ull);  /* 1.57079632673412561417e+00: first 33 bits of pi/2 */
  const double pio2_1t = icp_trig_from_bits(0x3DD0B4611A626331ull); /* 6.07710050650619224932e-11: pi/2 - pio2_1 */
  const double pio2_2 = icp_trig_from_bits(0x3DD0B4611A600000ull);  /* 6.07710050630396597660e-11: second 33 bits */
  const double pio2_2t = icp_trig_from_bits(0x3BA3198A2E037073ull); /* 2.02226624879595063154e-21 */
  const double pio2_3 = icp_trig_from_bits(0x3BA3198A2E000000ull);  /* 2.02226624871116645580e-21: third 33 bits */
  const double pio2_3t = icp_trig_from_bits(0x397B839A252049C1ull); /* 8.47842766036889956997e-32 */
  const uint64_t ux = icp_trig_bits(x);
  const int sign = (int)(ux >> 63);
  const uint32_t ix = (uint32_t)(ux >> 32) & 0x7fffffffu;
  double z, w, t, r, fn;
  int n;
  int medium = 0;
  if (ix <= 0x400f6a7au) {                     /* |x| ~<= 5pi/4 */
    if ((ix & 0xfffffu) == 0x921fbu) {         /* |x| ~= pi/2 or 2pi/2: cancellation, medium case */
      medium = 1;
    } else if (ix <= 0x4002d97cu) {            /* |x| ~<= 3pi/4 */
      if (!sign) {
        z = x - pio2_1; /* one round good to 85 bits */
        y[0] = z - pio2_1t;
        y[1] = (z - y[0]) - pio2_1t;
        return 1;
      }
      z = x + pio2_1;
      y[0] = z + pio2_1t;
      y[1] = (z - y[0]) + pio2_1t;
      return -1;
    } else {
      if (!sign) {
        z = x - 2 * pio2_1;
        y[0] = z - 2 * pio2_1t;
        y[1] = (z - y[0]) - 2 * pio2_1t;
        return 2;
      }
      z = x + 2 * pio2_1;
      y[0] = z + 2 * pio2_1t;
      y[1] = (z - y[0]) + 2 * pio2_1t;
      return -2;
    }
  } else if (ix <= 0x401c463bu) {              /* |x| ~<= 9pi/4 */
    if (ix <= 0x4015fdbcu) {                   /* |x| ~<= 7pi/4 */
      if (ix == 0x4012d97cu) {                 /* |x| ~= 3pi/2 */
        medium = 1;
      } else if (!sign) {
        z = x - 3 * pio2_1;
        y[0] = z - 3 * pio2_1t;
        y[1] = (z - y[0]) - 3 * pio2_1t;
        return 3;
      } else {
        z = x + 3 * pio2_1;
        y[0] = z + 3 * pio2_1t;
        y[1] = (z - y[0]) + 3 * pio2_1t;
        return -3;
      }
    } else {
      if (ix == 0x401921fbu) {                 /* |x| ~= 4pi/2 */
        medium = 1;
      } else if (!sign) {
        z = x - 4 * pio2_1;
        y[0] = z - 4 * pio2_1t;
        y[1] = (z - y[0]) - 4 * pio2_1t;
        return 4;
      } else {
        z = x + 4 * pio2_1;
        y[0] = z + 4 * pio2_1t;
        y[1] = (z - y[0]) + 4 * pio2_1t;
        return -4;
      }
    }
  }
  (void)medium;
  /* medium size: rint(x / (pi/2)) by the add-and-subtract trick (round to nearest) */
  fn = x * invpio2 + toint - toint;
  n = (int32_t)fn;
  r = x - fn * pio2_1;
  w = fn * pio2_1t; /* 1st round, good to 85 bits */
  /* (matters with directed rounding only; kept for operation-by-operation fidelity) */
  if (r - w < -pio4) {
    n--;
    fn--;
    r = x - fn * pio2_1;
    w = fn * pio2_1t;
  } else if (r - w > pio4) {
    n++;
    fn++;
    r = x - fn * pio2_1;
    w = fn * pio2_1t;
  }
  y[0] = r - w;
  {
    int ey = (int)(icp_trig_bits(y[0]) >> 52) & 0x7ff;
    const int ex = (int)(ix >> 20);
    if (ex - ey > 16) { /* 2nd round, good to 118 bits */
      t = r;
      w = fn * pio2_2;
      r = t - w;
      w = fn * pio2_2t - ((t - r) - w);
      y[0] = r - w;
      ey = (int)(icp_trig_bits(y[0]) >> 52) & 0x7ff;
      if (ex - ey > 49) { /* 3rd round, good to 151 bits, covers all cases */
        t = r;
        w = fn * pio2_3;
        r = t - w;
        w = fn * pio2_3t - ((t - r) - w);
        y[0] = r - w;
      }
    }
  }
  y[1] = (r - y[0]) - w;
  return n;
}

/* sin.c; *ok = 0 when x is outside the restated range (the value returned is then meaningless) */
ICP_TRIG_FN double icp_sin(double x, int *ok) {
  const uint32_t ix = (uint32_t)(icp_trig_bits(x) >> 32) & 0x7fffffffu;
  double y[2];
  int n;
  *ok = 1;
  if (ix <= 0x3fe921fbu) {                 /* |x| ~< pi/4 */
    if (ix < 0x3e500000u) return x;        /* |x| < 2^-26 */
    return icp_k_sin(x, 0.0, 0);
  }
  if (ix >= 0x7ff00000u) return x - x;     /* sin(Inf or NaN) is NaN */
  if (!icp_trig_in_range(x)) {
    *ok = 0;
    return 0.0;
  }
  n = icp_rem_pio2(x, y);
  switch (n & 3) {
    case 0: return icp_k_sin(y[0], y[1], 1);
    case 1: return icp_k_cos(y[0], y[1]);
    case 2: return -icp_k_sin(y[0], y[1], 1);
    default: return -icp_k_cos(y[0], y[1]);
  }
}

/* cos.c */
ICP_TRIG_FN double icp_cos(double x, int *ok) {
  const uint32_t ix = (uint32_t)(icp_trig_bits(x) >> 32) & 0x7fffffffu;
  double y[2];
  int n;
  *ok = 1;
  if (ix <= 0x3fe921fbu) {                 /* |x| ~< pi/4 */
    if (ix < 0x3e46a09eu) return 1.0;      /* |x| < 2^-27 * sqrt(2) */
    return icp_k_cos(x, 0.0);
  }
  if (ix >= 0x7ff00000u) return x - x;     /* cos(Inf or NaN) is NaN */
  if (!icp_trig_in_range(x)) {
    *ok = 0;
    return 0.0;
  }
  n = icp_rem_pio2(x, y);
  switch (n & 3) {
    case 0: return icp_k_cos(y[0], y[1]);
    case 1: return -icp_k_sin(y[0], y[1], 1);
    case 2: return -icp_k_cos(y[0], y[1]);
    default: return icp_k_sin(y[0], y[1], 1);
  }
}

#endif /* ICP_TRIG_H */
