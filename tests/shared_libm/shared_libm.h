/*
 * shared_libm.h -- ONE implementation of the five libm functions on the radiance path (cosf, sinf, atan2f, acosf, powf)
 * for BOTH sides of the radiance parity test.  TEST BUILDS ONLY: nothing that ships includes this file.
 *
 * Why: the oracle calls glibc, the kernels call the device library.  Both are good to an ulp, but not to the same ulp,
 * and on scene 1's curved mirrors an ulp in a bounce direction now and then decides whether a later ray hits -- which is
 * why the shipped builds are compared with a tolerance (tests/test_gpu_radiance.py).  To show that libm is the ONLY
 * source of that difference, a test build of the oracle (-DPT_SHARED_LIBM, oracle/Makefile: libpt_oracle_sharedlibm.so) and a
 * test build of the kernels (-DPTX_SHARED_LIBM, Makefile: libptx_backend_sharedlibm.so) replace the five calls by the functions
 * below: nothing but IEEE +, -, *, /, sqrt, floor and integer operations on double, no library call, no fused
 * multiply-add (both builds use -ffp-contract=off), so x86-64 and gfx950 return the same bits.  tests/test_gpu_shared_libm.py
 * then asks for bit-identical accumulation buffers.
 *
 * Accuracy is that of a plain double-precision evaluation rounded to float (well under 1 ulp of float): the pictures are
 * the same pictures.  Not a libm: no errno, NaN handling only as far as the path needs it (noted at each function).
 */
#ifndef PT_SHARED_LIBM_H
#define PT_SHARED_LIBM_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define PTM_FN __host__ __device__ static inline
#else
#define PTM_FN static inline
#endif

PTM_FN uint64_t ptm_bits(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
PTM_FN double ptm_from_bits(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
PTM_FN double ptm_abs(double x) { return ptm_from_bits(ptm_bits(x) & 0x7FFFFFFFFFFFFFFFull); }
PTM_FN int ptm_isnan(double x) { return (ptm_bits(x) & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull; }

#define PTM_PI 3.14159265358979323846
#define PTM_PI_2 1.57079632679489661923
#define PTM_PI_4 0.78539816339744830962

/* sin and cos of r in [-pi/4, pi/4]: Taylor series in r*r (r^16 / 16! < 1e-15) */
PTM_FN double ptm_sin_kernel(double r) {
  const double z = r * r;
  double p = -1.0 / 1307674368000.0;            /* -1/15! */
  p = p * z + 1.0 / 6227020800.0;               /*  1/13! */
  p = p * z - 1.0 / 39916800.0;                 /* -1/11! */
  p = p * z + 1.0 / 362880.0;                   /*  1/9!  */
  p = p * z - 1.0 / 5040.0;                     /* -1/7!  */
  p = p * z + 1.0 / 120.0;                      /*  1/5!  */
  p = p * z - 1.0 / 6.0;                        /* -1/3!  */
  return r + r * (z * p);
}
PTM_FN double ptm_cos_kernel(double r) {
  const double z = r * r;
  double p = 1.0 / 20922789888000.0;            /*  1/16! */
  p = p * z - 1.0 / 87178291200.0;              /* -1/14! */
  p = p * z + 1.0 / 479001600.0;                /*  1/12! */
  p = p * z - 1.0 / 3628800.0;                  /* -1/10! */
  p = p * z + 1.0 / 40320.0;                    /*  1/8!  */
  p = p * z - 1.0 / 720.0;                      /* -1/6!  */
  p = p * z + 1.0 / 24.0;                       /*  1/4!  */
  p = p * z - 0.5;
  return 1.0 + z * p;
}
/* x = k * pi/2 + r with r in [-pi/4, pi/4]; pi/2 in two parts (the path's arguments are below 7, so k <= 4 and
 * k * high part is exact) */
PTM_FN double ptm_reduce(double x, int* quadrant) {
  const double k = __builtin_floor(x * (2.0 / PTM_PI) + 0.5);
  const double hi = 1.5707963267341256, lo = 6.077100506506192e-11;   /* pi/2 = hi + lo, hi has 33 significant bits */
  const double r = (x - k * hi) - k * lo;
  const long long ki = (long long)k;
  *quadrant = (int)(ki & 3);
  return r;
}
PTM_FN float ptm_sinf(float xf) {
  const double x = (double)xf;
  if (ptm_isnan(x) || ptm_abs(x) > 1e9) return (float)(x - x);  /* NaN / out of the reduction's range: NaN */
  int q;
  const double r = ptm_reduce(x, &q);
  const double s = ptm_sin_kernel(r), c = ptm_cos_kernel(r);
  const double v = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
  return (float)v;
}
PTM_FN float ptm_cosf(float xf) {
  const double x = (double)xf;
  if (ptm_isnan(x) || ptm_abs(x) > 1e9) return (float)(x - x);
  int q;
  const double r = ptm_reduce(x, &q);
  const double s = ptm_sin_kernel(r), c = ptm_cos_kernel(r);
  const double v = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
  return (float)v;
}

/* atan(a) for a in [0, 1]: a > tan(pi/8) is folded to (a - 1) / (a + 1) + pi/4, then the odd series (|t| <= 0.4143: t^31 / 31 < 1e-13) */
PTM_FN double ptm_atan01(double a) {
  double t = a, base = 0.0;
  if (a > 0.41421356237309503) {
    t = (a - 1.0) / (a + 1.0);
    base = PTM_PI_4;
  }
  const double z = t * t;
  double p = 1.0 / 29.0;
  p = -1.0 / 27.0 + z * p;
  p = 1.0 / 25.0 + z * p;
  p = -1.0 / 23.0 + z * p;
  p = 1.0 / 21.0 + z * p;
  p = -1.0 / 19.0 + z * p;
  p = 1.0 / 17.0 + z * p;
  p = -1.0 / 15.0 + z * p;
  p = 1.0 / 13.0 + z * p;
  p = -1.0 / 11.0 + z * p;
  p = 1.0 / 9.0 + z * p;
  p = -1.0 / 7.0 + z * p;
  p = 1.0 / 5.0 + z * p;
  p = -1.0 / 3.0 + z * p;
  return base + (t + t * (z * p));
}
PTM_FN double ptm_atan2(double y, double x) {
  if (ptm_isnan(x) || ptm_isnan(y)) return x + y;
  const double ax = ptm_abs(x), ay = ptm_abs(y);
  const double mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
  double a = 0.0;
  if (mx > 0.0) a = ptm_atan01(mn / mx);       /* atan2(0, 0) = 0 here (glibc: +-0 or +-pi by the signs of the zeros) */
  if (ay > ax) a = PTM_PI_2 - a;
  if (x < 0.0) a = PTM_PI - a;
  return (y < 0.0) ? -a : a;
}
PTM_FN float ptm_atan2f(float y, float x) { return (float)ptm_atan2((double)y, (double)x); }
/* acos(x) = atan2(sqrt(1 - x^2), x); an argument a rounding beyond 1 in magnitude returns 0 or pi (glibc: NaN) */
PTM_FN float ptm_acosf(float xf) {
  const double x = (double)xf;
  double s = (1.0 - x) * (1.0 + x);
  if (s < 0.0) s = 0.0;
  return (float)ptm_atan2(__builtin_sqrt(s), x);
}

/* log(x) for finite x > 0: x = m * 2^e with m in [sqrt(1/2), sqrt(2)), log m = 2 atanh((m - 1) / (m + 1)) */
PTM_FN double ptm_log(double x) {
  uint64_t u = ptm_bits(x);
  int e = 0;
  if ((u >> 52) == 0) {                         /* subnormal: scale up */
    x = x * 18014398509481984.0;                /* 2^54 */
    u = ptm_bits(x);
    e = -54;
  }
  e += (int)(u >> 52) - 1023;
  double m = ptm_from_bits((u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);   /* [1, 2) */
  if (m > 1.4142135623730951) {
    m = m * 0.5;
    e += 1;
  }
  const double s = (m - 1.0) / (m + 1.0), z = s * s;   /* |s| <= 0.1716 */
  double p = 1.0 / 23.0;
  p = 1.0 / 21.0 + z * p;
  p = 1.0 / 19.0 + z * p;
  p = 1.0 / 17.0 + z * p;
  p = 1.0 / 15.0 + z * p;
  p = 1.0 / 13.0 + z * p;
  p = 1.0 / 11.0 + z * p;
  p = 1.0 / 9.0 + z * p;
  p = 1.0 / 7.0 + z * p;
  p = 1.0 / 5.0 + z * p;
  p = 1.0 / 3.0 + z * p;
  const double lm = 2.0 * (s + s * (z * p));
  return (double)e * 0.6931471803691238 + ((double)e * 1.9082149292705877e-10 + lm);   /* ln 2 in two parts */
}
/* exp(t) for |t| < 700 */
PTM_FN double ptm_exp(double t) {
  const double k = __builtin_floor(t * 1.4426950408889634 + 0.5);
  const double r = (t - k * 0.6931471803691238) - k * 1.9082149292705877e-10;   /* |r| <= 0.3466 */
  double p = 1.0 / 6227020800.0;                /* 1/13! */
  p = 1.0 / 479001600.0 + r * p;
  p = 1.0 / 39916800.0 + r * p;
  p = 1.0 / 3628800.0 + r * p;
  p = 1.0 / 362880.0 + r * p;
  p = 1.0 / 40320.0 + r * p;
  p = 1.0 / 5040.0 + r * p;
  p = 1.0 / 720.0 + r * p;
  p = 1.0 / 120.0 + r * p;
  p = 1.0 / 24.0 + r * p;
  p = 1.0 / 6.0 + r * p;
  p = 0.5 + r * p;
  p = 1.0 + r * p;
  p = 1.0 + r * p;
  const long long ki = (long long)k;
  return p * ptm_from_bits((uint64_t)(ki + 1023) << 52);   /* |t| < 700: the exponent stays in range */
}
/* powf as the path uses it: base in [-tiny, 1], exponent 5 or 1/2.2.  Zero, negative (integral exponent only) and NaN bases
 * are handled as libm does; overflow cannot occur for these arguments. */
PTM_FN float ptm_powf(float xf, float yf) {
  const double x = (double)xf, y = (double)yf;
  if (ptm_isnan(x) || ptm_isnan(y)) return (float)(x + y);
  if (y == 0.0) return 1.0f;
  if (x == 0.0) return y > 0.0 ? 0.0f : (float)(1.0 / ptm_abs(x));
  double sign = 1.0, ax = x;
  if (x < 0.0) {
    const double yi = __builtin_floor(y);
    if (yi != y) return (float)((x - x) / (x - x));   /* NaN */
    ax = -x;
    if ((yi * 0.5) != __builtin_floor(yi * 0.5)) sign = -1.0;
  }
  double t = y * ptm_log(ax);
  if (t > 700.0) t = 700.0;
  if (t < -700.0) return (float)(sign * 0.0);
  return (float)(sign * ptm_exp(t));
}

#undef cosf
#undef sinf
#undef atan2f
#undef acosf
#undef powf
#define cosf ptm_cosf
#define sinf ptm_sinf
#define atan2f ptm_atan2f
#define acosf ptm_acosf
#define powf ptm_powf

#endif /* PT_SHARED_LIBM_H */
