/*
 * ofdg_detmath.h -- the elementary functions the DEVICE sampler / realize path is DEFINED with.
 *
 * The reference builds its affines with libm's sin/cos (agg::trans_affine_rotation,
 * DataGenerator.cpp:304-318) and its Gaussian supports with libm's expf (WarpFields.cpp:101-112).
 * The host reference-stream path (OFDG_SAMPLER_REF, realize.cpp) keeps calling the host libm, so it
 * stays bit-identical to the reference on the same machine.  The device counter-sampler path has no
 * reference bit stream to reproduce (its random numbers are Philox, not the 45 mt19937 streams), but
 * it must be reproducible anywhere and checkable bit for bit: the GPU's libm (ocml) and glibc differ
 * in the last bit on a few arguments in a thousand, which flips rasteriser cells.  So that path is
 * defined with the functions below: fp64 +, -, * only (no FMA: every translation unit that includes
 * this header is compiled with -ffp-contract=off), therefore bit-identical on gfx950 and on any IEEE
 * host.  Accuracy: < 1 ULP (sin, cos; checked against libm in tests/test_host_logic.py), expf is the
 * fp64 result rounded once (equal to the correctly rounded float except within 2^-50 of a tie).
 *
 * Algorithms: Cody-Waite three-part pi/2 reduction + the classical degree-13/14 minimax kernels on
 * [-pi/4, pi/4] (fdlibm's k_sin / k_cos coefficients); exp by ln2 reduction + degree-11 Taylor-minimax.
 */
#ifndef OFDG_DETMATH_H_
#define OFDG_DETMATH_H_

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define OFDG_DM_FN __host__ __device__ static inline
#else
#define OFDG_DM_FN static inline
#endif

/* sin and cos of a (|a| < 2^19 * pi/2; larger or non-finite arguments return NaN). */
OFDG_DM_FN void ofdg_det_sincos(double a, double* s_out, double* c_out) {
  const double invpio2 = 6.36619772367581382433e-01;
  const double pio2_1 = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
  const double pio2_2 = 6.07710050630396597660e-11;  /* second 33 bits */
  const double pio2_3 = 2.02226624871116645580e-21;  /* third 33 bits */
  const double pio2_3t = 8.47842766036889956997e-32; /* pi/2 - (pio2_1 + pio2_2 + pio2_3) */
  const double big = 6755399441055744.0;             /* 1.5 * 2^52: adding it rounds to an integer */
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double ax = a < 0 ? -a : a;
  if (!(ax < 823549.6)) {  /* 2^19 * pi/2 (also catches NaN) */
    const double nan = (a - a) / (a - a);
    *s_out = nan; *c_out = nan;
    return;
  }
  /* n = nearest integer to a * 2/pi */
  const double fn = (a * invpio2 + big) - big;
  /* r + rt = a - fn * pi/2, head and tail */
  const double t = a - fn * pio2_1;  /* fn * pio2_1 is exact (33 x 20 bits) */
  const double w = fn * pio2_2;      /* exact */
  const double r = t - w;
  const double bb = r - t;           /* two-sum error of r = t - w */
  const double e = (t - (r - bb)) + (-w - bb);
  const double w3 = fn * pio2_3;
  const double y0 = r - w3;
  const double y1 = (((r - y0) - w3) + e) - fn * pio2_3t;
  /* kernels on [-pi/4, pi/4] */
  const double z = y0 * y0;
  double ks, kc;
  {
    const double v = z * y0;
    const double p = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    ks = y0 - ((z * (0.5 * y1 - v * p) - y1) - v * S1);
  }
  {
    const double ww = z * z;
    const double p = z * (C1 + z * (C2 + z * C3)) + (ww * ww) * (C4 + z * (C5 + z * C6));
    const double hz = 0.5 * z;
    const double om = 1.0 - hz;
    kc = om + (((1.0 - om) - hz) + (z * p - y0 * y1));
  }
  const int n = (int)(long long)fn & 3;
  *s_out = (n == 0) ? ks : (n == 1) ? kc : (n == 2) ? -ks : -kc;
  *c_out = (n == 0) ? kc : (n == 1) ? -ks : (n == 2) ? -kc : ks;
}
OFDG_DM_FN double ofdg_det_sin(double a) { double s, c; ofdg_det_sincos(a, &s, &c); return s; }
OFDG_DM_FN double ofdg_det_cos(double a) { double s, c; ofdg_det_sincos(a, &s, &c); return c; }

/* exp(x) in fp64 for |x| <= 745, relative error < 2^-52; 0 below, +inf above, NaN for NaN. */
OFDG_DM_FN double ofdg_det_exp(double x) {
  const double ln2_hi = 6.93147180369123816490e-01; /* 33 bits */
  const double ln2_lo = 1.90821492927058770002e-10;
  const double inv_ln2 = 1.44269504088896338700e+00;
  const double big = 6755399441055744.0;
  if (x != x) return x;
  if (x > 709.78) { const uint64_t ib = 0x7FF0000000000000ull; double inf; memcpy(&inf, &ib, 8); return inf; }
  if (x < -745.2) return 0.0;
  const double fk = (x * inv_ln2 + big) - big;
  const double r = (x - fk * ln2_hi) - fk * ln2_lo;  /* |r| <= ln2/2 */
  /* exp(r) = 1 + r + r^2/2! + ... + r^13/13!  (|r|^14/14! < 2^-58) */
  double p = 1.0 / 6227020800.0;
  p = p * r + 1.0 / 479001600.0;
  p = p * r + 1.0 / 39916800.0;
  p = p * r + 1.0 / 3628800.0;
  p = p * r + 1.0 / 362880.0;
  p = p * r + 1.0 / 40320.0;
  p = p * r + 1.0 / 5040.0;
  p = p * r + 1.0 / 720.0;
  p = p * r + 1.0 / 120.0;
  p = p * r + 1.0 / 24.0;
  p = p * r + 1.0 / 6.0;
  p = p * r + 0.5;
  p = p * r + 1.0;
  p = p * r + 1.0;
  /* scale by 2^k in two steps (k in [-1075, 1024]) through the exponent field */
  const int k = (int)(long long)fk;
  const int k1 = k / 2, k2 = k - k1;
  uint64_t b1 = (uint64_t)(int64_t)(k1 + 1023) << 52, b2 = (uint64_t)(int64_t)(k2 + 1023) << 52;
  double s1, s2;
  memcpy(&s1, &b1, 8);
  memcpy(&s2, &b2, 8);
  return (p * s1) * s2;
}
/* expf as the reference's Gaussian supports use it: the fp64 value rounded once to float. */
OFDG_DM_FN float ofdg_det_expf(float x) { return (float)ofdg_det_exp((double)x); }

#endif /* OFDG_DETMATH_H_ */
