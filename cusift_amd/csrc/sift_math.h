/*
 * sift_math.h -- the five transcendental functions of the SIFT extraction path, written out.
 *
 * The reference's device code calls CUDA's libm (expf, exp2f, atan2f, sinf, cosf: cuSIFT_D.cu:209-210,233,
 * 330,349,507).  CUDA's implementations are not available here, and the two libms that are -- glibc on the
 * host, OCML on gfx950 -- differ from it and from each other in the last bits.  Downstream of those bits sit
 * hard decisions (a 1/256 step of the texture-fraction model, a histogram bin edge), so "a few ulp" used to
 * become "a few keypoints per image that differ visibly" between the HIP kernels and the CPU oracle.
 *
 * This header fixes ONE evaluation of each function -- IEEE-754 single/double operations in a fixed order
 * (+, -, *, /, fma, round-to-nearest-even, comparisons, bit moves; nothing that a compiler may approximate)
 * -- and is compiled into BOTH the gfx950 kernels (hipcc, device code) and the CPU oracle (gcc, C99), both
 * with -ffp-contract=off.  Same operations on the same operands => the same bits on both sides, for every
 * input including denormals, infinities and NaN (tests/test_math.py checks the accuracy against float64 on the
 * CPU, tests/test_gpu_parity.py::test_math_device_equals_host checks bit identity on the device).
 *
 * Accuracy (measured, tests/test_math.py): sm_expf, sm_exp2f <= 1 ulp; sm_atan2f <= 2 ulp;
 * sm_sincosf <= 2 ulp below |x| = 128 (float Cody-Waite reduction), <= 1 ulp up to 1e9 (double precision).  Coefficients: tools/fit_math_polys.py (weighted Lawson iteration, then rounded to float).
 *
 * Plain C99 / C++ / HIP.  No state, no tables.
 */
#ifndef CUSIFT_SIFT_MATH_H
#define CUSIFT_SIFT_MATH_H

#if defined(__HIPCC__) || defined(__HIP__)
#define SM_FN static __host__ __device__ __forceinline__
#else
#define SM_FN static inline
#endif

SM_FN unsigned int sm_bits(float f) {
  unsigned int u;
  __builtin_memcpy(&u, &f, 4);
  return u;
}
SM_FN float sm_float(unsigned int u) {
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
}
SM_FN float sm_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
SM_FN float sm_abs(float x) { return sm_float(sm_bits(x) & 0x7fffffffu); }

/* p * 2^n for n in [-252, 254], p finite: two exact power-of-two factors, so the only rounding is the final one
 * (also when the result is denormal). */
SM_FN float sm_scale2(float p, int n) {
  const int n1 = n >> 1, n2 = n - n1; /* each in [-126, 127] */
  const float s1 = sm_float((unsigned int)(n1 + 127) << 23);
  const float s2 = sm_float((unsigned int)(n2 + 127) << 23);
  return (p * s1) * s2;
}

/* e^x.  n = rne(x * log2 e), r = x - n ln2 (two-constant Cody-Waite, |r| <= 0.3467),
 * e^r = 1 + r + r^2 P(r) with P of degree 5. */
SM_FN float sm_expf(float x) {
  if (!(x > -104.0f)) return (x == x) ? 0.0f : x; /* underflow to +0; NaN stays NaN */
  if (x > 88.8f) return sm_float(0x7f800000u);     /* overflow */
  const float n = __builtin_rintf(x * 1.4426950216293335f);
  float r = sm_fma(n, -0.693115234375f, x);            /* ln2 high part: 12 significant bits, n * it is exact */
  r = sm_fma(n, -3.194618329871446e-05f, r);           /* ln2 low part */
  float p = 0.0001979032385861501f;
  p = sm_fma(p, r, 0.0013944690581411123f);
  p = sm_fma(p, r, 0.008333497680723667f);
  p = sm_fma(p, r, 0.04166629537940025f);
  p = sm_fma(p, r, 0.1666666567325592f);
  p = sm_fma(p, r, 0.5f);
  const float r2 = r * r;
  p = sm_fma(p, r2, r);
  p = p + 1.0f;
  return sm_scale2(p, (int)n);
}

/* 2^x.  n = rne(x), r = x - n exactly, 2^r = 1 + r G(r) with G of degree 6. */
SM_FN float sm_exp2f(float x) {
  if (!(x > -150.0f)) return (x == x) ? 0.0f : x;
  if (x >= 128.0f) return sm_float(0x7f800000u);
  const float n = __builtin_rintf(x);
  const float r = x - n;
  float g = 1.5196151252894197e-05f;
  g = sm_fma(g, r, 0.00015466756303794682f);
  g = sm_fma(g, r, 0.0013333935057744384f);
  g = sm_fma(g, r, 0.009618038311600685f);
  g = sm_fma(g, r, 0.055504102259874344f);
  g = sm_fma(g, r, 0.24022650718688965f);
  g = sm_fma(g, r, 0.6931471824645996f);
  const float p = sm_fma(g, r, 1.0f);
  return sm_scale2(p, (int)n);
}

/* atan2(y, x) with the IEEE-754 / C99 special cases: atan2(+-0, x<0 or -0) = +-pi, atan2(+-0, x>0 or +0) = +-0,
 * infinities as limits, NaN in => NaN out.  (atan2f(+0, negative) == (float)pi exactly: the descriptor's
 * angle-index-8 path depends on it, cuSIFT_D.cu:233-235.)
 * a = min(|x|,|y|) / max(|x|,|y|) in [0, 1] (IEEE division), atan a = a + a s Q(s), s = a^2, Q of degree 7. */
SM_FN float sm_atan2f(float y, float x) {
  const float ax = sm_abs(x), ay = sm_abs(y);
  const int swap = ay > ax;
  const float mx = swap ? ay : ax;
  const float mn = swap ? ax : ay;
  float a = mn / mx;
  if (ax == ay) a = (ax == 0.0f) ? 0.0f : 1.0f; /* 0/0 and inf/inf; finite equal values give 1 anyway */
  const float s = a * a;
  float q = 0.0029205884784460068f;
  q = sm_fma(q, s, -0.016367513686418533f);
  q = sm_fma(q, s, 0.0432111918926239f);
  q = sm_fma(q, s, -0.07552158087491989f);
  q = sm_fma(q, s, 0.10665978491306305f);
  q = sm_fma(q, s, -0.14211048185825348f);
  q = sm_fma(q, s, 0.19993771612644196f);
  q = sm_fma(q, s, -0.33333152532577515f);
  const float t = q * s;
  float r = sm_fma(t, a, a);
  if (swap) r = (1.5707963705062866f - r) + -4.371138828673793e-08f;          /* pi/2 = hi + lo */
  if (sm_bits(x) & 0x80000000u) r = (3.1415927410125732f - r) + -8.742277657347586e-08f; /* pi = hi + lo */
  return sm_float(sm_bits(r) | (sm_bits(y) & 0x80000000u));
}

/* sin x and cos x for |x| >= 128: argument reduction and the two polynomials in DOUBLE precision
 * (n = rne(x 2/pi), r = x - n pi/2 with a two-constant pi/2, |r| <= pi/4; Taylor polynomials to r^15 / r^14:
 * truncation error < 1e-15), rounded to float at the end.  |x| >= 1e9, infinities and NaN give NaN.
 * Never taken by the extraction drivers (orientations lie in [0, 360) degrees => x in [0, 6.3)); kept out of line so
 * that its double-precision temporaries do not weigh on the register allocation of the keypoint kernels. */
#if defined(__HIPCC__) || defined(__HIP__)
static __host__ __device__ __attribute__((noinline)) unsigned long long sm_sincosf_large(float xf) {
#else
static __attribute__((noinline)) unsigned long long sm_sincosf_large(float xf) {
#endif
  /* returns (bits of cos << 32) | bits of sin: by value, so that the out-of-line call needs no stack */
  const double x = (double)xf;
  if (!(__builtin_fabs(x) < 1.0e9)) return 0x7fc000007fc00000ull;
  const double n = __builtin_rint(x * 0.63661977236758134308);
  double r = __builtin_fma(n, -1.57079632679489655800e+00, x);
  r = __builtin_fma(n, -6.12323399573676603587e-17, r);
  const double z = r * r;
  double ps = -7.6471637318198164759e-13; /* -1/15! */
  ps = __builtin_fma(ps, z, 1.6059043836821614599e-10);  /*  1/13! */
  ps = __builtin_fma(ps, z, -2.5052108385441718775e-08); /* -1/11! */
  ps = __builtin_fma(ps, z, 2.7557319223985890653e-06);  /*  1/9!  */
  ps = __builtin_fma(ps, z, -1.9841269841269841270e-04); /* -1/7!  */
  ps = __builtin_fma(ps, z, 8.3333333333333333333e-03);  /*  1/5!  */
  ps = __builtin_fma(ps, z, -1.6666666666666666667e-01); /* -1/3!  */
  const double s = __builtin_fma(ps * z, r, r);
  double pc = -1.1470745597729724714e-11; /* -1/14! */
  pc = __builtin_fma(pc, z, 2.0876756987868098979e-09);  /*  1/12! */
  pc = __builtin_fma(pc, z, -2.7557319223985890653e-07); /* -1/10! */
  pc = __builtin_fma(pc, z, 2.4801587301587301587e-05);  /*  1/8!  */
  pc = __builtin_fma(pc, z, -1.3888888888888888889e-03); /* -1/6!  */
  pc = __builtin_fma(pc, z, 4.1666666666666666667e-02);  /*  1/4!  */
  pc = __builtin_fma(pc, z, -0.5);
  const double c = __builtin_fma(pc, z, 1.0);
  const int q = (int)n & 3; /* |n| < 6.4e8 */
  const double ss = (q & 1) ? c : s;
  const double cc = (q & 1) ? s : c;
  const float sf = (float)((q & 2) ? -ss : ss);
  const float cf = (float)(((q + 1) & 2) ? -cc : cc);
  return ((unsigned long long)sm_bits(cf) << 32) | sm_bits(sf);
}

/* sin x and cos x.  |x| < 128: n = rne(x 2/pi) (|n| <= 82), r = x - n pi/2 by three fused multiply-adds with a
 * three-float pi/2 (each product is exact inside the fma), sin r = r + r^3 S(r^2), cos r = 1 - r^2/2 + r^4 C(r^2)
 * with S, C of degree 3 on |r| <= pi/4; quadrant from n.  Otherwise (incl. NaN): sm_sincosf_large. */
SM_FN void sm_sincosf(float x, float *sn, float *cs) {
  if (!(sm_abs(x) < 128.0f)) {
    const unsigned long long both = sm_sincosf_large(x);
    *sn = sm_float((unsigned int)both);
    *cs = sm_float((unsigned int)(both >> 32));
    return;
  }
  const float n = __builtin_rintf(x * 0.6366197466850281f);
  float r = sm_fma(n, -1.5707963705062866f, x);
  r = sm_fma(n, 4.371138828673793e-08f, r);
  r = sm_fma(n, 1.7151245100058819e-15f, r);
  const float z = r * r;
  float ps = 2.7171588499186328e-06f;
  ps = sm_fma(ps, z, -0.0001983921101782471f);
  ps = sm_fma(ps, z, 0.008333329111337662f);
  ps = sm_fma(ps, z, -0.1666666716337204f);
  const float s = sm_fma(ps * z, r, r);
  float pc = -2.7196659857509076e-07f;
  pc = sm_fma(pc, z, 2.479934846633114e-05f);
  pc = sm_fma(pc, z, -0.0013888883404433727f);
  pc = sm_fma(pc, z, 0.0416666679084301f);
  const float c = sm_fma(pc * z, z, sm_fma(-0.5f, z, 1.0f));
  const int q = (int)n & 3;
  const float ss = (q & 1) ? c : s;
  const float cc = (q & 1) ? s : c;
  *sn = (q & 2) ? -ss : ss;
  *cs = ((q + 1) & 2) ? -cc : cc;
}

#endif /* CUSIFT_SIFT_MATH_H */
