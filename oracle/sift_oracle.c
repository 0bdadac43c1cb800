/*
 * sift_oracle.c -- CPU restatement of the cuSIFT extraction hot path (see sift_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: the checker, never the thing measured or shipped.
 *
 * Arithmetic convention (all float32 unless noted).  The reference's device code was built by
 * nvcc, whose default (-fmad=true) contracts a*b+c into one fused multiply-add.  The filter
 * sums (ScaleDown, LaplaceMulti) are therefore restated as "first product, then a left-to-right
 * fmaf chain"; everything else is evaluated operation by operation with no contraction
 * (compile with -ffp-contract=off).  The same convention is used by the HIP kernels so that
 * the filter stages are bit-identical between oracle and device; the device code's transcendental
 * functions are shared with the kernels as well (sift_math.h, below).
 *
 * Build-time switch ORACLE_NO_FMA evaluates the filter sums without fusion (used once to
 * check the convention against the golden file).
 */
#include "sift_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* The transcendental functions of the DEVICE code (expf, exp2f, atan2f, sinf, cosf in cuSIFT_D.cu).  CUDA's libm is
 * not available; glibc's and the GPU's (OCML) differ from it and from each other in the last bits, and a last bit
 * occasionally decides a texture-fraction step or a histogram bin.  So one written-out evaluation of each function
 * (cusift_amd/csrc/sift_math.h: IEEE operations in a fixed order) is compiled into this oracle AND into the HIP
 * kernels; what remains between the two is then only the descriptor's summation order.  -DORACLE_LIBM builds the
 * same restatement on glibc's functions instead (libsift_oracle_libm.so): tests/test_oracle_golden.py pins BOTH
 * builds to the reference's golden file with the same gates, and tests/test_math.py bounds the written-out functions
 * against float64 -- that is what keeps this oracle tied to the reference rather than to the product.
 * The HOST-side tap tables (cuSIFT.cu:331,406: expf/powf in host code) use the host libm, as the reference did. */
#include "../cusift_amd/csrc/sift_math.h"
#ifdef ORACLE_LIBM
#define dev_expf expf
#define dev_exp2f exp2f
#define dev_atan2f atan2f
static inline void dev_sincosf(float x, float *s, float *c) {
  *s = sinf(x);
  *c = cosf(x);
}
#else
#define dev_expf sm_expf
#define dev_exp2f sm_exp2f
#define dev_atan2f sm_atan2f
#define dev_sincosf sm_sincosf
#endif

/* the written-out functions, exported for tests/test_math.py (accuracy) and the device bit-identity test */
float oracle_math_expf(float x) { return sm_expf(x); }
float oracle_math_exp2f(float x) { return sm_exp2f(x); }
float oracle_math_atan2f(float y, float x) { return sm_atan2f(y, x); }
void oracle_math_sincosf(float x, float *s, float *c) { sm_sincosf(x, s, c); }
/* array form: op 0 expf(a), 1 exp2f(a), 2 atan2f(a, b), 3 sincosf(a) -> (out, out2) */
void oracle_math_eval(int op, const float *a, const float *b, float *out, float *out2, int n) {
  for (int i = 0; i < n; i++) {
    if (op == 0) out[i] = sm_expf(a[i]);
    else if (op == 1) out[i] = sm_exp2f(a[i]);
    else if (op == 2) out[i] = sm_atan2f(a[i], b[i]);
    else sm_sincosf(a[i], out + i, out2 + i);
  }
}

#define NUM_SCALES 5 /* cuSIFT_D.h:8  */
#define LAPLACE_S 8  /* cuSIFT_D.h:23 (NUM_SCALES + 3) */
#define LAPLACE_R 4  /* cuSIFT_D.h:26 */

#ifdef ORACLE_NO_FMA
static inline float fma_(float a, float b, float c) { return a * b + c; }
#else
static inline float fma_(float a, float b, float c) { return fmaf(a, b, c); }
#endif

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------------------------------
 * ScaleDown: cuSIFT.cu:320-341 (taps) + cuSIFT_D.cu:37-182 (kernel).
 * Horizontal (cuSIFT_D.cu:111-113): out col c <- src cols clamp(2c-2 .. 2c+2), weights k0 k1 k2 k1 k0.
 * Vertical (cuSIFT_D.cu:75,123-125,144,155,166,177): the ring of 5 row buffers is read with
 * yRead = yStart + tx - 1, so out row r <- rows clamp(2r-1, 2r, 2r+1, 2r+2, 2r+3) with weights
 * (k1, k2, k1, k0, k0): there is no -2 tap and both +2 and +3 get k0.
 * ---------------------------------------------------------------------------------------- */
static void scale_down_taps(float k[5]) {
  const float variance = 0.5f; /* cuSIFT.cu:185 */
  float sum = 0.0f;
  for (int j = 0; j < 5; j++) {
    k[j] = (float)expf(-(double)(j - 2) * (j - 2) / 2.0 / variance);
    sum += k[j];
  }
  for (int j = 0; j < 5; j++) k[j] /= sum;
}

void oracle_scale_down(const float *src, int w, int h, int src_pitch, float *dst, int dst_pitch) {
  float k[5];
  scale_down_taps(k);
  const int ow = w / 2, oh = h / 2;
  float *rows = (float *)malloc(sizeof(float) * 5 * (size_t)(ow > 0 ? ow : 1));
  for (int r = 0; r < oh; r++) {
    /* rows[t] = horizontally filtered source row clamp(2r - 1 + t), t = 0..4 */
    for (int t = 0; t < 5; t++) {
      const float *s = src + (size_t)clampi(2 * r - 1 + t, 0, h - 1) * src_pitch;
      float *b = rows + (size_t)t * ow;
      for (int c = 0; c < ow; c++) {
        float a0 = s[clampi(2 * c - 2, 0, w - 1)], a1 = s[clampi(2 * c - 1, 0, w - 1)];
        float a2 = s[clampi(2 * c, 0, w - 1)], a3 = s[clampi(2 * c + 1, 0, w - 1)];
        float a4 = s[clampi(2 * c + 2, 0, w - 1)];
        /* k[0]*(in[2tx]+in[2tx+4]) + k[1]*(in[2tx+1]+in[2tx+3]) + k[2]*in[2tx+2] */
        float v = k[0] * (a0 + a4);
        v = fma_(k[1], a1 + a3, v);
        v = fma_(k[2], a2, v);
        b[c] = v;
      }
    }
    const float *bm1 = rows, *b0 = rows + ow, *bp1 = rows + 2 * (size_t)ow, *bp2 = rows + 3 * (size_t)ow,
                *bp3 = rows + 4 * (size_t)ow;
    float *d = dst + (size_t)r * dst_pitch;
    for (int c = 0; c < ow; c++) {
      /* k[2]*brow[centre] + k[0]*(brow[+3]+brow[+2]) + k[1]*(brow[-1]+brow[+1]) */
      float v = k[2] * b0[c];
      v = fma_(k[0], bp3[c] + bp2[c], v);
      v = fma_(k[1], bm1[c] + bp1[c], v);
      d[c] = v;
    }
  }
  free(rows);
}

/* ------------------------------------------------------------------------------------------
 * LaplaceMulti taps: cuSIFT.cu:239-240 (baseBlur, diffScale), cuSIFT.cu:400-412 (table).
 * ---------------------------------------------------------------------------------------- */
void oracle_laplace_taps(float init_blur, float taps[8 * 16]) {
  const float baseBlur = powf(2.0f, -1.0f / NUM_SCALES);
  const float diffScale = powf(2.0f, 1.0f / NUM_SCALES);
  float scale = baseBlur;
  memset(taps, 0, sizeof(float) * 8 * 16);
  for (int i = 0; i < LAPLACE_S; i++) {
    float kernelSum = 0.0f;
    float var = scale * scale - init_blur * init_blur;
    float *k = taps + 16 * i;
    if (var <= 1e-6f) {
      /* Rule of this build: the image already carries at least this much blur -> identity.
       * (The reference divides by var: NaN taps at var == 0, an inverted kernel at var < 0.) */
      k[LAPLACE_R] = 1.0f;
    } else {
      for (int j = -LAPLACE_R; j <= LAPLACE_R; j++) {
        k[j + LAPLACE_R] = (float)expf(-(double)j * j / 2.0 / var);
        kernelSum += k[j + LAPLACE_R];
      }
      for (int j = -LAPLACE_R; j <= LAPLACE_R; j++) k[j + LAPLACE_R] /= kernelSum;
    }
    scale *= diffScale;
  }
}

/* One 9-tap pass in the reference's order (cuSIFT_D.cu:536-540 / 544-548):
 * k[4]*c + k[3]*(m1+p1) + k[2]*(m2+p2) + k[1]*(m3+p3) + k[0]*(m4+p4) */
static inline float tap9(const float *k, float c, float m1, float p1, float m2, float p2, float m3, float p3,
                         float m4, float p4) {
  float v = k[4] * c;
  v = fma_(k[3], m1 + p1, v);
  v = fma_(k[2], m2 + p2, v);
  v = fma_(k[1], m3 + p3, v);
  v = fma_(k[0], m4 + p4, v);
  return v;
}

/* LaplaceMulti_D, cuSIFT_D.cu:525-553.  Texture coordinates (xp-3.5, yp+0.5+-k) hit texel centres,
 * so the fetches are exact reads with clamp addressing: vertical pass first (into sdata1), then
 * horizontal (sdata2), then DoG_s = L_s - L_{s+1} for s = 0..6; columns >= width are not written. */
void oracle_laplace_multi(const float *img, int w, int h, int pitch, float init_blur, float *dog) {
  float taps[8 * 16];
  oracle_laplace_taps(init_blur, taps);
  float *V = (float *)malloc(sizeof(float) * LAPLACE_S * (size_t)w);
  float *L = (float *)malloc(sizeof(float) * LAPLACE_S * (size_t)w);
  const size_t plane = (size_t)h * pitch;
  for (int y = 0; y < h; y++) {
    const float *r0 = img + (size_t)y * pitch;
    const float *rm[5], *rp[5];
    for (int k = 1; k <= 4; k++) {
      rm[k] = img + (size_t)clampi(y - k, 0, h - 1) * pitch;
      rp[k] = img + (size_t)clampi(y + k, 0, h - 1) * pitch;
    }
    for (int s = 0; s < LAPLACE_S; s++) {
      const float *k = taps + 16 * s;
      float *v = V + (size_t)s * w;
      for (int x = 0; x < w; x++)
        v[x] = tap9(k, r0[x], rm[1][x], rp[1][x], rm[2][x], rp[2][x], rm[3][x], rp[3][x], rm[4][x], rp[4][x]);
    }
    for (int s = 0; s < LAPLACE_S; s++) {
      const float *k = taps + 16 * s;
      const float *v = V + (size_t)s * w;
      float *l = L + (size_t)s * w;
      for (int x = 0; x < w; x++) {
#define VX(i) v[clampi((i), 0, w - 1)]
        /* sdata1[tx+4] is column x; sdata1[tx+3]/[tx+5] are x-1/x+1 ... (cuSIFT_D.cu:544-548) */
        l[x] = tap9(k, v[x], VX(x - 1), VX(x + 1), VX(x - 2), VX(x + 2), VX(x - 3), VX(x + 3), VX(x - 4), VX(x + 4));
#undef VX
      }
    }
    for (int s = 0; s < LAPLACE_S - 1; s++) {
      float *d = dog + (size_t)s * plane + (size_t)y * pitch;
      const float *a = L + (size_t)s * w, *b = L + (size_t)(s + 1) * w;
      for (int x = 0; x < w; x++) d[x] = a[x] - b[x];
    }
  }
  free(V);
  free(L);
}

/* ------------------------------------------------------------------------------------------
 * FindPointsMulti: cuSIFT.cu:424-455 (constants) + cuSIFT_D.cu:402-523 (kernel).
 * ---------------------------------------------------------------------------------------- */
void oracle_find_points_multi(const float *dog, int w, int h, int pitch, float peak_thresh, float edge_thresh,
                              float subsampling, oracle_sift_point *points, int max_pts, int *counter) {
  /* cuSIFT.cu:239-247: sigma = baseBlur*diffScale, factor = 1/NUM_SCALES; cuSIFT.cu:432-438 */
  const float baseBlur = powf(2.0f, -1.0f / NUM_SCALES);
  const float diffScale0 = powf(2.0f, 1.0f / NUM_SCALES);
  const double sigma = baseBlur * diffScale0;
  const float factor = 1.0f / NUM_SCALES;
  float scales[NUM_SCALES];
  {
    float scale = (float)sigma;
    const float diffScale = powf(2.0f, factor);
    for (int i = 0; i < NUM_SCALES; i++) {
      scales[i] = scale;
      scale *= diffScale;
    }
  }
  const float thr_pos = peak_thresh, thr_neg = -peak_thresh;
  const size_t size = (size_t)pitch * h;

  for (int y = 0; y < h; y++) {
    const int y0 = y - 1 < 0 ? 0 : y - 1, y2 = y + 1 > h - 1 ? h - 1 : y + 1;
    for (int x = 0; x < w; x++) {
      const int x0 = x - 1 < 0 ? 0 : x - 1, x2 = x + 1 > w - 1 ? w - 1 : x + 1;
      for (int s = 0; s < NUM_SCALES; s++) {
        const float *P = dog + size * s, *C = dog + size * (s + 1), *Q = dog + size * (s + 2);
        const float val = C[(size_t)y * pitch + x];
        /* cuSIFT_D.cu:451-470: strict test against the 26 neighbours (clamped addressing makes
         * border pixels compare against themselves and fail). fminf/fmaxf as in the kernel. */
        int is_min = 0, is_max = 0;
        if (val < thr_neg || val > thr_pos) {
          float mn = INFINITY, mx = -INFINITY;
          const int xs[3] = {x0, x, x2}, ys[3] = {y0, y, y2};
          for (int p = 0; p < 3; p++) {
            const float *pl = p == 0 ? P : (p == 1 ? C : Q);
            for (int j = 0; j < 3; j++)
              for (int i = 0; i < 3; i++) {
                if (p == 1 && i == 1 && j == 1) continue;
                float v = pl[(size_t)ys[j] * pitch + xs[i]];
                mn = fminf(mn, v);
                mx = fmaxf(mx, v);
              }
          }
          is_min = (val < thr_neg) && (val < mn);
          is_max = (val > thr_pos) && (val > mx);
        }
        if (!is_min && !is_max) continue;

        /* cuSIFT_D.cu:478-521 */
        const float *d1 = C + (size_t)y * pitch + x;
        float dxx = 2.0f * val - d1[-1] - d1[1];
        float dyy = 2.0f * val - d1[-pitch] - d1[pitch];
        float dxy = 0.25f * (d1[pitch + 1] + d1[-pitch - 1] - d1[-pitch + 1] - d1[pitch - 1]);
        float tra = dxx + dyy;
        float det = dxx * dyy - dxy * dxy;
        if (!(tra * tra < edge_thresh * det)) continue;
        float edge = (tra * tra) / det;
        float dx = 0.5f * (d1[1] - d1[-1]);
        float dy = 0.5f * (d1[pitch] - d1[-pitch]);
        const float *d0 = P + (size_t)y * pitch + x;
        const float *d2 = Q + (size_t)y * pitch + x;
        float ds = 0.5f * (d0[0] - d2[0]);
        float dss = 2.0f * val - d2[0] - d0[0];
        float dxs = 0.25f * (d2[1] + d0[-1] - d0[1] - d2[-1]);
        float dys = 0.25f * (d2[pitch] + d0[-pitch] - d2[-pitch] - d0[pitch]);
        float idxx = dyy * dss - dys * dys;
        float idxy = dys * dxs - dxy * dss;
        float idxs = dxy * dys - dyy * dxs;
        float idet = 1.0f / (idxx * dxx + idxy * dxy + idxs * dxs);
        float idyy = dxx * dss - dxs * dxs;
        float idys = dxy * dxs - dxx * dys;
        float idss = dxx * dyy - dxy * dxy;
        float pdx = idet * (idxx * dx + idxy * dy + idxs * ds);
        float pdy = idet * (idxy * dx + idyy * dy + idys * ds);
        float pds = idet * (idxs * dx + idys * dy + idss * ds);
        if (pdx < -0.5f || pdx > 0.5f || pdy < -0.5f || pdy > 0.5f || pds < -0.5f || pds > 0.5f) {
          pdx = dx / dxx;
          pdy = dy / dyy;
          pds = ds / dss;
        }
        float dval = 0.5f * (dx * pdx + dy * pdy + ds * pds);
        /* A point that is both a strict min and a strict max cannot exist; one append. */
        int idx = (*counter)++;
        if (idx >= max_pts) continue; /* reference clamps onto slot max_pts-1 (racy); we drop */
        oracle_sift_point *pt = points + idx;
        pt->coords2D[0] = x + pdx;
        pt->coords2D[1] = y + pdy;
        pt->scale = scales[s] * dev_exp2f(pds * factor);
        pt->sharpness = val + dval;
        pt->edgeness = edge;
        pt->subsampling = subsampling;
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * Texture model (CUDA programming guide, linear filtering, unnormalised coordinates, clamp):
 * xB = x - 0.5, i = floor(xB), alpha = frac(xB) kept with `frac_bits` fractional bits (8 on NVIDIA hardware),
 * T = w00 S[j][i] + w10 S[j][i+1] + w01 S[j+1][i] + w11 S[j+1][i+1].
 * The four weights are FIXED-POINT numbers of `frac_bits` bits as well (round 6; measured on the reference's golden
 * pair, tests/test_oracle_golden.py): with A = round(alpha 2^q), B = round(beta 2^q),
 *     W11 = floor(A B / 2^q + 1/2),  W10 = A - W11,  W01 = B - W11,  W00 = 2^q - A - B + W11,   w = W / 2^q
 * i.e. the product alpha*beta is rounded (half up) to q bits and the other three follow by subtraction, so the weights
 * always sum to one.  With the exact 16-bit products (1-a)(1-b), a(1-b), (1-a)b, ab that rounds 1 to 5 used, 35 % of
 * the 4,095 matched golden orientations were within 1e-3 degree (median 0.002); with this rule all 4,095 are within
 * 6.1e-5 degree and 86 % are bit-identical floats (round-half-even: 99.0 % within 1e-3, truncation: 53 %; rounding
 * each weight on its own: 98.3 %).  frac_bits = 0 keeps exact fp32 fractions and products.
 * The kernels pass pixel-index coordinates with no +0.5 (cuSIFT_D.cu:207-212,337-338), hence
 * every tap is displaced by (-1/2,-1/2) px; that displacement is part of the reference's results.
 * ---------------------------------------------------------------------------------------- */
float oracle_tex2d(const float *img, int w, int h, int pitch, float x, float y, int frac_bits) {
  float xb = x - 0.5f, yb = y - 0.5f;
  float fx = floorf(xb), fy = floorf(yb);
  float a = xb - fx, b = yb - fy;
  float w00, w10, w01, w11;
  if (frac_bits > 0 && frac_bits <= 11) { /* the range the C ABI accepts (cusift_params.tex_frac_bits); else exact */
    const float q = (float)(1 << frac_bits);
    const float A = floorf(a * q + 0.5f), B = floorf(b * q + 0.5f); /* integers in [0, q] */
    const float W11 = floorf(A * B / q + 0.5f);                     /* A B <= 2^(2q): exact in fp32 for q <= 11 */
    w11 = W11 / q;
    w10 = (A - W11) / q;
    w01 = (B - W11) / q;
    w00 = (q - A - B + W11) / q;
  } else {
    const float ia = 1.0f - a, ib = 1.0f - b;
    w00 = ia * ib, w10 = a * ib, w01 = ia * b, w11 = a * b;
  }
  /* clamp in float first so that huge |x| cannot overflow the int conversion */
  fx = fminf(fmaxf(fx, -1.0f), (float)w);
  fy = fminf(fmaxf(fy, -1.0f), (float)h);
  int i = (int)fx, j = (int)fy;
  int i0 = clampi(i, 0, w - 1), i1 = clampi(i + 1, 0, w - 1);
  int j0 = clampi(j, 0, h - 1), j1 = clampi(j + 1, 0, h - 1);
  float s00 = img[(size_t)j0 * pitch + i0], s10 = img[(size_t)j0 * pitch + i1];
  float s01 = img[(size_t)j1 * pitch + i0], s11 = img[(size_t)j1 * pitch + i1];
  /* interpolation order of this restatement: first product, then three fused multiply-adds
   * (the texture unit's own arithmetic is not documented; nvcc would fuse exactly like this) */
  float t = w00 * s00;
  t = fmaf(w10, s10, t);
  t = fmaf(w01, s01, t);
  t = fmaf(w11, s11, t);
  return t;
}

/* Diagnostic tap (tests/parity_utils.golden_gates): when a buffer is registered on the calling thread, every point bx
 * that oracle_compute_orientations processes leaves diag[2*bx] = second / first smoothed-histogram peak value and
 * diag[2*bx+1] = the orientation the SECOND peak would give (degrees; NaN when there is none).  The reference computes
 * both peaks and discards the second (`&& false`, cuSIFT_D.cu:380): a golden row that sits near our second peak is a
 * near-tie decided the other way by CUDA's arithmetic, not a modelling difference. */
static _Thread_local float *g_ori_diag = 0;
static _Thread_local int g_ori_diag_cap = 0;
void oracle_set_orientation_diag(float *buf, int n_points) {
  g_ori_diag = buf;
  g_ori_diag_cap = buf ? n_points : 0;
}

void oracle_compute_orientations(const float *img, int w, int h, int pitch, oracle_sift_point *points, int first,
                                 int last, int frac_bits) {
  for (int bx = first; bx < last; bx++) {
    oracle_sift_point *pt = points + bx;
    float hist[64];
    float hist_hi[32];
    float gauss[11];
    float i2sigma2 = -1.0f / (4.5f * pt->scale * pt->scale);
    for (int tx = 0; tx < 11; tx++) gauss[tx] = dev_expf(i2sigma2 * (tx - 5) * (tx - 5));
    for (int i = 0; i < 64; i++) hist[i] = 0.0f;
    for (int i = 0; i < 32; i++) hist_hi[i] = 0.0f;
    float xp = pt->coords2D[0] - 5.0f;
    float yp = pt->coords2D[1] - 5.0f;
    for (int tx = 0; tx < 121; tx++) { /* threads 121..127 have yd == 11 and skip */
      int yd = tx / 11;
      int xd = tx - yd * 11;
      float xf = xp + xd;
      float yf = yp + yd;
      float dx = oracle_tex2d(img, w, h, pitch, xf + 1.0f, yf, frac_bits) -
                 oracle_tex2d(img, w, h, pitch, xf - 1.0f, yf, frac_bits);
      float dy = oracle_tex2d(img, w, h, pitch, xf, yf + 1.0f, frac_bits) -
                 oracle_tex2d(img, w, h, pitch, xf, yf - 1.0f, frac_bits);
      int bin = (int)(16.0f * dev_atan2f(dy, dx) / 3.1416f + 16.5f);
      if (bin > 31 || bin < 0) bin = 0; /* < 0 only for non-finite input (memory safety) */
      float grad = sqrtf(dx * dx + dy * dy);
      /* reference: LDS float atomicAdd in arbitrary order.  This restatement (and the HIP kernel) fixes one:
       * samples 0..63 and 64..120 are summed separately in index order, then added. */
      if (tx < 64) hist[bin] += grad * gauss[xd] * gauss[yd];
      else hist_hi[bin] += grad * gauss[xd] * gauss[yd];
    }
    for (int i = 0; i < 32; i++) hist[i] = hist[i] + hist_hi[i];
    for (int tx = 0; tx < 32; tx++) {
      int x1m = (tx >= 1 ? tx - 1 : tx + 31);
      int x1p = (tx <= 30 ? tx + 1 : tx - 31);
      int x2m = (tx >= 2 ? tx - 2 : tx + 30);
      int x2p = (tx <= 29 ? tx + 2 : tx - 30);
      hist[tx + 32] = 6.0f * hist[tx] + 4.0f * (hist[x1m] + hist[x1p]) + (hist[x2m] + hist[x2p]);
    }
    for (int tx = 0; tx < 32; tx++) {
      int x1m = (tx >= 1 ? tx - 1 : tx + 31);
      int x1p = (tx <= 30 ? tx + 1 : tx - 31);
      float v = hist[32 + tx];
      hist[tx] = (v > hist[32 + x1m] && v >= hist[32 + x1p] ? v : 0.0f);
    }
    float maxval1 = 0.0f, maxval2 = 0.0f;
    int i1 = -1, i2 = -1;
    for (int i = 0; i < 32; i++) {
      float v = hist[i];
      if (v > maxval1) {
        maxval2 = maxval1;
        maxval1 = v;
        i2 = i1;
        i1 = i;
      } else if (v > maxval2) {
        maxval2 = v;
        i2 = i;
      }
    }
    if (g_ori_diag && bx < g_ori_diag_cap) {
      float o2 = NAN;
      if (i2 >= 0) {
        float w1 = hist[32 + ((i2 + 1) & 31)], w2 = hist[32 + ((i2 + 31) & 31)];
        float pk2 = i2 + 0.5f * (w1 - w2) / (2.0f * maxval2 - w1 - w2);
        o2 = 11.25f * (pk2 < 0.0f ? pk2 + 32.0f : pk2);
      }
      g_ori_diag[2 * bx] = maxval1 > 0.0f ? maxval2 / maxval1 : NAN;
      g_ori_diag[2 * bx + 1] = o2;
    }
    float val1 = hist[32 + ((i1 + 1) & 31)];
    float val2 = hist[32 + ((i1 + 31) & 31)];
    float peak = i1 + 0.5f * (val1 - val2) / (2.0f * maxval1 - val1 - val2);
    pt->orientation = 11.25f * (peak < 0.0f ? peak + 32.0f : peak);
  }
}

/* ------------------------------------------------------------------------------------------
 * ExtractSiftDescriptors_D: cuSIFT_D.cu:184-297.
 * Accumulation order: the device uses LDS float atomics (arbitrary order); the oracle walks
 * samples y-major (y = 0..15, tx = 0..15) and issues the 8 adds in source order.  Indices
 * outside [0,128) fall outside `buffer` on the device (into `sums`, overwritten before use)
 * and are dropped here.  rsqrtf(x) is restated as 1/sqrtf(x).
 * ---------------------------------------------------------------------------------------- */
static inline void desc_add(float *buffer, int idx, float v) {
  if (idx >= 0 && idx < 128) buffer[idx] += v;
}

void oracle_extract_descriptors(const float *img, int w, int h, int pitch, oracle_sift_point *points, int first,
                                int last, float subsampling, int frac_bits) {
  float gauss[16];
  for (int tx = 0; tx < 16; tx++) gauss[tx] = dev_expf(-(tx - 7.5f) * (tx - 7.5f) / 128.0f);
  for (int bx = first; bx < last; bx++) {
    oracle_sift_point *pt = points + bx;
    float buffer[128];
    for (int i = 0; i < 128; i++) buffer[i] = 0.0f;
    float theta = 2.0f * 3.1415f / 360.0f * pt->orientation;
    float sina, cosa;
    dev_sincosf(theta, &sina, &cosa);
    float scale = 12.0f / 16.0f * pt->scale;
    float ssina = scale * sina;
    float scosa = scale * cosa;
    for (int y = 0; y < 16; y++) {
      for (int tx = 0; tx < 16; tx++) {
        float xpos = pt->coords2D[0] + (tx - 7.5f) * scosa - (y - 7.5f) * ssina;
        float ypos = pt->coords2D[1] + (tx - 7.5f) * ssina + (y - 7.5f) * scosa;
        float dx = oracle_tex2d(img, w, h, pitch, xpos + cosa, ypos + sina, frac_bits) -
                   oracle_tex2d(img, w, h, pitch, xpos - cosa, ypos - sina, frac_bits);
        float dy = oracle_tex2d(img, w, h, pitch, xpos - sina, ypos + cosa, frac_bits) -
                   oracle_tex2d(img, w, h, pitch, xpos + sina, ypos - cosa, frac_bits);
        float grad = gauss[y] * gauss[tx] * sqrtf(dx * dx + dy * dy);
        float angf = 4.0f / 3.1415f * dev_atan2f(dy, dx) + 4.0f;

        int hori = (tx + 2) / 4 - 1;
        float horf = (tx - 1.5f) / 4.0f - hori;
        float ihorf = 1.0f - horf;
        int veri = (y + 2) / 4 - 1;
        float verf = (y - 1.5f) / 4.0f - veri;
        float iverf = 1.0f - verf;
        int angi = (int)angf;
        int angp = (angi < 7 ? angi + 1 : 0);
        angf -= angi;
        float iangf = 1.0f - angf;

        int hist = 8 * (4 * veri + hori);
        int p1 = angi + hist;
        int p2 = angp + hist;
        if (tx >= 2) {
          float grad1 = ihorf * grad;
          if (y >= 2) {
            float grad2 = iverf * grad1;
            desc_add(buffer, p1, iangf * grad2);
            desc_add(buffer, p2, angf * grad2);
          }
          if (y <= 13) {
            float grad2 = verf * grad1;
            desc_add(buffer, p1 + 32, iangf * grad2);
            desc_add(buffer, p2 + 32, angf * grad2);
          }
        }
        if (tx <= 14) { /* sic: 14, not 13 (cuSIFT_D.cu:243) */
          float grad1 = horf * grad;
          if (y >= 2) {
            float grad2 = iverf * grad1;
            desc_add(buffer, p1 + 8, iangf * grad2);
            desc_add(buffer, p2 + 8, angf * grad2);
          }
          if (y <= 13) {
            float grad2 = verf * grad1;
            desc_add(buffer, p1 + 40, iangf * grad2);
            desc_add(buffer, p2 + 40, angf * grad2);
          }
        }
      }
    }
    /* cuSIFT_D.cu:259-291: tree sums; idx<64: b[i]^2 + b[i+64]^2, then +32, +16, +8, +4, then 4 terms */
    for (int pass = 0; pass < 2; pass++) {
      float sums[64];
      for (int i = 0; i < 64; i++) sums[i] = buffer[i] * buffer[i] + buffer[i + 64] * buffer[i + 64];
      for (int i = 0; i < 32; i++) sums[i] = sums[i] + sums[i + 32];
      for (int i = 0; i < 16; i++) sums[i] = sums[i] + sums[i + 16];
      for (int i = 0; i < 8; i++) sums[i] = sums[i] + sums[i + 8];
      for (int i = 0; i < 4; i++) sums[i] = sums[i] + sums[i + 4];
      float tsum = sums[0] + sums[1] + sums[2] + sums[3];
      float r = 1.0f / sqrtf(tsum);
      if (pass == 0) {
        for (int i = 0; i < 128; i++) {
          buffer[i] = buffer[i] * r;
          if (buffer[i] > 0.2f) buffer[i] = 0.2f;
        }
      } else {
        for (int i = 0; i < 128; i++) pt->data[i] = buffer[i] * r;
      }
    }
    pt->coords2D[0] *= subsampling;
    pt->coords2D[1] *= subsampling;
    pt->scale *= subsampling;
  }
}

/* ConvertSiftToRootSift_D: cuSIFT_D.cu:299-317 (max(0.0, x) promotes to double there). */
void oracle_rootsift(oracle_sift_point *points, int n) {
  for (int p = 0; p < n; p++) {
    float sum = 0.0f;
    for (int i = 0; i < 128; i++) sum += points[p].data[i];
    for (int i = 0; i < 128; i++) {
      double m = points[p].data[i] > 0.0 ? (double)points[p].data[i] : 0.0;
      points[p].data[i] = sqrtf((float)(m / sum));
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * Driver: SiftData::Extract cuSIFT.cu:61-120, ExtractSiftLoop :175-202, ExtractSiftOctave :204-270.
 * Octaves are built finest->coarsest by ScaleDown and PROCESSED coarsest first.
 * ---------------------------------------------------------------------------------------- */
static int align_up(int a, int b) { return (a % b != 0) ? (a - a % b + b) : a; }

int oracle_extract(const float *img, int w, int h, const oracle_params *prm, oracle_sift_point *points) {
  const int N = prm->num_octaves < 1 ? 1 : prm->num_octaves;
  float **base = (float **)calloc((size_t)N, sizeof(float *));
  int *ws = (int *)calloc((size_t)N, sizeof(int)), *hs = (int *)calloc((size_t)N, sizeof(int));
  int *ps = (int *)calloc((size_t)N, sizeof(int));
  double *blur = (double *)calloc((size_t)N, sizeof(double));
  float *sub = (float *)calloc((size_t)N, sizeof(float));

  ws[0] = w;
  hs[0] = h;
  ps[0] = align_up(w, 128); /* cuImage.cu:11-13 */
  base[0] = (float *)calloc((size_t)ps[0] * h, sizeof(float));
  for (int y = 0; y < h; y++) memcpy(base[0] + (size_t)y * ps[0], img + (size_t)y * w, sizeof(float) * w);
  blur[0] = prm->init_blur;
  sub[0] = prm->subsampling;
  int built = 1;
  for (int o = 1; o < N; o++) {
    ws[o] = ws[o - 1] / 2;
    hs[o] = hs[o - 1] / 2;
    if (ws[o] < 1 || hs[o] < 1) break;
    ps[o] = align_up(ws[o], 128);
    base[o] = (float *)calloc((size_t)ps[o] * hs[o], sizeof(float));
    oracle_scale_down(base[o - 1], ws[o - 1], hs[o - 1], ps[o - 1], base[o], ps[o]);
    /* cuSIFT.cu:188: float totInitBlur = (float)sqrt(initBlur*initBlur + 0.5f*0.5f) / 2.0f; */
    float tot = (float)sqrt(blur[o - 1] * blur[o - 1] + 0.5f * 0.5f) / 2.0f;
    blur[o] = tot;
    sub[o] = sub[o - 1] * 2.0f;
    built = o + 1;
  }

  int counter = 0;
  for (int o = built - 1; o >= 0; o--) {
    if (!(prm->lowest_scale < sub[o] * 2.0f)) continue; /* cuSIFT.cu:194 */
    float *dog = (float *)calloc((size_t)7 * hs[o] * ps[o], sizeof(float));
    oracle_laplace_multi(base[o], ws[o], hs[o], ps[o], (float)blur[o], dog);
    int fst = counter; /* cuSIFT.cu:243 */
    oracle_find_points_multi(dog, ws[o], hs[o], ps[o], prm->peak_thresh, prm->edge_thresh, sub[o], points,
                             prm->max_pts, &counter);
    int tot = counter < prm->max_pts ? counter : prm->max_pts; /* cuSIFT.cu:252 */
    if (tot > fst) {
      oracle_compute_orientations(base[o], ws[o], hs[o], ps[o], points, fst, tot, prm->tex_frac_bits);
      oracle_extract_descriptors(base[o], ws[o], hs[o], ps[o], points, fst, tot, sub[o], prm->tex_frac_bits);
    }
    free(dog);
  }
  for (int o = 0; o < N; o++) free(base[o]);
  free(base);
  free(ws);
  free(hs);
  free(ps);
  free(blur);
  free(sub);
  return counter < prm->max_pts ? counter : prm->max_pts; /* cuSIFT.cu:110 */
}

/* ------------------------------------------------------------------------------------------
 * MatchSiftData: extras/matching.cu.  FLT_MAX is redefined to 999.0 there (:3).
 * ComputeDistance (:12-58): thread (ty=p1 in tile, tx=p2 in tile) sums pt1[itx]*pt2[itx] for i = 0..127 with
 * itx = (i + tx) & 127, i.e. the summation starts at element (p2 mod 16) and wraps; nvcc fuses a*b+sum.
 * Columns p2 >= numPts2 of the padded width hold -1 (dot) / 999 (L2).
 * FindMinCorr / FindMaxCorr: thread tx scans columns tx, tx+16, ... keeping (best, second, index) with strict
 * comparisons, then a tree over tx (len = 8,4,2,1) that keeps the lower tx on ties.
 * ---------------------------------------------------------------------------------------- */
#define MATCH_FLT_MAX 999.0f

typedef struct {
  float best, second;
  int idx;
} top2_t;

void oracle_match_sift_data(oracle_sift_point *sift1, int n1, const oracle_sift_point *sift2, int n2, int distance) {
  if (n1 <= 0 || n2 <= 0) return;
  const int corrWidth = ((n2 + 15) / 16) * 16;
  float *corr = (float *)malloc(sizeof(float) * (size_t)corrWidth);
  for (int p1 = 0; p1 < n1; p1++) {
    const float *a = sift1[p1].data;
    for (int p2 = 0; p2 < corrWidth; p2++) {
      if (p2 >= n2) {
        corr[p2] = distance == 1 ? MATCH_FLT_MAX : -1.0f;
        continue;
      }
      const float *b = sift2[p2].data;
      const int tx = p2 & 15;
      float sum = 0.0f;
      for (int i = 0; i < 128; i++) {
        const int itx = (i + tx) & 127;
        sum = fmaf(a[itx], b[itx], sum);
      }
      if (distance == 1) corr[p2] = sum > -1.0f ? 2 - 2 * sum : MATCH_FLT_MAX; /* :71-72 */
      else corr[p2] = sum;
    }
    top2_t t[16];
    for (int tx = 0; tx < 16; tx++) {
      t[tx].best = t[tx].second = distance == 1 ? MATCH_FLT_MAX : -1.0f;
      t[tx].idx = -1;
      for (int i = tx; i < corrWidth; i += 16) {
        const float val = corr[i];
        const int better = distance == 1 ? (val < t[tx].best) : (val > t[tx].best);
        const int better2 = distance == 1 ? (val < t[tx].second) : (val > t[tx].second);
        if (better) {
          t[tx].second = t[tx].best;
          t[tx].best = val;
          t[tx].idx = i;
        } else if (better2) {
          t[tx].second = val;
        }
      }
    }
    for (int len = 8; len > 0; len /= 2) {
      for (int tx = 0; tx < len; tx++) { /* tx < 8 in the kernel; entries >= len are dead afterwards */
        const float val = t[tx + len].best;
        const int i = t[tx + len].idx;
        const int better = distance == 1 ? (val < t[tx].best) : (val > t[tx].best);
        const int better2 = distance == 1 ? (val < t[tx].second) : (val > t[tx].second);
        if (better) {
          t[tx].second = t[tx].best;
          t[tx].best = val;
          t[tx].idx = i;
        } else if (better2) {
          t[tx].second = val;
        }
        const float val2 = t[tx + len].second;
        if (distance == 1 ? (val2 < t[tx].second) : (val2 > t[tx].second)) t[tx].second = val2;
      }
    }
    oracle_sift_point *pt = sift1 + p1;
    pt->score = t[0].best;
    if (distance == 1) pt->ambiguity = (float)(t[0].best / (t[0].second + 1e-6));       /* :222 (double constant) */
    else pt->ambiguity = (float)((1 - t[0].best) / (1 - t[0].second + 1e-6));            /* :143 */
    pt->match = t[0].idx;
    const int m = t[0].idx >= 0 && t[0].idx < n2 ? t[0].idx : 0;
    pt->match_xpos = sift2[m].coords2D[0];
    pt->match_ypos = sift2[m].coords2D[1];
  }
  free(corr);
}

int oracle_match_filter(const oracle_sift_point *sift1, int n1, float score_threshold, float ambiguity_threshold,
                        int *idx) {
  const float thresh2 = score_threshold * score_threshold;
  const float athresh2 = ambiguity_threshold * ambiguity_threshold;
  int n = 0;
  for (int i = 0; i < n1; i++)
    if (sift1[i].score < thresh2 && sift1[i].ambiguity < athresh2) { /* :336 */
      if (idx) idx[n] = i;
      n++;
    }
  return n;
}

/* ------------------------------------------------------------------------------------------------
 * Caller-side front-end: what the reference's programs do on the host before upload,
 * main.cpp:300-318 and test/detector.cpp:19-27: cv::Mat::convertTo(CV_32FC1) and the optional
 * cv::GaussianBlur(img, img, cv::Size(3, 3), 0.5).
 *
 * PARITY UNPINNED for the blur: OpenCV is a system dependency of the reference (find_package(OpenCV),
 * CMakeLists.txt:12-13, no version pinned) and is not in this image.  This restates the published algorithm of its
 * float path: getGaussianKernel(3, sigma) = exp(-x^2 / (2 sigma^2)) evaluated in double, normalised, stored as
 * float; separable, rows then columns, each as (S[-1] + S[+1]) * k1 + S[0] * k0 (the symmetric small-kernel
 * filters), border BORDER_REFLECT_101 (the default border type).
 * ------------------------------------------------------------------------------------------------ */
void oracle_u8_to_f32(const unsigned char *src, int w, int h, int src_pitch, float *dst, int dst_pitch) {
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) dst[(long)y * dst_pitch + x] = (float)src[(long)y * src_pitch + x];
}

static int reflect101(int i, int n) {
  if (n == 1) return 0;
  if (i < 0) return -i;
  if (i >= n) return 2 * n - 2 - i;
  return i;
}

void oracle_gaussian3x3(const float *src, int w, int h, int src_pitch, float *dst, int dst_pitch, float sigma) {
  const double e1 = exp(-1.0 / (2.0 * (double)sigma * (double)sigma));
  const double sum = 1.0 + 2.0 * e1;
  const float k0 = (float)(1.0 / sum), k1 = (float)(e1 / sum);
  float *rows = (float *)malloc(sizeof(float) * (size_t)w * (size_t)h);
  for (int y = 0; y < h; ++y) {
    const float *s = src + (long)y * src_pitch;
    for (int x = 0; x < w; ++x)
      rows[(long)y * w + x] = (s[reflect101(x - 1, w)] + s[reflect101(x + 1, w)]) * k1 + s[x] * k0;
  }
  for (int y = 0; y < h; ++y) {
    const float *r0 = rows + (long)reflect101(y - 1, h) * w, *r1 = rows + (long)y * w,
                *r2 = rows + (long)reflect101(y + 1, h) * w;
    for (int x = 0; x < w; ++x) dst[(long)y * dst_pitch + x] = (r0[x] + r2[x]) * k1 + r1[x] * k0;
  }
  free(rows);
}

/* ------------------------------------------------------------------------------------------------
 * RANSAC homography: FindHomography, extras/homography.cu:182-269, with its
 * kernels ComputeHomographies (:89-130, one 8x8 solve per hypothesis through InvertMatrix<8>, :3-87) and
 * TestHomographies (:135-178, inlier count of every hypothesis).  PARITY UNPINNED: the reference holds no test
 * or fixture for it (only main.cpp calls it), so this restatement is the only oracle.
 *
 * Arithmetic: where nvcc's default -fmad=true contracts `sum -= a*b` / `sum += a*b` the fused form is written
 * out (fmaf); the 1.0/x of the pivot scaling is a double division rounded to float, as the literal 1.0 makes it;
 * the inlier test multiplies with round-toward-zero (__fmul_rz, which also keeps nvcc from fusing).
 * Not reproduced: TestHomographies walks numPtsUp = iDivUp(numPts,16)*16 entries of d_coord, i.e. up to 15
 * uninitialised floats per coordinate row beyond numPts (:152,218-221); only the numPts real points are tested.
 * ------------------------------------------------------------------------------------------------ */
static float mul_rz(float a, float b) {
  const double p = (double)a * (double)b; /* exact: 24 + 24 bits */
  float f = (float)p;                     /* to nearest ... */
  if (fabs((double)f) > fabs(p)) f = nextafterf(f, 0.0f); /* ... then back towards zero */
  return f;
}

/* InvertMatrix<8>, extras/homography.cu:3-87: LU with implicit row scaling, then column-by-column
 * forward/back substitution of the identity. */
static void invert8(float m[8][8], float inv[8][8]) {
  int perm[8];
  float rhs[8], rowscale[8];
  int imax = 0;
  for (int i = 0; i < 8; i++) {
    perm[i] = 0;
    float big = 0.0f;
    for (int j = 0; j < 8; j++) {
      const float t = fabsf(m[i][j]);
      if (t > big) big = t;
    }
    rowscale[i] = big > 0.0f ? (float)(1.0 / (double)big) : 1e16f; /* :22-25 */
  }
  for (int j = 0; j < 8; j++) {
    for (int i = 0; i < j; i++) {
      float sum = m[i][j];
      for (int k = 0; k < i; k++) sum = fmaf(-m[i][k], m[k][j], sum);
      m[i][j] = sum;
    }
    float big = 0.0f;
    for (int i = j; i < 8; i++) {
      float sum = m[i][j];
      for (int k = 0; k < j; k++) sum = fmaf(-m[i][k], m[k][j], sum);
      m[i][j] = sum;
      const float dum = rowscale[i] * fabsf(sum);
      if (dum >= big) { /* :42-45: ties go to the later row */
        big = dum;
        imax = i;
      }
    }
    if (j != imax) {
      for (int k = 0; k < 8; k++) {
        const float t = m[imax][k];
        m[imax][k] = m[j][k];
        m[j][k] = t;
      }
      rowscale[imax] = rowscale[j];
    }
    perm[j] = imax;
    if (m[j][j] == 0.0f) m[j][j] = 1e-16f; /* :57-58 */
    if (j != 7) {
      const float dum = (float)(1.0 / (double)m[j][j]);
      for (int i = j + 1; i < 8; i++) m[i][j] *= dum;
    }
  }
  for (int c = 0; c < 8; c++) {
    for (int k = 0; k < 8; k++) rhs[k] = 0.0f;
    rhs[c] = 1.0f;
    int first = -1;
    for (int i = 0; i < 8; i++) {
      const int ip = perm[i];
      float sum = rhs[ip];
      rhs[ip] = rhs[i];
      if (first != -1)
        for (int k = first; k < i; k++) sum = fmaf(-m[i][k], rhs[k], sum);
      else if (sum != 0.0f)
        first = i;
      rhs[i] = sum;
    }
    for (int i = 7; i >= 0; i--) {
      float sum = rhs[i];
      for (int k = i + 1; k < 8; k++) sum = fmaf(-m[i][k], rhs[k], sum);
      rhs[i] = sum / m[i][i];
    }
    for (int i = 0; i < 8; i++) inv[i][c] = rhs[i];
  }
}

/* ComputeHomographies, extras/homography.cu:89-130.  coord = [4][num_pts] (x1, y1, x2, y2 rows),
 * rand_pts = [4][num_loops] point indices, homo = [8][num_loops]. */
void oracle_compute_homographies(const float *coord, int num_pts, const int *rand_pts, int num_loops, float *homo) {
  for (int idx = 0; idx < num_loops; idx++) {
    float a[8][8], ia[8][8], b[8];
    for (int i = 0; i < 4; i++) {
      const int pt = rand_pts[i * num_loops + idx];
      const float x1 = coord[pt], y1 = coord[pt + num_pts];
      const float x2 = coord[pt + 2 * num_pts], y2 = coord[pt + 3 * num_pts];
      float *r1 = a[2 * i], *r2 = a[2 * i + 1];
      r1[0] = x1, r1[1] = y1, r1[2] = 1.0f, r1[3] = r1[4] = r1[5] = 0.0f, r1[6] = -x2 * x1, r1[7] = -x2 * y1;
      r2[0] = r2[1] = r2[2] = 0.0f, r2[3] = x1, r2[4] = y1, r2[5] = 1.0f, r2[6] = -y2 * x1, r2[7] = -y2 * y1;
      b[2 * i] = x2;
      b[2 * i + 1] = y2;
    }
    invert8(a, ia);
    for (int j = 0; j < 8; j++) {
      float sum = 0.0f;
      for (int i = 0; i < 8; i++) sum = fmaf(ia[j][i], b[i], sum);
      homo[j * num_loops + idx] = sum;
    }
  }
}

/* TestHomographies, extras/homography.cu:135-178: inliers of every hypothesis over the num_pts points. */
void oracle_test_homographies(const float *coord, int num_pts, const float *homo, int num_loops, float thresh2,
                              int *counts) {
  for (int idx = 0; idx < num_loops; idx++) {
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = homo[i * num_loops + idx];
    int cnt = 0;
    for (int i = 0; i < num_pts; i++) {
      const float x1 = coord[i], y1 = coord[i + num_pts], x2 = coord[i + 2 * num_pts], y2 = coord[i + 3 * num_pts];
      const float nomx = mul_rz(a[0], x1) + mul_rz(a[1], y1) + a[2];
      const float nomy = mul_rz(a[3], x1) + mul_rz(a[4], y1) + a[5];
      const float deno = mul_rz(a[6], x1) + mul_rz(a[7], y1) + 1.0f;
      const float errx = mul_rz(x2, deno) - nomx;
      const float erry = mul_rz(y2, deno) - nomy;
      const float err2 = mul_rz(errx, errx) + mul_rz(erry, erry);
      if (err2 < mul_rz(thresh2, mul_rz(deno, deno))) cnt++;
    }
    counts[idx] = cnt;
  }
}

/* The device part + the final selection of FindHomography (extras/homography.cu:237-258) for given samples:
 * rand_pts = [4][num_loops] indices into `pts` (the host draws them with rand(), :222-235 -- the caller's job).
 * homography[9] (h[8] = 1, :184-186); returns the index of the winning hypothesis (first maximum, :249-254).
 * all_homo ([8][num_loops]) and all_counts ([num_loops]) may be NULL. */
int oracle_find_homography(const oracle_sift_point *pts, int num_pts, const int *rand_pts, int num_loops, float thresh,
                           float homography[9], int *num_matches, float *all_homo, int *all_counts) {
  static const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  memcpy(homography, ident, sizeof(ident));
  *num_matches = 0;
  if (num_pts < 1 || num_loops < 1) return -1;
  float *coord = (float *)malloc(sizeof(float) * 4 * (size_t)num_pts);
  float *homo = (float *)malloc(sizeof(float) * 8 * (size_t)num_loops);
  int *counts = (int *)malloc(sizeof(int) * (size_t)num_loops);
  for (int i = 0; i < num_pts; i++) { /* :237-240 */
    coord[i] = pts[i].coords2D[0];
    coord[i + num_pts] = pts[i].coords2D[1];
    coord[i + 2 * num_pts] = pts[i].match_xpos;
    coord[i + 3 * num_pts] = pts[i].match_ypos;
  }
  oracle_compute_homographies(coord, num_pts, rand_pts, num_loops, homo);
  oracle_test_homographies(coord, num_pts, homo, num_loops, thresh * thresh, counts);
  int best = -1, best_count = -1;
  for (int i = 0; i < num_loops; i++)
    if (counts[i] > best_count) {
      best_count = counts[i];
      best = i;
    }
  *num_matches = best_count;
  for (int j = 0; j < 8; j++) homography[j] = homo[j * num_loops + best];
  if (all_homo) memcpy(all_homo, homo, sizeof(float) * 8 * (size_t)num_loops);
  if (all_counts) memcpy(all_counts, counts, sizeof(int) * (size_t)num_loops);
  free(coord);
  free(homo);
  free(counts);
  return best;
}
