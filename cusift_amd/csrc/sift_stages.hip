// sift_stages.hip -- the stage entry points of the C ABI (include/cusift_amd.h) and their launch wrappers: one function per
// kernel of the reference (cuSIFT.cu:313-455) plus the fused detection, the band forms of the strip tiling, the matcher,
// the homography and the packing of SiftData.
#include "sift_host.h"

// ------------------------------------------------------------------------------------------------
// front-end
// ------------------------------------------------------------------------------------------------
extern "C" int cusift_u8_to_f32(cusift_ctx *ctx, float *d_dst, int dst_pitch, size_t dst_stride,
                                const unsigned char *d_src, int w, int h, int src_pitch_bytes,
                                size_t src_stride_bytes, int n_images) {
  TRY(enter(ctx));
  if (!d_dst || !d_src) return fail(CUSIFT_ERR_INVALID, "u8_to_f32: missing data");
  if (n_images < 1 || n_images > 65535 || w < 1 || h < 1 || h > 65535 || dst_pitch < w || src_pitch_bytes < w)
    return fail(CUSIFT_ERR_INVALID, "u8_to_f32: bad geometry");
  const int vec_ok = (dst_pitch % 4 == 0) && (((uintptr_t)d_dst % 16) == 0) && (dst_stride % 4 == 0) &&
                     (src_pitch_bytes % 4 == 0) && (((uintptr_t)d_src % 4) == 0) && (src_stride_bytes % 4 == 0);
  dim3 grid(idiv_up(idiv_up(w, 4), 256), h, n_images);
  hipLaunchKernelGGL(u8_to_f32_kernel, grid, dim3(256), 0, ctx->stream, d_dst, dst_pitch, (long)dst_stride, d_src, w, h,
                     src_pitch_bytes, (long)src_stride_bytes, vec_ok);
  return check_launch("u8_to_f32");
}

extern "C" int cusift_image_u8_h2d(cusift_ctx *ctx, float *d_dst, int dst_pitch, const unsigned char *h_src, int w,
                                   int h) {
  if (!ctx || !d_dst || !h_src || w < 1 || h < 1 || dst_pitch < w) return fail(CUSIFT_ERR_INVALID, "bad argument");
  TRY(enter(ctx));
  const size_t spitch = align_up_sz((size_t)w, 4);
  const size_t bytes = spitch * h;
  if (bytes > ctx->u8_stage_bytes) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->u8_stage) HIP_TRY(hipFree(ctx->u8_stage));
    ctx->u8_stage = nullptr;
    ctx->u8_stage_bytes = 0;
    hipError_t e = hipMalloc((void **)&ctx->u8_stage, bytes);
    if (e != hipSuccess) return fail(CUSIFT_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    ctx->u8_stage_bytes = bytes;
  }
  HIP_TRY(hipMemcpy2DAsync(ctx->u8_stage, spitch, h_src, (size_t)w, (size_t)w, h, hipMemcpyHostToDevice, ctx->stream));
  TRY(cusift_u8_to_f32(ctx, d_dst, dst_pitch, (size_t)dst_pitch * h, ctx->u8_stage, w, h, (int)spitch, bytes, 1));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return CUSIFT_OK;
}

extern "C" int cusift_gaussian3x3(cusift_ctx *ctx, float *d_dst, int dst_pitch, size_t dst_stride, const float *d_src,
                                  int w, int h, int src_pitch, size_t src_stride, int n_images, float sigma) {
  TRY(enter(ctx));
  if (!d_dst || !d_src || d_dst == d_src) return fail(CUSIFT_ERR_INVALID, "gaussian3x3: need distinct src and dst");
  if (n_images < 1 || n_images > 65535 || w < 1 || h < 1 || h > 65535 || dst_pitch < w || src_pitch < w ||
      !(sigma > 0.0f))
    return fail(CUSIFT_ERR_INVALID, "gaussian3x3: bad argument");
  // cv::getGaussianKernel(3, sigma, CV_32F): exp(-x^2/(2 sigma^2)) in double, normalised, stored as float
  const double e1 = exp(-1.0 / (2.0 * (double)sigma * (double)sigma));
  const double sum = 1.0 + 2.0 * e1;
  const float k0 = (float)(1.0 / sum), k1 = (float)(e1 / sum);
  dim3 grid(idiv_up(w, 256), h, n_images);
  hipLaunchKernelGGL(gaussian3x3_kernel, grid, dim3(256), 0, ctx->stream, d_dst, dst_pitch, (long)dst_stride, d_src, w,
                     h, src_pitch, (long)src_stride, k0, k1);
  return check_launch("gaussian3x3");
}

// ------------------------------------------------------------------------------------------------
// stage entry points
// ------------------------------------------------------------------------------------------------
int scale_down_impl(cusift_ctx *ctx, float *d_dst, int dst_pitch, size_t dst_stride, const float *d_src, int w,
                           int h, int src_pitch, size_t src_stride, int n_images, float variance, RowWindow src_rw,
                           int dst_row0, int r_begin, int r_end, bool band) {
  TRY(enter(ctx));
  if (!d_dst || !d_src) return fail(CUSIFT_ERR_INVALID, "ScaleDown: missing data");  // cuSIFT.cu:315-318
  if (!(variance > 0.0f)) return fail(CUSIFT_ERR_INVALID, "ScaleDown: variance must be > 0");
  const int ow = w / 2, oh = r_end - r_begin;
  if (n_images < 1 || ow < 1 || oh < 1 || src_pitch < w || dst_pitch < ow)
    return fail(CUSIFT_ERR_INVALID, "ScaleDown: bad geometry w=%d h=%d", w, h);
  ScaleDownTaps T;
  scale_down_taps(T, variance);
  const bool fast = w >= 4 && (src_pitch % 4 == 0) && (((uintptr_t)d_src % 16) == 0) &&
                    (src_stride % 4 == 0) && (dst_pitch % 2 == 0) && (((uintptr_t)d_dst % 8) == 0) &&
                    (dst_stride % 2 == 0) && ((size_t)h * src_pitch * sizeof(float) < (1ull << 31)) &&
                    (band || !ctx->knobs.force_generic);
  if (band && !fast)
    return fail(CUSIFT_ERR_INVALID, "ScaleDown (band): needs w >= 4, 16-byte aligned source rows, band < 2 GiB");
  StageTimer t(ctx, CUSIFT_STAGE_SCALEDOWN);
  if (fast) {
    const int strips = idiv_up(ow, 124);  // kDownStrip
    // measured (tools/probe_rows.py, 64 images): 1920x1080 -> 960x540 streams from HBM and likes short chunks
    // (r = 4: 0.137 ms, r = 32: 0.150 ms); the smaller levels are served by the Infinity Cache and like tall ones
    int rlo = 4, rhi = (long)oh * strips * n_images > 200000 ? 4 : 32;
    rows_bounds(ctx, kKnobScaleDown, rlo, rhi);
    const int rows = pick_rows(ctx, oh, strips, n_images, rlo, rhi);
    dim3 grid(idiv_up(strips, kWavesPerBlock), idiv_up(oh, rows), n_images);
    hipLaunchKernelGGL(scale_down_fast_kernel, grid, dim3(256), 0, ctx->stream, d_dst, dst_pitch, (long)dst_stride,
                       d_src, w, h, src_pitch, (long)src_stride, rows, T, src_rw, dst_row0, r_begin, r_end);
  } else {
    const int strips = idiv_up(ow, 64);
    const int rows = pick_rows(ctx, oh, strips, n_images, 4, 16);
    dim3 grid(strips, idiv_up(idiv_up(oh, rows), kWavesPerBlock), n_images);
    hipLaunchKernelGGL(scale_down_kernel, grid, dim3(256), 0, ctx->stream, d_dst, dst_pitch, (long)dst_stride, d_src,
                       w, h, src_pitch, (long)src_stride, rows, T);
  }
  return check_launch("scale_down");
}

// The ScaleDown chain of a small call -- levels 1 .. n from level 0 -- in ONE launch (pyramid_small_kernel).
// A call takes it up to kPyramidSmallPixels source pixels (one 1080p frame: four launches of 6-10 us become one of
// ~10); beyond that the 2.9x re-reads of the source cost more than the dispatches.  Not with the stage timers on (they
// count one ScaleDown per octave).
constexpr size_t kPyramidSmallPixels = (size_t)5 << 19;  // 2.6 Mpixel
bool wants_small_pyramid(const cusift_ctx *ctx, int n_images, int w, int h) {
  if (ctx->knobs.small_pyramid == 0 || ctx->knobs.force_generic) return false;
  if (ctx->knobs.small_pyramid > 0) return true;
  return !ctx->timing && (size_t)n_images * (size_t)w * (size_t)h <= kPyramidSmallPixels;
}

// describe_all_kernel may form the lists' running sums itself (no join_counts_kernel: a small call saves the dispatch) --
// but it then finds every keypoint's list by a walk over the lists' counters, which a large call pays per keypoint
// (64 x 1080p of `blobs`, 581 k keypoints, one stream: 2.066 ms with the self-join against 1.99 with the 5-us join kernel and
// its precomputed ends).  So: small calls only.
bool wants_self_join(const cusift_ctx *, int n_images, int w, int h) {
  return (size_t)n_images * (size_t)w * (size_t)h <= kPyramidSmallPixels;
}

int pyramid_small_impl(cusift_ctx *ctx, const float *const *base, const int *w, const int *h, const int *pitch,
                              const size_t *stride, int n_levels, int n_images, float variance, unsigned int *d_zero,
                              int n_zero) {
  TRY(enter(ctx));
  if (n_levels < 1 || n_levels > kMaxPyramidLevels || n_images < 1)
    return fail(CUSIFT_ERR_INVALID, "ScaleDown (levels): 1..%d levels", kMaxPyramidLevels);
  if (!(variance > 0.0f)) return fail(CUSIFT_ERR_INVALID, "ScaleDown: variance must be > 0");
  PyramidLevels P;
  memset(&P, 0, sizeof(P));
  P.n = n_levels;
  P.tile = 64 >> n_levels;  // 32, 16, 8, 4: a workgroup needs about 60 x 60 pixels of level 1 whatever the depth
  for (int k = 0; k <= n_levels; ++k) {
    if (!base[k] || w[k] < 1 || h[k] < 1 || pitch[k] < w[k]) return fail(CUSIFT_ERR_INVALID, "ScaleDown (levels): bad level %d", k);
    if (k > 0 && (w[k] != w[k - 1] / 2 || h[k] != h[k - 1] / 2)) return fail(CUSIFT_ERR_INVALID, "ScaleDown (levels): level %d is not half of level %d", k, k - 1);
    P.base[k] = const_cast<float *>(base[k]);
    P.w[k] = w[k];
    P.h[k] = h[k];
    P.pitch[k] = pitch[k];
    P.stride[k] = (long)stride[k];
  }
  // LDS: the needed squares of every level + the largest H, for the largest workgroup: a side grows as 2 s + 3 going
  // down a level, + 1 for the odd remainder an owner at the far edge takes on
  size_t floats = 0, h_max = 0;
  int side = P.tile;
  for (int k = n_levels; k >= 1; --k) {
    floats += (size_t)side * side;
    h_max = std::max(h_max, (size_t)(2 * side + 3) * side);
    side = 2 * side + 4;
  }
  floats += h_max;
  ScaleDownTaps T;
  scale_down_taps(T, variance);
  dim3 grid(idiv_up(w[n_levels], P.tile), idiv_up(h[n_levels], P.tile), n_images);
  StageTimer t(ctx, CUSIFT_STAGE_SCALEDOWN);
  hipLaunchKernelGGL(pyramid_small_kernel, grid, dim3(256), floats * sizeof(float), ctx->stream, P, T, d_zero, n_zero);
  return check_launch("scale_down (levels)");
}

extern "C" int cusift_scale_down_levels(cusift_ctx *ctx, const float *d_src, int w, int h, int src_pitch,
                                        size_t src_stride, float *const *d_levels, const int *pitches,
                                        const size_t *strides, int n_levels, int n_images, float variance) {
  if (!d_src || !d_levels || !pitches || !strides) return fail(CUSIFT_ERR_INVALID, "ScaleDown (levels): NULL argument");
  if (n_levels < 1 || n_levels > kMaxPyramidLevels)
    return fail(CUSIFT_ERR_INVALID, "ScaleDown (levels): 1..%d levels", kMaxPyramidLevels);
  const float *base[kMaxPyramidLevels + 1];
  int ws[kMaxPyramidLevels + 1], hs[kMaxPyramidLevels + 1], ps[kMaxPyramidLevels + 1];
  size_t st[kMaxPyramidLevels + 1];
  base[0] = d_src, ws[0] = w, hs[0] = h, ps[0] = src_pitch, st[0] = src_stride;
  for (int k = 1; k <= n_levels; ++k) {
    base[k] = d_levels[k - 1], ws[k] = ws[k - 1] / 2, hs[k] = hs[k - 1] / 2, ps[k] = pitches[k - 1], st[k] = strides[k - 1];
    if (ws[k] < 1 || hs[k] < 1) return fail(CUSIFT_ERR_INVALID, "ScaleDown (levels): level %d of %dx%d is empty", k, w, h);
    if (n_images > 1 && st[k] < (size_t)hs[k] * ps[k]) return fail(CUSIFT_ERR_INVALID, "ScaleDown (levels): stride of level %d too small", k);
  }
  if (n_images > 1 && src_stride < (size_t)h * src_pitch) return fail(CUSIFT_ERR_INVALID, "ScaleDown (levels): src_stride too small");
  return pyramid_small_impl(ctx, base, ws, hs, ps, st, n_levels, n_images, variance, nullptr, 0);
}

extern "C" int cusift_scale_down(cusift_ctx *ctx, float *d_dst, int dst_pitch, size_t dst_stride, const float *d_src,
                                 int w, int h, int src_pitch, size_t src_stride, int n_images, float variance) {
  if (h / 2 < 1) return fail(CUSIFT_ERR_INVALID, "ScaleDown: bad geometry w=%d h=%d", w, h);
  return scale_down_impl(ctx, d_dst, dst_pitch, dst_stride, d_src, w, h, src_pitch, src_stride, n_images, variance,
                         RowWindow{0, h}, 0, 0, h / 2, false);
}

extern "C" int cusift_scale_down_band(cusift_ctx *ctx, float *d_dst, int dst_pitch, int dst_row0, int r_begin,
                                      int r_end, const float *d_src, int w, int h_src, int src_pitch, int src_row0,
                                      int h_src_global, float variance) {
  if (h_src < 1 || h_src_global < 2 || src_row0 < 0 || src_row0 + h_src > h_src_global || r_begin < dst_row0 ||
      r_end <= r_begin || r_end > h_src_global / 2)
    return fail(CUSIFT_ERR_INVALID, "ScaleDown (band): bad row geometry");
  return scale_down_impl(ctx, d_dst, dst_pitch, 0, d_src, w, h_src, src_pitch, 0, 1, variance,
                         RowWindow{src_row0, h_src_global}, dst_row0, r_begin, r_end, true);
}

extern "C" int cusift_laplace_taps(float init_blur, float taps[8 * 16]) {
  if (!taps) return fail(CUSIFT_ERR_INVALID, "taps is NULL");
  laplace_taps_table(init_blur, taps);
  return CUSIFT_OK;
}

extern "C" int cusift_laplace_multi(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, size_t img_stride,
                                    float init_blur, float *d_dog, size_t dog_stride, int n_images) {
  TRY(enter(ctx));
  if (!d_img || !d_dog) return fail(CUSIFT_ERR_INVALID, "LaplaceMulti: missing data");
  if (n_images < 1 || w < 1 || h < 1 || pitch < w) return fail(CUSIFT_ERR_INVALID, "LaplaceMulti: bad geometry");
  if (n_images > 1 && dog_stride < (size_t)kNumDog * h * pitch)
    return fail(CUSIFT_ERR_INVALID, "LaplaceMulti: dog_stride too small");
  float taps[8 * 16];
  laplace_taps_table(init_blur, taps);
  LaplaceTaps T;
  for (int s = 0; s < kNumLevels; ++s)
    for (int j = 0; j < 5; ++j) T.k[s][j] = taps[16 * s + j];
  const int vec_ok = (pitch % 4 == 0) && (((uintptr_t)d_img % 16) == 0) && (((uintptr_t)d_dog % 16) == 0) &&
                     (img_stride % 4 == 0) && (dog_stride % 4 == 0) && (((size_t)h * pitch) % 4 == 0);
  const int strips = idiv_up(w, kBlurStrip);
  // short chunks: the halo rows they re-read come from L2, and the chip sustains a visibly higher store rate when
  // many short waves write than when few long ones do (tools/ab_laplace_rows.sh with non-temporal stores, 64x1080p, all
  // octaves at one r: r = 3 0.263 ms per launch, 6: 0.242, 8: 0.234, 12: 0.239, 16: 0.243, 32: 0.270)
  int rlo = 3, rhi = 8;
  rows_bounds(ctx, kKnobLaplace, rlo, rhi);
  const int rows = pick_rows(ctx, h, strips, n_images, rlo, rhi);
  dim3 grid(strips, idiv_up(idiv_up(h, rows), kWavesPerBlock), n_images);
  // fast path: 16-byte aligned rows (any width >= 4), 32-bit buffer offsets
  const bool fast = vec_ok && w >= 4 && ((size_t)h * pitch * sizeof(float) < (1ull << 31)) && !ctx->knobs.force_generic;
  StageTimer t(ctx, CUSIFT_STAGE_LAPLACE);
  if (fast) {
    LaplaceTapsPk TP;
    for (int q = 0; q < kNumLevels / 2; ++q)
      for (int j = 0; j < 5; ++j) {
        TP.k[q][j].x = taps[16 * (2 * q) + j];
        TP.k[q][j].y = taps[16 * (2 * q + 1) + j];
      }
    int wpb = kWavesPerBlock;
    if (ctx->knobs.laplace_waves > 0) wpb = std::min(4, ctx->knobs.laplace_waves);  // experiments only
    dim3 fgrid(strips, idiv_up(idiv_up(h, rows), wpb), n_images);
    // DoG planes are written once and read much later (by FindPointsMulti): non-temporal stores keep them from
    // displacing the source rows' halo in L2 -- measured on one box (tools/ab_laplace_aux.sh): 4.35 -> 4.63 TB/s for
    // this kernel and 3.87 -> 4.18 TB/s for the FindPointsMulti that follows
    const int aux = ctx->knobs.laplace_aux >= 0 ? ctx->knobs.laplace_aux : 2;  // experiments: cache policy of the stores
#define LAUNCH_LAPLACE(A)                                                                                         \
  hipLaunchKernelGGL(laplace_multi_fast_kernel<A>, fgrid, dim3(64 * wpb), 0, ctx->stream, d_img, d_dog, w, h, pitch, \
                     (long)img_stride, (long)dog_stride, rows, TP)
    if (aux == 2) LAUNCH_LAPLACE(2);
    else if (aux == 16) LAUNCH_LAPLACE(16);
    else if (aux == 18) LAUNCH_LAPLACE(18);
    else LAUNCH_LAPLACE(0);
#undef LAUNCH_LAPLACE
  } else {
    hipLaunchKernelGGL(laplace_multi_kernel, grid, dim3(256), 0, ctx->stream, d_img, d_dog, w, h, pitch,
                       (long)img_stride, (long)dog_stride, rows, vec_ok, T);
  }
  return check_launch("laplace_multi");
}

extern "C" int cusift_find_points_multi(cusift_ctx *ctx, const float *d_dog, int w, int h, int pitch,
                                        size_t dog_stride, float peak_thresh, float edge_thresh, float subsampling,
                                        cusift_point *d_points, int max_pts, unsigned int *d_counters, int n_images) {
  TRY(enter(ctx));
  if (!d_dog || !d_points || !d_counters)
    return fail(CUSIFT_ERR_INVALID, "FindPointsMulti: missing data");  // cuSIFT.cu:425-428
  if (n_images < 1 || w < 1 || h < 1 || pitch < w || max_pts < 1)
    return fail(CUSIFT_ERR_INVALID, "FindPointsMulti: bad geometry");
  FindParams P;
  find_params(P, peak_thresh, edge_thresh, subsampling);
  const int vec_ok = (pitch % 2 == 0) && (((uintptr_t)d_dog % 8) == 0) && (dog_stride % 2 == 0) &&
                     (((size_t)h * pitch) % 2 == 0);
  const int strips = idiv_up(w, kFindStrip);
  int rlo = 4, rhi = 16;  // tools/probe_rows.py, 64x1080p: r = 16 0.770 ms, r = 32 0.803 ms
  rows_bounds(ctx, kKnobFindPoints, rlo, rhi);
  const int rows = pick_rows(ctx, h, strips, n_images, rlo, rhi);
  dim3 grid(strips, idiv_up(idiv_up(h, rows), kWavesPerBlock), n_images);
  const bool fast = vec_ok && w >= 2 &&
                    ((size_t)kNumDog * h * pitch * sizeof(float) < (1ull << 31)) && !ctx->knobs.force_generic;
  StageTimer t(ctx, CUSIFT_STAGE_FINDPOINTS);
  if (fast) {
    dim3 fgrid(idiv_up(strips, kWavesPerBlock), idiv_up(h, rows), n_images);  // 4 waves = 4 adjacent strips
    hipLaunchKernelGGL(find_points_fast_kernel, fgrid, dim3(256), 0, ctx->stream, d_dog, w, h, pitch,
                       (long)dog_stride, d_points, max_pts, d_counters, rows, P);
  }
  else
    hipLaunchKernelGGL(find_points_kernel, grid, dim3(256), 0, ctx->stream, d_dog, w, h, pitch, (long)dog_stride,
                     d_points, max_pts, d_counters, rows, vec_ok, P);
  return check_launch("find_points_multi");
}

bool detect_fused_ok(const float *d_img, int w, int h, int pitch, size_t img_stride) {
  return (pitch % 4 == 0) && (((uintptr_t)d_img % 16) == 0) && (img_stride % 4 == 0) && w >= 4 && h >= 3 &&
         ((size_t)h * pitch * sizeof(float) < (1ull << 31));
}

// Chunk height of the fused detection: centre rows per wave (see the comment in detect_impl).
int detect_rows(const cusift_ctx *ctx, int rows_total, int strips, int n_images, int concurrent) {
  int rows_lo = 2, rows_hi = concurrent >= 3 ? 240 : (concurrent >= 2 ? 112 : 64);
  rows_bounds(ctx, kKnobDetect, rows_lo, rows_hi);
  const double wave_rows = (double)rows_total * strips * n_images;
  double coef = concurrent >= 2 ? 0.09 : 0.05;
  if (ctx->knobs.detect_rows_coef > 0.0) coef = ctx->knobs.detect_rows_coef;  // tuning experiments only
  double r = coef * sqrt(wave_rows);
  // Three or more batches in flight (round 5, the detections of one batch are a chain of launches now): the other batches'
  // kernels fill whatever a launch leaves idle, so a LARGE launch is best cut into about as many chunks as the chip holds
  // waves at once -- 1.17 x (CUs x 4 SIMDs x 2 waves) = 2400 on MI355X -- i.e. chunk height proportional to the work, not
  // to its square root: the window fill and the two extra rows of a chunk are then paid ~5 times per strip instead of ~16.
  // tools/ab_pyramid.py with the lab build's CUSIFT_DETECT_ROWS_COEF / _HI, four streams, ms per call, rows of octave 0:
  // 64 frames 67: 0.977, 119: 0.955, 164: 0.947, 223: 0.945, 298: 0.952, 446: 1.000;  32 frames 47: 0.505, 105: 0.489,
  // 158: 0.489;  16 frames 33: 0.274, 74: 0.271, 111: 0.287;  8 frames 24: 0.154, 79: 0.182;  three streams, 64 frames
  // 67: 0.982, 134: 0.960, 186: 0.959;  two streams 67: 1.018, 119: 1.059, 223: 1.103 (stays on the square-root rule).
  if (concurrent >= 3 && ctx->knobs.detect_rows_coef <= 0.0) r = std::max(r, wave_rows / (1.17 * ctx->num_cus * 8.0));
  int rows = std::max(rows_lo, std::min(rows_hi, (int)lround(r)));
  // equal chunks: 1080 rows at 37 per chunk are 29 x 37 + 7 -- thirty chunks of 36 end together (a lone caller's
  // 64 x 1080p 1.108 -> 1.102 ms, three interleaved runs each; four streams: unchanged)
  if (ctx->knobs.detect_rows_coef <= 0.0) rows = std::max(rows_lo, idiv_up(rows_total, idiv_up(rows_total, rows)));
  return rows;
}

int detect_impl(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, size_t img_stride, float init_blur,
                       float peak_thresh, float edge_thresh, float subsampling, cusift_point *d_points, int max_pts,
                       unsigned int *d_counters, int n_images, RowWindow rw, int cy_begin, int cy_end,
                       int concurrent, bool heads, bool side, const DownOut *down) {
  // heads: `d_points` is a staging list of the context (kStagedRecBytes per keypoint); side: the launch goes to the
  // context's side stream (cusift_extract_batch); down: the kernel also writes the next octave's image (heads only,
  // whole images only: see down_emit_ok)
  TRY(enter(ctx));
  if (!d_img || !d_points || !d_counters) return fail(CUSIFT_ERR_INVALID, "DetectMulti: missing data");
  if (n_images < 1 || w < 1 || h < 1 || pitch < w || max_pts < 1)
    return fail(CUSIFT_ERR_INVALID, "DetectMulti: bad geometry");
  if (!detect_fused_ok(d_img, w, h, pitch, img_stride))
    return fail(CUSIFT_ERR_INVALID, "DetectMulti: needs 16-byte aligned rows (pitch %% 4 == 0), w >= 4, h >= 3, image < 2 GiB");
  float taps[8 * 16];
  laplace_taps_table(init_blur, taps);
  LaplaceTapsPk TP;
  for (int q = 0; q < kNumLevels / 2; ++q)
    for (int j = 0; j < 5; ++j) {
      TP.k[q][j].x = taps[16 * (2 * q) + j];
      TP.k[q][j].y = taps[16 * (2 * q + 1) + j];
    }
  FindParams P;
  find_params(P, peak_thresh, edge_thresh, subsampling);
  const int rows_total = cy_end - cy_begin;
  const int strips = idiv_up(w, 240);  // kDetStrip
  // Chunk height.  A chunk of r centre rows costs r + 2 blurred rows (+ an 8-row window fill), so tall chunks waste
  // the least arithmetic -- but the launch ends with a tail in which the last chunks run on a part-empty chip, and
  // that tail grows with r.  Minimising (r + c)/r * work + k * r gives r ~ sqrt(work): r = coef * sqrt(rows * strips *
  // images).  Measured on MI355X, 64 x 1080p (tools/ab_detect_rows.sh, profiles/r02_ab/): a launch that has the GPU to
  // itself was fastest at coef 0.022-0.035 with the round-1 kernel and is at 0.05 since the candidates are refined in
  // batches (0.03: 0.873 ms, 0.04: 0.862, 0.05: 0.850, 0.07: 0.883; a chunk's fill and its two extra rows weigh more
  // now that a row with a candidate no longer costs 3,300 cycles); with consecutive batches on
  // several streams -- the throughput mode -- the other batches' kernels fill the tail and taller chunks win: two
  // streams 0.05 -> 1.451, 0.08 -> 1.414 ms per step; four streams 0.05 -> 1.414, 0.08 -> 1.370, 0.1 -> 1.379,
  // 0.13 -> 1.382.  The caller says which case it is (cusift_params.concurrent_batches).
  int rows = detect_rows(ctx, rows_total, strips, n_images, concurrent);
  if (down) {
    if (!heads || rw.row0 != 0 || rw.hg != h || cy_begin != 0 || cy_end != h || !down->dst || h / 2 < 1 || w / 2 < 1 ||
        (down->pitch % 2) != 0 || (down->stride % 2) != 0 || (((uintptr_t)down->dst) % 8) != 0 || down->pitch < w / 2)
      return fail(CUSIFT_ERR_INVALID, "DetectMulti: the next octave can only be emitted from a whole image into a staged list, 8-byte aligned rows");
    rows = std::max(rows, 2);  // the chunk [0, rows) clipped to [1, h-1) must not be empty: it owns output row 0
  }
  // Single-wave workgroups: a workgroup's wave slots and LDS are released only when its slowest wave ends, and the
  // threshold pre-test makes the waves' run times uneven -- measured 64x1080p, r = 16: 4 waves per workgroup 0.693 ms,
  // 2: 0.645 ms, 1: 0.630 ms (tools/probe_rows.py with CUSIFT_DETECT_WAVES).
  int wpb = 1;
  if (ctx->knobs.detect_waves > 0) wpb = std::min(4, ctx->knobs.detect_waves);  // experiments only
  dim3 grid(strips, idiv_up(idiv_up(rows_total, rows), wpb), n_images);
  const size_t cube_bytes = (size_t)wpb * kDetectWaveLdsFloats * sizeof(float);  // the wave's candidate list
  // levels 0 and 1 both identity (initBlur >= their sigma)?  then the kernel passes them through
  bool ident0 = true;
  for (int lv = 0; lv < 2; ++lv)
    for (int j = 0; j < 9; ++j) ident0 = ident0 && (taps[16 * lv + j] == (j == kBlurRadius ? 1.0f : 0.0f));
  StageTimer t(ctx, CUSIFT_STAGE_DETECT);
  constexpr int kWhole = (int)sizeof(cusift_point);
  const bool ident = ident0 && !ctx->knobs.no_ident;
  auto kernel = down ? (ident ? detect_fused_kernel<true, kStagedRecBytes, true> : detect_fused_kernel<false, kStagedRecBytes, true>)
              : heads ? (ident ? detect_fused_kernel<true, kStagedRecBytes, false> : detect_fused_kernel<false, kStagedRecBytes, false>)
                      : (ident ? detect_fused_kernel<true, kWhole, false> : detect_fused_kernel<false, kWhole, false>);
  DownOut dn;
  memset(&dn, 0, sizeof(dn));
  if (down) dn = *down;
  hipLaunchKernelGGL(kernel, grid, dim3(64 * wpb), cube_bytes, side ? ctx->side : ctx->stream, d_img, w, h, pitch,
                     (long)img_stride, d_points, max_pts, d_counters, rows, TP, P, rw, cy_begin, cy_end, dn);
  return check_launch("detect_multi");
}

// The fused detection of several octaves of a batch in ONE launch (detect_multi_kernel): whole images, general taps,
// keypoint HEADS to a staging list per octave.  `octaves` in launch order (largest first: the small ones fill its tail).

int detect_multi_impl(cusift_ctx *ctx, const MultiOctave *octaves, int n_octaves, float peak_thresh,
                             float edge_thresh, int max_pts, int n_images, int concurrent, unsigned int *d_queue) {
  TRY(enter(ctx));
  if (n_octaves < 1 || n_octaves > kMaxMultiOctaves) return fail(CUSIFT_ERR_INVALID, "DetectMulti (octaves): 1..%d octaves", kMaxMultiOctaves);
  DetectTable tab;
  memset(&tab, 0, sizeof(tab));
  tab.n = n_octaves;
  long blocks = 0;
  for (int k = 0; k < n_octaves; ++k) {
    const MultiOctave &m = octaves[k];
    if (!detect_fused_ok(m.img, m.w, m.h, m.pitch, m.img_stride))
      return fail(CUSIFT_ERR_INVALID, "DetectMulti (octaves): octave %d needs 16-byte aligned rows, w >= 4, h >= 3", k);
    DetectOctave &o = tab.o[k];
    float taps[8 * 16];
    laplace_taps_table(m.init_blur, taps);
    for (int q = 0; q < kNumLevels / 2; ++q)
      for (int j = 0; j < 5; ++j) {
        o.T.k[q][j].x = taps[16 * (2 * q) + j];
        o.T.k[q][j].y = taps[16 * (2 * q + 1) + j];
      }
    bool ident0 = !ctx->knobs.no_ident;  // levels 0 and 1 both identity (initBlur >= their sigma)?
    for (int lv = 0; lv < 2; ++lv)
      for (int j = 0; j < 9; ++j) ident0 = ident0 && (taps[16 * lv + j] == (j == kBlurRadius ? 1.0f : 0.0f));
    o.ident = ident0 ? 1 : 0;
    find_params(o.P, peak_thresh, edge_thresh, m.subsampling);
    o.img = m.img;
    o.img_stride = (long)m.img_stride;
    o.lists = reinterpret_cast<char *>(m.lists);
    o.counters = m.counters;
    o.w = m.w;
    o.h = m.h;
    o.pitch = m.pitch;
    o.row0 = m.hg < 0 ? 0 : m.row0;
    o.hg = m.hg < 0 ? m.h : m.hg;
    o.cy_begin = m.hg < 0 ? 0 : m.cy_begin;
    o.cy_end = m.hg < 0 ? m.h : m.cy_end;
    const int rows_total = o.cy_end - o.cy_begin;
    o.strips = idiv_up(m.w, 240);  // kDetStrip
    o.rows_per_wave = detect_rows(ctx, rows_total, o.strips, n_images, concurrent);
    o.chunks = idiv_up(rows_total, o.rows_per_wave);
    o.first_block = (int)blocks;
    blocks += (long)o.strips * o.chunks * n_images;
  }
  if (blocks > 0x7fffffffL) return fail(CUSIFT_ERR_INVALID, "DetectMulti (octaves): too many workgroups");
  StageTimer t(ctx, CUSIFT_STAGE_DETECT);
  hipLaunchKernelGGL(detect_multi_kernel<kStagedRecBytes>, dim3((unsigned int)blocks), dim3(64),
                     kDetectWaveLdsFloats * sizeof(float), ctx->stream, tab, max_pts, d_queue);
  return check_launch("detect_multi (octaves)");
}

extern "C" int cusift_detect_multi(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, size_t img_stride,
                                   float init_blur, float peak_thresh, float edge_thresh, float subsampling,
                                   cusift_point *d_points, int max_pts, unsigned int *d_counters, int n_images) {
  return detect_impl(ctx, d_img, w, h, pitch, img_stride, init_blur, peak_thresh, edge_thresh, subsampling, d_points,
                     max_pts, d_counters, n_images, RowWindow{0, h}, 0, h);
}

extern "C" int cusift_detect_multi_down(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, size_t img_stride,
                                        float init_blur, float peak_thresh, float edge_thresh, float subsampling,
                                        void *d_heads, int max_pts, unsigned int *d_counters, int n_images,
                                        float *d_next, int next_pitch, size_t next_stride, float variance) {
  if (!d_next) return fail(CUSIFT_ERR_INVALID, "DetectMulti (down): d_next is NULL");
  if (!(variance > 0.0f)) return fail(CUSIFT_ERR_INVALID, "ScaleDown: variance must be > 0");
  if (n_images > 1 && next_stride < (size_t)(h / 2) * next_pitch)
    return fail(CUSIFT_ERR_INVALID, "DetectMulti (down): next_stride too small");
  DownOut dn;
  dn.dst = d_next;
  dn.pitch = next_pitch;
  dn.stride = (long)next_stride;
  scale_down_taps(dn.T, variance);
  return detect_impl(ctx, d_img, w, h, pitch, img_stride, init_blur, peak_thresh, edge_thresh, subsampling,
                     reinterpret_cast<cusift_point *>(d_heads), max_pts, d_counters, n_images, RowWindow{0, h}, 0, h, 1, true,
                     false, &dn);
}

extern "C" int cusift_detect_band(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, int row0, int h_global,
                                  int cy_begin, int cy_end, float init_blur, float peak_thresh, float edge_thresh,
                                  float subsampling, cusift_point *d_points, int max_pts, unsigned int *d_counter) {
  if (row0 < 0 || h < 1 || row0 + h > h_global || cy_begin < row0 || cy_end > row0 + h || cy_end <= cy_begin)
    return fail(CUSIFT_ERR_INVALID, "Detect (band): bad row geometry");
  // centres need 4 blur rows + 1 extremum row of true data on either side, unless the band ends at the image border
  if ((row0 > 0 && cy_begin - row0 < 5) || (row0 + h < h_global && row0 + h - cy_end < 5))
    return fail(CUSIFT_ERR_INVALID, "Detect (band): centres [%d,%d) need 5 halo rows inside the band [%d,%d)", cy_begin,
                cy_end, row0, row0 + h);
  return detect_impl(ctx, d_img, w, h, pitch, (size_t)h * pitch, init_blur, peak_thresh, edge_thresh, subsampling,
                     d_points, max_pts, d_counter, 1, RowWindow{row0, h_global}, cy_begin, cy_end);
}

int keypoint_grid_x(int max_pts, int n_images) {
  // persistent grid: enough waves to fill 256 CUs x 32 wave slots, never more than max_pts per image
  int per_image = std::max(1, (256 * 32 * 2) / std::max(1, n_images));
  return std::max(1, std::min(max_pts, std::min(per_image, 4096)));
}

int orientations_impl(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, size_t img_stride,
                             cusift_point *d_points, int max_pts, const unsigned int *d_first,
                             const unsigned int *d_counters, int tex_frac_bits, int n_images, RowWindow rw) {
  TRY(enter(ctx));
  if (!d_img || !d_points || !d_counters) return fail(CUSIFT_ERR_INVALID, "ComputeOrientations: missing data");
  if (n_images < 1 || w < 1 || h < 1 || pitch < w || max_pts < 1)
    return fail(CUSIFT_ERR_INVALID, "ComputeOrientations: bad geometry");
  float q, inv_q;
  frac_consts(tex_frac_bits, q, inv_q);
  dim3 grid(keypoint_grid_x(max_pts, n_images), n_images);
  StageTimer t(ctx, CUSIFT_STAGE_ORIENT);
  hipLaunchKernelGGL(orientations_kernel, grid, dim3(64), 0, ctx->stream, d_img, w, h, pitch, (long)img_stride,
                     d_points, max_pts, d_first, d_counters, q, inv_q, rw);
  return check_launch("compute_orientations");
}

int descriptors_impl(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, size_t img_stride,
                            cusift_point *d_points, int max_pts, const unsigned int *d_first,
                            const unsigned int *d_counters, float subsampling, int tex_frac_bits, int n_images,
                            RowWindow rw, int root_sift, unsigned int *d_flags) {
  TRY(enter(ctx));
  if (!d_img || !d_points || !d_counters) return fail(CUSIFT_ERR_INVALID, "ExtractSiftDescriptors: missing data");
  if (n_images < 1 || w < 1 || h < 1 || pitch < w || max_pts < 1)
    return fail(CUSIFT_ERR_INVALID, "ExtractSiftDescriptors: bad geometry");
  float q, inv_q;
  frac_consts(tex_frac_bits, q, inv_q);
  dim3 grid(keypoint_grid_x(max_pts, n_images), n_images);
  StageTimer t(ctx, CUSIFT_STAGE_DESCR);
  hipLaunchKernelGGL(descriptors_kernel, grid, dim3(64), 0, ctx->stream, d_img, w, h, pitch, (long)img_stride,
                     d_points, max_pts, d_first, d_counters, subsampling, q, inv_q, rw, root_sift, d_flags);
  return check_launch("extract_descriptors");
}

extern "C" int cusift_compute_orientations(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch,
                                           size_t img_stride, cusift_point *d_points, int max_pts,
                                           const unsigned int *d_first, const unsigned int *d_counters,
                                           int tex_frac_bits, int n_images) {
  return orientations_impl(ctx, d_img, w, h, pitch, img_stride, d_points, max_pts, d_first, d_counters, tex_frac_bits,
                           n_images, RowWindow{0, h});
}

extern "C" int cusift_extract_descriptors(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch,
                                          size_t img_stride, cusift_point *d_points, int max_pts,
                                          const unsigned int *d_first, const unsigned int *d_counters,
                                          float subsampling, int tex_frac_bits, int n_images) {
  return descriptors_impl(ctx, d_img, w, h, pitch, img_stride, d_points, max_pts, d_first, d_counters, subsampling,
                          tex_frac_bits, n_images, RowWindow{0, h});
}

extern "C" int cusift_describe_band(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, int row0,
                                    int h_global, cusift_point *d_points, int max_pts, const unsigned int *d_first,
                                    const unsigned int *d_counter, float subsampling, int tex_frac_bits,
                                    int root_sift, unsigned int *d_flags) {
  if (row0 < 0 || h < 1 || row0 + h > h_global) return fail(CUSIFT_ERR_INVALID, "Describe (band): bad row geometry");
  const RowWindow rw{row0, h_global};
  TRY(orientations_impl(ctx, d_img, w, h, pitch, (size_t)h * pitch, d_points, max_pts, d_first, d_counter,
                        tex_frac_bits, 1, rw));
  return descriptors_impl(ctx, d_img, w, h, pitch, (size_t)h * pitch, d_points, max_pts, d_first, d_counter,
                          subsampling, tex_frac_bits, 1, rw, root_sift, d_flags);
}

// Detection + description of SEVERAL octave bands of one strip-tiled image: one detection launch for all of them
// (detect_multi_kernel), a join, one description launch (describe_bands_kernel) -- instead of a counter copy and three
// launches per octave.  bands[0] is the finest octave, bands[k] the next coarser (subsampling doubles).  On entry
// *d_counter counts the keypoints already in d_points (a root's collapsed coarse octaves, described): they stay where
// they are and the bands' keypoints follow, coarsest band first -- the list order of cusift_tiled_process.
extern "C" int cusift_extract_bands(cusift_ctx *ctx, const cusift_band *bands, int n_bands, float peak_thresh,
                                    float edge_thresh, cusift_point *d_points, int max_pts, unsigned int *d_counter,
                                    int tex_frac_bits, int root_sift, unsigned int *d_flags) {
  TRY(enter(ctx));
  if (!bands || !d_points || !d_counter) return fail(CUSIFT_ERR_INVALID, "ExtractBands: missing data");
  if (n_bands < 1 || n_bands > kMaxMultiOctaves || max_pts < 1)
    return fail(CUSIFT_ERR_INVALID, "ExtractBands: 1..%d bands", kMaxMultiOctaves);
  for (int k = 0; k < n_bands; ++k) {
    const cusift_band &b = bands[k];
    if (!b.d_img || b.row0 < 0 || b.h < 1 || b.row0 + b.h > b.h_global || b.cy_begin < b.row0 || b.cy_end > b.row0 + b.h ||
        b.cy_end <= b.cy_begin)
      return fail(CUSIFT_ERR_INVALID, "ExtractBands: band %d has bad row geometry", k);
    // centres need 4 blur rows + 1 extremum row of true data on either side, unless the band ends at the image border
    if ((b.row0 > 0 && b.cy_begin - b.row0 < 5) || (b.row0 + b.h < b.h_global && b.row0 + b.h - b.cy_end < 5))
      return fail(CUSIFT_ERR_INVALID, "ExtractBands: band %d: centres [%d,%d) need 5 halo rows inside [%d,%d)", k, b.cy_begin,
                  b.cy_end, b.row0, b.row0 + b.h);
    if (k > 0 && !(b.subsampling == 2.0f * bands[k - 1].subsampling))
      return fail(CUSIFT_ERR_INVALID, "ExtractBands: band %d is not the next octave of band %d", k, k - 1);
  }
  // scratch in the arena: [counters of the bands | running sums | a list of heads per band]; a no-op after
  // cusift_ctx_reserve_bands (the tiled driver reserves at create: growing here synchronises the stream and frees the
  // old arena in the middle of a rank's collective sequence)
  const size_t list_bytes = (size_t)max_pts * kStagedRecBytes;
  const size_t lists_off = 512;
  TRY(ensure_arena(ctx, bands_arena_bytes(n_bands, max_pts)));
  ctx->seg_clean_ptr = nullptr;  // (this driver lays the arena out its own way)
  unsigned int *seg_counts = (unsigned int *)ctx->arena;
  unsigned int *seg_end = seg_counts + 32;
  HIP_TRY(hipMemsetAsync(seg_counts, 0, 128, ctx->stream));
  MultiOctave mo[kMaxMultiOctaves];
  OctaveTable T;
  BandWindows BW;
  SegmentTable G;
  memset(&T, 0, sizeof(T));
  memset(&BW, 0, sizeof(BW));
  memset(&G, 0, sizeof(G));
  T.n_oct = n_bands;
  G.n_seg = n_bands + 1;
  G.base[0] = nullptr;  // what is in the list already: in place
  G.count[0] = d_counter;
  for (int k = 0; k < n_bands; ++k) {
    const cusift_band &b = bands[k];
    cusift_point *list = reinterpret_cast<cusift_point *>(ctx->arena + lists_off + (size_t)k * list_bytes);
    mo[k] = MultiOctave{b.d_img, b.w, b.h, b.pitch, (size_t)b.h * b.pitch, b.init_blur, b.subsampling, list, seg_counts + k,
                        b.row0, b.h_global, b.cy_begin, b.cy_end};
    T.base[k] = b.d_img;
    T.stride[k] = 0;
    T.w[k] = b.w;
    T.h[k] = b.h;
    T.pitch[k] = b.pitch;
    T.sub[k] = b.subsampling;
    BW.row0[k] = b.row0;
    BW.hg[k] = b.h_global;
    const int r = n_bands - k;  // list order: coarsest band first, behind segment 0
    G.base[r] = reinterpret_cast<const char *>(list);
    G.count[r] = seg_counts + k;
  }
  TRY(detect_multi_impl(ctx, mo, n_bands, peak_thresh, edge_thresh, max_pts, 1, 1, nullptr));
  hipLaunchKernelGGL(join_counts_kernel, dim3(1), dim3(256), 0, ctx->stream, d_counter, G, seg_end, 1, max_pts, ctx->d_queue, 0);
  TRY(check_launch("join_counts"));
  float q, inv_q;
  frac_consts(tex_frac_bits, q, inv_q);
  StageTimer t(ctx, CUSIFT_STAGE_DESCRIBE_ALL);
  hipLaunchKernelGGL(describe_bands_kernel, dim3(keypoint_grid_x(max_pts, 1)), dim3(64), 0, ctx->stream, T, BW, d_points,
                     max_pts, G, (const unsigned int *)seg_end, q, inv_q, root_sift, d_flags);
  return check_launch("describe_bands");
}

extern "C" int cusift_rootsift(cusift_ctx *ctx, cusift_point *d_points, int num_pts) {
  TRY(enter(ctx));
  if (!d_points) return fail(CUSIFT_ERR_INVALID, "ConvertSiftToRootSift: missing data");
  if (num_pts <= 0) return CUSIFT_OK;
  dim3 grid(std::min(num_pts, 256 * 32));
  hipLaunchKernelGGL(rootsift_kernel, grid, dim3(64), 0, ctx->stream, d_points, num_pts);
  return check_launch("rootsift");
}

extern "C" int cusift_math_eval(cusift_ctx *ctx, int op, const float *d_a, const float *d_b, float *d_out,
                                float *d_out2, size_t n) {
  TRY(enter(ctx));
  if (op < 0 || op > 5 || !d_a || !d_out || ((op == 2 || op == 4 || op == 5) && !d_b) || ((op == 3 || op == 5) && !d_out2))
    return fail(CUSIFT_ERR_INVALID, "math_eval: bad argument");
  if (n == 0) return CUSIFT_OK;
  const unsigned int blocks = (unsigned int)std::min<size_t>((n + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(math_eval_kernel, dim3(blocks), dim3(256), 0, ctx->stream, op, d_a, d_b, d_out, d_out2, (long)n);
  return check_launch("math_eval");
}

// ------------------------------------------------------------------------------------------------
// matcher
// ------------------------------------------------------------------------------------------------
extern "C" int cusift_match(cusift_ctx *ctx, cusift_point *d_sift1, int num_pts1, const cusift_point *d_sift2,
                            int num_pts2, int distance) {
  TRY(enter(ctx));
  if (num_pts1 <= 0 || num_pts2 <= 0) return CUSIFT_OK;  // extras/matching.cu:241-242: nothing to match
  if (!d_sift1 || !d_sift2) return fail(CUSIFT_ERR_INVALID, "MatchSiftData: missing data");
  if (distance != 0 && distance != 1) return fail(CUSIFT_ERR_INVALID, "MatchSiftData: distance must be 0 or 1");
  // Column splits: aim at >= 4 workgroups per CU, keep >= 4 LDS tiles (128 columns) per split.
  const int row_blocks = idiv_up(num_pts1, 64);
  int splits = std::max(1, std::min(idiv_up(4 * ctx->num_cus, row_blocks), idiv_up(num_pts2, 128)));
  if (ctx->knobs.match_splits > 0) splits = std::min(ctx->knobs.match_splits, idiv_up(num_pts2, 32));
  splits = std::min(splits, 65535);
  // the kernel addresses a split's columns through a buffer resource with 32-bit byte offsets: a split may span at most
  // 2^31 / 588 records (3.65 M) -- more points than that force further splits
  constexpr int kMaxColsPerSplit = (int)((0x7fffffffu / sizeof(cusift_point)) / 32 * 32);
  splits = std::max(splits, idiv_up(num_pts2, kMaxColsPerSplit));
  if (splits > 65535) return fail(CUSIFT_ERR_INVALID, "MatchSiftData: too many points in image 2 (%d)", num_pts2);
  const int cols_per_split = idiv_up(idiv_up(num_pts2, splits), 32) * 32;
  splits = idiv_up(num_pts2, cols_per_split);
  const int n1_pad = row_blocks * 64;
  MatchPartial *partials = nullptr;
  if (splits > 1) {
    const size_t bytes = sizeof(MatchPartial) * (size_t)splits * n1_pad;
    if (bytes > ctx->match_scratch_bytes) {
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      ctx->scratch_gen++;
      if (ctx->match_scratch) HIP_TRY(hipFree(ctx->match_scratch));
      ctx->match_scratch = nullptr;
      ctx->match_scratch_bytes = 0;
      hipError_t e = hipMalloc((void **)&ctx->match_scratch, bytes);
      if (e != hipSuccess) return fail(CUSIFT_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
      ctx->match_scratch_bytes = bytes;
    }
    partials = ctx->match_scratch;
  }
  if (distance)
    hipLaunchKernelGGL(match_kernel<true>, dim3(row_blocks, splits), dim3(256), 0, ctx->stream, d_sift1, num_pts1,
                       d_sift2, num_pts2, cols_per_split, partials, n1_pad);
  else
    hipLaunchKernelGGL(match_kernel<false>, dim3(row_blocks, splits), dim3(256), 0, ctx->stream, d_sift1, num_pts1,
                       d_sift2, num_pts2, cols_per_split, partials, n1_pad);
  if (splits > 1)
    hipLaunchKernelGGL(match_merge_kernel, dim3(idiv_up(num_pts1, 256)), dim3(256), 0, ctx->stream, d_sift1, num_pts1,
                       d_sift2, num_pts2, distance, partials, n1_pad, splits);
  return check_launch("match");
}

// ------------------------------------------------------------------------------------------------
// RANSAC homography
// ------------------------------------------------------------------------------------------------
extern "C" int cusift_find_homography(cusift_ctx *ctx, const cusift_point *d_sift, int num_pts, const int *h_rand_pts,
                                      int num_loops, float thresh, float h_homography[9], int *num_matches,
                                      float *h_all_homo, int *h_all_counts) {
  TRY(enter(ctx));
  if (!h_homography || !num_matches) return fail(CUSIFT_ERR_INVALID, "FindHomography: NULL output");
  static const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // extras/homography.cu:184-187
  memcpy(h_homography, ident, sizeof(ident));
  *num_matches = 0;
  if (!d_sift || !h_rand_pts) return fail(CUSIFT_ERR_INVALID, "FindHomography: missing data");
  if (num_pts < 1 || num_loops < 1) return fail(CUSIFT_ERR_INVALID, "FindHomography: num_pts and num_loops must be >= 1");
  for (long i = 0; i < 4L * num_loops; ++i)
    if (h_rand_pts[i] < 0 || h_rand_pts[i] >= num_pts)
      return fail(CUSIFT_ERR_INVALID, "FindHomography: sample index %d out of range [0, %d)", h_rand_pts[i], num_pts);
  const size_t coord_b = align_up_sz(sizeof(float) * 4 * (size_t)num_pts, 256);
  const size_t rand_b = align_up_sz(sizeof(int) * 4 * (size_t)num_loops, 256);
  const size_t homo_b = align_up_sz(sizeof(float) * 8 * (size_t)num_loops, 256);
  const size_t cnt_b = align_up_sz(sizeof(int) * (size_t)num_loops, 256);
  const size_t bytes = coord_b + rand_b + homo_b + cnt_b;
  if (bytes > ctx->homo_scratch_bytes) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->homo_scratch) HIP_TRY(hipFree(ctx->homo_scratch));
    ctx->homo_scratch = nullptr;
    ctx->homo_scratch_bytes = 0;
    hipError_t e = hipMalloc((void **)&ctx->homo_scratch, bytes);
    if (e != hipSuccess) return fail(CUSIFT_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    ctx->homo_scratch_bytes = bytes;
  }
  float *d_coord = (float *)ctx->homo_scratch;
  int *d_rand = (int *)(ctx->homo_scratch + coord_b);
  float *d_homo = (float *)(ctx->homo_scratch + coord_b + rand_b);
  int *d_counts = (int *)(ctx->homo_scratch + coord_b + rand_b + homo_b);
  HIP_TRY(hipMemcpyAsync(d_rand, h_rand_pts, sizeof(int) * 4 * (size_t)num_loops, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(homography_gather_kernel, dim3(idiv_up(num_pts, 256)), dim3(256), 0, ctx->stream, d_sift, num_pts,
                     d_coord);
  hipLaunchKernelGGL(homography_solve_kernel, dim3(idiv_up(num_loops, 64)), dim3(64), 0, ctx->stream, d_coord, num_pts,
                     d_rand, num_loops, d_homo);
  hipLaunchKernelGGL(homography_test_kernel, dim3(num_loops), dim3(64), 0, ctx->stream, d_coord, num_pts, d_homo,
                     num_loops, thresh * thresh, d_counts);
  TRY(check_launch("find_homography"));
  std::vector<int> counts((size_t)num_loops);
  std::vector<float> homo(8 * (size_t)num_loops);
  HIP_TRY(hipMemcpyAsync(counts.data(), d_counts, sizeof(int) * (size_t)num_loops, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(homo.data(), d_homo, sizeof(float) * 8 * (size_t)num_loops, hipMemcpyDeviceToHost,
                         ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  int best = -1, best_count = -1;  // extras/homography.cu:249-254: first maximum
  for (int i = 0; i < num_loops; ++i)
    if (counts[i] > best_count) {
      best_count = counts[i];
      best = i;
    }
  *num_matches = best_count;
  for (int j = 0; j < 8; ++j) h_homography[j] = homo[(size_t)j * num_loops + best];
  if (h_all_homo) memcpy(h_all_homo, homo.data(), sizeof(float) * homo.size());
  if (h_all_counts) memcpy(h_all_counts, counts.data(), sizeof(int) * counts.size());
  return CUSIFT_OK;
}

extern "C" int cusift_memcpy2d_d2h(cusift_ctx *ctx, void *h_dst, size_t dst_pitch, const void *d_src,
                                   size_t src_pitch, size_t width_bytes, size_t rows) {
  if (!ctx || !h_dst || !d_src) return fail(CUSIFT_ERR_INVALID, "NULL argument");
  TRY(enter(ctx));
  if (rows == 0 || width_bytes == 0) return CUSIFT_OK;
  HIP_TRY(hipMemcpy2DAsync(h_dst, dst_pitch, d_src, src_pitch, width_bytes, rows, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return CUSIFT_OK;
}

// Canonical order of extracted records (host side).  Octave blocks arrive coarsest first, but inside an octave the
// order is that of an atomic append -- racy here as in the reference (atomicInc, cuSIFT_D.cu:512).  Callers that
// need run-to-run identical ARRAYS (not just sets) sort: octave (coarsest first, as emitted), then y, x, scale; the
// remaining fields break exact ties, so equal sets give equal arrays.
extern "C" int cusift_sort_points_host(cusift_point *h_points, int num_pts) {
  if (num_pts <= 0) return CUSIFT_OK;
  if (!h_points) return fail(CUSIFT_ERR_INVALID, "sort: h_points is NULL");
  // every key is compared as a BIT PATTERN mapped to an unsigned integer that orders like the float (negative values
  // reversed, then offset): a strict weak ordering whatever the values -- a NaN location or scale (1/0 in the
  // refinement of a degenerate DoG neighbourhood) sorts after every number instead of breaking the sort's contract
  auto key = [](float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  };
  std::stable_sort(h_points, h_points + num_pts, [&key](const cusift_point &a, const cusift_point &b) {
    const uint32_t ka[] = {~key(a.subsampling), key(a.coords2D[1]), key(a.coords2D[0]), key(a.scale)};
    const uint32_t kb[] = {~key(b.subsampling), key(b.coords2D[1]), key(b.coords2D[0]), key(b.scale)};
    for (int i = 0; i < 4; ++i)
      if (ka[i] != kb[i]) return ka[i] < kb[i];
    // exact ties of location and scale (two scales of one pixel refined onto the same point): the rest of the
    // extracted fields, bytewise -- a total order even where an orientation is NaN (flat patch)
    return memcmp(&a.sharpness, &b.sharpness, 3 * sizeof(float)) < 0;  // sharpness, edgeness, orientation
  });
  return CUSIFT_OK;
}

extern "C" int cusift_pack_points(cusift_ctx *ctx, const cusift_point *d_points, const unsigned int *d_counters,
                                  int n_images, int max_pts, cusift_point *d_packed, size_t capacity,
                                  unsigned int *d_offsets) {
  TRY(enter(ctx));
  if (!d_points || !d_counters || !d_packed) return fail(CUSIFT_ERR_INVALID, "pack: missing data");
  if (n_images < 1 || n_images > kMaxFlatImages || max_pts < 1)
    return fail(CUSIFT_ERR_INVALID, "pack: n_images must be in [1, %d]", kMaxFlatImages);
  const size_t cap = std::min(capacity, (size_t)0xffffffffu);
  dim3 grid((unsigned int)std::max<size_t>(1, std::min<size_t>(std::max<size_t>(cap, 1), 256 * 32)));
  hipLaunchKernelGGL(pack_points_kernel, grid, dim3(64), 0, ctx->stream, d_points, d_counters, n_images, max_pts,
                     d_packed, (unsigned int)cap, d_offsets);
  return check_launch("pack_points");
}

static_assert(sizeof(cusift_compact_point) == 160, "the compact wire record is 160 bytes");

extern "C" int cusift_pack_points_compact(cusift_ctx *ctx, const cusift_point *d_points, const unsigned int *d_counters,
                                          int n_images, int max_pts, cusift_compact_point *d_packed, size_t capacity,
                                          unsigned int *d_offsets) {
  TRY(enter(ctx));
  if (!d_points || !d_counters || !d_packed) return fail(CUSIFT_ERR_INVALID, "pack (compact): missing data");
  if (n_images < 1 || n_images > kMaxFlatImages || max_pts < 1)
    return fail(CUSIFT_ERR_INVALID, "pack (compact): n_images must be in [1, %d]", kMaxFlatImages);
  const size_t cap = std::min(capacity, (size_t)0xffffffffu);
  dim3 grid((unsigned int)std::max<size_t>(1, std::min<size_t>(std::max<size_t>(cap, 1), 256 * 32)));
  hipLaunchKernelGGL(pack_points_compact_kernel, grid, dim3(64), 0, ctx->stream, d_points, d_counters, n_images, max_pts,
                     d_packed, (unsigned int)cap, d_offsets);
  return check_launch("pack_points_compact");
}

static_assert(sizeof(cusift_trimmed_point) == 540, "the trimmed wire record is 135 floats");

extern "C" int cusift_pack_points_trimmed(cusift_ctx *ctx, const cusift_point *d_points, const unsigned int *d_counters,
                                          int n_images, int max_pts, cusift_trimmed_point *d_packed, size_t capacity,
                                          unsigned int *d_offsets) {
  TRY(enter(ctx));
  if (!d_points || !d_counters || !d_packed) return fail(CUSIFT_ERR_INVALID, "pack (trimmed): missing data");
  if (n_images < 1 || n_images > kMaxFlatImages || max_pts < 1)
    return fail(CUSIFT_ERR_INVALID, "pack (trimmed): n_images must be in [1, %d]", kMaxFlatImages);
  const size_t cap = std::min(capacity, (size_t)0xffffffffu);
  dim3 grid((unsigned int)std::max<size_t>(1, std::min<size_t>(std::max<size_t>(cap, 1), 256 * 32)));
  hipLaunchKernelGGL(pack_points_trimmed_kernel, grid, dim3(64), 0, ctx->stream, d_points, d_counters, n_images, max_pts,
                     d_packed, (unsigned int)cap, d_offsets);
  return check_launch("pack_points_trimmed");
}

extern "C" int cusift_expand_trimmed(cusift_ctx *ctx, const cusift_trimmed_point *d_trimmed, size_t n,
                                     cusift_point *d_points) {
  TRY(enter(ctx));
  if (n == 0) return CUSIFT_OK;
  if (!d_trimmed || !d_points) return fail(CUSIFT_ERR_INVALID, "expand (trimmed): NULL argument");
  dim3 grid((unsigned int)std::min<size_t>(n, 256 * 32));
  hipLaunchKernelGGL(expand_trimmed_kernel, grid, dim3(64), 0, ctx->stream, d_trimmed, n, d_points);
  return check_launch("expand_trimmed");
}

extern "C" int cusift_expand_trimmed_host(const cusift_trimmed_point *h_trimmed, size_t n, cusift_point *h_points) {
  if (n == 0) return CUSIFT_OK;
  if (!h_trimmed || !h_points) return fail(CUSIFT_ERR_INVALID, "expand (trimmed): NULL argument");
  for (size_t i = 0; i < n; ++i) {
    const cusift_trimmed_point &t = h_trimmed[i];
    cusift_point &p = h_points[i];
    memset(&p, 0, sizeof(p));
    memcpy(&p.coords2D[0], &t.coords2D[0], 6 * sizeof(float));  // coords2D, scale, sharpness, edgeness, orientation
    p.subsampling = t.subsampling;
    memcpy(p.data, t.data, sizeof(p.data));
  }
  return CUSIFT_OK;
}

extern "C" int cusift_expand_points_host(const cusift_compact_point *h_compact, size_t n, cusift_point *h_points) {
  if (n == 0) return CUSIFT_OK;
  if (!h_compact || !h_points) return fail(CUSIFT_ERR_INVALID, "expand: NULL argument");
  for (size_t i = 0; i < n; ++i) {
    const cusift_compact_point &c = h_compact[i];
    cusift_point &p = h_points[i];
    memset(&p, 0, sizeof(p));
    p.coords2D[0] = c.coords2D[0];
    p.coords2D[1] = c.coords2D[1];
    p.scale = c.scale;
    p.sharpness = c.sharpness;
    p.edgeness = c.edgeness;
    p.orientation = c.orientation;
    p.subsampling = c.subsampling;
    for (int k = 0; k < 128; ++k) p.data[k] = (float)c.q[k] * c.desc_step;
  }
  return CUSIFT_OK;
}

