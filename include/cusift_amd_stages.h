/*
 * cusift_amd_stages.h -- stage entry points (one per kernel of the reference), the caller-side front-end, diagnostics.
 * Part of the C ABI of libcusift_amd.so; conventions and the map of the four headers: cusift_amd.h.
 */
#ifndef CUSIFT_AMD_STAGES_H
#define CUSIFT_AMD_STAGES_H

#include "cusift_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* How many extractions of this context ran octave 0's detection on the context's second stream (see
 * CUSIFT_POLICY_SIDE_STREAM below). */
unsigned long cusift_ctx_forks(cusift_ctx *ctx);
/* Per-stage GPU timing with HIP events on the context's stream (TimerGPU, cutils.h:94-114, used at
 * cuSIFT.cu:64,177,208,238,249).  Stages: 0 ScaleDown, 1 LaplaceMulti, 2 FindPointsMulti,
 * 3 ComputeOrientations, 4 ExtractSiftDescriptors, 5 whole extract call, 6 fused detection, 7 orientation+descriptor of all octaves in one launch.  Accumulates
 * milliseconds and launch counts until reset.  cusift_ctx_timing_read blocks. */
enum { CUSIFT_STAGE_SCALEDOWN = 0, CUSIFT_STAGE_LAPLACE = 1, CUSIFT_STAGE_FINDPOINTS = 2,
       CUSIFT_STAGE_ORIENT = 3, CUSIFT_STAGE_DESCR = 4, CUSIFT_STAGE_TOTAL = 5, CUSIFT_STAGE_DETECT = 6,
       CUSIFT_STAGE_DESCRIBE_ALL = 7, CUSIFT_NUM_STAGES = 8 };
int cusift_ctx_timing_enable(cusift_ctx *ctx, int on);
int cusift_ctx_timing_read(cusift_ctx *ctx, float ms[CUSIFT_NUM_STAGES], int launches[CUSIFT_NUM_STAGES]);
int cusift_ctx_timing_reset(cusift_ctx *ctx);

/* Introspection for DESIGN.md / tuning: resident workgroups per CU of a named kernel ("detect_fused",
 * "laplace_multi", "find_points", "scale_down", "describe_all", "orientations", "descriptors") according to
 * hipOccupancyMaxActiveBlocksPerMultiprocessor. */
int cusift_kernel_occupancy(const char *kernel, int *blocks_per_cu, int *threads_per_block);

/* ---- caller-side front-end on the device ------------------------------------ */
/* The reference's callers decode 8-bit images with OpenCV, convertTo(CV_32FC1) and optionally
 * cv::GaussianBlur(img, img, Size(3,3), 0.5) on the HOST, then upload 4 bytes per pixel (main.cpp:300-318,
 * test/detector.cpp:19-27).  These do the same after uploading 1 byte per pixel.
 * cusift_image_u8_h2d: dense 8-bit host rows (w bytes) -> pitched float device image (exact conversion); blocking.
 * cusift_u8_to_f32:    the conversion alone on device-resident 8-bit images (batch form); asynchronous.
 * cusift_gaussian3x3:  3x3 separable Gaussian as cv::GaussianBlur(Size(3,3), sigma) evaluates its float path
 *                      (symmetric small filters, BORDER_REFLECT_101); d_dst != d_src; asynchronous. */
int cusift_image_u8_h2d(cusift_ctx *ctx, float *d_dst, int dst_pitch, const unsigned char *h_src, int w, int h);
int cusift_u8_to_f32(cusift_ctx *ctx, float *d_dst, int dst_pitch, size_t dst_stride, const unsigned char *d_src,
                     int w, int h, int src_pitch_bytes, size_t src_stride_bytes, int n_images);
int cusift_gaussian3x3(cusift_ctx *ctx, float *d_dst, int dst_pitch, size_t dst_stride, const float *d_src, int w,
                       int h, int src_pitch, size_t src_stride, int n_images, float sigma);

/* ---- stage entry points (the reference's launch wrappers) --------------------------------- */
/* The ScaleDown CHAIN of ExtractSiftLoop (cuSIFT.cu:175-192) -- level k (w >> k, h >> k) from level k - 1 for
 * k = 1 .. n_levels <= 4 -- in ONE launch: same pixels, bit for bit, as n_levels calls of cusift_scale_down.  What the
 * drivers use for small calls (one 1080p frame: four dependent launches of 6-10 us become one of ~20); it re-reads the
 * source 2.9 times, so it is not the way to scale down a large batch.  d_levels[k - 1], pitches[k - 1], strides[k - 1]
 * (host arrays): level k's device buffer, floats per row and floats between images. */
int cusift_scale_down_levels(cusift_ctx *ctx, const float *d_src, int w, int h, int src_pitch, size_t src_stride,
                             float *const *d_levels, const int *pitches, const size_t *strides, int n_levels,
                             int n_images, float variance);
/* SiftData::LaplaceMulti, cuSIFT.cu:399-422 + LaplaceMulti_D cuSIFT_D.cu:525-553: 8 blurs + 7 DoG
 * planes, planar [7][h][pitch] per image (`dog_stride` floats between images, >= 7*h*pitch). */
int cusift_laplace_multi(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, size_t img_stride,
                         float init_blur, float *d_dog, size_t dog_stride, int n_images);
/* The 8 x 9 tap table LaplaceMulti uploads (cuSIFT.cu:400-413), row stride 16 floats; host-only. */
int cusift_laplace_taps(float init_blur, float taps[8 * 16]);
/* SiftData::FindPointsMulti, cuSIFT.cu:424-455 + FindPointsMulti_D cuSIFT_D.cu:402-523.
 * Appends at d_points[i*max_pts + atomicAdd(d_counters[i])]; overflow is dropped, counters keep counting. */
int cusift_find_points_multi(cusift_ctx *ctx, const float *d_dog, int w, int h, int pitch, size_t dog_stride,
                             float peak_thresh, float edge_thresh, float subsampling, cusift_point *d_points,
                             int max_pts, unsigned int *d_counters, int n_images);
/* LaplaceMulti + FindPointsMulti fused (cuSIFT.cu:239-247 calls them back to back): same results as the two
 * stages above, but the 7 DoG planes stay in registers -- no DoG buffer, 4 B/px of HBM traffic instead of 60.
 * Needs 16-byte aligned rows (pitch % 4 == 0), w % 4 == 0 and an image < 2 GiB; returns CUSIFT_ERR_INVALID
 * otherwise (the drivers then fall back to the two-stage path). */
int cusift_detect_multi(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, size_t img_stride,
                        float init_blur, float peak_thresh, float edge_thresh, float subsampling,
                        cusift_point *d_points, int max_pts, unsigned int *d_counters, int n_images);
/* cusift_detect_multi that ALSO writes the next octave's image -- ScaleDown (cuSIFT.cu:185,313-353 + ScaleDown_D
 * cuSIFT_D.cu:37-182: 5 x 5 low-pass of variance `variance` and decimation, the asymmetric vertical taps included), bit
 * for bit what cusift_scale_down writes -- from the row window the blur streams through anyway: no second read of the
 * image, no launch of its own.  The keypoints go to a list of 64-byte HEADS (the first 16 floats of a SiftPoint:
 * coords2D .. subsampling), `max_pts` per image: finest-first detection cannot append to SiftData in list order, the
 * octave driver joins such lists (cusift_extract_batch with CUSIFT_POLICY_PYRAMID_IN_DETECT).  d_next: (w/2) x (h/2),
 * next_pitch floats per row (even), next_stride floats between images (even), 8-byte aligned. */
int cusift_detect_multi_down(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, size_t img_stride,
                             float init_blur, float peak_thresh, float edge_thresh, float subsampling, void *d_heads,
                             int max_pts, unsigned int *d_counters, int n_images, float *d_next, int next_pitch,
                             size_t next_stride, float variance);
/* SiftData::ComputeOrientations, cuSIFT.cu:355-365 + ComputeOrientations_D cuSIFT_D.cu:319-396.
 * Processes points [d_first[i], min(d_counters[i], max_pts)) of every image; d_first may be NULL (= 0). */
int cusift_compute_orientations(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, size_t img_stride,
                                cusift_point *d_points, int max_pts, const unsigned int *d_first,
                                const unsigned int *d_counters, int tex_frac_bits, int n_images);
/* SiftData::ExtractSiftDescriptors, cuSIFT.cu:367-377 + ExtractSiftDescriptors_D cuSIFT_D.cu:184-297
 * (also scales coords2D and scale by `subsampling`, cuSIFT_D.cu:292-296). */
int cusift_extract_descriptors(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, size_t img_stride,
                               cusift_point *d_points, int max_pts, const unsigned int *d_first,
                               const unsigned int *d_counters, float subsampling, int tex_frac_bits, int n_images);

/* The device build of the five transcendental functions the kernels use in place of CUDA's libm (expf, exp2f,
 * atan2f, sinf/cosf: cuSIFT_D.cu:209-210,233,330,349,507), array form: op 0 expf(a), 1 exp2f(a), 2 atan2f(a, b),
 * 3 sincosf(a) -> (out, out2).  They are written out in IEEE operations (cusift_amd/csrc/sift_math.h) so that a host
 * build of the same header gives the same bits; this entry point exists so that callers and tests can verify that
 * on their device.  op 4: the descriptor stage's angle coordinate 4/3.1415f * atan2f(a, b) + 4 as its kernel forms it
 * (a fit of degree 4 in t^2, within 4e-6 of a bin of the exact form -- the bound the kernel's comment states and the
 * tests assert --, with the reference's operations where the value decides: index 8).  op 5: the orientation stage's
 * histogram bin (int)(16 atan2f(a, b) / 3.1416f + 16.5f), 32 -> 0 (cuSIFT_D.cu:349-351) as its kernel finds it -- by
 * counting the bin edges below the gradient's direction inside its octant, no angle -- out = that bin, + 64 where the
 * sample lies within the stated margin of an edge (the kernel then evaluates the formula itself), out2 = the formula's
 * bin: equal wherever out < 64, for every input.
 * Asynchronous. */
int cusift_math_eval(cusift_ctx *ctx, int op, const float *d_a, const float *d_b, float *d_out, float *d_out2,
                     size_t n);


#ifdef __cplusplus
}
#endif
#endif /* CUSIFT_AMD_STAGES_H */
