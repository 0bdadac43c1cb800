// sift_host.h -- what the host-side translation units of libcusift_amd.so share (not installed): the kernels' prototypes,
// the context object, the octave plan and the helpers of the launch wrappers.
//   sift_context.hip  errors, context + arena + policy + stage timers, memory helpers
//   sift_stages.hip   the C ABI's stage entry points and their launch wrappers (front-end, ScaleDown, LaplaceMulti,
//                     FindPointsMulti, fused detection, orientation, descriptors, bands, matcher, homography, packing)
//   sift_driver.hip   the octave driver (cusift_extract_batch), its recorded graph, the single-image entry points
#pragma once

#include <hip/hip_runtime.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "sift_internal.h"
#include "sift_types.h"


namespace cusift {
// kernels (sift_stencils.hip, sift_keypoints.hip)
__global__ void scale_down_kernel(float *, int, long, const float *, int, int, int, long, int, ScaleDownTaps);
__global__ void scale_down_fast_kernel(float *, int, long, const float *, int, int, int, long, int, ScaleDownTaps,
                                       RowWindow, int, int, int);
__global__ void laplace_multi_kernel(const float *, float *, int, int, int, long, long, int, int, LaplaceTaps);
template <int kStoreAux>
__global__ void laplace_multi_fast_kernel(const float *, float *, int, int, int, long, long, int, LaplaceTapsPk);
__global__ void find_points_fast_kernel(const float *, int, int, int, long, cusift_point *, int, unsigned int *, int,
                                        FindParams);
template <bool kIdent0, int kRecBytes, bool kDown>
__global__ void detect_fused_kernel(const float *, int, int, int, long, cusift_point *, int, unsigned int *, int,
                                    LaplaceTapsPk, FindParams, RowWindow, int, int, DownOut);
template <int kRecBytes>
__global__ void detect_multi_kernel(DetectTable, int, unsigned int *);
__global__ void pyramid_small_kernel(PyramidLevels, ScaleDownTaps, unsigned int *, int);
__global__ void find_points_kernel(const float *, int, int, int, long, cusift_point *, int, unsigned int *, int, int,
                                   FindParams);
__global__ void orientations_kernel(const float *, int, int, int, long, cusift_point *, int, const unsigned int *,
                                    const unsigned int *, float, float, RowWindow);
__global__ void descriptors_kernel(const float *, int, int, int, long, cusift_point *, int, const unsigned int *,
                                   const unsigned int *, float, float, float, RowWindow, int, unsigned int *);
__global__ void describe_all_kernel(OctaveTable, cusift_point *, int, unsigned int *, int, float, float, int,
                                    unsigned int *, SegmentTable, const unsigned int *);
__global__ void join_counts_kernel(unsigned int *, SegmentTable, unsigned int *, int, int, unsigned int *, int);
__global__ void describe_bands_kernel(OctaveTable, BandWindows, cusift_point *, int, SegmentTable, const unsigned int *,
                                      float, float, int, unsigned int *);
__global__ void rootsift_kernel(cusift_point *, int);
template <bool kL2>
__global__ void match_kernel(cusift_point *, int, const cusift_point *, int, int, MatchPartial *, int);
__global__ void match_merge_kernel(cusift_point *, int, const cusift_point *, int, int, const MatchPartial *, int, int);
__global__ void homography_gather_kernel(const cusift_point *, int, float *);
__global__ void homography_solve_kernel(const float *, int, const int *, int, float *);
__global__ void homography_test_kernel(const float *, int, const float *, int, float, int *);
__global__ void u8_to_f32_kernel(float *, int, long, const unsigned char *, int, int, int, long, int);
__global__ void gaussian3x3_kernel(float *, int, long, const float *, int, int, int, long, float, float);
__global__ void math_eval_kernel(int, const float *, const float *, float *, float *, long);
__global__ void pack_points_kernel(const cusift_point *, const unsigned int *, int, int, cusift_point *, unsigned int,
                                   unsigned int *);
__global__ void pack_points_trimmed_kernel(const cusift_point *, const unsigned int *, int, int, cusift_trimmed_point *,
                                           unsigned int, unsigned int *);
__global__ void expand_trimmed_kernel(const cusift_trimmed_point *, size_t, cusift_point *);
__global__ void pack_points_compact_kernel(const cusift_point *, const unsigned int *, int, int, cusift_compact_point *,
                                           unsigned int, unsigned int *);
}  // namespace cusift

using namespace cusift;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
// sets the thread's error text (cusift_last_error) and returns `code` (cusift_fail: sift_internal.h, printf-checked)
#define fail(...) cusift_fail(__VA_ARGS__)

#define HIP_TRY(expr)                                                                                        \
  do {                                                                                                       \
    hipError_t e_ = (expr);                                                                                  \
    if (e_ != hipSuccess)                                                                                    \
      return fail(CUSIFT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

#define TRY(expr)              \
  do {                         \
    int rc_ = (expr);          \
    if (rc_ != CUSIFT_OK) return rc_; \
  } while (0)

static inline int idiv_up(int a, int b) { return (a + b - 1) / b; }
static inline int ialign_up(int a, int b) { return idiv_up(a, b) * b; }  // cutils.h:17
static inline size_t align_up_sz(size_t a, size_t b) { return (a + b - 1) / b * b; }

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct TimedSpan {
  hipEvent_t start, stop;
  int stage;
};

// Launch policy of a context.  The PRODUCT build reads no environment variable for it (CUSIFT_OCTAVE_OVERLAP, for an
// unchanged caller of the C++ shim, is read by the shim: include/cuSIFT.h); everything a test needs to force a driver path is
// set per context through cusift_ctx_set_policy; the tuning overrides of the A/B tools (rows per wave, waves per
// workgroup, cache policy of the DoG stores, ...) exist only in a -DCUSIFT_LAB build, which reads them from the
// environment ONCE, when a context is created -- nothing on a launch path looks at the environment in either build.
// 0 / negative = "not set".
enum { kKnobScaleDown = 0, kKnobLaplace, kKnobFindPoints, kKnobDetect, kKnobStages };
struct Knobs {
  // ---- policy (cusift_ctx_set_policy) ----
  int octave_overlap = 0;                            // CUSIFT_POLICY_SIDE_STREAM: 0 never (default), 1 eligible calls, 2 eligible calls
                                                     // after the concurrency probe, 3 every call (tests)
  int stage_all = -1;                                // CUSIFT_POLICY_OCTAVE_LISTS: -1 by size (default), 0 never, 1 whenever it fits
  bool force_generic = false;                        // CUSIFT_POLICY_GENERIC_KERNELS
  bool no_multi = false;                             // CUSIFT_POLICY_LAUNCH_PER_OCTAVE: the coarser octaves one launch each, even with lists
  int match_splits = 0;                              // CUSIFT_POLICY_MATCH_SPLITS
  bool tiled_per_octave = false;                     // CUSIFT_POLICY_TILED_PER_OCTAVE (read by cusift_tiled_create)
  int pyramid_in_detect = -1;                        // CUSIFT_POLICY_PYRAMID_IN_DETECT: -1 by size, 0 never, 1 octave 0, 2 every octave
  // ---- tuning (CUSIFT_LAB builds only) ----
  int rows_per_wave = 0;                             // CUSIFT_ROWS_PER_WAVE: every stencil stage
  int rows_lo[kKnobStages] = {0}, rows_hi[kKnobStages] = {0};  // CUSIFT_<STAGE>_ROWS_LO / _HI
  double detect_rows_coef = 0.0;                     // CUSIFT_DETECT_ROWS_COEF
  int detect_waves = 0, laplace_waves = 0;           // CUSIFT_DETECT_WAVES, CUSIFT_LAPLACE_WAVES (waves per workgroup)
  int laplace_aux = -1;                              // CUSIFT_LAPLACE_AUX: cache policy of the DoG stores
  bool no_ident = false;                             // CUSIFT_NO_IDENT
  bool side_debug = false;                           // CUSIFT_SIDE_DEBUG: print the side stream's concurrency probe
  int stage_all_mb = 0;                              // CUSIFT_STAGE_ALL_MB: largest staging for all octaves (0: default)
  int small_pyramid = -1;                            // CUSIFT_SMALL_PYRAMID: 0 never, 1 always, default by size
  bool unordered_coarse = false;                     // CUSIFT_UNORDERED_COARSE: a TIMING experiment, results are wrong unless
                                                     // the same batch is extracted over and over (sift_driver.hip)
};

struct cusift_ctx {
  int device = 0;
  Knobs knobs;
  int num_cus = 256;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  // scratch arena (one allocation, grown on demand, never shrunk)
  char *arena = nullptr;
  size_t arena_bytes = 0;
  // DoG planes of the two-stage path ([n][7][h0][p0] floats); allocated only when that path runs
  float *dog = nullptr;
  size_t dog_bytes = 0;
  // per-split partial results of the matcher (cusift_match)
  MatchPartial *match_scratch = nullptr;
  size_t match_scratch_bytes = 0;
  // coordinates / samples / hypotheses / counts of cusift_find_homography
  char *homo_scratch = nullptr;
  size_t homo_scratch_bytes = 0;
  // staging buffer for 8-bit uploads (cusift_image_u8_h2d)
  unsigned char *u8_stage = nullptr;
  size_t u8_stage_bytes = 0;
  // small persistent device scratch for the blocking single-image entry points
  unsigned int *d_counter1 = nullptr;
  unsigned int *h_counter1 = nullptr;  // pinned mailbox the count of cusift_extract is copied to (no staged pageable copy)
  int extract_guess = 0;  // records cusift_extract copies back BEFORE it knows the count (the previous call's, + 1/8)
  unsigned int *d_queue = nullptr;  // kQueueShards work cursors of describe_all_kernel, 128 bytes apart
  int describe_grid = 0;  // resident blocks of describe_all_kernel on this device (occupancy query, cached)
  // octave 0's detection beside the coarser octaves (cusift_extract_batch): a second stream and its fork / join events
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool side_failed = false;  // no stream was found that runs beside the context's stream: never fork
  bool side_probed = false;  // `side` passed the concurrency probe (policy value 2 accepts no other)
  bool recording = false;  // inside cusift_graph_create's capture
  unsigned long forks = 0;  // extractions that took the side stream
  // the lists' counters in the arena that the last extraction's join_counts_kernel left zero (stream order): the next
  // extraction that uses exactly them skips its memset.  Anything else that writes the arena resets this.
  const void *seg_clean_ptr = nullptr;
  size_t seg_clean_bytes = 0;
  unsigned long scratch_gen = 0;  // bumped whenever arena / DoG / matcher scratch is re-allocated (recorded graphs check it)
  // timing
  bool timing = false;
  std::vector<TimedSpan> spans;       // recorded, not yet folded
  std::vector<hipEvent_t> event_pool;  // free events
  float ms[CUSIFT_NUM_STAGES] = {0};
  int launches[CUSIFT_NUM_STAGES] = {0};
};


struct StageTimer {
  cusift_ctx *ctx;
  int stage;
  hipEvent_t start = nullptr, stop = nullptr;
  StageTimer(cusift_ctx *c, int s) : ctx(c), stage(s) {
    if (!ctx->timing) return;
    start = take();
    stop = take();
    if (start) (void)hipEventRecord(start, ctx->stream);
  }
  ~StageTimer() {
    if (!ctx->timing || !start || !stop) return;
    (void)hipEventRecord(stop, ctx->stream);
    ctx->spans.push_back({start, stop, stage});
  }
  hipEvent_t take() {
    if (!ctx->event_pool.empty()) {
      hipEvent_t e = ctx->event_pool.back();
      ctx->event_pool.pop_back();
      return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
  }
};

// Octave geometry of one extraction (cuSIFT.cu:76-91,175-190).
struct Plan {
  int n_oct = 0;
  int w[kMaxOctaves], h[kMaxOctaves], p[kMaxOctaves];
  double blur[kMaxOctaves];
  float sub[kMaxOctaves];
  // arena offsets in bytes
  size_t base_off[kMaxOctaves];  // octave >= 1 base images (n * h*p floats each); [0] unused
  size_t first_off = 0, total = 0;
  // staging lists of keypoint heads ([octave][image][max_pts] x kStagedRecBytes), see cusift_extract_batch:
  // staged_octaves == 0: none; 1: octave 0's (searched on the side stream); n_oct: every octave's
  size_t staged_off = 0, seg_end_off = 0;
  int staged_octaves = 0;
  bool fork = false;  // octave 0's detection on the context's side stream
};

// Largest staging a context allocates: for octave 0 alone (the side stream), for all octaves (one detection launch; a
// batch beyond it keeps the in-place lists)
constexpr size_t kMaxStagedBytes = (size_t)1 << 30, kMaxStagedAllBytes = (size_t)1 << 30;


struct MultiOctave {
  const float *img;
  int w, h, pitch;
  size_t img_stride;
  float init_blur, subsampling;
  cusift_point *lists;
  unsigned int *counters;
  // a band of a larger image (cusift_extract_bands): global row of local row 0, global rows, centre rows; -1: whole image
  int row0 = 0, hg = -1, cy_begin = 0, cy_end = 0;
};

// defined in sift_context.hip / sift_stages.hip / sift_driver.hip
int fold_spans(cusift_ctx *ctx);
int make_plan(Plan &pl, int n_images, int w, int h, int pitch, const cusift_params *prm, bool fork = false,
              bool stage_all = false, size_t stage_all_limit = kMaxStagedAllBytes);
int ensure_dog(cusift_ctx *ctx, size_t bytes);
int ensure_arena(cusift_ctx *ctx, size_t bytes);
int pick_rows(const cusift_ctx *ctx, int h, int strips, int n_images, int lo, int hi);
void rows_bounds(const cusift_ctx *ctx, int stage, int &lo, int &hi);
void scale_down_taps(ScaleDownTaps &T, float variance);
void laplace_taps_table(float init_blur, float taps[8 * 16]);
void find_params(FindParams &P, float peak_thresh, float edge_thresh, float subsampling);
void frac_consts(int frac_bits, float &q, float &inv_q);
int enter(cusift_ctx *ctx);
int check_launch(const char *what);
bool wants_side_stream(const cusift_ctx *ctx, const cusift_params *prm, int n_images, int w, int h);
size_t stage_all_limit(const cusift_ctx *ctx);
bool wants_stage_all(const cusift_ctx *ctx, const cusift_params *prm, int n_images, int w, int h);
int wants_pyramid_in_detect(const cusift_ctx *ctx, const cusift_params *prm, int n_images, int w, int h);
int ensure_side_stream(cusift_ctx *ctx);
int ctx_create_impl(cusift_ctx **out, int device, void *hip_stream, bool borrow);
size_t bands_arena_bytes(int n_bands, int max_pts);
int scale_down_impl(cusift_ctx *ctx, float *d_dst, int dst_pitch, size_t dst_stride, const float *d_src, int w, int h, int src_pitch, size_t src_stride, int n_images, float variance, RowWindow src_rw, int dst_row0, int r_begin, int r_end, bool band);
bool wants_small_pyramid(const cusift_ctx *ctx, int n_images, int w, int h);
bool wants_self_join(const cusift_ctx *ctx, int n_images, int w, int h);
int pyramid_small_impl(cusift_ctx *ctx, const float *const *base, const int *w, const int *h, const int *pitch, const size_t *stride, int n_levels, int n_images, float variance, unsigned int *d_zero, int n_zero);
bool detect_fused_ok(const float *d_img, int w, int h, int pitch, size_t img_stride);
int detect_rows(const cusift_ctx *ctx, int rows_total, int strips, int n_images, int concurrent);
int detect_impl(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, size_t img_stride, float init_blur, float peak_thresh, float edge_thresh, float subsampling, cusift_point *d_points, int max_pts, unsigned int *d_counters, int n_images, RowWindow rw, int cy_begin, int cy_end, int concurrent = 1, bool heads = false, bool side = false, const DownOut *down = nullptr);
int detect_multi_impl(cusift_ctx *ctx, const MultiOctave *octaves, int n_octaves, float peak_thresh, float edge_thresh, int max_pts, int n_images, int concurrent, unsigned int *d_queue);
int keypoint_grid_x(int max_pts, int n_images);
int orientations_impl(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, size_t img_stride, cusift_point *d_points, int max_pts, const unsigned int *d_first, const unsigned int *d_counters, int tex_frac_bits, int n_images, RowWindow rw);
int descriptors_impl(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, size_t img_stride, cusift_point *d_points, int max_pts, const unsigned int *d_first, const unsigned int *d_counters, float subsampling, int tex_frac_bits, int n_images, RowWindow rw, int root_sift = 0, unsigned int *d_flags = nullptr);
size_t two_stage_dog_bytes(const cusift_ctx *ctx, const Plan &pl, const cusift_params *prm, const float *d_imgs, size_t image_stride, const char *arena_base, int n_images);
