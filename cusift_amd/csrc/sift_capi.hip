// sift_capi.hip -- the C ABI (include/cusift_amd.h) over the gfx950 kernels: context, scratch arena,
// launch wrappers and the octave driver.  Host logic follows cuSIFT.cu (cited per function); nothing
// here allocates, frees or reads back between stages once the arena is sized.
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
template <bool kIdent0, int kRecBytes>
__global__ void detect_fused_kernel(const float *, int, int, int, long, cusift_point *, int, unsigned int *, int,
                                    LaplaceTapsPk, FindParams, RowWindow, int, int);
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
__global__ void join_counts_kernel(unsigned int *, SegmentTable, unsigned int *, int, int, unsigned int *);
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
__global__ void pack_points_compact_kernel(const cusift_point *, const unsigned int *, int, int, cusift_compact_point *,
                                           unsigned int, unsigned int *);
}  // namespace cusift

using namespace cusift;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

int cusift_fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

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

// Tuning overrides for experiments (tools/ab_*.sh, tools/probe_rows.py), read from the environment ONCE, when a
// context is created -- nothing on a launch path looks at the environment.  0 / negative = "not set".
enum { kKnobScaleDown = 0, kKnobLaplace, kKnobFindPoints, kKnobDetect, kKnobStages };
struct Knobs {
  int rows_per_wave = 0;                             // CUSIFT_ROWS_PER_WAVE: every stencil stage
  int rows_lo[kKnobStages] = {0}, rows_hi[kKnobStages] = {0};  // CUSIFT_<STAGE>_ROWS_LO / _HI
  double detect_rows_coef = 0.0;                     // CUSIFT_DETECT_ROWS_COEF
  int detect_waves = 0, laplace_waves = 0;           // CUSIFT_DETECT_WAVES, CUSIFT_LAPLACE_WAVES (waves per workgroup)
  int laplace_aux = -1;                              // CUSIFT_LAPLACE_AUX: cache policy of the DoG stores
  bool no_ident = false, force_generic = false;      // CUSIFT_NO_IDENT, CUSIFT_FORCE_GENERIC
  int match_splits = 0;                              // CUSIFT_MATCH_SPLITS
  int octave_overlap = -1;                           // CUSIFT_OCTAVE_OVERLAP: 0 never, 1 lone callers (default), 2 always
  bool side_debug = false;                           // CUSIFT_SIDE_DEBUG: print the side stream's concurrency probe
  int stage_all = -1;                                // CUSIFT_STAGE_ALL: 0 never, 1 whenever it fits (default), 2 = 1
  bool no_multi = false;                             // CUSIFT_NO_MULTI: the coarser octaves one launch each, even with lists
  int stage_all_mb = 0;                              // CUSIFT_STAGE_ALL_MB: largest staging for all octaves (0: default)
  int small_pyramid = -1;                            // CUSIFT_SMALL_PYRAMID: 0 never, 1 always, default by size
};

static Knobs read_knobs() {
  auto text = [](const char *name) -> const char * { return getenv(name); };
  auto num = [&](const char *name, int unset) { const char *e = text(name); return e ? atoi(e) : unset; };
  Knobs k;
  k.rows_per_wave = num("CUSIFT_ROWS_PER_WAVE", 0);
  static const char *const stage[kKnobStages] = {"SCALEDOWN", "LAPLACE", "FINDPOINTS", "DETECT"};
  for (int i = 0; i < kKnobStages; ++i) {
    char name[64];
    snprintf(name, sizeof(name), "CUSIFT_%s_ROWS_LO", stage[i]);
    k.rows_lo[i] = num(name, 0);
    snprintf(name, sizeof(name), "CUSIFT_%s_ROWS_HI", stage[i]);
    k.rows_hi[i] = num(name, 0);
  }
  if (const char *e = text("CUSIFT_DETECT_ROWS_COEF")) k.detect_rows_coef = atof(e);
  k.detect_waves = num("CUSIFT_DETECT_WAVES", 0);
  k.laplace_waves = num("CUSIFT_LAPLACE_WAVES", 0);
  k.laplace_aux = num("CUSIFT_LAPLACE_AUX", -1);
  k.no_ident = text("CUSIFT_NO_IDENT") != nullptr;
  k.force_generic = text("CUSIFT_FORCE_GENERIC") != nullptr;
  k.match_splits = num("CUSIFT_MATCH_SPLITS", 0);
  k.octave_overlap = num("CUSIFT_OCTAVE_OVERLAP", -1);
  k.side_debug = text("CUSIFT_SIDE_DEBUG") != nullptr;
  k.stage_all = num("CUSIFT_STAGE_ALL", -1);
  k.no_multi = text("CUSIFT_NO_MULTI") != nullptr;
  k.stage_all_mb = num("CUSIFT_STAGE_ALL_MB", 0);
  k.small_pyramid = num("CUSIFT_SMALL_PYRAMID", -1);
  return k;
}

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
  unsigned int *d_queue = nullptr;  // kQueueShards work cursors of describe_all_kernel, 128 bytes apart
  int describe_grid = 0;  // resident blocks of describe_all_kernel on this device (occupancy query, cached)
  // octave 0's detection beside the coarser octaves (cusift_extract_batch): a second stream and its fork / join events
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool side_failed = false;  // no stream was found that runs beside the context's stream: never fork
  bool recording = false;  // inside cusift_graph_create's capture
  unsigned long forks = 0;  // extractions that took the side stream
  unsigned long scratch_gen = 0;  // bumped whenever arena / DoG / matcher scratch is re-allocated (recorded graphs check it)
  // timing
  bool timing = false;
  std::vector<TimedSpan> spans;       // recorded, not yet folded
  std::vector<hipEvent_t> event_pool;  // free events
  float ms[CUSIFT_NUM_STAGES] = {0};
  int launches[CUSIFT_NUM_STAGES] = {0};
};

namespace {

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

int fold_spans(cusift_ctx *ctx) {
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (auto &s : ctx->spans) {
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, s.start, s.stop));
    ctx->ms[s.stage] += t;
    ctx->launches[s.stage] += 1;
    ctx->event_pool.push_back(s.start);
    ctx->event_pool.push_back(s.stop);
  }
  ctx->spans.clear();
  return CUSIFT_OK;
}

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

int make_plan(Plan &pl, int n_images, int w, int h, int pitch, const cusift_params *prm, bool fork = false,
              bool stage_all = false, size_t stage_all_limit = kMaxStagedAllBytes) {
  if (!prm) return fail(CUSIFT_ERR_INVALID, "params is NULL");
  if (n_images < 1 || w < 1 || h < 1 || pitch < w)
    return fail(CUSIFT_ERR_INVALID, "bad geometry n=%d w=%d h=%d pitch=%d", n_images, w, h, pitch);
  if (n_images > 65535) return fail(CUSIFT_ERR_INVALID, "at most 65535 images per batch (grid.z), got %d", n_images);
  if (prm->max_pts < 1) return fail(CUSIFT_ERR_INVALID, "max_pts must be >= 1");
  int n = std::max(1, std::min(prm->num_octaves, kMaxOctaves));
  pl.w[0] = w;
  pl.h[0] = h;
  pl.p[0] = pitch;
  pl.blur[0] = prm->init_blur;
  pl.sub[0] = prm->subsampling;
  pl.n_oct = 1;
  for (int o = 1; o < n; ++o) {
    int ww = pl.w[o - 1] / 2, hh = pl.h[o - 1] / 2;  // integer division, cuSIFT.cu:182
    if (ww < 1 || hh < 1) break;
    pl.w[o] = ww;
    pl.h[o] = hh;
    pl.p[o] = ialign_up(ww, 128);  // cuSIFT.cu:183
    // cuSIFT.cu:188: float totInitBlur = (float)sqrt(initBlur*initBlur + 0.5f*0.5f) / 2.0f;
    float tot = (float)sqrt(pl.blur[o - 1] * pl.blur[o - 1] + 0.5f * 0.5f) / 2.0f;
    pl.blur[o] = tot;
    pl.sub[o] = pl.sub[o - 1] * 2.0f;
    pl.n_oct = o + 1;
  }
  size_t off = 0;
  pl.base_off[0] = 0;
  for (int o = 1; o < pl.n_oct; ++o) {
    pl.base_off[o] = off;
    off = align_up_sz(off + (size_t)n_images * pl.h[o] * pl.p[o] * sizeof(float), 256);
  }
  pl.first_off = off;  // per-octave snapshots of the counters (fstPts), or the segments' counters
  off = align_up_sz(off + (size_t)n_images * kMaxOctaves * sizeof(unsigned int), 256);
  pl.seg_end_off = off;  // join_counts_kernel's running sums, [image][segment]
  off = align_up_sz(off + (size_t)n_images * kMaxOctaves * sizeof(unsigned int), 256);
  const size_t per_octave = (size_t)n_images * prm->max_pts * kStagedRecBytes;
  pl.fork = fork && pl.n_oct >= 2 && per_octave <= kMaxStagedBytes;
  pl.staged_octaves = (stage_all && pl.n_oct >= 2 && per_octave * pl.n_oct <= stage_all_limit) ? pl.n_oct : (pl.fork ? 1 : 0);
  if (pl.staged_octaves) {
    pl.staged_off = off;
    off = align_up_sz(off + per_octave * pl.staged_octaves, 256);
  }
  pl.total = off;
  return CUSIFT_OK;
}

int ensure_dog(cusift_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->dog_bytes) return CUSIFT_OK;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  ctx->scratch_gen++;
  if (ctx->dog) HIP_TRY(hipFree(ctx->dog));
  ctx->dog = nullptr;
  ctx->dog_bytes = 0;
  hipError_t e = hipMalloc((void **)&ctx->dog, bytes);
  if (e != hipSuccess) return fail(CUSIFT_ERR_NOMEM, "DoG hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  ctx->dog_bytes = bytes;
  return CUSIFT_OK;
}

int ensure_arena(cusift_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->arena_bytes) return CUSIFT_OK;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  ctx->scratch_gen++;
  if (ctx->arena) HIP_TRY(hipFree(ctx->arena));
  ctx->arena = nullptr;
  ctx->arena_bytes = 0;
  hipError_t e = hipMalloc((void **)&ctx->arena, bytes);
  if (e != hipSuccess) return fail(CUSIFT_ERR_NOMEM, "arena hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  ctx->arena_bytes = bytes;
  return CUSIFT_OK;
}

// rows each wave marches: as large as possible (less halo re-read) while the launch still has
// >= ~2 waves per SIMD on 256 CUs.
int pick_rows(const cusift_ctx *ctx, int h, int strips, int n_images, int lo, int hi) {
  if (ctx->knobs.rows_per_wave > 0) return ctx->knobs.rows_per_wave;  // tuning/experiments only
  const long target_waves = 256L * 4 * 2 * 2;
  long r = (long)h * strips * n_images / target_waves;
  if (r < lo) r = lo;
  if (r > hi) r = hi;
  return (int)r;
}

// [lo, hi] of pick_rows for a stage, overridable for tuning experiments (Knobs)
void rows_bounds(const cusift_ctx *ctx, int stage, int &lo, int &hi) {
  if (ctx->knobs.rows_hi[stage] > 0) hi = ctx->knobs.rows_hi[stage];
  if (ctx->knobs.rows_lo[stage] > 0) lo = ctx->knobs.rows_lo[stage];
  lo = std::min(lo, hi);
}

void scale_down_taps(ScaleDownTaps &T, float variance) {
  // cuSIFT.cu:320-341 (the pyramid passes variance = 0.5, cuSIFT.cu:185)
  float k[5], sum = 0.0f;
  for (int j = 0; j < 5; j++) {
    k[j] = (float)expf(-(double)(j - 2) * (j - 2) / 2.0 / variance);
    sum += k[j];
  }
  for (int j = 0; j < 5; j++) k[j] /= sum;
  T.k[0] = k[0];
  T.k[1] = k[1];
  T.k[2] = k[2];
}

void laplace_taps_table(float init_blur, float taps[8 * 16]) {
  // cuSIFT.cu:239-240,400-412.  Rule of this build: var <= 1e-6 => identity (the reference produces
  // NaN taps at var == 0 and an inverted kernel at var < 0; see DESIGN.md "degenerate initBlur").
  const float baseBlur = powf(2.0f, -1.0f / kNumScales);
  const float diffScale = powf(2.0f, 1.0f / kNumScales);
  float scale = baseBlur;
  memset(taps, 0, sizeof(float) * 8 * 16);
  for (int i = 0; i < kNumLevels; i++) {
    float kernelSum = 0.0f;
    float var = scale * scale - init_blur * init_blur;
    float *k = taps + 16 * i;
    if (var <= 1e-6f) {
      k[kBlurRadius] = 1.0f;
    } else {
      for (int j = -kBlurRadius; j <= kBlurRadius; j++) {
        k[j + kBlurRadius] = (float)expf(-(double)j * j / 2.0 / var);
        kernelSum += k[j + kBlurRadius];
      }
      for (int j = -kBlurRadius; j <= kBlurRadius; j++) k[j + kBlurRadius] /= kernelSum;
    }
    scale *= diffScale;
  }
}

void find_params(FindParams &P, float peak_thresh, float edge_thresh, float subsampling) {
  // cuSIFT.cu:239-247 (sigma = baseBlur*diffScale, factor = 1/NUM_SCALES), cuSIFT.cu:432-444
  const float baseBlur = powf(2.0f, -1.0f / kNumScales);
  const float diffScale0 = powf(2.0f, 1.0f / kNumScales);
  const double sigma = baseBlur * diffScale0;
  const float factor = 1.0f / kNumScales;
  float scale = (float)sigma;
  const float diffScale = powf(2.0f, factor);
  for (int i = 0; i < kNumScales; i++) {
    P.scales[i] = scale;
    scale *= diffScale;
  }
  P.thr_pos = peak_thresh;
  P.thr_neg = -peak_thresh;
  P.edge_limit = edge_thresh;
  P.factor = factor;
  P.subsampling = subsampling;
}

void frac_consts(int frac_bits, float &q, float &inv_q) {
  if (frac_bits > 0 && frac_bits < 24) {
    q = (float)(1 << frac_bits);
    inv_q = 1.0f / q;
  } else {
    q = 0.0f;
    inv_q = 0.0f;
  }
}

// Every ctx-taking entry point starts here: a NULL check and the selection of the context's device BEFORE any
// allocation or launch (a process may hold contexts on several GPUs; the current device is per thread).
int enter(cusift_ctx *ctx) {
  if (!ctx) return fail(CUSIFT_ERR_INVALID, "ctx is NULL");
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != ctx->device) HIP_TRY(hipSetDevice(ctx->device));
  return CUSIFT_OK;
}

int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CUSIFT_ERR_HIP, "%s launch failed: %s", what, hipGetErrorString(e));
  return CUSIFT_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// process / device
// ------------------------------------------------------------------------------------------------
extern "C" const char *cusift_last_error(void) { return g_err.c_str(); }
extern "C" const char *cusift_version(void) { return "cusift_amd 0.1 (gfx950)"; }

extern "C" int cusift_device_count(int *count) {
  if (!count) return fail(CUSIFT_ERR_INVALID, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) n = 0;
  *count = n;
  return CUSIFT_OK;
}

extern "C" int cusift_init(int device) {
  // InitCuda, cutils.h:71-92: clamp into [0, n-1], select
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail(CUSIFT_ERR_NO_DEVICE, "no HIP device available");
  device = std::max(0, std::min(n - 1, device));
  HIP_TRY(hipSetDevice(device));
  return CUSIFT_OK;
}

// Octave 0 beside the coarser octaves.  One extraction on one stream is a chain of launches of very different sizes:
// octave 0's detection (3/4 of the pixels), then four ScaleDowns and four detections that each fill the chip for a few
// microseconds and end in a tail.  A caller that keeps several batches in flight (cusift_params.concurrent_batches >= 2)
// covers those tails with the other batches' kernels; a lone caller -- ExtractSift as the reference calls it -- cannot.
// For it the driver forks: octave 0's detection goes to a second stream of the context and appends HEADS (64 bytes)
// to a staging list in the arena, the ScaleDown chain and the coarser detections run on the context's stream as
// before, the streams join, and describe_all_kernel moves the staged keypoints behind the coarser ones while it
// describes them -- so SiftData comes out coarsest octave first, and saturates coarsest first, exactly as before.
// Measured on MI355X (tools/probe_octave_overlap.py, one stream, back to back): 64 x 1080p 1.527 -> 1.356 ms, 16: 0.478 ->
// 0.429, 4: 0.213 -> 0.200 -- but ONE frame 0.132 -> 0.138 ms, and its recorded graph 0.148 -> 0.187: the two
// cross-stream waits cost more than a frame's tails.  So the fork is taken from kSideStreamMinPixels up (three 1080p
// frames), never inside a recording, never with the stage timers on (they bracket launches on one stream).
constexpr size_t kSideStreamMinPixels = 6u << 20;
static bool wants_side_stream(const cusift_ctx *ctx, const cusift_params *prm, int n_images, int w, int h) {
  const int mode = ctx->knobs.octave_overlap;
  if (mode == 0 || ctx->timing || ctx->knobs.force_generic || ctx->side_failed) return false;
  if (!prm || !prm->fused_detect || n_images < 1 || n_images > kMaxFlatImages) return false;
  if (mode == 2) return true;  // tests: whatever the size, also inside a recording
  if (ctx->recording) return false;
  return prm->concurrent_batches < 2 && (size_t)n_images * (size_t)w * (size_t)h >= kSideStreamMinPixels;
}

// Every octave's keypoints to staging lists, joined by describe_all_kernel: detections no longer have to run, or end, in
// list order -- ALL octaves are searched by ONE launch (detect_multi_impl; two with octave 0 on the side stream), the
// large octave's workgroups first and the small ones in its tail.  What it buys is dispatches and tails (MI355X,
// 1080p, ms per call back to back on one stream: 1 frame 0.130 -> 0.082, 4: 0.210 -> 0.145, 16: 0.476 -> 0.389,
// 64: 1.52 -> 1.41; with four calls in flight: 1 frame 0.062 -> 0.043, 4: 0.111 -> 0.106, 16: 0.338 -> 0.343,
// 64: 1.165 -> 1.20 -- there the other batches fill the tails already and octave 0 is better off in its own, tuned
// instantiation).  So: a lone caller whenever the lists fit, a pipelining caller up to eight 1080p frames' worth of
// pixels per call.  Not with the per-octave stage sequence, the generic kernels or the stage timers on.
constexpr size_t kListsMaxPixelsPipelined = 16u << 20;
static size_t stage_all_limit(const cusift_ctx *ctx) {
  return ctx->knobs.stage_all_mb > 0 ? (size_t)ctx->knobs.stage_all_mb << 20 : kMaxStagedAllBytes;
}
static bool wants_stage_all(const cusift_ctx *ctx, const cusift_params *prm, int n_images, int w, int h) {
  if (ctx->knobs.stage_all == 0 || ctx->knobs.force_generic) return false;
  if (!prm || !prm->fused_detect || n_images < 1 || n_images > kMaxFlatImages) return false;
  if (ctx->knobs.stage_all > 0) return true;  // tests: whenever the lists fit
  if (ctx->timing) return false;  // the stage timers bracket the reference's launch-per-octave sequence
  return prm->concurrent_batches < 2 || (size_t)n_images * (size_t)w * (size_t)h <= kListsMaxPixelsPipelined;
}

// A second stream only helps if the device runs it BESIDE the context's stream.  HIP maps streams onto a few hardware
// queues (GPU_MAX_HW_QUEUES, 4 by default) in the order of their first use, and the queues onto the compute pipes of
// the command processor round robin.  Two streams on one hardware queue run one after the other; two queues on one
// pipe run their kernels at the same time but the pipe serves one queue's packets while the other queue's wait, a few
// microseconds each -- worse than one queue for a chain of short launches (measured, 64 x 1080p: 1.29-1.35 ms beside each other, 1.55
// on one queue, 1.88 on one pipe: four streams in use before this one and 8 hardware queues).  Which case a new stream
// lands in depends on what else the process has created -- so the context finds out: a chain of sixteen 5 us probe
// kernels on its own stream is timed alone and while a 120 us probe runs on the candidate, and the candidate is kept
// only if the chain takes about as long in both cases.  Up to four candidates (consecutive hardware queues sit on
// different pipes); if none passes, the context never forks.  One-time cost: under a millisecond and a wait for the
// context's stream, at the first call that would fork.
__global__ void __launch_bounds__(64) spin_kernel(long ticks) {
  const long t0 = (long)wall_clock64();  // the 100 MHz constant clock
  while ((long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

static int ensure_side_stream(cusift_ctx *ctx) {
  if (ctx->side) return CUSIFT_OK;
  if (ctx->side_failed) return fail(CUSIFT_ERR_HIP, "no side stream runs beside the context's stream");
  if (!ctx->ev_fork) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  if (!ctx->ev_join) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  constexpr int kChain = 16;
  constexpr long kShort = 500, kLong = 12000;  // 5 us, 120 us
  int rc = CUSIFT_OK;
  hipStream_t rejected[4] = {nullptr, nullptr, nullptr, nullptr};  // kept alive until the end so that each try is a NEW queue
  int n_rejected = 0;
  for (int attempt = 0; attempt < 4 && !ctx->side && rc == CUSIFT_OK; ++attempt) {
    hipStream_t cand = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&cand, hipStreamNonBlocking);
    if (e != hipSuccess) {
      rc = fail(CUSIFT_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
      break;
    }
    float alone = 1e30f, beside_ms = 1e30f;
    auto chain = [&](bool with_candidate, float &best) -> hipError_t {
      hipError_t err;
      if ((err = hipStreamSynchronize(cand)) != hipSuccess) return err;
      if ((err = hipStreamSynchronize(ctx->stream)) != hipSuccess) return err;
      if ((err = hipEventRecord(e0, ctx->stream)) != hipSuccess) return err;
      if (with_candidate) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, cand, kLong);
      for (int k = 0; k < kChain; ++k) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, ctx->stream, kShort);
      if ((err = hipEventRecord(e1, ctx->stream)) != hipSuccess) return err;
      if ((err = hipEventSynchronize(e1)) != hipSuccess) return err;
      float ms = 0.f;
      if ((err = hipEventElapsedTime(&ms, e0, e1)) != hipSuccess) return err;
      best = std::min(best, ms);
      return hipStreamSynchronize(cand);
    };
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, cand, 1L);  // first use: the candidate's hardware queue is created here
    for (int rep = 0; rep < 2 && e == hipSuccess; ++rep) {
      e = chain(false, alone);
      if (e == hipSuccess) e = chain(true, beside_ms);
    }
    if (e != hipSuccess) {
      (void)hipStreamDestroy(cand);
      rc = fail(CUSIFT_ERR_HIP, "side stream probe failed: %s", hipGetErrorString(e));
      break;
    }
    const bool beside = beside_ms < 1.25f * std::max(alone, 0.08f);
    if (ctx->knobs.side_debug)
      fprintf(stderr, "cusift: side stream candidate %d: probe chain alone %.1f us, beside the candidate %.1f us -> %s\n",
              attempt, alone * 1e3f, beside_ms * 1e3f, beside ? "kept" : "rejected");
    if (beside)
      ctx->side = cand;
    else
      rejected[n_rejected++] = cand;
  }
  if (!ctx->side && rc == CUSIFT_OK && ctx->knobs.octave_overlap == 2 && n_rejected > 0)
    ctx->side = rejected[--n_rejected];  // forced (tests of the forked driver): any stream will do
  for (int i = 0; i < n_rejected; ++i) (void)hipStreamDestroy(rejected[i]);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc == CUSIFT_OK && !ctx->side) rc = fail(CUSIFT_ERR_HIP, "no side stream runs beside the context's stream");
  if (rc != CUSIFT_OK) ctx->side_failed = true;  // the callers go on with one stream
  return rc;
}

extern "C" void cusift_default_params(cusift_params *p) {
  if (!p) return;
  p->num_octaves = 5;
  p->init_blur = 0.0;
  p->peak_thresh = 3.0f;
  p->edge_thresh = 10.0f;  // the only value the reference uses (test/detector.cpp:46)
  p->lowest_scale = 0.0f;
  p->subsampling = 1.0f;
  p->max_pts = 1024;  // SiftData ctor default (cuSIFT.h:56)
  p->tex_frac_bits = 8;
  p->fused_detect = 1;
  p->root_sift = 0;
  p->concurrent_batches = 1;
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
static int ctx_create_impl(cusift_ctx **out, int device, void *hip_stream, bool borrow) {
  if (!out) return fail(CUSIFT_ERR_INVALID, "out is NULL");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail(CUSIFT_ERR_NO_DEVICE, "no HIP device available");
  if (device < 0 || device >= n) return fail(CUSIFT_ERR_INVALID, "device %d out of range [0,%d)", device, n);
  HIP_TRY(hipSetDevice(device));
  cusift_ctx *ctx = new cusift_ctx();
  ctx->device = device;
  ctx->knobs = read_knobs();
  (void)hipDeviceGetAttribute(&ctx->num_cus, hipDeviceAttributeMultiprocessorCount, device);
  if (ctx->num_cus < 1) ctx->num_cus = 256;
  if (borrow) {
    ctx->stream = (hipStream_t)hip_stream;  // NULL = the null stream
  } else {
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete ctx;
      return fail(CUSIFT_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    ctx->owns_stream = true;
  }
  hipError_t e = hipMalloc((void **)&ctx->d_counter1, 256);
  if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_queue, kQueueShards * 128);
  if (e != hipSuccess) {
    if (ctx->d_counter1) (void)hipFree(ctx->d_counter1);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return fail(CUSIFT_ERR_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
  }
  *out = ctx;
  return CUSIFT_OK;
}

extern "C" int cusift_ctx_create(cusift_ctx **out, int device, void *hip_stream) {
  return ctx_create_impl(out, device, hip_stream, hip_stream != nullptr);
}

extern "C" int cusift_ctx_create_borrowed(cusift_ctx **out, int device, void *hip_stream) {
  return ctx_create_impl(out, device, hip_stream, true);
}

extern "C" int cusift_ctx_destroy(cusift_ctx *ctx) {
  if (!ctx) return CUSIFT_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto &s : ctx->spans) {
    (void)hipEventDestroy(s.start);
    (void)hipEventDestroy(s.stop);
  }
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  if (ctx->arena) (void)hipFree(ctx->arena);
  if (ctx->dog) (void)hipFree(ctx->dog);
  if (ctx->u8_stage) (void)hipFree(ctx->u8_stage);
  if (ctx->homo_scratch) (void)hipFree(ctx->homo_scratch);
  if (ctx->match_scratch) (void)hipFree(ctx->match_scratch);
  if (ctx->d_counter1) (void)hipFree(ctx->d_counter1);
  if (ctx->d_queue) (void)hipFree(ctx->d_queue);
  if (ctx->side) {
    (void)hipStreamSynchronize(ctx->side);
    (void)hipStreamDestroy(ctx->side);
  }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return CUSIFT_OK;
}

extern "C" int cusift_ctx_synchronize(cusift_ctx *ctx) {
  TRY(enter(ctx));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return CUSIFT_OK;
}

extern "C" void *cusift_ctx_stream(cusift_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
extern "C" int cusift_ctx_device(cusift_ctx *ctx) { return ctx ? ctx->device : -1; }

extern "C" int cusift_ctx_wait(cusift_ctx *ctx, cusift_ctx *other) {
  TRY(enter(ctx));
  if (!other) return fail(CUSIFT_ERR_INVALID, "other is NULL");
  if (other == ctx || other->stream == ctx->stream) return CUSIFT_OK;  // same stream: already ordered
  if (other->device != ctx->device) return fail(CUSIFT_ERR_INVALID, "contexts live on different devices");
  hipEvent_t ev = nullptr;
  HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, other->stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ev, 0);
  (void)hipEventDestroy(ev);  // destruction is deferred until the event has completed
  if (e != hipSuccess) return fail(CUSIFT_ERR_HIP, "cusift_ctx_wait: %s", hipGetErrorString(e));
  return CUSIFT_OK;
}

extern "C" int cusift_ctx_reserve(cusift_ctx *ctx, int n_images, int w, int h, const cusift_params *p) {
  TRY(enter(ctx));
  Plan pl;
  TRY(make_plan(pl, n_images, w, h, ialign_up(w, 128), p, wants_side_stream(ctx, p, n_images, w, h),
                wants_stage_all(ctx, p, n_images, w, h), stage_all_limit(ctx)));
  // + one pitched upload image for cusift_extract_host
  return ensure_arena(ctx, pl.total + align_up_sz((size_t)h * ialign_up(w, 128) * sizeof(float), 256));
}

// scratch of cusift_extract_bands: [counters of the bands | running sums | a list of heads per band]
static size_t bands_arena_bytes(int n_bands, int max_pts) { return 512 + (size_t)max_pts * kStagedRecBytes * (size_t)n_bands; }

extern "C" int cusift_ctx_reserve_bands(cusift_ctx *ctx, int n_bands, int max_pts) {
  TRY(enter(ctx));
  if (n_bands < 0 || n_bands > kMaxMultiOctaves || max_pts < 1)
    return fail(CUSIFT_ERR_INVALID, "reserve_bands: 0..%d bands, max_pts >= 1", kMaxMultiOctaves);
  return n_bands ? ensure_arena(ctx, bands_arena_bytes(n_bands, max_pts)) : CUSIFT_OK;
}

extern "C" size_t cusift_ctx_arena_bytes(cusift_ctx *ctx) { return ctx ? ctx->arena_bytes + ctx->dog_bytes : 0; }
extern "C" unsigned long cusift_ctx_forks(cusift_ctx *ctx) { return ctx ? ctx->forks : 0; }

extern "C" int cusift_ctx_timing_enable(cusift_ctx *ctx, int on) {
  TRY(enter(ctx));
  if (!on && ctx->timing) TRY(fold_spans(ctx));
  ctx->timing = on != 0;
  return CUSIFT_OK;
}

extern "C" int cusift_ctx_timing_read(cusift_ctx *ctx, float ms[CUSIFT_NUM_STAGES], int launches[CUSIFT_NUM_STAGES]) {
  TRY(enter(ctx));
  TRY(fold_spans(ctx));
  for (int i = 0; i < CUSIFT_NUM_STAGES; ++i) {
    if (ms) ms[i] = ctx->ms[i];
    if (launches) launches[i] = ctx->launches[i];
  }
  return CUSIFT_OK;
}

extern "C" int cusift_ctx_timing_reset(cusift_ctx *ctx) {
  TRY(enter(ctx));
  TRY(fold_spans(ctx));
  for (int i = 0; i < CUSIFT_NUM_STAGES; ++i) {
    ctx->ms[i] = 0.f;
    ctx->launches[i] = 0;
  }
  return CUSIFT_OK;
}

extern "C" int cusift_kernel_occupancy(const char *kernel, int *blocks_per_cu, int *threads_per_block) {
  if (!kernel || !blocks_per_cu || !threads_per_block) return fail(CUSIFT_ERR_INVALID, "NULL argument");
  const std::string k(kernel);
  int n = 0, t = 256;
  hipError_t e = hipErrorInvalidValue;
  if (k == "detect_fused")  // single-wave workgroups, a 10.5 KB candidate list each
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, detect_fused_kernel<false, (int)sizeof(cusift_point)>, t = 64,
                                                     kDetectWaveLdsFloats * sizeof(float));
  else if (k == "laplace_multi") e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, laplace_multi_fast_kernel<0>, t, 0);
  else if (k == "find_points") e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, find_points_fast_kernel, t, 0);
  else if (k == "scale_down") e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, scale_down_fast_kernel, t, 0);
  else if (k == "describe_all") e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, describe_all_kernel, t = 64, 0);
  else if (k == "orientations") e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, orientations_kernel, t = 64, 0);
  else if (k == "descriptors") e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, descriptors_kernel, t = 64, 0);
  else return fail(CUSIFT_ERR_INVALID, "unknown kernel '%s'", kernel);
  if (e != hipSuccess) return fail(CUSIFT_ERR_HIP, "occupancy query failed: %s", hipGetErrorString(e));
  *blocks_per_cu = n;
  *threads_per_block = t;
  return CUSIFT_OK;
}

// ------------------------------------------------------------------------------------------------
// memory helpers
// ------------------------------------------------------------------------------------------------
extern "C" int cusift_malloc(void **d_ptr, size_t bytes) {
  if (!d_ptr) return fail(CUSIFT_ERR_INVALID, "d_ptr is NULL");
  *d_ptr = nullptr;
  hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 1);
  if (e != hipSuccess) return fail(CUSIFT_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  return CUSIFT_OK;
}

extern "C" int cusift_free(void *d_ptr) {
  if (d_ptr) HIP_TRY(hipFree(d_ptr));
  return CUSIFT_OK;
}

extern "C" int cusift_malloc_host(void **h_ptr, size_t bytes) {
  if (!h_ptr) return fail(CUSIFT_ERR_INVALID, "h_ptr is NULL");
  *h_ptr = nullptr;
  hipError_t e = hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) return fail(CUSIFT_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  return CUSIFT_OK;
}

extern "C" int cusift_free_host(void *h_ptr) {
  if (h_ptr) HIP_TRY(hipHostFree(h_ptr));
  return CUSIFT_OK;
}

extern "C" int cusift_memset(cusift_ctx *ctx, void *d_ptr, int value, size_t bytes) {
  if (!ctx || !d_ptr) return fail(CUSIFT_ERR_INVALID, "NULL argument");
  TRY(enter(ctx));
  HIP_TRY(hipMemsetAsync(d_ptr, value, bytes, ctx->stream));
  return CUSIFT_OK;
}

extern "C" int cusift_memcpy_h2d(cusift_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
  if (!ctx || !d_dst || !h_src) return fail(CUSIFT_ERR_INVALID, "NULL argument");
  TRY(enter(ctx));
  HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return CUSIFT_OK;
}

extern "C" int cusift_memcpy_d2h(cusift_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
  if (!ctx || !h_dst || !d_src) return fail(CUSIFT_ERR_INVALID, "NULL argument");
  TRY(enter(ctx));
  HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return CUSIFT_OK;
}

extern "C" int cusift_memcpy_d2d(cusift_ctx *ctx, void *d_dst, const void *d_src, size_t bytes) {
  if (!ctx || !d_dst || !d_src) return fail(CUSIFT_ERR_INVALID, "NULL argument");
  TRY(enter(ctx));
  if (bytes == 0) return CUSIFT_OK;
  HIP_TRY(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return CUSIFT_OK;
}

extern "C" int cusift_image_h2d(cusift_ctx *ctx, float *d_dst, int dst_pitch, const float *h_src, int w, int h) {
  if (!ctx || !d_dst || !h_src || w < 1 || h < 1 || dst_pitch < w) return fail(CUSIFT_ERR_INVALID, "bad argument");
  TRY(enter(ctx));
  // cuImage::HostToDevice, cuImage.cu:83-92: dense host rows -> pitched device rows
  HIP_TRY(hipMemcpy2DAsync(d_dst, sizeof(float) * dst_pitch, h_src, sizeof(float) * w, sizeof(float) * w, h,
                           hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return CUSIFT_OK;
}

extern "C" int cusift_image_d2h(cusift_ctx *ctx, float *h_dst, const float *d_src, int src_pitch, int w, int h) {
  if (!ctx || !h_dst || !d_src || w < 1 || h < 1 || src_pitch < w) return fail(CUSIFT_ERR_INVALID, "bad argument");
  TRY(enter(ctx));
  HIP_TRY(hipMemcpy2DAsync(h_dst, sizeof(float) * w, d_src, sizeof(float) * src_pitch, sizeof(float) * w, h,
                           hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return CUSIFT_OK;
}

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
static int scale_down_impl(cusift_ctx *ctx, float *d_dst, int dst_pitch, size_t dst_stride, const float *d_src, int w,
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
static bool wants_small_pyramid(const cusift_ctx *ctx, int n_images, int w, int h) {
  if (ctx->knobs.small_pyramid == 0 || ctx->knobs.force_generic) return false;
  if (ctx->knobs.small_pyramid > 0) return true;
  return !ctx->timing && (size_t)n_images * (size_t)w * (size_t)h <= kPyramidSmallPixels;
}

static int pyramid_small_impl(cusift_ctx *ctx, const float *const *base, const int *w, const int *h, const int *pitch,
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

static bool detect_fused_ok(const float *d_img, int w, int h, int pitch, size_t img_stride) {
  return (pitch % 4 == 0) && (((uintptr_t)d_img % 16) == 0) && (img_stride % 4 == 0) && w >= 4 && h >= 3 &&
         ((size_t)h * pitch * sizeof(float) < (1ull << 31));
}

// Chunk height of the fused detection: centre rows per wave (see the comment in detect_impl).
static int detect_rows(const cusift_ctx *ctx, int rows_total, int strips, int n_images, int concurrent) {
  int rows_lo = 2, rows_hi = concurrent >= 2 ? 112 : 64;
  rows_bounds(ctx, kKnobDetect, rows_lo, rows_hi);
  const double wave_rows = (double)rows_total * strips * n_images;
  double coef = concurrent >= 2 ? 0.09 : 0.05;
  if (ctx->knobs.detect_rows_coef > 0.0) coef = ctx->knobs.detect_rows_coef;  // tuning experiments only
  return std::max(rows_lo, std::min(rows_hi, (int)lround(coef * sqrt(wave_rows))));
}

static int detect_impl(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, size_t img_stride, float init_blur,
                       float peak_thresh, float edge_thresh, float subsampling, cusift_point *d_points, int max_pts,
                       unsigned int *d_counters, int n_images, RowWindow rw, int cy_begin, int cy_end,
                       int concurrent = 1, bool heads = false, bool side = false) {
  // heads: `d_points` is a staging list of the context (kStagedRecBytes per keypoint); side: the launch goes to the
  // context's side stream (cusift_extract_batch)
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
  const int rows = detect_rows(ctx, rows_total, strips, n_images, concurrent);
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
  auto kernel = heads ? (ident ? detect_fused_kernel<true, kStagedRecBytes> : detect_fused_kernel<false, kStagedRecBytes>)
                      : (ident ? detect_fused_kernel<true, kWhole> : detect_fused_kernel<false, kWhole>);
  hipLaunchKernelGGL(kernel, grid, dim3(64 * wpb), cube_bytes, side ? ctx->side : ctx->stream, d_img, w, h, pitch,
                     (long)img_stride, d_points, max_pts, d_counters, rows, TP, P, rw, cy_begin, cy_end);
  return check_launch("detect_multi");
}

// The fused detection of several octaves of a batch in ONE launch (detect_multi_kernel): whole images, general taps,
// keypoint HEADS to a staging list per octave.  `octaves` in launch order (largest first: the small ones fill its tail).
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

static int detect_multi_impl(cusift_ctx *ctx, const MultiOctave *octaves, int n_octaves, float peak_thresh,
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

static int keypoint_grid_x(int max_pts, int n_images) {
  // persistent grid: enough waves to fill 256 CUs x 32 wave slots, never more than max_pts per image
  int per_image = std::max(1, (256 * 32 * 2) / std::max(1, n_images));
  return std::max(1, std::min(max_pts, std::min(per_image, 4096)));
}

static int orientations_impl(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, size_t img_stride,
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

static int descriptors_impl(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, size_t img_stride,
                            cusift_point *d_points, int max_pts, const unsigned int *d_first,
                            const unsigned int *d_counters, float subsampling, int tex_frac_bits, int n_images,
                            RowWindow rw, int root_sift = 0, unsigned int *d_flags = nullptr) {
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
  hipLaunchKernelGGL(join_counts_kernel, dim3(1), dim3(256), 0, ctx->stream, d_counter, G, seg_end, 1, max_pts, ctx->d_queue);
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
  if (op < 0 || op > 3 || !d_a || !d_out || (op == 2 && !d_b) || (op == 3 && !d_out2))
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

// ------------------------------------------------------------------------------------------------
// drivers
// ------------------------------------------------------------------------------------------------
// Bytes of DoG planes the two-stage path needs: the largest searched octave that does not take the fused detection
// (0 when every octave does).  `arena_base`: where octaves >= 1 live (their alignment is what matters).
static size_t two_stage_dog_bytes(const cusift_ctx *ctx, const Plan &pl, const cusift_params *prm, const float *d_imgs,
                                  size_t image_stride, const char *arena_base, int n_images) {
  size_t need = 0;
  const bool generic = ctx->knobs.force_generic;
  for (int o = 0; o < pl.n_oct; ++o) {
    if (!(prm->lowest_scale < pl.sub[o] * 2.0f)) continue;
    const float *b = o == 0 ? d_imgs : (const float *)(arena_base + pl.base_off[o]);
    const size_t st = o == 0 ? image_stride : (size_t)pl.h[o] * pl.p[o];
    if (!prm->fused_detect || generic || !detect_fused_ok(b, pl.w[o], pl.h[o], pl.p[o], st))
      need = std::max(need, (size_t)n_images * kNumDog * pl.h[o] * pl.p[o] * sizeof(float));
  }
  return need;
}

extern "C" int cusift_extract_batch(cusift_ctx *ctx, const float *d_imgs, int n_images, int w, int h, int pitch,
                                    size_t image_stride, const cusift_params *prm, cusift_point *d_points,
                                    unsigned int *d_counters) {
  TRY(enter(ctx));
  if (!d_imgs || !d_points || !d_counters) return fail(CUSIFT_ERR_INVALID, "extract: missing data");
  if (n_images > 1 && image_stride < (size_t)h * pitch) return fail(CUSIFT_ERR_INVALID, "image_stride too small");
  Plan pl;
  TRY(make_plan(pl, n_images, w, h, pitch, prm, wants_side_stream(ctx, prm, n_images, w, h),
                wants_stage_all(ctx, prm, n_images, w, h), stage_all_limit(ctx)));
  TRY(ensure_arena(ctx, pl.total));
  if (const size_t dog_need = two_stage_dog_bytes(ctx, pl, prm, d_imgs, image_stride, ctx->arena, n_images))
    TRY(ensure_dog(ctx, dog_need));  // sized once for the largest two-stage octave, before anything is enqueued

  StageTimer total(ctx, CUSIFT_STAGE_TOTAL);

  const float *base[kMaxOctaves];
  size_t stride[kMaxOctaves];
  base[0] = d_imgs;
  stride[0] = image_stride;
  for (int o = 1; o < pl.n_oct; ++o) {
    base[o] = (const float *)(ctx->arena + pl.base_off[o]);
    stride[o] = (size_t)pl.h[o] * pl.p[o];
  }
  unsigned int *first = (unsigned int *)(ctx->arena + pl.first_off);
  // With fused_detect the keypoint stages run once, after the last octave's detection, over the flattened list
  // of all keypoints of the batch (describe_all_kernel); otherwise per octave like the reference.  The DETECTION
  // kernel is chosen per octave: the fused one wherever it applies (16-byte aligned rows, w >= 4, h >= 3), the
  // two-stage pair for an octave where it does not (a 2x1 coarsest octave, a caller's odd pitch) -- one such octave
  // no longer demotes the others.
  const bool generic = ctx->knobs.force_generic;
  const bool flat = prm->fused_detect && n_images <= kMaxFlatImages && !generic;
  auto searched = [&](int o) { return prm->lowest_scale < pl.sub[o] * 2.0f; };  // cuSIFT.cu:194
  auto fused_ok = [&](int o) { return detect_fused_ok(base[o], pl.w[o], pl.h[o], pl.p[o], stride[o]); };

  // Where the octaves' keypoints go (sift_types.h: SegmentTable).  One stream searching coarsest first leaves SiftData
  // in list order by itself; the moment two detections may overlap -- octave 0 on the side stream, the coarser octaves
  // in one launch -- they append to lists of their own (record heads in the arena) and describe_all_kernel joins them.
  //   stage_all   every searched octave to its own list: needs the fused kernel for every searched octave
  //   forked      octave 0 to a list of its own and to the side stream; the coarser ones in place (or staged too)
  bool stage_all = flat && pl.staged_octaves == pl.n_oct;
  bool any_coarser = false;
  for (int o = 0; o < pl.n_oct; ++o) {
    if (!searched(o)) continue;
    stage_all = stage_all && fused_ok(o);
    any_coarser = any_coarser || o > 0;
  }
  bool forked = flat && pl.fork && pl.staged_octaves >= 1 && searched(0) && fused_ok(0) && any_coarser;
  if (forked && ensure_side_stream(ctx) != CUSIFT_OK) forked = false;  // no stream runs beside this one: one stream
  const size_t list_bytes = (size_t)n_images * prm->max_pts * kStagedRecBytes;
  char *const lists = ctx->arena + pl.staged_off;
  unsigned int *const seg_counts = first;  // [octave][image]: `first` is free when the keypoint stages run once
  unsigned int *const seg_end = (unsigned int *)(ctx->arena + pl.seg_end_off);
  auto list_of = [&](int o) { return reinterpret_cast<cusift_point *>(lists + (size_t)o * list_bytes); };
  SegmentTable G;
  memset(&G, 0, sizeof(G));
  if (stage_all) {
    G.n_seg = pl.n_oct;
    for (int r = 0; r < pl.n_oct; ++r) {  // list order: coarsest octave first
      const int o = pl.n_oct - 1 - r;
      G.base[r] = reinterpret_cast<const char *>(list_of(o));
      G.count[r] = seg_counts + (size_t)o * n_images;
    }
  } else if (forked) {
    G.n_seg = 2;
    G.base[0] = nullptr;  // the coarser octaves: in place, the caller's counter
    G.count[0] = d_counters;
    G.base[1] = reinterpret_cast<const char *>(list_of(0));
    G.count[1] = seg_counts;
  }
  // A small call's dispatches are most of its time, so its housekeeping rides along: the ScaleDown chain in one launch
  // (which also clears the lists' counters), all octaves in one detection launch (which also clears describe_all's work
  // cursors), and describe_all_kernel joins the lists itself -- pyramid, detection, description: three dispatches.
  const bool small_pyramid = pl.n_oct >= 2 && wants_small_pyramid(ctx, n_images, w, h);
  int n_one_launch = 0;  // octaves the one detection launch would take
  if (stage_all && !ctx->knobs.no_multi)
    for (int o = forked ? 1 : 0; o < pl.n_oct && n_one_launch < kMaxMultiOctaves; ++o) n_one_launch += searched(o) ? 1 : 0;
  const bool one_launch = n_one_launch >= 2;
  const bool self_join = stage_all && one_launch;              // no join_counts_kernel: describe_all_kernel joins
  const bool pyramid_clears = small_pyramid && stage_all && !forked;  // (a forked octave 0 may count before the pyramid runs)
  // cuSIFT.cu:69: point counter = 0 (with every octave staged the join writes it instead)
  if (!stage_all) HIP_TRY(hipMemsetAsync(d_counters, 0, sizeof(unsigned int) * n_images, ctx->stream));
  const size_t n_seg_counts = (size_t)n_images * (stage_all ? pl.n_oct : 1);
  if (G.n_seg && !pyramid_clears) {
    // (a multiple of 64 bytes: the runtime fills an odd size with two dispatches; the region is kMaxOctaves x n_images)
    HIP_TRY(hipMemsetAsync(seg_counts, 0, std::min(align_up_sz(sizeof(unsigned int) * n_seg_counts, 64),
                                                   sizeof(unsigned int) * n_images * kMaxOctaves), ctx->stream));
  }

  if (forked) {
    ctx->forks++;
    HIP_TRY(hipEventRecord(ctx->ev_fork, ctx->stream));
    HIP_TRY(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
    const int rc = detect_impl(ctx, d_imgs, w, h, pitch, image_stride, (float)pl.blur[0], prm->peak_thresh,
                               prm->edge_thresh, pl.sub[0], list_of(0), prm->max_pts, seg_counts, n_images,
                               RowWindow{0, h}, 0, h, 1, true, true);
    const hipError_t e = hipEventRecord(ctx->ev_join, ctx->side);
    if (rc != CUSIFT_OK || e != hipSuccess) {
      (void)hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0);  // never leave the side stream forked (a capture would not end)
      if (rc != CUSIFT_OK) return rc;
      HIP_TRY(e);
    }
  }
  // the side stream rejoins the context's stream however the work in between ends
  auto on_main = [&]() -> int {
    // ExtractSiftLoop, cuSIFT.cu:175-192: build the pyramid finest -> coarsest -- a small call's first levels in one launch
    int built = 0;
    if (small_pyramid) {
      built = std::min(pl.n_oct - 1, kMaxPyramidLevels);
      TRY(pyramid_small_impl(ctx, base, pl.w, pl.h, pl.p, stride, built, n_images, 0.5f,
                             pyramid_clears ? seg_counts : nullptr, pyramid_clears ? (int)n_seg_counts : 0));
    }
    for (int o = built + 1; o < pl.n_oct; ++o)
      TRY(cusift_scale_down(ctx, const_cast<float *>(base[o]), pl.p[o], stride[o], base[o - 1], pl.w[o - 1], pl.h[o - 1],
                            pl.p[o - 1], stride[o - 1], n_images, 0.5f));  // cuSIFT.cu:185
    // With a list per octave all octaves (but a forked octave 0) are searched by ONE launch, largest first
    bool in_one_launch[kMaxOctaves] = {false};
    if (one_launch) {
      MultiOctave mo[kMaxMultiOctaves];
      int n_mo = 0;
      for (int o = forked ? 1 : 0; o < pl.n_oct && n_mo < kMaxMultiOctaves; ++o)
        if (searched(o)) {
          mo[n_mo++] = MultiOctave{base[o], pl.w[o], pl.h[o], pl.p[o], stride[o], (float)pl.blur[o], pl.sub[o], list_of(o),
                                   seg_counts + (size_t)o * n_images};
          in_one_launch[o] = true;
        }
      TRY(detect_multi_impl(ctx, mo, n_mo, prm->peak_thresh, prm->edge_thresh, prm->max_pts, n_images,
                            forked ? 1 : prm->concurrent_batches, ctx->d_queue));
    }
    // ... and search it coarsest first (the recursion unwinds: cuSIFT.cu:190-196)
    for (int o = pl.n_oct - 1; o >= (forked ? 1 : 0); --o) {
      if (!searched(o) || in_one_launch[o]) continue;
      // ExtractSiftOctave, cuSIFT.cu:204-270
      unsigned int *fst = first + (size_t)o * n_images;  // cuSIFT.cu:243 (fstPts), kept on the device
      if (!flat)
        HIP_TRY(hipMemcpyAsync(fst, d_counters, sizeof(unsigned int) * n_images, hipMemcpyDeviceToDevice, ctx->stream));
      if (prm->fused_detect && !generic && fused_ok(o)) {
        TRY(detect_impl(ctx, base[o], pl.w[o], pl.h[o], pl.p[o], stride[o], (float)pl.blur[o], prm->peak_thresh,
                        prm->edge_thresh, pl.sub[o], stage_all ? list_of(o) : d_points, prm->max_pts,
                        stage_all ? seg_counts + (size_t)o * n_images : d_counters, n_images, RowWindow{0, pl.h[o]}, 0,
                        pl.h[o], forked ? 1 : prm->concurrent_batches, stage_all));
      } else {
        const size_t dstride = (size_t)kNumDog * pl.h[o] * pl.p[o];
        float *dog = ctx->dog;
        TRY(cusift_laplace_multi(ctx, base[o], pl.w[o], pl.h[o], pl.p[o], stride[o], (float)pl.blur[o], dog, dstride,
                                 n_images));
        TRY(cusift_find_points_multi(ctx, dog, pl.w[o], pl.h[o], pl.p[o], dstride, prm->peak_thresh, prm->edge_thresh,
                                     pl.sub[o], d_points, prm->max_pts, d_counters, n_images));
      }
      if (flat) continue;
      TRY(cusift_compute_orientations(ctx, base[o], pl.w[o], pl.h[o], pl.p[o], stride[o], d_points, prm->max_pts, fst,
                                      d_counters, prm->tex_frac_bits, n_images));
      TRY(descriptors_impl(ctx, base[o], pl.w[o], pl.h[o], pl.p[o], stride[o], d_points, prm->max_pts, fst, d_counters,
                           pl.sub[o], prm->tex_frac_bits, n_images, RowWindow{0, pl.h[o]}, prm->root_sift));
    }
    return CUSIFT_OK;
  };
  const int rc_main = on_main();
  if (forked) {
    const hipError_t e = hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0);
    if (rc_main == CUSIFT_OK) HIP_TRY(e);
  }
  if (rc_main != CUSIFT_OK) return rc_main;
  if (flat) {
    OctaveTable T;
    memset(&T, 0, sizeof(T));
    T.n_oct = pl.n_oct;
    for (int o = 0; o < pl.n_oct; ++o) {
      T.base[o] = base[o];
      T.stride[o] = (long)stride[o];
      T.w[o] = pl.w[o];
      T.h[o] = pl.h[o];
      T.pitch[o] = pl.p[o];
      T.sub[o] = pl.sub[o];
    }
    float q, inv_q;
    frac_consts(prm->tex_frac_bits, q, inv_q);
    // persistent grid = exactly the blocks that are resident at once (a larger static grid would run in
    // rounds and leave the second round's items waiting); items are interleaved over the blocks
    if (ctx->describe_grid == 0) {
      int per_cu = 0, cus = 0;
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, describe_all_kernel, 64, 0));
      HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
      ctx->describe_grid = std::max(1, per_cu) * std::max(1, cus);
    }
    const long cap = (long)n_images * prm->max_pts;
    // a multiple of the shard count (the kernel deals items to shards by workgroup index)
    const long want = std::max(1L, std::min(cap, (long)ctx->describe_grid));
    dim3 grid((unsigned int)std::max<long>(kQueueShards, want / kQueueShards * kQueueShards));
    unsigned int *queue = ctx->d_queue;  // the kernel's work cursors, zero at launch
    if (self_join) {
      // (the detection launch cleared the cursors; describe_all_kernel joins the lists itself)
    } else if (G.n_seg) {
      hipLaunchKernelGGL(join_counts_kernel, dim3(1), dim3(256), 0, ctx->stream, d_counters, G, seg_end, n_images,
                         prm->max_pts, queue);
      TRY(check_launch("join_counts"));
    } else {
      HIP_TRY(hipMemsetAsync(queue, 0, kQueueShards * 128, ctx->stream));
    }
    StageTimer t(ctx, CUSIFT_STAGE_DESCRIBE_ALL);
    hipLaunchKernelGGL(describe_all_kernel, grid, dim3(64), 0, ctx->stream, T, d_points, prm->max_pts, d_counters,
                       n_images, q, inv_q, prm->root_sift, queue, G,
                       self_join ? (const unsigned int *)nullptr : (const unsigned int *)seg_end);
    TRY(check_launch("describe_all"));
  }
  return CUSIFT_OK;
}

// ------------------------------------------------------------------------------------------------
// replayable extraction: the launch sequence of cusift_extract_batch recorded once as a hipGraph
// ------------------------------------------------------------------------------------------------
struct cusift_graph {
  cusift_ctx *ctx = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  unsigned long scratch_gen = 0;  // the recording refers to the context's scratch (arena, DoG planes of the
                                  // two-stage path) as it was: any later re-allocation invalidates it
  int nodes = 0;
};

extern "C" int cusift_graph_create(cusift_ctx *ctx, cusift_graph **out, const float *d_imgs, int n_images, int w, int h,
                                   int pitch, size_t image_stride, const cusift_params *prm, cusift_point *d_points,
                                   unsigned int *d_counters) {
  if (!ctx || !out) return fail(CUSIFT_ERR_INVALID, "NULL argument");
  TRY(enter(ctx));
  *out = nullptr;
  if (!ctx->stream) return fail(CUSIFT_ERR_INVALID, "graph capture needs a real stream (the context borrows the null stream)");
  // For the length of this call the context is "recording": no stage-timer events (they are not part of a recording)
  // and the one-stream launch sequence (wants_side_stream) -- the plan below is the one cusift_extract_batch will make.
  struct Recording {
    cusift_ctx *c;
    bool timing;
    explicit Recording(cusift_ctx *ctx) : c(ctx), timing(ctx->timing) { c->timing = false; c->recording = true; }
    ~Recording() { c->timing = timing; c->recording = false; }
  } recording(ctx);
  Plan pl;
  // the side stream is found (and probed: that waits) before the capture starts; the fork and the join become edges
  const bool fork = wants_side_stream(ctx, prm, n_images, w, h) && ensure_side_stream(ctx) == CUSIFT_OK;
  TRY(make_plan(pl, n_images, w, h, pitch, prm, fork, wants_stage_all(ctx, prm, n_images, w, h), stage_all_limit(ctx)));
  // everything that allocates or synchronises happens before the capture starts
  TRY(ensure_arena(ctx, pl.total));
  // the DoG planes of every octave that takes the two-stage path (see cusift_extract_batch), sized up front
  const size_t dog_need = two_stage_dog_bytes(ctx, pl, prm, d_imgs, image_stride, ctx->arena, n_images);
  if (dog_need) TRY(ensure_dog(ctx, dog_need));
  if (ctx->describe_grid == 0) {
    int per_cu = 0, cus = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, describe_all_kernel, 64, 0));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
    ctx->describe_grid = std::max(1, per_cu) * std::max(1, cus);
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  cusift_graph *g = new cusift_graph();
  g->ctx = ctx;
  hipError_t e = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) {
    delete g;
    return fail(CUSIFT_ERR_HIP, "hipStreamBeginCapture failed: %s", hipGetErrorString(e));
  }
  const int rc = cusift_extract_batch(ctx, d_imgs, n_images, w, h, pitch, image_stride, prm, d_points, d_counters);
  e = hipStreamEndCapture(ctx->stream, &g->graph);
  if (rc != CUSIFT_OK || e != hipSuccess || !g->graph) {
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    if (rc != CUSIFT_OK) return rc;
    return fail(CUSIFT_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
  }
  e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(g->graph);
    delete g;
    return fail(CUSIFT_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
  }
  size_t n_nodes = 0;
  (void)hipGraphGetNodes(g->graph, nullptr, &n_nodes);
  g->nodes = (int)n_nodes;
  g->scratch_gen = ctx->scratch_gen;
  *out = g;
  return CUSIFT_OK;
}

extern "C" int cusift_graph_launch(cusift_graph *g) {
  if (!g || !g->exec) return fail(CUSIFT_ERR_INVALID, "graph is NULL");
  if (g->ctx->scratch_gen != g->scratch_gen)
    return fail(CUSIFT_ERR_INVALID,
                "the context's scratch (arena / DoG planes) was re-allocated after this graph was recorded; record it again");
  HIP_TRY(hipSetDevice(g->ctx->device));
  HIP_TRY(hipGraphLaunch(g->exec, g->ctx->stream));
  return CUSIFT_OK;
}

extern "C" int cusift_graph_nodes(cusift_graph *g) { return g ? g->nodes : 0; }

extern "C" int cusift_graph_destroy(cusift_graph *g) {
  if (!g) return CUSIFT_OK;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
  return CUSIFT_OK;
}

extern "C" int cusift_extract(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, const cusift_params *prm,
                              cusift_point *d_points, cusift_point *h_points, int *num_pts) {
  TRY(enter(ctx));
  if (!num_pts) return fail(CUSIFT_ERR_INVALID, "num_pts is NULL");
  *num_pts = 0;
  TRY(cusift_extract_batch(ctx, d_img, 1, w, h, pitch, (size_t)h * pitch, prm, d_points, ctx->d_counter1));
  unsigned int cnt = 0;
  HIP_TRY(hipMemcpyAsync(&cnt, ctx->d_counter1, sizeof(cnt), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  // cuSIFT.cu:107-110
  const int n = cnt < (unsigned int)prm->max_pts ? (int)cnt : prm->max_pts;
  *num_pts = n;
  if (h_points && n > 0) {  // SiftData::Synchronize, cuSIFT.cu:52-59
    HIP_TRY(hipMemcpyAsync(h_points, d_points, sizeof(cusift_point) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  return CUSIFT_OK;
}

extern "C" int cusift_extract_host(cusift_ctx *ctx, const float *h_img, int w, int h, const cusift_params *prm,
                                   cusift_point *d_points, cusift_point *h_points, int *num_pts) {
  TRY(enter(ctx));
  if (!h_img) return fail(CUSIFT_ERR_INVALID, "image is NULL");
  if (w < 1 || h < 1) return fail(CUSIFT_ERR_INVALID, "bad image size %dx%d", w, h);
  const int pitch = ialign_up(w, 128);  // cuImage::AllocateWithHostMemory, cuImage.cu:11-13
  Plan pl;
  TRY(make_plan(pl, 1, w, h, pitch, prm, wants_side_stream(ctx, prm, 1, w, h),
                wants_stage_all(ctx, prm, 1, w, h), stage_all_limit(ctx)));  // the plan cusift_extract_batch will make
  const size_t img_bytes = align_up_sz((size_t)h * pitch * sizeof(float), 256);
  TRY(ensure_arena(ctx, pl.total + img_bytes));
  float *d_img = (float *)(ctx->arena + pl.total);
  HIP_TRY(hipMemcpy2DAsync(d_img, sizeof(float) * pitch, h_img, sizeof(float) * w, sizeof(float) * w, h,
                           hipMemcpyHostToDevice, ctx->stream));
  return cusift_extract(ctx, d_img, w, h, pitch, prm, d_points, h_points, num_pts);
}
