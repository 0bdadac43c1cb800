// sift_context.hip -- the context of the C ABI (include/cusift_amd.h): errors, device selection, the scratch arena, the
// launch policy, the side stream, the stage timers and the memory helpers.  Host logic follows cuSIFT.cu / cutils.h (cited
// per function); nothing here allocates, frees or reads back between stages once the arena is sized.
#include "sift_host.h"

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

int cusift_fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

Knobs read_knobs() {
  Knobs k;
  // (the product build reads NO environment variable here: CUSIFT_OCTAVE_OVERLAP, the one it read until round 5 for
  // unchanged callers of the C++ shim, is now read by that shim -- include/cuSIFT.h -- and set through cusift_ctx_set_policy)
#ifdef CUSIFT_LAB
  if (const char *e = getenv("CUSIFT_OCTAVE_OVERLAP")) k.octave_overlap = std::max(0, std::min(3, atoi(e)));
  auto text = [](const char *name) -> const char * { return getenv(name); };
  auto num = [&](const char *name, int unset) { const char *e = text(name); return e ? atoi(e) : unset; };
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
  k.side_debug = text("CUSIFT_SIDE_DEBUG") != nullptr;
  k.stage_all_mb = num("CUSIFT_STAGE_ALL_MB", 0);
  k.small_pyramid = num("CUSIFT_SMALL_PYRAMID", -1);
  // the policy keys as well, for the A/B scripts that drive bench.py from the shell
  k.force_generic = text("CUSIFT_FORCE_GENERIC") != nullptr;
  k.match_splits = num("CUSIFT_MATCH_SPLITS", 0);
  k.stage_all = num("CUSIFT_STAGE_ALL", -1);
  k.no_multi = text("CUSIFT_NO_MULTI") != nullptr;
  k.pyramid_in_detect = num("CUSIFT_PYRAMID_IN_DETECT", -1);
  k.unordered_coarse = text("CUSIFT_UNORDERED_COARSE") != nullptr;
#endif
  return k;
}

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

int make_plan(Plan &pl, int n_images, int w, int h, int pitch, const cusift_params *prm, bool fork,
              bool stage_all, size_t stage_all_limit) {
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
  ctx->seg_clean_ptr = nullptr;
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
  // (<= 11 bits: the texture model's weight product A B / 2^q is then exact in fp32 -- sift_keypoints.hip, bilinear_weights)
  if (frac_bits > 0 && frac_bits <= 11) {
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
bool wants_side_stream(const cusift_ctx *ctx, const cusift_params *prm, int n_images, int w, int h) {
  const int mode = ctx->knobs.octave_overlap;  // 0 (default): the context has not been asked to fork
  if (mode == 0 || ctx->timing || ctx->knobs.force_generic || ctx->side_failed) return false;
  if (!prm || !prm->fused_detect || n_images < 1 || n_images > kMaxFlatImages) return false;
  if (mode == 3) return true;  // tests: whatever the size, also inside a recording
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
size_t stage_all_limit(const cusift_ctx *ctx) {
  return ctx->knobs.stage_all_mb > 0 ? (size_t)ctx->knobs.stage_all_mb << 20 : kMaxStagedAllBytes;
}
bool wants_stage_all(const cusift_ctx *ctx, const cusift_params *prm, int n_images, int w, int h) {
  if (ctx->knobs.stage_all == 0 || ctx->knobs.force_generic) return false;
  if (!prm || !prm->fused_detect || n_images < 1 || n_images > kMaxFlatImages) return false;
  if (ctx->knobs.stage_all > 0) return true;  // tests: whenever the lists fit
  if (wants_pyramid_in_detect(ctx, prm, n_images, w, h) > 0) return true;  // finest first needs a list per octave
  if (ctx->timing) return false;  // the stage timers bracket the reference's launch-per-octave sequence
  return prm->concurrent_batches < 2 || (size_t)n_images * (size_t)w * (size_t)h <= kListsMaxPixelsPipelined;
}

// The pyramid as a by-product of the detection (CUSIFT_POLICY_PYRAMID_IN_DETECT; detect_fused_kernel<.., kDown>): octave
// o's detection writes octave o + 1's image from its own row window, so the ScaleDown launches -- 0.2 ms of HBM-bound
// re-reading per 64 x 1080p, a fifth of a lone caller's step -- disappear for ~5 % more vector instructions in the
// detection, and the octaves are searched finest first (lists per octave).  What it costs is the one-launch detection
// of the coarser octaves: a chain of dependent launches has a tail per octave, which a caller with several batches in
// flight fills with the other batches' kernels and a lone caller does not.  Measured on MI355X (tools/ab_pyramid.py,
// profiles/r05/ab_pyramid_by_size.txt; 1080p frames per call, ms per call, ScaleDown chain first -> every octave):
//   four calls in flight  1: 0.0409 -> 0.0390   3: 0.0776 -> 0.0725   8: 0.1645 -> 0.1450   16: 0.288 -> 0.267   64: 1.033 -> 0.976
//   a lone caller         1: 0.0649 -> 0.0973  16: 0.342 -> 0.392    32: 0.645 -> 0.662    48: 0.941 -> 0.932   64: 1.241 -> 1.153
// A lone caller's middle ground is "octave 0 only" (1): octave 0's detection hands octave 1 over -- the large ScaleDown is
// the one worth saving -- and the coarser octaves keep their short ScaleDown chain and their ONE launch: 32 frames
// 0.645 -> 0.640, 48: 0.941 -> 0.910 (every octave: 0.932), 64: 1.241 -> 1.152 (every octave: 1.154); 24: 0.496 -> 0.505.
// So by default: a pipelining caller (concurrent_batches >= 2) every octave from one 1080p frame's worth of pixels up
// (2,000,000: a single 1920 x 1080 frame, 2,073,600 pixels, qualifies -- until round 6 the limit was 2 << 20 = 2,097,152
// and it did not), a lone caller octave 0 only from 64,000,000 pixels per call (31 frames of 1080p).  0: never; 1: octave 0
// only; 2: every octave.
constexpr size_t kPyramidInDetectMinPixelsPipelined = 2000000, kPyramidInDetectMinPixelsLone = 64000000;
int wants_pyramid_in_detect(const cusift_ctx *ctx, const cusift_params *prm, int n_images, int w, int h) {
  const int mode = ctx->knobs.pyramid_in_detect;
  if (mode == 0 || ctx->knobs.force_generic || ctx->knobs.stage_all == 0) return 0;
  if (!prm || !prm->fused_detect || prm->num_octaves < 2 || n_images < 1 || n_images > kMaxFlatImages) return 0;
  // (the stage timers do not change this: every launch of the finest-first sequence is a detection launch and is
  // bracketed as one -- the ScaleDown stage then simply reports no launches)
  if (mode > 0) return std::min(mode, 2);
  const size_t px = (size_t)n_images * (size_t)w * (size_t)h;
  if (prm->concurrent_batches >= 2) return px >= kPyramidInDetectMinPixelsPipelined ? 2 : 0;
  return px >= kPyramidInDetectMinPixelsLone ? 1 : 0;
}

// OPT-IN since round 4 (cusift_ctx_set_policy(ctx, CUSIFT_POLICY_SIDE_STREAM, 1 | 2), or CUSIFT_OCTAVE_OVERLAP=1 | 2 in the
// environment of an unchanged caller): whether a second stream runs beside the first depends on what else the process
// has created, and a context that decided that by itself -- by timing -- was a silent performance cliff under a
// profiler, on a busy box, or in a host with streams of its own.  1 trusts the caller: the stream is created and used.
// 2 keeps the probe below as an explicit request.
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

int ensure_side_stream(cusift_ctx *ctx) {
  // (policy value 2 is "after the probe": a stream taken without it -- under value 1 or 3 -- does not qualify; it is
  // let go, once idle, and the probe runs)
  if (ctx->side && ctx->knobs.octave_overlap == 2 && !ctx->side_probed) {
    HIP_TRY(hipStreamSynchronize(ctx->side));
    HIP_TRY(hipStreamDestroy(ctx->side));
    ctx->side = nullptr;
  }
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
  if (ctx->knobs.octave_overlap != 2) {  // 1 / 3: no probe -- the caller says a second stream is worth having
    ctx->side_probed = false;
    hipError_t e = hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (e != hipSuccess) {
      ctx->side = nullptr;
      ctx->side_failed = true;
      return fail(CUSIFT_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    return CUSIFT_OK;
  }
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
    if (beside) {
      ctx->side = cand;
      ctx->side_probed = true;
    } else
      rejected[n_rejected++] = cand;
  }
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
int ctx_create_impl(cusift_ctx **out, int device, void *hip_stream, bool borrow) {
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
  if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->h_counter1, 256, hipHostMallocDefault);
  if (e != hipSuccess) {
    if (ctx->d_counter1) (void)hipFree(ctx->d_counter1);
    if (ctx->d_queue) (void)hipFree(ctx->d_queue);
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
  if (ctx->h_counter1) (void)hipHostFree(ctx->h_counter1);
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
size_t bands_arena_bytes(int n_bands, int max_pts) { return 512 + (size_t)max_pts * kStagedRecBytes * (size_t)n_bands; }

extern "C" int cusift_ctx_reserve_bands(cusift_ctx *ctx, int n_bands, int max_pts) {
  TRY(enter(ctx));
  if (n_bands < 0 || n_bands > kMaxMultiOctaves || max_pts < 1)
    return fail(CUSIFT_ERR_INVALID, "reserve_bands: 0..%d bands, max_pts >= 1", kMaxMultiOctaves);
  return n_bands ? ensure_arena(ctx, bands_arena_bytes(n_bands, max_pts)) : CUSIFT_OK;
}

extern "C" size_t cusift_ctx_arena_bytes(cusift_ctx *ctx) { return ctx ? ctx->arena_bytes + ctx->dog_bytes : 0; }
extern "C" unsigned long cusift_ctx_forks(cusift_ctx *ctx) { return ctx ? ctx->forks : 0; }

extern "C" int cusift_ctx_set_policy(cusift_ctx *ctx, int key, int value) {
  if (!ctx) return fail(CUSIFT_ERR_INVALID, "ctx is NULL");
  Knobs &k = ctx->knobs;
  switch (key) {
    case CUSIFT_POLICY_SIDE_STREAM:
      if (value < 0 || value > 3) return fail(CUSIFT_ERR_INVALID, "CUSIFT_POLICY_SIDE_STREAM: 0..3");
      if (value != k.octave_overlap) ctx->side_failed = false;  // a new request gets a new try
      k.octave_overlap = value;
      return CUSIFT_OK;
    case CUSIFT_POLICY_OCTAVE_LISTS:
      if (value < -1 || value > 1) return fail(CUSIFT_ERR_INVALID, "CUSIFT_POLICY_OCTAVE_LISTS: -1, 0 or 1");
      k.stage_all = value;
      return CUSIFT_OK;
    case CUSIFT_POLICY_GENERIC_KERNELS: k.force_generic = value != 0; return CUSIFT_OK;
    case CUSIFT_POLICY_LAUNCH_PER_OCTAVE: k.no_multi = value != 0; return CUSIFT_OK;
    case CUSIFT_POLICY_MATCH_SPLITS:
      if (value < 0) return fail(CUSIFT_ERR_INVALID, "CUSIFT_POLICY_MATCH_SPLITS: >= 0");
      k.match_splits = value;
      return CUSIFT_OK;
    case CUSIFT_POLICY_TILED_PER_OCTAVE: k.tiled_per_octave = value != 0; return CUSIFT_OK;
    case CUSIFT_POLICY_PYRAMID_IN_DETECT:
      if (value < -1 || value > 2) return fail(CUSIFT_ERR_INVALID, "CUSIFT_POLICY_PYRAMID_IN_DETECT: -1, 0, 1 or 2");
      k.pyramid_in_detect = value;
      return CUSIFT_OK;
  }
  return fail(CUSIFT_ERR_INVALID, "unknown policy key %d", key);
}

extern "C" int cusift_ctx_get_policy(cusift_ctx *ctx, int key, int *value) {
  if (!ctx || !value) return fail(CUSIFT_ERR_INVALID, "ctx / value is NULL");
  const Knobs &k = ctx->knobs;
  switch (key) {
    case CUSIFT_POLICY_SIDE_STREAM: *value = k.octave_overlap; return CUSIFT_OK;
    case CUSIFT_POLICY_OCTAVE_LISTS: *value = k.stage_all; return CUSIFT_OK;
    case CUSIFT_POLICY_GENERIC_KERNELS: *value = k.force_generic; return CUSIFT_OK;
    case CUSIFT_POLICY_LAUNCH_PER_OCTAVE: *value = k.no_multi; return CUSIFT_OK;
    case CUSIFT_POLICY_MATCH_SPLITS: *value = k.match_splits; return CUSIFT_OK;
    case CUSIFT_POLICY_TILED_PER_OCTAVE: *value = k.tiled_per_octave; return CUSIFT_OK;
    case CUSIFT_POLICY_PYRAMID_IN_DETECT: *value = k.pyramid_in_detect; return CUSIFT_OK;
  }
  return fail(CUSIFT_ERR_INVALID, "unknown policy key %d", key);
}

// ---- events on a context's stream: what TimerGPU (cutils.h:94-114) is made of ----
struct cusift_event {
  hipEvent_t ev = nullptr;
  int device = 0;
};

extern "C" int cusift_event_create(cusift_ctx *ctx, cusift_event **out) {
  if (!out) return fail(CUSIFT_ERR_INVALID, "out is NULL");
  *out = nullptr;
  TRY(enter(ctx));
  cusift_event *e = new cusift_event();
  e->device = ctx->device;
  hipError_t r = hipEventCreate(&e->ev);
  if (r != hipSuccess) {
    delete e;
    return fail(CUSIFT_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(r));
  }
  *out = e;
  return CUSIFT_OK;
}

extern "C" int cusift_event_record(cusift_event *ev, cusift_ctx *ctx) {
  if (!ev) return fail(CUSIFT_ERR_INVALID, "event is NULL");
  TRY(enter(ctx));
  HIP_TRY(hipEventRecord(ev->ev, ctx->stream));
  return CUSIFT_OK;
}

extern "C" int cusift_event_wait(cusift_event *ev, cusift_ctx *ctx) {
  if (!ev) return fail(CUSIFT_ERR_INVALID, "event is NULL");
  TRY(enter(ctx));
  if (ev->device != ctx->device) return fail(CUSIFT_ERR_INVALID, "event and context are on different devices");
  HIP_TRY(hipStreamWaitEvent(ctx->stream, ev->ev, 0));
  return CUSIFT_OK;
}

extern "C" int cusift_event_elapsed_ms(cusift_event *start, cusift_event *stop, float *ms) {
  if (!start || !stop || !ms) return fail(CUSIFT_ERR_INVALID, "event / ms is NULL");
  HIP_TRY(hipSetDevice(stop->device));
  HIP_TRY(hipEventSynchronize(stop->ev));
  HIP_TRY(hipEventElapsedTime(ms, start->ev, stop->ev));
  return CUSIFT_OK;
}

extern "C" int cusift_event_destroy(cusift_event *ev) {
  if (!ev) return CUSIFT_OK;
  (void)hipSetDevice(ev->device);
  (void)hipEventDestroy(ev->ev);
  delete ev;
  return CUSIFT_OK;
}

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
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, detect_fused_kernel<false, (int)sizeof(cusift_point), false>, t = 64,
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

