/*
 * cusift_amd.h -- C ABI of the MI355X-native SIFT extraction path (libcusift_amd.so).
 *
 * This is the drop-in boundary for the hot path of danielsuo/cuSIFT: plain pointers and sizes,
 * no C++/HIP/torch types.  The reference has no FFI of its own (it is a C++ static library), so each
 * entry point below names the reference interface it stands in for (file:line relative to the
 * reference root).  The C++ header include/cuSIFT.h re-creates the reference's classes
 * (SiftPoint / SiftData / cuImage / ExtractSift ...) on top of exactly these functions; Python binds
 * them with ctypes (cusift_amd/capi.py).  See INTEGRATION.md.
 *
 * Map of this header (107 entry points; the path's own boundary is the first group):
 *   THE DROP-IN CORE, what include/cuSIFT.h is built on (22): cusift_init / _device_count / _last_error / _version,
 *     cusift_ctx_create / _create_borrowed / _destroy / _synchronize / _stream / _device / _reserve, cusift_malloc /
 *     _free / _malloc_host / _free_host / _memcpy_h2d / _memcpy_d2h / _image_h2d / _image_d2h, cusift_default_params,
 *     cusift_extract / _extract_host / _extract_batch, cusift_scale_down, cusift_rootsift.
 *   STAGE ENTRY POINTS, one per kernel of the reference, for callers that drive the stages themselves and for the parity
 *     tests (15): cusift_scale_down_levels / _laplace_multi / _laplace_taps / _find_points_multi / _detect_multi /
 *     _compute_orientations / _extract_descriptors / _math_eval / _kernel_occupancy; band forms of the strip tiling:
 *     _scale_down_band / _detect_band / _describe_band / _extract_bands; front-end: _image_u8_h2d / _u8_to_f32 / _gaussian3x3.
 *   CONTEXT SERVICES (14): _ctx_wait, _ctx_reserve_bands, _ctx_arena_bytes, _ctx_forks, _ctx_set_policy / _get_policy,
 *     _ctx_timing_enable / _read / _reset, cusift_event_*, cusift_graph_* (replay of one launch sequence).
 *   HOST-TO-HOST PIPELINE (5): cusift_pipe_*.
 *   SIFTDATA ON THE WIRE AND AT REST (10): cusift_pack_points (+ _compact, _trimmed), cusift_expand_* , _sort_points_host,
 *     _memset, _memcpy_d2d, _memcpy2d_d2h.
 *   MORE THAN ONE GPU (37): cusift_comm_* (communicator over RCCL, bound at run time), cusift_allgatherv_* /
 *     _compact_gathered, cusift_exchange_halos / _rows, cusift_tiled_* (one large image over the ranks).
 *   NEXT ROWS OF SURVEY 8f (2): cusift_match, cusift_find_homography.
 *
 * Conventions
 *  - every function returns CUSIFT_OK (0) or a negative cusift_status; cusift_last_error() gives text.
 *    (The reference prints and exit(-1)s, cutils.h:24-48; the C++ shim keeps that behaviour.)
 *  - images are float32, row-pitched; `pitch` is in FLOATS (cuImage.h:11, cuImage.cu:11-13).
 *  - "d_" pointers are device (HBM) addresses, "h_" pointers are host addresses.
 *  - batch entry points take `n_images` images laid out `image_stride` floats apart and write
 *    `max_pts` records per image into d_points[i*max_pts ...] and one counter per image.
 *  - all work is enqueued on the context's HIP stream; only calls documented as blocking wait.
 */
#ifndef CUSIFT_AMD_H
#define CUSIFT_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cusift_status {
  CUSIFT_OK = 0,
  CUSIFT_ERR_INVALID = -1,   /* bad argument (NULL buffer, non-positive size, misaligned pitch ...) */
  CUSIFT_ERR_HIP = -2,       /* a HIP runtime call failed; see cusift_last_error() */
  CUSIFT_ERR_NO_DEVICE = -3, /* no GPU visible */
  CUSIFT_ERR_NOMEM = -4
} cusift_status;

/* SiftPoint, cuSIFT.h:10-30.  147 x 4 B = 588 B, no padding; callers index h_data[i] directly
 * (test/detector.cpp:56) so the layout is ABI.  Extraction writes coords2D, scale, sharpness,
 * edgeness, orientation, subsampling and data; the other fields are left as they were. */
typedef struct cusift_point {
  float coords2D[2];
  float scale;
  float sharpness;
  float edgeness;
  float orientation; /* degrees, [0,360) */
  float score;
  float ambiguity;
  int match;
  float match_xpos;
  float match_ypos;
  float match_error;
  float subsampling;
  float empty[3];
  float data[128];
  float coords3D[3];
} cusift_point;

/* The public parameter fields of SiftData (cuSIFT.h:44-51) plus Extract()'s `subsampling`
 * argument (cuSIFT.cu:61) and the capacity given to the SiftData constructor (cuSIFT.cu:13-15).
 * Use cusift_default_params() -- the reference leaves these uninitialised. */
typedef struct cusift_params {
  int num_octaves;     /* SiftData::numOctaves */
  double init_blur;    /* SiftData::initBlur: blur already present in the input image */
  float peak_thresh;   /* SiftData::peakThresh, absolute DoG units on the caller's 0..255 scale */
  float edge_thresh;   /* SiftData::edgeThresh, used raw: keep iff tr^2 < edge_thresh*det (cuSIFT_D.cu:486) */
  float lowest_scale;  /* SiftData::lowestScale: octave o is searched iff lowest_scale < 2*subsampling*2^o */
  float subsampling;   /* Extract(..., float subsampling = 1.0f) */
  int max_pts;         /* capacity per image (SiftData::maxPts) */
  int tex_frac_bits;   /* bilinear fraction bits of the texture-unit model: 8 = as the reference ran, 0 = fp32 */
  int fused_detect;    /* 1 (default): the drivers run LaplaceMulti+FindPointsMulti as one kernel that keeps the
                          DoG planes on chip, and orientation+descriptor of all octaves as one launch after the
                          last detection (identical results); 0: the reference's per-octave stage sequence */
  int root_sift;       /* 0 (default): SIFT descriptors; 1: the drivers emit RootSIFT -- ConvertSiftToRootSift
                          (cuSIFT.cu:383-395) applied in the descriptor kernel's epilogue, the fusion the reference
                          leaves as a TODO (cuSIFT.cu:122-134,376-379); same bits as extract + cusift_rootsift() */
  int concurrent_batches; /* scheduling hint, results do not depend on it: how many extractions the caller keeps in
                          flight on this device at once (one context + stream each).  1 (default): this call has the
                          GPU to itself -- the detection launches use short row chunks so that their tails stay
                          short; >= 2: other batches fill the tails, so tall chunks (less redundant blurring at chunk
                          borders) are faster.  cusift_amd.batch.PipelinedExtractor sets it to its stream count.
                          With 1 the driver searches all octaves with ONE launch whenever a keypoint list per octave
                          fits the arena (with >= 2: up to eight 1080p frames' worth of pixels per call), and a call of
                          three 1080p frames' worth of pixels or more also runs octave 0's detection on a second stream
                          of the context, beside the ScaleDown chain (cusift_ctx_forks).  Same SiftData either way,
                          coarsest octave first. */
} cusift_params;

typedef struct cusift_ctx cusift_ctx; /* opaque: device, stream, scratch arena, timers */

/* ---- process / device ------------------------------------------------------------------ */
const char *cusift_last_error(void);
const char *cusift_version(void);
int cusift_device_count(int *count);
/* InitCuda(devNum), cutils.h:71-92: clamps dev into [0, n-1] and selects it. */
int cusift_init(int device);
void cusift_default_params(cusift_params *p);

/* ---- context --------------------------------------------------------------------------- */
/* `hip_stream` may be NULL (the context creates and owns a non-blocking stream) or an existing
 * hipStream_t (borrowed; e.g. torch.cuda.current_stream().cuda_stream). The reference has one
 * implicit global context: default stream + file-scope device symbols (cuSIFT_D.cu:13-20). */
int cusift_ctx_create(cusift_ctx **out, int device, void *hip_stream);
/* Same, but `hip_stream` is always borrowed, and NULL means the device's default (null) stream -- what
 * torch.cuda.current_stream().cuda_stream is (0) unless the caller switched streams. */
int cusift_ctx_create_borrowed(cusift_ctx **out, int device, void *hip_stream);
int cusift_ctx_destroy(cusift_ctx *ctx);
int cusift_ctx_synchronize(cusift_ctx *ctx); /* blocking */
void *cusift_ctx_stream(cusift_ctx *ctx);
int cusift_ctx_device(cusift_ctx *ctx);
/* Stream ordering between two contexts of one device without HIP headers: everything enqueued on `ctx` after this
 * call waits for everything enqueued on `other` so far (event record + stream wait; no host wait).  Lets a caller run
 * the all-gatherv of batch i on a second context while batch i+1 is extracted on the first. */
int cusift_ctx_wait(cusift_ctx *ctx, cusift_ctx *other);
/* Pre-size the scratch arena for batches of n_images w x h images (otherwise grown on demand). */
int cusift_ctx_reserve(cusift_ctx *ctx, int n_images, int w, int h, const cusift_params *p);
/* Pre-size the arena for cusift_extract_bands over up to n_bands bands of one image with max_pts records (the strip
 * tiling's per-rank step; cusift_tiled_create calls it on every rank so that no extraction allocates). */
int cusift_ctx_reserve_bands(cusift_ctx *ctx, int n_bands, int max_pts);
/* Bytes of HBM currently held by the arena. */
size_t cusift_ctx_arena_bytes(cusift_ctx *ctx);
/* How many extractions of this context ran octave 0's detection on the context's second stream (see
 * CUSIFT_POLICY_SIDE_STREAM below). */
unsigned long cusift_ctx_forks(cusift_ctx *ctx);
/* Launch policy of a context (new; the reference has one fixed launch sequence, cuSIFT.cu:175-270).  Results never
 * depend on it -- only which kernels run on which stream in which order.  The defaults are deterministic functions of
 * the call's size and of cusift_params.concurrent_batches; nothing is decided by timing unless asked for (value 2 of
 * the first key).  The only environment variable the library's extraction code reads is CUSIFT_OCTAVE_OVERLAP (= the
 * first key's value, read when a context is created) so that an unchanged caller of the C++ shim can opt in. */
enum {
  /* Octave 0's detection on a second stream of the context beside the ScaleDown chain and the coarser octaves, for
   * callers that keep ONE batch in flight (concurrent_batches < 2) and calls of >= 6 Mpixel: 0 never (default),
   * 1 yes, 2 yes if a one-time probe finds that the device runs the two streams side by side (HIP maps streams onto
   * a few hardware queues; which one a new stream lands on depends on what else the process created), 3 every call
   * whatever its size (tests). */
  CUSIFT_POLICY_SIDE_STREAM = 0,
  /* Keypoints of every octave to lists of their own, joined by the description kernel, so that all octaves are
   * searched by one launch: -1 by size (default: lone callers always, pipelining callers up to 16 Mpixel per call),
   * 0 never, 1 whenever the lists fit. */
  CUSIFT_POLICY_OCTAVE_LISTS = 1,
  CUSIFT_POLICY_GENERIC_KERNELS = 2,   /* 1: the generic (any pitch / alignment) kernels even where the fast ones apply */
  CUSIFT_POLICY_LAUNCH_PER_OCTAVE = 3, /* 1: with lists, still one detection launch per coarser octave */
  CUSIFT_POLICY_MATCH_SPLITS = 4,      /* cusift_match: column splits of the second set (0: by size) */
  CUSIFT_POLICY_TILED_PER_OCTAVE = 5,  /* cusift_tiled_*: one detection + description launch per tiled octave */
  /* The pyramid as a by-product of the detection: the fused detection of octave o writes octave o + 1's image from the
   * row window it streams through anyway (ScaleDown's arithmetic, cuSIFT_D.cu:37-182, bit for bit), so the octaves are
   * searched finest first -- into lists of their own, which therefore must fit (CUSIFT_POLICY_OCTAVE_LISTS) -- and no
   * ScaleDown launch re-reads the images.  -1 by size (default), 0 never (the ScaleDown chain first, as the reference:
   * cuSIFT.cu:175-192), 1 octave 0 only (then the chain and the coarser octaves as with 0), 2 every octave. */
  CUSIFT_POLICY_PYRAMID_IN_DETECT = 6
};
int cusift_ctx_set_policy(cusift_ctx *ctx, int key, int value);
int cusift_ctx_get_policy(cusift_ctx *ctx, int key, int *value);
/* An event on a context's stream: record, then the GPU time between two of them (blocks until `stop` has happened).
 * What the reference's TimerGPU does with cudaEvents (cutils.h:94-114); include/cuSIFT.h builds TimerGPU on these. */
typedef struct cusift_event cusift_event;
int cusift_event_create(cusift_ctx *ctx, cusift_event **out);
int cusift_event_record(cusift_event *ev, cusift_ctx *ctx);
int cusift_event_elapsed_ms(cusift_event *start, cusift_event *stop, float *ms);
int cusift_event_destroy(cusift_event *ev);
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

/* ---- device memory helpers (so a host program needs no HIP headers) ----------------------- */
/* cuImage::Allocate / SiftData ctor: cudaMallocPitch / cudaMalloc (cuImage.cu:30, cuSIFT.cu:29) */
int cusift_malloc(void **d_ptr, size_t bytes);
int cusift_free(void *d_ptr);
int cusift_memset(cusift_ctx *ctx, void *d_ptr, int value, size_t bytes);
/* SiftData::Synchronize (cuSIFT.cu:52-59) and raw copies; blocking. */
int cusift_memcpy_h2d(cusift_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int cusift_memcpy_d2h(cusift_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
/* cudaMemcpyDeviceToDevice as AddSiftData uses it when it grows a SiftData (extras/debug.cpp:436-441); blocking. */
int cusift_memcpy_d2d(cusift_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
/* cuImage::HostToDevice / DeviceToHost (cuImage.cu:83-117): dense host rows (w floats) <-> pitched device rows. */
int cusift_image_h2d(cusift_ctx *ctx, float *d_dst, int dst_pitch, const float *h_src, int w, int h);
int cusift_image_d2h(cusift_ctx *ctx, float *h_dst, const float *d_src, int src_pitch, int w, int h);
/* Pinned host memory for overlap of uploads (new; the reference uses pageable malloc). */
int cusift_malloc_host(void **h_ptr, size_t bytes);
int cusift_free_host(void *h_ptr);

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
/* ScaleDown(res, src, variance), cuSIFT.cu:313-353 + ScaleDown_D cuSIFT_D.cu:37-182.  `variance` sets the
 * 5-tap Gaussian exp(-(j-2)^2/(2*variance)) (the pyramid uses 0.5, cuSIFT.cu:185).
 * dst is (w/2) x (h/2); writes are bounds-checked (the reference's are not). */
int cusift_scale_down(cusift_ctx *ctx, float *d_dst, int dst_pitch, size_t dst_stride, const float *d_src, int w,
                      int h, int src_pitch, size_t src_stride, int n_images, float variance);
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
/* SiftData::ConvertSiftToRootSift, cuSIFT.cu:383-395 + cuSIFT_D.cu:299-317. */
int cusift_rootsift(cusift_ctx *ctx, cusift_point *d_points, int num_pts);

/* The device build of the five transcendental functions the kernels use in place of CUDA's libm (expf, exp2f,
 * atan2f, sinf/cosf: cuSIFT_D.cu:209-210,233,330,349,507), array form: op 0 expf(a), 1 exp2f(a), 2 atan2f(a, b),
 * 3 sincosf(a) -> (out, out2).  They are written out in IEEE operations (cusift_amd/csrc/sift_math.h) so that a host
 * build of the same header gives the same bits; this entry point exists so that callers and tests can verify that
 * on their device.  op 4: the descriptor stage's angle coordinate 4/3.1415f * atan2f(a, b) + 4 as its kernel forms it
 * (a degree-4 fit, within 4e-6 of the exact form, with the reference's operations where the value decides -- index 8).
 * Asynchronous. */
int cusift_math_eval(cusift_ctx *ctx, int op, const float *d_a, const float *d_b, float *d_out, float *d_out2,
                     size_t n);

/* ---- band ("tile") forms: one large image strip-tiled over several GPUs (BASELINE configs[4]) -- */
/* A band is `h` rows of device memory whose local row 0 is row `row0` of a global image with `h_global` rows
 * (same width).  Row addressing is "clamp to the global image, then translate", so a band that carries enough
 * halo rows gives bit-identical results to the whole image for the rows it owns.  New functionality: the
 * reference has no tiling (its scratch arena is sized for the whole image, cuSIFT.cu:81-98).
 * cusift_scale_down_band: computes global rows [r_begin, r_end) of the half-size image into a destination band
 *   that starts at global row dst_row0, from a source band {src_row0, h_src_global}; needs source rows
 *   2r-1 .. 2r+3 (cuSIFT_D.cu:75,123-125) inside the source band.
 * cusift_detect_band: fused LaplaceMulti+FindPointsMulti with extremum centres restricted to global rows
 *   [cy_begin, cy_end); keypoint rows are written in global coordinates.  Needs >= 5 halo rows of true data on each
 *   side that is not the image border (4 blur + 1 extremum; CUSIFT_ERR_INVALID otherwise).
 * cusift_describe_band: ComputeOrientations + ExtractSiftDescriptors for keypoints in global coordinates
 *   (root_sift as cusift_params.root_sift).
 *   d_flags (may be NULL): one counter, incremented for every keypoint whose sampling footprint (orientation window,
 *   rotated descriptor grid, +-1 px taps, bilinear 2x2) reaches beyond the band on a side that is not the image
 *   border -- such a keypoint samples clamped rows instead of the neighbour's and would differ from the whole image;
 *   the caller must treat a non-zero count as an error (cusift_amd.tiling.StripExtractor.check does). */
int cusift_scale_down_band(cusift_ctx *ctx, float *d_dst, int dst_pitch, int dst_row0, int r_begin, int r_end,
                           const float *d_src, int w, int h_src, int src_pitch, int src_row0, int h_src_global,
                           float variance);
int cusift_detect_band(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, int row0, int h_global,
                       int cy_begin, int cy_end, float init_blur, float peak_thresh, float edge_thresh,
                       float subsampling, cusift_point *d_points, int max_pts, unsigned int *d_counter);
int cusift_describe_band(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, int row0, int h_global,
                         cusift_point *d_points, int max_pts, const unsigned int *d_first,
                         const unsigned int *d_counter, float subsampling, int tex_frac_bits, int root_sift,
                         unsigned int *d_flags);
/* One octave of a strip-tiled image as a rank holds it: `h` local rows (global rows row0 .. row0 + h - 1 of h_global),
 * extremum centres restricted to global rows [cy_begin, cy_end). */
typedef struct cusift_band {
  const float *d_img;
  int w, h, pitch;
  int row0, h_global, cy_begin, cy_end;
  float init_blur, subsampling;
} cusift_band;
/* cusift_detect_band + cusift_describe_band for SEVERAL consecutive octaves of one image (bands[0] the finest) with one
 * detection launch and one description launch.  *d_counter (device) counts what d_points holds already -- the coarser
 * octaves of a root rank, described: left alone -- and the bands' keypoints are appended behind it, coarsest band
 * first; the same records as the per-octave calls leave.  At most 8 bands.  d_flags as in cusift_describe_band. */
int cusift_extract_bands(cusift_ctx *ctx, const cusift_band *bands, int n_bands, float peak_thresh, float edge_thresh,
                         cusift_point *d_points, int max_pts, unsigned int *d_counter, int tex_frac_bits,
                         int root_sift, unsigned int *d_flags);

/* ---- matcher (first consumer of SiftData) ----------------------------- */
/* MatchSiftData(data1, data2, distance, ...), extras/matching.cu:232-362: for every point of d_sift1 the best
 * and second-best point of d_sift2 under `distance` (0 = MatchSiftDistanceDotProduct, 1 = MatchSiftDistanceL2,
 * extras/matching.h:10-13); writes score, ambiguity, match, match_xpos, match_ypos of d_sift1 (extras/matching.cu:
 * 140-150,219-229).  The score/ambiguity thresholds of the reference are a host-side filter over those fields
 * (:318-349) and stay on the caller's side (include/matching.h does it).  Asynchronous on the context's stream. */
int cusift_match(cusift_ctx *ctx, cusift_point *d_sift1, int num_pts1, const cusift_point *d_sift2, int num_pts2,
                 int distance);
/* cudaMemcpy2D device->host (extras/matching.cu:311-315 copies the 5 match fields of every record); blocking. */
int cusift_memcpy2d_d2h(cusift_ctx *ctx, void *h_dst, size_t dst_pitch, const void *d_src, size_t src_pitch,
                        size_t width_bytes, size_t rows);

/* ---- RANSAC homography from matched SiftData ------------------------------ */
/* The device part and the final selection of FindHomography(data, homography, numMatches, numLoops, minScore,
 * maxAmbiguity, thresh), extras/homography.cu:182-269: for num_loops hypotheses -- h_rand_pts[i*num_loops + l] is
 * the i-th (of 4) sample of hypothesis l, an index into d_sift; the reference draws them on the host with rand()
 * from the points that pass minScore/maxAmbiguity (:208-235), and so does include/homography.h -- solve the 8x8
 * system (ComputeHomographies :89-130), count the points with reprojection error < thresh (TestHomographies
 * :135-178, over coords2D -> match_xpos/ypos of ALL num_pts records) and return the first hypothesis with the
 * most inliers: h_homography[0..7], h_homography[8] = 1, *num_matches = its count.  h_all_homo ([8][num_loops])
 * and h_all_counts ([num_loops]) may be NULL.  Blocking. */
int cusift_find_homography(cusift_ctx *ctx, const cusift_point *d_sift, int num_pts, const int *h_rand_pts,
                           int num_loops, float thresh, float h_homography[9], int *num_matches, float *h_all_homo,
                           int *h_all_counts);

/* Packs a batch's SiftData for an exchange (all-gatherv over RCCL): the valid records of all images back to back in
 * image order into d_packed (room for `capacity` records; anything beyond is dropped) and the exclusive prefix sums
 * of the valid counts into d_offsets[0 .. n_images] (may be NULL).  n_images <= 256.  Asynchronous. */
int cusift_pack_points(cusift_ctx *ctx, const cusift_point *d_points, const unsigned int *d_counters, int n_images,
                       int max_pts, cusift_point *d_packed, size_t capacity, unsigned int *d_offsets);

/* Compact wire format for SiftData that has to cross PCIe or a network (new; optional -- the exact 588-byte records
 * stay the default everywhere).  Extraction writes 7 header fields + the 128-float descriptor of a record; the other 12
 * floats are left as they were (SURVEY: cuSIFT.cu:24,29).  A compact record carries the 7 fields EXACTLY and the
 * descriptor as 128 bytes with one quantisation step per record: data[i] ~= q[i] * desc_step, desc_step =
 * max(data) / 255, q[i] = min(255, floor(data[i] / desc_step + 0.5)) -- 160 B instead of 588, |error| <= desc_step / 2
 * per element (<= 1e-3 for a SIFT descriptor, whose elements are <= ~0.5: an L2 distance of a few 1e-3, NOT within the
 * 1e-4 parity bar, which is why this is a wire format and not the SiftData).  A descriptor without a finite positive
 * maximum (flat patch: NaN) travels as desc_step = that maximum (NaN or 0) and q = 0.
 * cusift_pack_points_compact: as cusift_pack_points, compacting on the way (d_packed holds `capacity` compact records).
 * cusift_expand_points_host: compact records -> SiftPoint records on the host (data[i] = q[i] * desc_step; a NaN step
 *   gives the NaN descriptor back; the 12 fields extraction never writes are zeroed). */
typedef struct cusift_compact_point {
  float coords2D[2];
  float scale;
  float sharpness;
  float edgeness;
  float orientation;
  float subsampling;
  float desc_step;
  unsigned char q[128];
} cusift_compact_point; /* 160 bytes */
int cusift_pack_points_compact(cusift_ctx *ctx, const cusift_point *d_points, const unsigned int *d_counters,
                               int n_images, int max_pts, cusift_compact_point *d_packed, size_t capacity,
                               unsigned int *d_offsets);
int cusift_expand_points_host(const cusift_compact_point *h_compact, size_t n, cusift_point *h_points);

/* Trimmed wire format (new; optional): the 135 floats of a record that extraction WRITES -- the seven header fields and
 * the descriptor -- and nothing else: 540 B instead of 588, bit-exact (the other 12 floats of a SiftPoint are whatever
 * the caller's buffer held, cuSIFT.cu:24,29, so nothing is lost).  8 % fewer bytes per record over PCIe / xGMI.
 * cusift_pack_points_trimmed: as cusift_pack_points.  cusift_expand_trimmed (device, asynchronous) /
 * cusift_expand_trimmed_host: trimmed -> SiftPoint records, the 12 unwritten floats zeroed. */
typedef struct cusift_trimmed_point {
  float coords2D[2];
  float scale;
  float sharpness;
  float edgeness;
  float orientation;
  float subsampling;
  float data[128];
} cusift_trimmed_point; /* 540 bytes */
int cusift_pack_points_trimmed(cusift_ctx *ctx, const cusift_point *d_points, const unsigned int *d_counters,
                               int n_images, int max_pts, cusift_trimmed_point *d_packed, size_t capacity,
                               unsigned int *d_offsets);
int cusift_expand_trimmed(cusift_ctx *ctx, const cusift_trimmed_point *d_trimmed, size_t n, cusift_point *d_points);
int cusift_expand_trimmed_host(const cusift_trimmed_point *h_trimmed, size_t n, cusift_point *h_points);

/* Canonical order of extracted records, on the HOST copy: octave blocks coarsest first (as emitted), inside an octave
 * by y, x, scale.  The append order inside an octave is that of an atomic counter -- racy in the reference as well
 * (atomicInc, cuSIFT_D.cu:512) -- so callers that need run-to-run identical arrays, not just identical sets, sort. */
int cusift_sort_points_host(cusift_point *h_points, int num_pts);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI (BASELINE configs[3], [4]) --------------------------------
 * New functionality: the reference is single-GPU, single-image (no collective call sites).
 * A communicator wraps one ncclComm_t bound to a context: every exchange is enqueued on that context's stream.
 * RCCL is loaded at run time from the directory of the process's HIP runtime (or $CUSIFT_RCCL_LIB, or the library
 * named by cusift_comm_use_library -- any library exporting the nine nccl* entry points used here; the tests bind an
 * in-process transport that way, next to the real RCCL); a program that never creates a communicator never needs it.
 *   rank 0:  cusift_comm_get_unique_id(id)  -> hand the 128 bytes to every rank (MPI_Bcast, a TCP store, a file ...)
 *   all:     cusift_comm_create(&comm, ctx, id, rank, world)      (collective: ncclCommInitRank)
 */
#define CUSIFT_UNIQUE_ID_BYTES 128
typedef struct cusift_comm cusift_comm;
/* The library cusift_comm_get_unique_id / cusift_comm_create bind from now on (NULL or "": the default search). */
int cusift_comm_use_library(const char *path);
int cusift_comm_get_unique_id(char id[CUSIFT_UNIQUE_ID_BYTES]);
int cusift_comm_create(cusift_comm **out, cusift_ctx *ctx, const char id[CUSIFT_UNIQUE_ID_BYTES], int rank, int world);
int cusift_comm_destroy(cusift_comm *comm);
int cusift_comm_rank(cusift_comm *comm, int *rank, int *world);
cusift_ctx *cusift_comm_ctx(cusift_comm *comm); /* the context (device + stream) the communicator is bound to */
/* What the bound LIBRARY reports for this communicator: ncclCommCount, ncclCommUserRank, ncclGetVersion (major * 10000 +
 * minor * 100 + patch) -- each -1 if the library does not export the call.  A benchmark line that carries lib_ranks
 * proves "RCCL saw N ranks" without trusting the caller's own bookkeeping (cusift_comm_rank). */
int cusift_comm_info(cusift_comm *comm, int *lib_ranks, int *lib_rank, int *lib_version);
/* Path of the library bound last ("" before the first communicator call). */
const char *cusift_comm_library(void);
/* Tests / world == 1: route the local shard (and rows addressed to this rank) through ncclSend/ncclRecv to self too,
 * so that one GPU exercises the grouped p2p path.  Also settable with CUSIFT_COMM_SELF_P2P=1. */
int cusift_comm_set_self_p2p(cusift_comm *comm, int on);
/* Pre-sizes what the all-gatherv needs so that no call of the loop allocates or synchronises: `tickets` exchanges may
 * be in flight at once (begin() without its finish(); default 4), each with n_images_max count slots per rank;
 * stage_records only matters with self_p2p (the staging copy of the local shard). */
int cusift_comm_reserve(cusift_comm *comm, int n_images_max, int tickets, size_t stage_records);
/* 0 (default): exact sizes -- finish() reads the gathered counts on the host and posts ncclSend/ncclRecv of exactly the
 * valid records.  1: whole regions travel (region_cap records per peer whatever the counts), posted by begin(): the
 * exchange needs no host read at all, at the price of the bytes; for small capacities (a tiled image's merge). */
int cusift_comm_set_fixed_size(cusift_comm *comm, int on);
/* Wire format of the gathered records.  0 (default): cusift_point (588 B, exact).  1: they travel -- and arrive -- as
 * cusift_compact_point (160 B: exact header fields, 8-bit descriptor; see cusift_pack_points_compact): 3.7x fewer bytes
 * over xGMI when the exchange, not the extraction, bounds a step.  2: as cusift_trimmed_point (540 B: the 135 floats
 * extraction writes, EXACT; cusift_expand_trimmed makes SiftPoint records of them).  d_gathered then holds
 * world * region_cap records of that format. */
int cusift_comm_set_wire_format(cusift_comm *comm, int format);
/* Diagnostic: how many finish() calls found their counts not yet there, i.e. the host was ahead of the GPU (in a
 * GPU-bound pipelined loop that is the normal case and costs nothing: the device has the caller's other steps queued). */
unsigned long long cusift_comm_host_waits(cusift_comm *comm);
/* ... and the wall time those finish() calls spent waiting, in milliseconds. */
double cusift_comm_host_wait_ms(cusift_comm *comm);
/* Diagnostic: synchronising HIP calls (hipStreamSynchronize) this communicator has made so far -- they only happen while
 * its buffers are (re)sized: after cusift_comm_reserve the number stays put through any number of begin / finish. */
unsigned long long cusift_comm_hip_syncs(cusift_comm *comm);

/* All-gatherv of SiftData: every rank ends up with the valid records of ALL ranks' images.  d_gathered is `world`
 * REGIONS of region_cap records (cusift_point, or cusift_compact_point after cusift_comm_set_wire_format(comm, 1));
 * region r holds rank r's records packed back to back in image order (h_totals[r] of them).  Fixed region starts are what lets a rank pack its shard into place before anybody's counts are known.
 *   begin  (asynchronous, no host wait): orders the exchange after everything enqueued on `producer` so far (the
 *          context that extracted d_points; NULL: the caller has ordered the streams), clamps the per-image counters on
 *          the device, packs the local shard straight into region `rank` of d_gathered -- after which d_points /
 *          d_counters are free again: cusift_ctx_wait(producer, cusift_comm_ctx(comm)) before overwriting them --,
 *          ncclAllGather of the counts (n_images_max slots per rank: the largest image count of any rank, the same value
 *          on every rank) and a kernel that publishes them to pinned host memory.
 *   finish (of the oldest begin): READS the counts on the host -- the sizes of ncclSend/ncclRecv are host arguments --
 *          which is no wait when the caller has enqueued a step or more of other work since begin() (up to `tickets`
 *          begins may be outstanding); writes h_counts[world][n_images_max] and h_totals[world] (either may be NULL) and
 *          posts ONE ncclGroup of ncclSend/ncclRecv: each shard travels directly to each peer over its xGMI link.
 *          CUSIFT_ERR_NOMEM (on every rank alike, nothing sent) if a rank's total exceeds region_cap.
 * Nothing here allocates once cusift_comm_reserve() has been called.  n_images <= 256.
 * cusift_allgatherv() = begin + finish.  cusift_compact_gathered: the regions back to back in rank order (world
 * asynchronous device copies on ctx's stream), for consumers that want one list. */
int cusift_allgatherv_begin(cusift_comm *comm, cusift_ctx *producer, const cusift_point *d_points,
                            const unsigned int *d_counters, int n_images, int max_pts, int n_images_max,
                            void *d_gathered, size_t region_cap);
int cusift_allgatherv_finish(cusift_comm *comm, unsigned int *h_counts, size_t *h_totals);
int cusift_allgatherv(cusift_comm *comm, cusift_ctx *producer, const cusift_point *d_points,
                      const unsigned int *d_counters, int n_images, int max_pts, int n_images_max,
                      void *d_gathered, size_t region_cap, unsigned int *h_counts, size_t *h_totals);
int cusift_compact_gathered(cusift_ctx *ctx, const cusift_point *d_gathered, size_t region_cap, int world,
                            const size_t *h_totals, cusift_point *d_out, size_t capacity);
/* Expand on arrival: regions gathered in the trimmed wire format (cusift_comm_set_wire_format(comm, 2): 540 B per record
 * over xGMI) -> the same regions as SiftPoint records (588 B, what the reference's SiftData holds; the 12 floats
 * extraction never writes -- uninitialised in the reference, cuSIFT.cu:24,29 -- are zero), region r at
 * d_points + r * region_cap, h_totals[r] records each (as cusift_allgatherv_finish returned them).  One launch for all
 * ranks on the communicator's stream, behind the exchange; asynchronous.  Every rank then ends the step holding the
 * SiftData of all ranks' images exactly as with the 588-byte wire format, for 8 % fewer bytes per link. */
int cusift_expand_gathered(cusift_comm *comm, const cusift_trimmed_point *d_gathered, size_t region_cap,
                           const size_t *h_totals, cusift_point *d_points);

/* Rows of a pitched float image between ranks, as one ncclGroup: op i sends send_rows[i] rows starting at local row
 * send_row[i] of d_band to peers[i] and receives recv_rows[i] rows from it into local row recv_row[i] (`pitch` floats
 * per row; the peer must post the mirror image; d_band holds `band_rows` rows and every op is checked against that --
 * RCCL reads and writes the rows on the device, where a range outside the allocation is a fault).  Asynchronous.
 * cusift_exchange_halos is the strip tiling's per-octave step (cusift_*_band entry points, BASELINE configs[4]): a band
 * is [top_halo rows of the neighbour above][own_rows][bottom_halo rows of the neighbour below]; the first / last
 * `send_rows` owned rows go to rank-1 / rank+1 and their counterparts arrive in the halo rows (every interior rank
 * uses the same send_rows == its neighbours' halo depth; rank 0 has top_halo = 0, the last rank bottom_halo = 0). */
int cusift_exchange_rows(cusift_comm *comm, float *d_band, int pitch, int band_rows, int n_ops, const int *peers,
                         const int *send_row, const int *send_rows, const int *recv_row, const int *recv_rows);
int cusift_exchange_halos(cusift_comm *comm, float *d_band, int pitch, int top_halo, int own_rows, int bottom_halo,
                          int send_rows);

/* ---- host to host: frames in host memory in, SiftData in pinned host memory out ----------------------------------
 * The reference's entry point takes a HOST image and leaves SiftData on the host, one image at a time, every step
 * blocking (SiftData::Extract + Synchronize, cuSIFT.cu:61-120,52-59).  At this build's rates a caller is bound by PCIe,
 * and gets what the link gives only if upload, extraction and read-back of consecutive batches overlap: this object
 * is that pipeline for a C / C++ caller (no HIP, no Python): `depth` batches in flight, batches alternating over two
 * extraction streams, an upload stream and a copy stream (four streams: one per hardware queue); the 8-bit form converts on the device
 * (cusift_u8_to_f32; main.cpp:300-318 converts on the host and uploads 4x the bytes).
 *   create   n_images = the largest batch; frames are dense rows of w pixels, unsigned char (CUSIFT_PIPE_U8) or float
 *            (CUSIFT_PIPE_F32, values 0..255 as the reference expects); depth 2..8; records_capacity = records one
 *            batch's SiftData may hold (0: n_images * max_pts -- generous: pinned memory per slot).
 *   submit   asynchronous: enqueues one batch ([n_images][h][w], pinned memory recommended -- pageable memory makes
 *            the upload synchronous); fails if `depth` batches are in flight already.  h_frames is READ by the upload,
 *            which runs some time after submit returns: the frames must stay untouched until THIS batch has been
 *            collected (there is no earlier signal; a caller that recycles frame buffers needs depth + 1 of them).
 *   collect  the OLDEST batch in flight: blocks until its records are on the host; *h_records = its `*total` valid
 *            records back to back in image order (SiftPoint layout), h_offsets[0 .. n] = their exclusive prefix sums
 *            per image (image i: records [h_offsets[i], h_offsets[i + 1])).  The pointers stay valid until the NEXT
 *            cusift_pipe_collect, through any number of submits in between (the pipeline owns depth + 1 pinned result
 *            buffers: the one handed out last is never the next to be written).
 *   errors   a batch with more records than records_capacity: CUSIFT_ERR_NOMEM from the submit / collect that finds
 *            out -- the reference saturates at maxPts per image instead (cuSIFT.cu:110), which max_pts still does; the
 *            batch capacity is this pipeline's own limit.  After ANY error from submit or collect other than
 *            CUSIFT_ERR_INVALID (bad arguments, nothing enqueued) the pipeline is FAILED: work of unknown state is in
 *            flight, every later submit / collect refuses, the batches in flight are lost; destroy it. */
typedef struct cusift_pipe cusift_pipe;
enum { CUSIFT_PIPE_U8 = 0, CUSIFT_PIPE_F32 = 1 };
int cusift_pipe_create(cusift_pipe **out, int device, int n_images, int w, int h, const cusift_params *prm,
                       int input_format, int depth, size_t records_capacity);
int cusift_pipe_submit(cusift_pipe *pipe, const void *h_frames, int n_images);
int cusift_pipe_collect(cusift_pipe *pipe, const cusift_point **h_records, const unsigned int **h_offsets, int *n_images,
                        size_t *total);
int cusift_pipe_in_flight(cusift_pipe *pipe);
int cusift_pipe_destroy(cusift_pipe *pipe);

/* ---- one large image strip-tiled over the ranks (BASELINE configs[4]) -------------------------------------------
 * The rank-side driver of the tiling: plan, bands, per-octave ScaleDown -> halo exchange -> band detection and
 * description, coarse-octave collapse onto rank 0, footprint check.  Mirrors the octave loop of cuSIFT.cu:175-202 (the
 * reference itself has no tiling); the union of the ranks' SiftData equals the whole-image extraction bit for bit.
 * Rank k owns base rows [k*H/world, (k+1)*H/world).  Everything runs on ctx's stream; `comm` (NULL for world == 1 or
 * for extractors driven with cusift_tiled_exchange_virtual) must be bound to the same context or stream.
 *   cusift_tiled_create(&t, ctx, comm, rank, world, W, H, &params, 0)     allocates the bands (halo_rows 0 = 48)
 *   cusift_tiled_extract(t, d_strip, strip_pitch, d_points, d_counter)    collective, asynchronous: my owned base rows
 *                                                                         in, my SiftData (params.max_pts records) out
 *   cusift_tiled_check(t, &flagged)                                       blocking; an error if a keypoint's sampling
 *                                                                         footprint left the halo
 *   cusift_allgatherv(comm, ...)                                          merged SiftData on every rank
 * cusift_tiled_extract is cusift_tiled_load, then for each octave o cusift_tiled_build_octave(o) (o > 0) and
 * cusift_tiled_exchange(o), then cusift_tiled_process; the steps are public so that P extractors of one process can be
 * stepped together with cusift_tiled_exchange_virtual (device copies instead of RCCL: a plan run on one GPU).
 * cusift_tiled_plan is the row geometry alone (no GPU): octave sizes, the first collapsed octave (== n_octaves: none),
 * the rows [own_begin, own_end) rank `rank` owns in `octave` and the rows [band_begin, band_end) its band holds. */
#define CUSIFT_TILED_DEFAULT_HALO 48
typedef struct cusift_tiled cusift_tiled;
int cusift_tiled_plan(int W, int H, int world, int num_octaves, int halo_rows, int rank, int octave, int *n_octaves,
                      int *collapse_octave, int *w, int *h, int *pitch, int *own_begin, int *own_end, int *band_begin,
                      int *band_end);
int cusift_tiled_create(cusift_tiled **out, cusift_ctx *ctx, cusift_comm *comm, int rank, int world, int W, int H,
                        const cusift_params *p, int halo_rows);
int cusift_tiled_destroy(cusift_tiled *t);
int cusift_tiled_info(cusift_tiled *t, int *n_octaves, int *collapse_octave, int *root, int *halo_rows);
int cusift_tiled_band(cusift_tiled *t, int octave, float **d_band, int *w, int *h_global, int *pitch, int *own_begin,
                      int *own_end, int *band_begin, int *band_end);
int cusift_tiled_load(cusift_tiled *t, const float *d_strip, int strip_pitch);
int cusift_tiled_build_octave(cusift_tiled *t, int octave);
int cusift_tiled_exchange(cusift_tiled *t, int octave);
int cusift_tiled_exchange_virtual(cusift_tiled **ranks, int n, int octave);
int cusift_tiled_process(cusift_tiled *t, cusift_point *d_points, unsigned int *d_counter);
int cusift_tiled_extract(cusift_tiled *t, const float *d_strip, int strip_pitch, cusift_point *d_points,
                         unsigned int *d_counter);
int cusift_tiled_check(cusift_tiled *t, unsigned int *flagged);

/* ---- drivers ------------------------------------------------------------------------------ */
/* Batch form of ExtractSiftLoop/ExtractSiftOctave (cuSIFT.cu:175-270) on device-resident images.
 * Asynchronous on the context's stream: no host read-back, no allocation once the arena is sized.
 * d_points: n_images*max_pts records; d_counters: n_images counters (zeroed by this call); on
 * completion image i holds min(d_counters[i], max_pts) points, octave blocks coarsest first. */
int cusift_extract_batch(cusift_ctx *ctx, const float *d_imgs, int n_images, int w, int h, int pitch,
                         size_t image_stride, const cusift_params *p, cusift_point *d_points,
                         unsigned int *d_counters);
/* Replayable form of cusift_extract_batch for a caller that extracts again and again from the SAME buffers and
 * geometry (a video pipeline: new frame copied into d_imgs, results read from d_points): the launch sequence is
 * recorded once into a hipGraph and replayed with one call, which removes the per-launch host cost that bounds
 * single-image latency (a 5-octave 1080p extraction is 11 short dependent kernels).  New: the reference has no
 * counterpart (it re-creates textures, symbols and buffers on every call, cuSIFT.cu:61-120).
 * The context must own or borrow a real stream (not the null stream).  The recording refers to the context's
 * scratch arena: a later call that grows the arena invalidates it (cusift_graph_launch then fails with
 * CUSIFT_ERR_INVALID).  cusift_graph_launch is asynchronous on the context's stream. */
typedef struct cusift_graph cusift_graph;
int cusift_graph_create(cusift_ctx *ctx, cusift_graph **out, const float *d_imgs, int n_images, int w, int h,
                        int pitch, size_t image_stride, const cusift_params *p, cusift_point *d_points,
                        unsigned int *d_counters);
int cusift_graph_launch(cusift_graph *g);
int cusift_graph_nodes(cusift_graph *g); /* kernel/memset/copy nodes in the recording */
int cusift_graph_destroy(cusift_graph *g);
/* The legacy ExtractSift(siftData, cuImage&, numOctaves, initBlur, thresh, lowestScale, subsampling)
 * (main.cpp:99-103,324-328; cuSIFT.cu:123-134): image already on the device.  Blocking; writes
 * *num_pts = min(count, max_pts) (cuSIFT.cu:107-110) and, if h_points != NULL, copies that many
 * records to the host (SiftData::Synchronize, cuSIFT.cu:52-59). */
int cusift_extract(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, const cusift_params *p,
                   cusift_point *d_points, cusift_point *h_points, int *num_pts);
/* SiftData::Extract(float *im, w, h, subsampling), cuSIFT.cu:61-120: dense host image in, uploads
 * (cuImage ctor, cuImage.cu:53-60), extracts, synchronises.  Blocking. */
int cusift_extract_host(cusift_ctx *ctx, const float *h_img, int w, int h, const cusift_params *p,
                        cusift_point *d_points, cusift_point *h_points, int *num_pts);

#ifdef __cplusplus
}
#endif
#endif /* CUSIFT_AMD_H */
