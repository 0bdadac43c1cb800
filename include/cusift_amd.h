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
 * Map of the C ABI (111 entry points in four headers; cusift_amd_all.h includes them all):
 *   cusift_amd.h (THIS FILE, 46): the drop-in boundary -- everything include/cuSIFT.h is built on (process / device,
 *     context, device memory helpers, cusift_extract / _extract_host / _scale_down / _rootsift / _sort_points_host,
 *     cusift_event_* for TimerGPU), the batch driver a throughput caller needs (cusift_extract_batch, cusift_graph_*,
 *     cusift_ctx_wait / _reserve / _set_policy) and the host-to-host pipeline (cusift_pipe_*).
 *   cusift_amd_stages.h: one entry point per kernel of the reference (LaplaceMulti, FindPointsMulti, the fused
 *     detection, ComputeOrientations, ExtractSiftDescriptors ...), the caller-side front-end (8-bit frames, 3 x 3
 *     pre-blur), the stage timers and other diagnostics -- for callers that drive the stages themselves and for the
 *     parity tests.
 *   cusift_amd_multigpu.h: SiftData on the wire (pack / trimmed / compact records), the communicator over RCCL, the
 *     all-gatherv of SiftData, halo exchange, the strip tiling of one large image over the ranks and its band kernels.
 *   cusift_amd_extras.h: the next rows of SURVEY 8f -- cusift_match, cusift_find_homography.
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
  int tex_frac_bits;   /* fixed-point bits of the texture-unit model (fractions AND the four bilinear weights):
                          8 = as the reference ran, 1..11 accepted, anything else = exact fp32 fractions and weights */
  int fused_detect;    /* 1 (default): the drivers run LaplaceMulti+FindPointsMulti as one kernel that keeps the
                          DoG planes on chip, and orientation+descriptor of all octaves as one launch after the
                          last detection (identical results); 0: the reference's per-octave stage sequence */
  int root_sift;       /* 0 (default): SIFT descriptors; 1: the drivers emit RootSIFT -- ConvertSiftToRootSift
                          (cuSIFT.cu:383-395) applied in the descriptor kernel's epilogue, the fusion the reference
                          leaves as a TODO (cuSIFT.cu:122-134,376-379); same bits as extract + cusift_rootsift() */
  int concurrent_batches; /* scheduling hint, results do not depend on it: how many extractions the caller keeps in
                          flight on this device at once (one context + stream each).  1 (default): this call has the
                          GPU to itself -- short row chunks in the detection launches (their tails stay short), all
                          octaves searched by ONE launch whenever a keypoint list per octave fits the arena, and from
                          64 million pixels per call (31 frames of 1080p) octave 1 as a by-product of octave 0's detection (CUSIFT_POLICY_PYRAMID_IN_DETECT).
                          >= 2: other batches fill the tails -- taller chunks (less redundant blurring at chunk borders),
                          the pyramid in the detections from 2 million pixels per call (one 1080p frame); >= 3: chunk height proportional to the
                          work (about as many chunks per launch as the chip holds waves).
                          cusift_amd.batch.PipelinedExtractor sets it to its stream count.  Same SiftData either way,
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
/* Bytes of HBM currently held by the arena. */
size_t cusift_ctx_arena_bytes(cusift_ctx *ctx);
/* Launch policy of a context (new; the reference has one fixed launch sequence, cuSIFT.cu:175-270).  Results never
 * depend on it -- only which kernels run on which stream in which order.  The defaults are deterministic functions of
 * the call's size and of cusift_params.concurrent_batches; nothing is decided by timing unless asked for (value 2 of
 * the first key).  The library's extraction code reads NO environment variable (sift_comm.hip reads CUSIFT_RCCL_LIB to
 * find the collective library): an unchanged caller of the C++ shim opts in to the first key through
 * CUSIFT_OCTAVE_OVERLAP, which the shim itself reads (include/cuSIFT.h) and hands to cusift_ctx_set_policy. */
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
   * ScaleDown launch re-reads the images.  -1 by size (default: a caller with cusift_params.concurrent_batches >= 2 every
   * octave from 2 Mpixel per call; a lone caller octave 0 only from 64 Mpixel -- its one-launch detection of the coarser
   * octaves is worth more than their short ScaleDown launches, and below 64 Mpixel more than octave 0's too), 0 never (the ScaleDown chain first, as the reference:
   * cuSIFT.cu:175-192), 1 octave 0 only (then the chain and the coarser octaves as with 0), 2 every octave. */
  CUSIFT_POLICY_PYRAMID_IN_DETECT = 6
};
/* (Changing a policy may change the launch plan and with it the size of the scratch arena: the next extraction then
 * re-allocates it once -- a stream synchronisation -- unless cusift_ctx_reserve is called again first.  A recorded
 * cusift_graph refuses to replay after such a re-allocation.) */
int cusift_ctx_set_policy(cusift_ctx *ctx, int key, int value);
int cusift_ctx_get_policy(cusift_ctx *ctx, int key, int *value);
/* An event on a context's stream: record, then the GPU time between two of them (blocks until `stop` has happened).
 * What the reference's TimerGPU does with cudaEvents (cutils.h:94-114); include/cuSIFT.h builds TimerGPU on these. */
typedef struct cusift_event cusift_event;
int cusift_event_create(cusift_ctx *ctx, cusift_event **out);
int cusift_event_record(cusift_event *ev, cusift_ctx *ctx);
/* Everything enqueued on `ctx` after this call waits for the work that preceded the event's last record (no host wait):
 * the fine-grained form of cusift_ctx_wait -- e.g. "the next extraction into these records waits for the pack that read
 * them", not for everything else the other context has queued since (tests/cpp/scaling_bench.cpp). */
int cusift_event_wait(cusift_event *ev, cusift_ctx *ctx);
int cusift_event_elapsed_ms(cusift_event *start, cusift_event *stop, float *ms);
int cusift_event_destroy(cusift_event *ev);

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

/* ---- the two stage entry points of the reference's public surface (the rest: cusift_amd_stages.h) ---------- */
/* ScaleDown(res, src, variance), cuSIFT.cu:313-353 + ScaleDown_D cuSIFT_D.cu:37-182.  `variance` sets the
 * 5-tap Gaussian exp(-(j-2)^2/(2*variance)) (the pyramid uses 0.5, cuSIFT.cu:185).
 * dst is (w/2) x (h/2); writes are bounds-checked (the reference's are not). */
int cusift_scale_down(cusift_ctx *ctx, float *d_dst, int dst_pitch, size_t dst_stride, const float *d_src, int w,
                      int h, int src_pitch, size_t src_stride, int n_images, float variance);
/* SiftData::ConvertSiftToRootSift, cuSIFT.cu:383-395 + cuSIFT_D.cu:299-317. */
int cusift_rootsift(cusift_ctx *ctx, cusift_point *d_points, int num_pts);

/* Canonical order of extracted records, on the HOST copy: octave blocks coarsest first (as emitted), inside an octave
 * by y, x, scale.  The append order inside an octave is that of an atomic counter -- racy in the reference as well
 * (atomicInc, cuSIFT_D.cu:512) -- so callers that need run-to-run identical arrays, not just identical sets, sort. */
int cusift_sort_points_host(cusift_point *h_points, int num_pts);

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
 * records to the host (SiftData::Synchronize, cuSIFT.cu:52-59).  h_points must hold max_pts records (the
 * reference's SiftData does): rows [num_pts, max_pts) are unspecified afterwards -- the copy starts before the
 * count is known, sized by the context's previous call.  Pinned h_points (cusift_malloc_host) make it one DMA. */
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
