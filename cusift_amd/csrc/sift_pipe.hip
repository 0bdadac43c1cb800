// sift_pipe.hip -- the HOST-TO-HOST pipeline behind the C ABI (cusift_pipe_*): batches of frames in host memory in,
// their SiftData in pinned host memory out, with the upload of batch i+1, the extraction of batch i and the read-back of
// batch i-1 running at the same time.
//
// Why: the reference's entry point takes a HOST image and leaves SiftData on the host (SiftData::Extract uploads, runs
// and -- via Synchronize -- copies back, cuSIFT.cu:61-120,52-59), one image at a time, everything blocking.  At this
// build's rates the GPU needs ~17 us for a 1080p frame while its 8-bit pixels need ~36 us over PCIe and its float pixels
// ~145 us: a drop-in caller is bound by the link, and gets what the link can give only if the three phases overlap.
// bench.py's `host_in` leg shows that overlap with torch as plumbing; this file is the same pipeline for a C or C++
// caller -- no Python, no torch, no HIP in the caller.  (the caller-side front-end; the
// 8-bit conversion is cusift_u8_to_f32, main.cpp:300-318.)
//
// Host code only: the kernels are the product's (u8_to_f32, the batch driver, pack_points).  One object owns
//   * `depth` SLOTS -- per slot the upload staging, the float batch, the records + counters, the packed records, and
//     their pinned host twins (records, offsets);
//   * two extraction contexts, each with a stream of its own (consecutive batches alternate over them: the launch tails
//     of one batch are filled by the next);
//   * an upload stream and a copy stream -- four streams in all, one per hardware queue (cusift_pipe_create).
// A batch moves through: H2D (upload stream) -> [8-bit -> float] + extraction + pack + offsets D2H (its context's
// stream) -> records D2H (copy stream; exactly sized, so it is enqueued once the batch's counts have
// arrived on the host -- by the submit / collect calls that follow, never by a blocked thread).
#include <hip/hip_runtime.h>

#include <string.h>

#include <algorithm>
#include <memory>
#include <vector>

#include "sift_internal.h"
#include "sift_types.h"

namespace {

#define HIP_TRY(expr)                                                                                                \
  do {                                                                                                               \
    hipError_t e_ = (expr);                                                                                          \
    if (e_ != hipSuccess)                                                                                            \
      return cusift_fail(CUSIFT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

#define TRY(expr)                     \
  do {                                \
    int rc_ = (expr);                 \
    if (rc_ != CUSIFT_OK) return rc_; \
  } while (0)

inline int ialign_up(int a, int b) { return (a + b - 1) / b * b; }

struct Slot {
  unsigned char *d_u8 = nullptr;   // [n][h][w] bytes (8-bit input only)
  float *d_img = nullptr;          // [n][h][pitch] floats
  cusift_point *d_points = nullptr;  // [n][max_pts]
  unsigned int *d_counters = nullptr;  // [n]
  cusift_point *d_packed = nullptr;  // [capacity]
  unsigned int *d_offsets = nullptr;  // [n + 1]
  cusift_point *h_records = nullptr;  // pinned, [capacity]: the host buffer this batch lands in (one of depth + 1, below)
  unsigned int *h_offsets = nullptr;  // pinned, [n + 1]
  hipEvent_t ev_up = nullptr, ev_counts = nullptr, ev_copied = nullptr;
  int n = 0;            // images of the batch in flight
  bool busy = false;    // submitted, not yet collected
  bool copying = false;  // its records' D2H has been enqueued
  size_t total = 0;     // valid records (known once the counts have arrived)
};

}  // namespace

struct cusift_pipe {
  int device = 0, n_max = 0, w = 0, h = 0, pitch = 0, format = 0, depth = 0;
  size_t capacity = 0;
  cusift_params prm;
  std::vector<cusift_ctx *> ex;  // extraction contexts (own streams)
  hipStream_t up = nullptr, copy = nullptr;
  std::vector<Slot> slots;
  // The pinned host buffers are one MORE than the slots and are dealt out round robin at submit: with at most `depth`
  // batches in flight, the buffer of the batch collected last is never the next to be written -- a collected view stays
  // valid through any number of submits, until the next collect (round 4 documented "until depth further submits", which
  // the slot-indexed buffers did not hold: at full depth the very next submit re-used the collected slot).
  std::vector<cusift_point *> h_records;   // [depth + 1] pinned, [capacity] each
  std::vector<unsigned int *> h_offsets;   // [depth + 1] pinned, [n + 1] each
  unsigned long submitted = 0, collected = 0;
  bool failed = false;
};

namespace {

void free_pipe(cusift_pipe *p) {
  (void)hipSetDevice(p->device);
  (void)hipDeviceSynchronize();
  for (Slot &s : p->slots) {
    if (s.d_u8) (void)hipFree(s.d_u8);
    if (s.d_img) (void)hipFree(s.d_img);
    if (s.d_points) (void)hipFree(s.d_points);
    if (s.d_counters) (void)hipFree(s.d_counters);
    if (s.d_packed) (void)hipFree(s.d_packed);
    if (s.d_offsets) (void)hipFree(s.d_offsets);
    for (hipEvent_t e : {s.ev_up, s.ev_counts, s.ev_copied})
      if (e) (void)hipEventDestroy(e);
  }
  for (cusift_point *h : p->h_records)
    if (h) (void)hipHostFree(h);
  for (unsigned int *h : p->h_offsets)
    if (h) (void)hipHostFree(h);
  for (cusift_ctx *c : p->ex) (void)cusift_ctx_destroy(c);
  if (p->up) (void)hipStreamDestroy(p->up);
  if (p->copy) (void)hipStreamDestroy(p->copy);
  delete p;
}

// Enqueue the exactly-sized read-back of every batch whose counts have arrived, oldest first (the copy stream keeps
// them in order).  `must`: the slot whose counts are waited for if they are not there yet (collect), or -1.
int progress(cusift_pipe *p, long must) {
  for (unsigned long k = p->collected; k < p->submitted; ++k) {
    Slot &s = p->slots[k % p->depth];
    if (s.copying) continue;
    hipError_t q = hipEventQuery(s.ev_counts);
    if (q == hipErrorNotReady) {
      if ((long)k != must) break;  // later batches' counts come later still
      HIP_TRY(hipEventSynchronize(s.ev_counts));
    } else if (q != hipSuccess) {
      return cusift_fail(CUSIFT_ERR_HIP, "pipe: hipEventQuery failed: %s", hipGetErrorString(q));
    }
    s.total = s.h_offsets[s.n];
    if (s.total > p->capacity)
      return cusift_fail(CUSIFT_ERR_NOMEM, "pipe: a batch holds %zu records, the slots were sized for %zu", s.total,
                         p->capacity);
    if (s.total)
      HIP_TRY(hipMemcpyAsync(s.h_records, s.d_packed, s.total * sizeof(cusift_point), hipMemcpyDeviceToHost, p->copy));
    HIP_TRY(hipEventRecord(s.ev_copied, p->copy));
    s.copying = true;
  }
  return CUSIFT_OK;
}

}  // namespace

extern "C" int cusift_pipe_create(cusift_pipe **out, int device, int n_images, int w, int h, const cusift_params *prm,
                                  int input_format, int depth, size_t records_capacity) {
  if (!out) return cusift_fail(CUSIFT_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (!prm) return cusift_fail(CUSIFT_ERR_INVALID, "params is NULL");
  if (n_images < 1 || n_images > cusift::kMaxFlatImages || w < 1 || h < 1 || prm->max_pts < 1)
    return cusift_fail(CUSIFT_ERR_INVALID, "pipe: 1..%d images per batch, w, h, max_pts >= 1", cusift::kMaxFlatImages);
  if (input_format != CUSIFT_PIPE_U8 && input_format != CUSIFT_PIPE_F32)
    return cusift_fail(CUSIFT_ERR_INVALID, "pipe: input_format is CUSIFT_PIPE_U8 or CUSIFT_PIPE_F32");
  if (depth < 2 || depth > 8) return cusift_fail(CUSIFT_ERR_INVALID, "pipe: depth 2..8 batches in flight");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) return cusift_fail(CUSIFT_ERR_NO_DEVICE, "no HIP device available");
  if (device < 0 || device >= n_dev) return cusift_fail(CUSIFT_ERR_INVALID, "device %d out of range [0,%d)", device, n_dev);
  HIP_TRY(hipSetDevice(device));
  cusift_pipe *p = new cusift_pipe();
  p->device = device;
  p->n_max = n_images;
  p->w = w;
  p->h = h;
  p->pitch = ialign_up(w, 128);  // cuImage's rule (cuImage.cu:16-21), what the fast kernels want
  p->format = input_format;
  p->depth = depth;
  p->prm = *prm;
  p->capacity = records_capacity ? records_capacity : (size_t)n_images * prm->max_pts;
  // FOUR streams in all -- two extraction streams (consecutive batches alternate; a batch's records are packed on its own
  // extraction stream), the upload stream and the copy stream: HIP maps streams onto four hardware queues, and a fifth
  // or sixth busy stream shares a queue with one of these -- an upload then waits behind another batch's kernels.
  // Measured, 64 x 1080p 8-bit frames, a fresh process each (tools/probe_pipe_depth.py): three extraction streams + a pack
  // stream 2.64 ms per batch at depth 4 (3.02 at depth 3), two + a pack stream 2.55, two with the pack on the extraction
  // stream 2.51 (2.49 at depth 3) -- 0.93 of the upload's own time; the link-bound pipeline has no use for a third batch
  // in extraction (1.1 ms of GPU time per 2.3 ms of upload).
  const int n_ex = std::min(depth, 2);
  p->prm.concurrent_batches = n_ex;
  int rc = CUSIFT_OK;
  auto hip = [&](hipError_t e, const char *what) {
    if (e != hipSuccess && rc == CUSIFT_OK) rc = cusift_fail(CUSIFT_ERR_HIP, "pipe: %s failed: %s", what, hipGetErrorString(e));
    return e == hipSuccess;
  };
  for (int i = 0; i < n_ex && rc == CUSIFT_OK; ++i) {
    cusift_ctx *c = nullptr;
    rc = cusift_ctx_create(&c, device, nullptr);
    if (rc == CUSIFT_OK) {
      p->ex.push_back(c);
      rc = cusift_ctx_reserve(c, n_images, w, h, &p->prm);
    }
  }
  if (rc == CUSIFT_OK) hip(hipStreamCreateWithFlags(&p->up, hipStreamNonBlocking), "hipStreamCreate");
  if (rc == CUSIFT_OK) hip(hipStreamCreateWithFlags(&p->copy, hipStreamNonBlocking), "hipStreamCreate");
  p->slots.resize(depth);
  const size_t px = (size_t)n_images * h * w, fpx = (size_t)n_images * h * p->pitch;
  for (int j = 0; j < depth && rc == CUSIFT_OK; ++j) {
    Slot &s = p->slots[j];
    if (input_format == CUSIFT_PIPE_U8) hip(hipMalloc((void **)&s.d_u8, px), "hipMalloc");
    hip(hipMalloc((void **)&s.d_img, fpx * sizeof(float)), "hipMalloc");
    if (rc == CUSIFT_OK && p->pitch != w) hip(hipMemsetAsync(s.d_img, 0, fpx * sizeof(float), p->up), "hipMemset");
    hip(hipMalloc((void **)&s.d_points, (size_t)n_images * prm->max_pts * sizeof(cusift_point)), "hipMalloc");
    hip(hipMalloc((void **)&s.d_counters, sizeof(unsigned int) * n_images), "hipMalloc");
    hip(hipMalloc((void **)&s.d_packed, std::max<size_t>(1, p->capacity) * sizeof(cusift_point)), "hipMalloc");
    hip(hipMalloc((void **)&s.d_offsets, sizeof(unsigned int) * (n_images + 1)), "hipMalloc");
    for (hipEvent_t *e : {&s.ev_up, &s.ev_counts, &s.ev_copied})
      hip(hipEventCreateWithFlags(e, hipEventDisableTiming), "hipEventCreate");
  }
  p->h_records.assign(depth + 1, nullptr);
  p->h_offsets.assign(depth + 1, nullptr);
  for (int j = 0; j <= depth && rc == CUSIFT_OK; ++j) {
    hip(hipHostMalloc((void **)&p->h_records[j], std::max<size_t>(1, p->capacity) * sizeof(cusift_point), hipHostMallocDefault),
        "hipHostMalloc");
    hip(hipHostMalloc((void **)&p->h_offsets[j], sizeof(unsigned int) * (n_images + 1), hipHostMallocDefault), "hipHostMalloc");
  }
  if (rc == CUSIFT_OK) hip(hipStreamSynchronize(p->up), "hipStreamSynchronize");
  if (rc != CUSIFT_OK) {
    free_pipe(p);
    return rc;
  }
  *out = p;
  return CUSIFT_OK;
}

extern "C" int cusift_pipe_destroy(cusift_pipe *p) {
  if (!p) return CUSIFT_OK;
  free_pipe(p);
  return CUSIFT_OK;
}

extern "C" int cusift_pipe_in_flight(cusift_pipe *p) { return p ? (int)(p->submitted - p->collected) : 0; }

extern "C" int cusift_pipe_submit(cusift_pipe *p, const void *h_frames, int n_images) {
  if (!p || !h_frames) return cusift_fail(CUSIFT_ERR_INVALID, "pipe / frames is NULL");
  if (p->failed) return cusift_fail(CUSIFT_ERR_HIP, "pipe: an earlier call failed; destroy the pipeline");
  if (n_images < 1 || n_images > p->n_max)
    return cusift_fail(CUSIFT_ERR_INVALID, "pipe: 1..%d images per batch, got %d", p->n_max, n_images);
  if ((int)(p->submitted - p->collected) >= p->depth)
    return cusift_fail(CUSIFT_ERR_INVALID, "pipe: %d batches in flight already: collect the oldest first", p->depth);
  HIP_TRY(hipSetDevice(p->device));
  Slot &s = p->slots[p->submitted % p->depth];
  cusift_ctx *ctx = p->ex[p->submitted % p->ex.size()];
  hipStream_t st = (hipStream_t)cusift_ctx_stream(ctx);
  // From the first enqueue on, ANY failure leaves work for this slot in flight that nothing accounts for (`submitted` has
  // not moved): the pipeline is then marked failed -- every later call refuses -- instead of re-using the slot.
#define PIPE_HIP(expr)                                                                                                   \
  do {                                                                                                                   \
    hipError_t e_ = (expr);                                                                                              \
    if (e_ != hipSuccess) {                                                                                              \
      p->failed = true;                                                                                                  \
      return cusift_fail(CUSIFT_ERR_HIP, "pipe: %s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    }                                                                                                                    \
  } while (0)
  // (the slot's previous batch was collected: its upload, extraction, pack and copy have all completed)
  const size_t row = (size_t)p->w * (p->format == CUSIFT_PIPE_U8 ? 1 : sizeof(float));
  if (p->format == CUSIFT_PIPE_U8 || p->pitch == p->w) {
    void *dst = p->format == CUSIFT_PIPE_U8 ? (void *)s.d_u8 : (void *)s.d_img;
    PIPE_HIP(hipMemcpyAsync(dst, h_frames, row * p->h * n_images, hipMemcpyHostToDevice, p->up));
  } else {  // dense host rows -> pitched device rows (cuImage::HostToDevice, cuImage.cu:83-92)
    PIPE_HIP(hipMemcpy2DAsync(s.d_img, (size_t)p->pitch * sizeof(float), h_frames, row, row, (size_t)p->h * n_images,
                              hipMemcpyHostToDevice, p->up));
  }
  PIPE_HIP(hipEventRecord(s.ev_up, p->up));
  PIPE_HIP(hipStreamWaitEvent(st, s.ev_up, 0));
  int rc = CUSIFT_OK;
  if (p->format == CUSIFT_PIPE_U8)
    rc = cusift_u8_to_f32(ctx, s.d_img, p->pitch, (size_t)p->h * p->pitch, s.d_u8, p->w, p->h, p->w, (size_t)p->h * p->w,
                          n_images);
  if (rc == CUSIFT_OK)
    rc = cusift_extract_batch(ctx, s.d_img, n_images, p->w, p->h, p->pitch, (size_t)p->h * p->pitch, &p->prm, s.d_points,
                              s.d_counters);
  if (rc != CUSIFT_OK) {
    p->failed = true;
    return rc;
  }
  // the records are packed, and the offsets read back, on the batch's own extraction stream (behind the extraction)
  hipStream_t ps = st;
  rc = cusift_pack_points(ctx, s.d_points, s.d_counters, n_images, p->prm.max_pts, s.d_packed, p->capacity, s.d_offsets);
  if (rc != CUSIFT_OK) {
    p->failed = true;
    return rc;
  }
  // this batch's host buffers: the next of depth + 1 (never the one handed out by the last collect)
  s.h_records = p->h_records[p->submitted % (p->depth + 1)];
  s.h_offsets = p->h_offsets[p->submitted % (p->depth + 1)];
  PIPE_HIP(hipMemcpyAsync(s.h_offsets, s.d_offsets, sizeof(unsigned int) * (n_images + 1), hipMemcpyDeviceToHost, ps));
  PIPE_HIP(hipEventRecord(s.ev_counts, ps));
#undef PIPE_HIP
  s.n = n_images;
  s.busy = true;
  s.copying = false;
  s.total = 0;
  p->submitted++;
  rc = progress(p, -1);  // read-backs of older batches whose counts are in
  if (rc != CUSIFT_OK) p->failed = true;
  return rc;
}

extern "C" int cusift_pipe_collect(cusift_pipe *p, const cusift_point **h_records, const unsigned int **h_offsets,
                                   int *n_images, size_t *total) {
  if (!p) return cusift_fail(CUSIFT_ERR_INVALID, "pipe is NULL");
  if (p->failed) return cusift_fail(CUSIFT_ERR_HIP, "pipe: an earlier call failed; destroy the pipeline");
  if (p->collected == p->submitted) return cusift_fail(CUSIFT_ERR_INVALID, "pipe: nothing in flight");
  HIP_TRY(hipSetDevice(p->device));
  int rc = progress(p, (long)p->collected);
  if (rc != CUSIFT_OK) {
    p->failed = true;
    return rc;
  }
  Slot &s = p->slots[p->collected % p->depth];
  {
    const hipError_t e = hipEventSynchronize(s.ev_copied);
    if (e != hipSuccess) {
      p->failed = true;
      return cusift_fail(CUSIFT_ERR_HIP, "pipe: hipEventSynchronize failed: %s", hipGetErrorString(e));
    }
  }
  if (h_records) *h_records = s.h_records;
  if (h_offsets) *h_offsets = s.h_offsets;
  if (n_images) *n_images = s.n;
  if (total) *total = s.total;
  s.busy = false;
  p->collected++;
  return CUSIFT_OK;
}
