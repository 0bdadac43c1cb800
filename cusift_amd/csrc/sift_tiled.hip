// sift_tiled.hip -- ONE large image strip-tiled over several GPUs (BASELINE configs[4]: a single 8192x8192 image over
// 8 MI355X, merged SiftData): the rank-side driver behind the C ABI (cusift_tiled_*).  Host code only -- the kernels
// are the band forms of the ordinary ones (cusift_scale_down_band / cusift_detect_band / cusift_describe_band) and the
// exchanges are sift_comm.hip's grouped ncclSend/ncclRecv.
//
// New functionality: the reference has no tiling (its arena is sized for the whole image, cuSIFT.cu:81-98).  What is
// mirrored is its octave loop, cuSIFT.cu:175-202: ScaleDown finest -> coarsest (:185), then the octaves searched
// coarsest first (:190-196), fstPts snapshot before each (:243), orientation + descriptor on the new points (:253-258).
// The equality target is the single-GPU result on the whole image, which this scheme reproduces bit for bit.
//
// Partition.  Rank k owns base rows [b_k, b_{k+1}) with b_k = k*H / P (any H: strips differ by at most one row); in
// octave o it owns rows [b_k >> o, b_{k+1} >> o) and a keypoint belongs to the rank that owns its integer detection
// row, so nothing is found twice.  Every octave band carries `halo` rows of true neighbour data above and below (none at
// the real image border): 4 rows for the 9-tap blur, 1 for the extremum test, the rest for the orientation / descriptor
// footprint.  A keypoint whose footprint leaves the halo would sample clamped rows instead of the neighbour's: the
// band kernels count those on the device and cusift_tiled_check() turns a non-zero count into an error.
// Coarse octaves collapse onto rank 0: from the first octave in which some rank would own fewer
// rows than the halo, every rank ships its owned rows of that octave to the root, which runs the ordinary whole-image
// driver on it for that octave and all coarser ones (its initBlur, its subsampling: identical arithmetic).
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <vector>

#include "sift_internal.h"
#include "sift_types.h"

using cusift::kMaxOctaves;

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

// Row geometry of every (rank, octave).  Pure host arithmetic (cusift_tiled_plan exposes it without a GPU).
struct StripPlan {
  int W = 0, H = 0, world = 1, halo = 0, n_oct = 0, root = 0, collapse = 0;
  int w[kMaxOctaves], h[kMaxOctaves], pitch[kMaxOctaves];
  std::vector<long> bounds;  // base rows: rank k owns [bounds[k], bounds[k + 1])

  void own(int rank, int o, int &a, int &b) const {
    a = (int)(bounds[rank] >> o);
    b = (int)(bounds[rank + 1] >> o);
  }
  // Global rows [lo, hi) of octave o held by `rank`: owned rows + halo (tiled octaves), owned rows only (the collapse
  // octave: they are shipped to the root, no neighbour data is needed).
  void band(int rank, int o, int &lo, int &hi) const {
    int a, b;
    own(rank, o, a, b);
    if (o >= collapse || world == 1) {
      lo = a;
      hi = b;
    } else {
      lo = std::max(0, a - halo);
      hi = std::min(h[o], b + halo);
    }
  }
  bool tiled(int o) const { return o < collapse; }
  int n_bands() const { return std::min(collapse + 1, n_oct); }
};

int make_strip_plan(StripPlan &pl, int W, int H, int world, int num_octaves, int halo) {
  if (world < 1 || H < world || W < 1)
    return cusift_fail(CUSIFT_ERR_INVALID, "tiled: need world >= 1 and at least one base row per rank (W=%d H=%d world=%d)",
                       W, H, world);
  if (halo < 8) return cusift_fail(CUSIFT_ERR_INVALID, "tiled: halo must be >= 8 rows (4 blur + 1 extremum + the ScaleDown taps)");
  pl.W = W;
  pl.H = H;
  pl.world = world;
  pl.halo = halo;
  pl.w[0] = W;
  pl.h[0] = H;
  pl.n_oct = 1;
  const int n = std::max(1, std::min(num_octaves, kMaxOctaves));
  for (int o = 1; o < n; ++o) {
    const int ww = pl.w[o - 1] / 2, hh = pl.h[o - 1] / 2;  // integer division, cuSIFT.cu:182
    if (ww < 1 || hh < 1) break;
    pl.w[o] = ww;
    pl.h[o] = hh;
    pl.n_oct = o + 1;
  }
  for (int o = 0; o < pl.n_oct; ++o) pl.pitch[o] = ialign_up(pl.w[o], 128);  // cuSIFT.cu:183
  pl.bounds.resize((size_t)world + 1);
  for (int k = 0; k <= world; ++k) pl.bounds[k] = (long)k * H / world;
  pl.root = 0;
  pl.collapse = pl.n_oct;
  for (int o = 0; o < pl.n_oct; ++o) {
    int thinnest = 1 << 30;
    for (int k = 0; k < world; ++k) {
      int a, b;
      pl.own(k, o, a, b);
      thinnest = std::min(thinnest, b - a);
    }
    // a rank that owns fewer rows than the halo cannot serve its neighbour's halo; and the band kernels need w >= 4 and
    // h >= 3: narrower / flatter octaves run whole on the root as well (with one rank, too)
    if ((world > 1 && thinnest < halo) || pl.w[o] < 4 || pl.h[o] < 3) {
      pl.collapse = o;
      break;
    }
  }
  return CUSIFT_OK;
}

}  // namespace

struct cusift_tiled {
  cusift_ctx *ctx = nullptr;
  cusift_comm *comm = nullptr;
  hipStream_t stream = nullptr;
  int device = 0, rank = 0;
  cusift_params prm;
  StripPlan pl;
  double blur[kMaxOctaves];
  float sub[kMaxOctaves];
  float *bands[kMaxOctaves] = {nullptr};
  float *full = nullptr;  // the whole collapse octave, on the root
  unsigned int *d_small = nullptr;  // [0] first (fstPts), [1] flags
  bool per_octave = false;          // $CUSIFT_TILED_PER_OCTAVE (read at creation): the tiled octaves one at a time
  bool loaded = false;
  ~cusift_tiled() {
    (void)hipSetDevice(device);
    for (float *b : bands)
      if (b) (void)hipFree(b);
    if (full) (void)hipFree(full);
    if (d_small) (void)hipFree(d_small);
  }
};

static int tiled_enter(cusift_tiled *t) {
  if (!t) return cusift_fail(CUSIFT_ERR_INVALID, "tiled is NULL");
  HIP_TRY(hipSetDevice(t->device));
  return CUSIFT_OK;
}

// ------------------------------------------------------------------------------------------------
// plan (host only)
// ------------------------------------------------------------------------------------------------
extern "C" int cusift_tiled_plan(int W, int H, int world, int num_octaves, int halo_rows, int rank, int octave,
                                 int *n_octaves, int *collapse_octave, int *w, int *h, int *pitch, int *own_begin,
                                 int *own_end, int *band_begin, int *band_end) {
  StripPlan pl;
  TRY(make_strip_plan(pl, W, H, world, num_octaves, halo_rows > 0 ? halo_rows : CUSIFT_TILED_DEFAULT_HALO));
  if (n_octaves) *n_octaves = pl.n_oct;
  if (collapse_octave) *collapse_octave = pl.collapse;
  if (rank < 0 || rank >= world || octave < 0 || octave >= pl.n_oct)
    return cusift_fail(CUSIFT_ERR_INVALID, "tiled plan: rank %d / octave %d out of range", rank, octave);
  int a, b, lo, hi;
  pl.own(rank, octave, a, b);
  pl.band(rank, octave, lo, hi);
  if (w) *w = pl.w[octave];
  if (h) *h = pl.h[octave];
  if (pitch) *pitch = pl.pitch[octave];
  if (own_begin) *own_begin = a;
  if (own_end) *own_end = b;
  if (band_begin) *band_begin = lo;
  if (band_end) *band_end = hi;
  return CUSIFT_OK;
}

// ------------------------------------------------------------------------------------------------
// one rank
// ------------------------------------------------------------------------------------------------
extern "C" int cusift_tiled_create(cusift_tiled **out, cusift_ctx *ctx, cusift_comm *comm, int rank, int world, int W,
                                   int H, const cusift_params *prm, int halo_rows) {
  if (!out) return cusift_fail(CUSIFT_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (!ctx || !prm) return cusift_fail(CUSIFT_ERR_INVALID, "tiled: ctx / params is NULL");
  if (rank < 0 || rank >= world) return cusift_fail(CUSIFT_ERR_INVALID, "tiled: bad rank %d of %d", rank, world);
  if (prm->max_pts < 1) return cusift_fail(CUSIFT_ERR_INVALID, "tiled: max_pts must be >= 1");
  if (comm) {
    int cr = -1, cw = -1;
    TRY(cusift_comm_rank(comm, &cr, &cw));
    if (cr != rank || cw != world)
      return cusift_fail(CUSIFT_ERR_INVALID, "tiled: the communicator is rank %d of %d, not %d of %d", cr, cw, rank, world);
    // kernels and exchanges must share ONE stream: a halo received on another stream would race the detection
    if (cusift_comm_ctx(comm) != ctx && cusift_ctx_stream(cusift_comm_ctx(comm)) != cusift_ctx_stream(ctx))
      return cusift_fail(CUSIFT_ERR_INVALID,
                         "tiled: the communicator is bound to another context / stream than the extraction's");
  }
  std::unique_ptr<cusift_tiled> t(new cusift_tiled());
  t->ctx = ctx;
  t->comm = comm;
  t->stream = (hipStream_t)cusift_ctx_stream(ctx);
  t->device = cusift_ctx_device(ctx);
  t->rank = rank;
  t->prm = *prm;
  TRY(make_strip_plan(t->pl, W, H, world, prm->num_octaves, halo_rows > 0 ? halo_rows : CUSIFT_TILED_DEFAULT_HALO));
  const StripPlan &pl = t->pl;
  // cuSIFT.cu:188: float totInitBlur = (float)sqrt(initBlur*initBlur + 0.5f*0.5f) / 2.0f, recursively
  t->blur[0] = prm->init_blur;
  t->sub[0] = prm->subsampling;
  for (int o = 1; o < pl.n_oct; ++o) {
    t->blur[o] = (float)sqrt(t->blur[o - 1] * t->blur[o - 1] + 0.5f * 0.5f) / 2.0f;
    t->sub[o] = t->sub[o - 1] * 2.0f;
  }
  HIP_TRY(hipSetDevice(t->device));
  for (int o = 0; o < pl.n_bands(); ++o) {
    int lo, hi;
    pl.band(rank, o, lo, hi);
    const size_t bytes = std::max<size_t>(1, (size_t)(hi - lo) * pl.pitch[o]) * sizeof(float);
    HIP_TRY(hipMalloc((void **)&t->bands[o], bytes));
    HIP_TRY(hipMemsetAsync(t->bands[o], 0, bytes, t->stream));
  }
  if (pl.collapse < pl.n_oct && rank == pl.root) {
    const size_t bytes = (size_t)pl.h[pl.collapse] * pl.pitch[pl.collapse] * sizeof(float);
    HIP_TRY(hipMalloc((void **)&t->full, bytes));
    HIP_TRY(hipMemsetAsync(t->full, 0, bytes, t->stream));
    // the whole-image driver's scratch, sized now so that the extraction itself never allocates
    cusift_params sub = *prm;
    sub.num_octaves = pl.n_oct - pl.collapse;
    TRY(cusift_ctx_reserve(ctx, 1, pl.w[pl.collapse], pl.h[pl.collapse], &sub));
  }
  // every rank: the scratch of cusift_extract_bands for its tiled octaves (the root's arena is the larger of the two)
  TRY(cusift_ctx_reserve_bands(ctx, std::min(std::min(pl.collapse, pl.n_oct), 8), prm->max_pts));
  int per_octave = 0;
  TRY(cusift_ctx_get_policy(ctx, CUSIFT_POLICY_TILED_PER_OCTAVE, &per_octave));
  t->per_octave = per_octave != 0;
  HIP_TRY(hipMalloc((void **)&t->d_small, 256));
  HIP_TRY(hipMemsetAsync(t->d_small, 0, 256, t->stream));
  *out = t.release();
  return CUSIFT_OK;
}

extern "C" int cusift_tiled_destroy(cusift_tiled *t) {
  if (!t) return CUSIFT_OK;
  (void)hipSetDevice(t->device);
  (void)hipStreamSynchronize(t->stream);
  delete t;
  return CUSIFT_OK;
}

extern "C" int cusift_tiled_info(cusift_tiled *t, int *n_octaves, int *collapse_octave, int *root, int *halo_rows) {
  if (!t) return cusift_fail(CUSIFT_ERR_INVALID, "tiled is NULL");
  if (n_octaves) *n_octaves = t->pl.n_oct;
  if (collapse_octave) *collapse_octave = t->pl.collapse;
  if (root) *root = t->pl.root;
  if (halo_rows) *halo_rows = t->pl.halo;
  return CUSIFT_OK;
}

extern "C" int cusift_tiled_band(cusift_tiled *t, int octave, float **d_band, int *w, int *h_global, int *pitch,
                                 int *own_begin, int *own_end, int *band_begin, int *band_end) {
  if (!t) return cusift_fail(CUSIFT_ERR_INVALID, "tiled is NULL");
  if (octave < 0 || octave >= t->pl.n_oct) return cusift_fail(CUSIFT_ERR_INVALID, "tiled: octave %d out of range", octave);
  int a, b, lo, hi;
  t->pl.own(t->rank, octave, a, b);
  t->pl.band(t->rank, octave, lo, hi);
  if (d_band) *d_band = octave < t->pl.n_bands() ? t->bands[octave] : nullptr;
  if (w) *w = t->pl.w[octave];
  if (h_global) *h_global = t->pl.h[octave];
  if (pitch) *pitch = t->pl.pitch[octave];
  if (own_begin) *own_begin = a;
  if (own_end) *own_end = b;
  if (band_begin) *band_begin = lo;
  if (band_end) *band_end = hi;
  return CUSIFT_OK;
}

// This rank's owned base rows (b_{k+1} - b_k rows of W floats, `strip_pitch` floats apart, device memory) -> band 0.
extern "C" int cusift_tiled_load(cusift_tiled *t, const float *d_strip, int strip_pitch) {
  TRY(tiled_enter(t));
  const StripPlan &pl = t->pl;
  if (!d_strip || strip_pitch < pl.W) return cusift_fail(CUSIFT_ERR_INVALID, "tiled: bad strip");
  int a, b, lo, hi;
  pl.own(t->rank, 0, a, b);
  pl.band(t->rank, 0, lo, hi);
  HIP_TRY(hipMemcpy2DAsync(t->bands[0] + (size_t)(a - lo) * pl.pitch[0], sizeof(float) * pl.pitch[0], d_strip,
                           sizeof(float) * strip_pitch, sizeof(float) * pl.W, (size_t)(b - a), hipMemcpyDeviceToDevice,
                           t->stream));
  HIP_TRY(hipMemsetAsync(t->d_small, 0, 256, t->stream));
  t->loaded = true;
  return CUSIFT_OK;
}

// ScaleDown the owned rows of octave o from the band of octave o-1 (cuSIFT.cu:185).
extern "C" int cusift_tiled_build_octave(cusift_tiled *t, int o) {
  TRY(tiled_enter(t));
  const StripPlan &pl = t->pl;
  if (o < 1 || o >= pl.n_bands()) return cusift_fail(CUSIFT_ERR_INVALID, "tiled: octave %d is not built from a band", o);
  int a, b, lo, hi, slo, shi;
  pl.own(t->rank, o, a, b);
  pl.band(t->rank, o, lo, hi);
  pl.band(t->rank, o - 1, slo, shi);
  if (b <= a) return CUSIFT_OK;
  return cusift_scale_down_band(t->ctx, t->bands[o], pl.pitch[o], lo, a, b, t->bands[o - 1], pl.w[o - 1], shi - slo,
                                pl.pitch[o - 1], slo, pl.h[o - 1], 0.5f);
}

// Tiled octave: `halo` owned rows go to each neighbour, the neighbour's come in (one ncclGroup).  The collapse octave:
// every rank's owned rows -> the root's whole-octave image.
extern "C" int cusift_tiled_exchange(cusift_tiled *t, int o) {
  TRY(tiled_enter(t));
  const StripPlan &pl = t->pl;
  if (o < 0 || o >= pl.n_bands()) return cusift_fail(CUSIFT_ERR_INVALID, "tiled: octave %d has no band", o);
  int a, b, lo, hi;
  pl.own(t->rank, o, a, b);
  pl.band(t->rank, o, lo, hi);
  if (pl.tiled(o)) {
    if (pl.world == 1) return CUSIFT_OK;
    if (!t->comm) return cusift_fail(CUSIFT_ERR_INVALID, "tiled: world %d needs a communicator", pl.world);
    return cusift_exchange_halos(t->comm, t->bands[o], pl.pitch[o], a - lo, b - a, hi - b, pl.halo);
  }
  // o == collapse
  if (t->rank == pl.root) {
    if (b > a)
      HIP_TRY(hipMemcpyAsync(t->full + (size_t)a * pl.pitch[o], t->bands[o] + (size_t)(a - lo) * pl.pitch[o],
                             sizeof(float) * (size_t)(b - a) * pl.pitch[o], hipMemcpyDeviceToDevice, t->stream));
    std::vector<int> peer, srow, srows, rrow, rrows;
    for (int k = 0; k < pl.world; ++k) {
      int ka, kb;
      pl.own(k, o, ka, kb);
      if (k == pl.root || kb <= ka) continue;
      peer.push_back(k);
      srow.push_back(0);
      srows.push_back(0);
      rrow.push_back(ka);
      rrows.push_back(kb - ka);
    }
    if (peer.empty()) return CUSIFT_OK;
    if (!t->comm) return cusift_fail(CUSIFT_ERR_INVALID, "tiled: world %d needs a communicator", pl.world);
    return cusift_exchange_rows(t->comm, t->full, pl.pitch[o], pl.h[o], (int)peer.size(), peer.data(), srow.data(),
                                srows.data(), rrow.data(), rrows.data());
  }
  if (b <= a) return CUSIFT_OK;
  if (!t->comm) return cusift_fail(CUSIFT_ERR_INVALID, "tiled: world %d needs a communicator", pl.world);
  const int peer = pl.root, srow = a - lo, srows = b - a, zero = 0;
  return cusift_exchange_rows(t->comm, t->bands[o], pl.pitch[o], hi - lo, 1, &peer, &srow, &srows, &zero, &zero);
}

// The exchange of octave o among P extractors of ONE process (no communicator): device copies, enqueued on the
// receiving extractor's stream after everything the sending one has enqueued.  Used to run a plan on one GPU.
extern "C" int cusift_tiled_exchange_virtual(cusift_tiled **ranks, int n, int o) {
  if (!ranks || n < 1 || !ranks[0]) return cusift_fail(CUSIFT_ERR_INVALID, "tiled: no extractors");
  const StripPlan &pl = ranks[0]->pl;
  if (n != pl.world) return cusift_fail(CUSIFT_ERR_INVALID, "tiled: %d extractors for a world of %d", n, pl.world);
  if (o < 0 || o >= pl.n_bands()) return cusift_fail(CUSIFT_ERR_INVALID, "tiled: octave %d has no band", o);
  for (int k = 0; k < n; ++k)
    if (!ranks[k] || ranks[k]->rank != k || ranks[k]->device != ranks[0]->device)
      return cusift_fail(CUSIFT_ERR_INVALID, "tiled: extractor %d is missing, mis-ranked or on another device", k);
  TRY(tiled_enter(ranks[0]));
  const size_t row = sizeof(float) * pl.pitch[o];
  if (pl.tiled(o)) {
    for (int k = 0; k < n; ++k) {
      cusift_tiled *t = ranks[k];
      int a, b, lo, hi;
      pl.own(k, o, a, b);
      pl.band(k, o, lo, hi);
      if (k > 0) {  // the neighbour above sends its last owned rows down
        cusift_tiled *s = ranks[k - 1];
        int sa, sb, slo, shi;
        pl.own(k - 1, o, sa, sb);
        pl.band(k - 1, o, slo, shi);
        TRY(cusift_ctx_wait(t->ctx, s->ctx));
        HIP_TRY(hipMemcpyAsync(t->bands[o], s->bands[o] + (size_t)(sb - slo - (a - lo)) * pl.pitch[o], row * (a - lo),
                               hipMemcpyDeviceToDevice, t->stream));
        TRY(cusift_ctx_wait(s->ctx, t->ctx));  // the sender's next load must not overtake this copy
      }
      if (k < n - 1) {  // the neighbour below sends its first owned rows up
        cusift_tiled *s = ranks[k + 1];
        int sa, sb, slo, shi;
        pl.own(k + 1, o, sa, sb);
        pl.band(k + 1, o, slo, shi);
        TRY(cusift_ctx_wait(t->ctx, s->ctx));
        HIP_TRY(hipMemcpyAsync(t->bands[o] + (size_t)(b - lo) * pl.pitch[o], s->bands[o] + (size_t)(sa - slo) * pl.pitch[o],
                               row * (hi - b), hipMemcpyDeviceToDevice, t->stream));
        TRY(cusift_ctx_wait(s->ctx, t->ctx));
      }
    }
    return CUSIFT_OK;
  }
  cusift_tiled *root = ranks[pl.root];
  for (int k = 0; k < n; ++k) {
    int a, b, lo, hi;
    pl.own(k, o, a, b);
    pl.band(k, o, lo, hi);
    if (b <= a) continue;
    TRY(cusift_ctx_wait(root->ctx, ranks[k]->ctx));
    HIP_TRY(hipMemcpyAsync(root->full + (size_t)a * pl.pitch[o], ranks[k]->bands[o] + (size_t)(a - lo) * pl.pitch[o],
                           row * (b - a), hipMemcpyDeviceToDevice, root->stream));
    TRY(cusift_ctx_wait(ranks[k]->ctx, root->ctx));
  }
  return CUSIFT_OK;
}

// Detection + description of everything this rank owns, coarsest octave first (cuSIFT.cu:190-196).
extern "C" int cusift_tiled_process(cusift_tiled *t, cusift_point *d_points, unsigned int *d_counter) {
  TRY(tiled_enter(t));
  if (!d_points || !d_counter) return cusift_fail(CUSIFT_ERR_INVALID, "tiled: missing output");
  const StripPlan &pl = t->pl;
  const cusift_params &p = t->prm;
  HIP_TRY(hipMemsetAsync(d_counter, 0, sizeof(unsigned int), t->stream));  // cuSIFT.cu:69
  // Root only: octaves >= collapse as ONE whole-image extraction of the collapse octave (its initBlur, its subsampling,
  // the remaining octave count) -- the ordinary driver, so the ordinary results.  First: coarse octaves lead the list.
  if (pl.collapse < pl.n_oct && t->rank == pl.root) {
    const int oc = pl.collapse;
    cusift_params sub = p;
    sub.num_octaves = pl.n_oct - oc;
    sub.init_blur = t->blur[oc];
    sub.subsampling = t->sub[oc];
    TRY(cusift_extract_batch(t->ctx, t->full, 1, pl.w[oc], pl.h[oc], pl.pitch[oc], (size_t)pl.h[oc] * pl.pitch[oc], &sub,
                             d_points, d_counter));
  }
  unsigned int *d_first = t->d_small, *d_flags = t->d_small + 1;
  // The tiled octaves of this rank: ONE detection launch and ONE description launch for all of them
  // (cusift_extract_bands) when they are consecutive octaves -- they are, unless lowest_scale drops octaves from the
  // middle, which it cannot -- and few enough; otherwise octave by octave, coarsest first.
  {
    cusift_band bands[16];
    int n_bands = 0, first_o = -1, last_o = -1;
    for (int o = 0; o < std::min(pl.collapse, pl.n_oct); ++o) {
      if (!(p.lowest_scale < t->sub[o] * 2.0f)) continue;  // cuSIFT.cu:194
      int a, b, lo, hi;
      pl.own(t->rank, o, a, b);
      pl.band(t->rank, o, lo, hi);
      if (b <= a) continue;
      if (first_o < 0) first_o = o;
      last_o = o;
      if (n_bands < 16)
        bands[n_bands] = cusift_band{t->bands[o], pl.w[o], hi - lo, pl.pitch[o], lo, pl.h[o], a, b, (float)t->blur[o], t->sub[o]};
      ++n_bands;
    }
    const bool consecutive = n_bands > 0 && last_o - first_o + 1 == n_bands;
    if (consecutive && n_bands <= 8 && !t->per_octave)
      return cusift_extract_bands(t->ctx, bands, n_bands, p.peak_thresh, p.edge_thresh, d_points, p.max_pts, d_counter,
                                  p.tex_frac_bits, p.root_sift, d_flags);
    if (n_bands == 0) return CUSIFT_OK;
  }
  for (int o = std::min(pl.collapse, pl.n_oct) - 1; o >= 0; --o) {
    if (!(p.lowest_scale < t->sub[o] * 2.0f)) continue;  // cuSIFT.cu:194
    int a, b, lo, hi;
    pl.own(t->rank, o, a, b);
    pl.band(t->rank, o, lo, hi);
    if (b <= a) continue;
    // ExtractSiftOctave (cuSIFT.cu:204-270) on this rank's band, centres restricted to the owned rows
    HIP_TRY(hipMemcpyAsync(d_first, d_counter, sizeof(unsigned int), hipMemcpyDeviceToDevice, t->stream));  // fstPts, :243
    TRY(cusift_detect_band(t->ctx, t->bands[o], pl.w[o], hi - lo, pl.pitch[o], lo, pl.h[o], a, b, (float)t->blur[o],
                           p.peak_thresh, p.edge_thresh, t->sub[o], d_points, p.max_pts, d_counter));
    TRY(cusift_describe_band(t->ctx, t->bands[o], pl.w[o], hi - lo, pl.pitch[o], lo, pl.h[o], d_points, p.max_pts,
                             d_first, d_counter, t->sub[o], p.tex_frac_bits, p.root_sift, d_flags));
  }
  return CUSIFT_OK;
}

// The whole rank-side sequence: strip in, this rank's SiftData out.  Asynchronous on the context's stream; collective
// (every rank of the communicator calls it).  Merge with cusift_allgatherv.
extern "C" int cusift_tiled_extract(cusift_tiled *t, const float *d_strip, int strip_pitch, cusift_point *d_points,
                                    unsigned int *d_counter) {
  TRY(cusift_tiled_load(t, d_strip, strip_pitch));
  const StripPlan &pl = t->pl;
  for (int o = 0; o < pl.n_bands(); ++o) {
    if (o > 0) TRY(cusift_tiled_build_octave(t, o));
    TRY(cusift_tiled_exchange(t, o));
  }
  return cusift_tiled_process(t, d_points, d_counter);
}

// Blocking: how many keypoints of the last extraction sampled rows beyond the halo (their descriptors would differ from
// the whole image's).  Non-zero is an error -- loudly, never silently different; *flagged receives the count either way.
extern "C" int cusift_tiled_check(cusift_tiled *t, unsigned int *flagged) {
  TRY(tiled_enter(t));
  unsigned int f = 0;
  HIP_TRY(hipMemcpyAsync(&f, t->d_small + 1, sizeof(f), hipMemcpyDeviceToHost, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  if (flagged) *flagged = f;
  if (f)
    return cusift_fail(CUSIFT_ERR_INVALID,
                       "strip tiling: %u keypoint(s) of rank %d sample rows beyond the %d-row halo (scale too large for "
                       "the halo); results would differ from the whole image -- use a larger halo or fewer ranks",
                       f, t->rank, t->pl.halo);
  return CUSIFT_OK;
}
