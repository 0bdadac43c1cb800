// sift_comm.hip -- the multi-GPU step behind the C ABI: one process per GPU, RCCL over xGMI.
//
// The reference is single-GPU (no collective call sites); BASELINE configs[3]/[4] add
//   * a batch sharded over the GPUs of a node with an ALL-GATHERV OF SiftData (variable-length lists of 588-byte
//     SiftPoint records, cuSIFT.h:10-30), and
//   * one large image strip-tiled over the GPUs with a HALO EXCHANGE of octave rows between neighbours.
// Both are written against RCCL directly (ncclAllGather for the counts, then ONE group of ncclSend/ncclRecv per step):
// xGMI is point-to-point (7 links per GPU), so every shard / halo travels over its own link instead of hopping round
// a ring.  Images are independent, so the extraction itself needs no collective.
//
// RCCL is bound at run time (dlopen), from the directory of the HIP runtime the process already holds: a PyTorch
// process gets the wheel's librccl.so (built against the wheel's libamdhip64), a plain C++ program the system's
// /opt/rocm/lib/librccl.so.1.  libcusift_amd.so therefore has no link-time dependency on RCCL, and a program that
// never creates a communicator never loads it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
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

// ---- RCCL, bound at run time ---------------------------------------------------------------------------------
struct Rccl {
  void *handle = nullptr;
  std::string path, error;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl &rccl_state() {
  static Rccl r;
  return r;
}

bool try_open(Rccl &r, const std::string &name) {
  void *h = dlopen(name.c_str(), RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    r.error += name + ": " + dlerror() + "; ";
    return false;
  }
  r.handle = h;
  r.path = name;
  return true;
}

int load_rccl() {
  static std::once_flag once;
  Rccl &r = rccl_state();
  std::call_once(once, [&]() {
    std::vector<std::string> cands;
    if (const char *e = getenv("CUSIFT_RCCL_LIB")) cands.push_back(e);
    // next to the HIP runtime this process already uses (a torch wheel ships both; so does /opt/rocm/lib)
    Dl_info info;
    if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
      std::string dir(info.dli_fname);
      const size_t slash = dir.rfind('/');
      if (slash != std::string::npos) {
        dir.resize(slash + 1);
        cands.push_back(dir + "librccl.so.1");
        cands.push_back(dir + "librccl.so");
      }
    }
    cands.push_back("librccl.so.1");
    cands.push_back("librccl.so");
    for (const auto &c : cands)
      if (try_open(r, c)) break;
    if (!r.handle) return;
#define BIND(field, sym)                                       \
  r.field = (decltype(r.field))dlsym(r.handle, sym);           \
  if (!r.field) r.error += std::string("missing symbol ") + sym + "; ";
    BIND(GetUniqueId, "ncclGetUniqueId")
    BIND(CommInitRank, "ncclCommInitRank")
    BIND(CommDestroy, "ncclCommDestroy")
    BIND(AllGather, "ncclAllGather")
    BIND(Send, "ncclSend")
    BIND(Recv, "ncclRecv")
    BIND(GroupStart, "ncclGroupStart")
    BIND(GroupEnd, "ncclGroupEnd")
    BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
  });
  if (!r.handle) return cusift_fail(CUSIFT_ERR_INVALID, "RCCL not found (%s); set CUSIFT_RCCL_LIB", r.error.c_str());
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.Send || !r.Recv || !r.GroupStart ||
      !r.GroupEnd || !r.GetErrorString)
    return cusift_fail(CUSIFT_ERR_INVALID, "RCCL at %s is incomplete: %s", r.path.c_str(), r.error.c_str());
  return CUSIFT_OK;
}

#define NCCL_TRY(expr)                                                                                              \
  do {                                                                                                              \
    ncclResult_t r_ = (expr);                                                                                       \
    if (r_ != ncclSuccess)                                                                                          \
      return cusift_fail(CUSIFT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, rccl_state().GetErrorString(r_), __FILE__, \
                         __LINE__);                                                                                 \
  } while (0)

static_assert(CUSIFT_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "cusift_comm ids are ncclUniqueId");
static_assert(sizeof(cusift_point) % 4 == 0, "records travel as 32-bit words");
constexpr size_t kWordsPerPoint = sizeof(cusift_point) / 4;  // 147

// valid[i] = min(counters[i], max_pts) for the rank's images, 0 for the padding slots up to n_slots
__global__ void clamp_counts_kernel(const unsigned int *__restrict__ counters, int n_images, int max_pts,
                                    unsigned int *__restrict__ valid, int n_slots) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_slots) return;
  unsigned int c = 0;
  if (i < n_images) {
    c = counters[i];
    if (c > (unsigned int)max_pts) c = (unsigned int)max_pts;
  }
  valid[i] = c;
}

}  // namespace

struct cusift_comm {
  cusift_ctx *ctx = nullptr;  // device + stream every exchange is enqueued on (not owned)
  hipStream_t stream = nullptr;
  int device = 0, rank = 0, world = 1;
  ncclComm_t nccl = nullptr;
  int self_p2p = 0;  // world 1 / tests: route the local shard through ncclSend/ncclRecv to self as well
  // counts of the all-gatherv in flight
  int n_slots = 0;                         // count slots per rank of the buffers below
  unsigned int *d_local = nullptr;         // [n_slots]
  unsigned int *d_all = nullptr;           // [world * n_slots]
  unsigned int *h_all = nullptr;           // pinned, [world * n_slots]
  hipEvent_t counts_ready = nullptr;
  bool pending = false;
  const cusift_point *p_points = nullptr;  // arguments of the pending begin()
  const unsigned int *p_counters = nullptr;
  int p_images = 0, p_max_pts = 0, p_slots = 0;
  cusift_point *d_stage = nullptr;         // self_p2p only: the packed local shard before it is "sent"
  size_t stage_cap = 0;
};

static int comm_enter(cusift_comm *c) {
  if (!c) return cusift_fail(CUSIFT_ERR_INVALID, "comm is NULL");
  HIP_TRY(hipSetDevice(c->device));
  return CUSIFT_OK;
}

static int ensure_slots(cusift_comm *c, int n_slots) {
  if (n_slots <= c->n_slots) return CUSIFT_OK;
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->d_local) HIP_TRY(hipFree(c->d_local));
  if (c->d_all) HIP_TRY(hipFree(c->d_all));
  if (c->h_all) HIP_TRY(hipHostFree(c->h_all));
  c->d_local = c->d_all = c->h_all = nullptr;
  c->n_slots = 0;
  HIP_TRY(hipMalloc((void **)&c->d_local, sizeof(unsigned int) * n_slots));
  HIP_TRY(hipMalloc((void **)&c->d_all, sizeof(unsigned int) * (size_t)n_slots * c->world));
  HIP_TRY(hipHostMalloc((void **)&c->h_all, sizeof(unsigned int) * (size_t)n_slots * c->world, hipHostMallocDefault));
  c->n_slots = n_slots;
  return CUSIFT_OK;
}

// ------------------------------------------------------------------------------------------------
// communicator
// ------------------------------------------------------------------------------------------------
extern "C" int cusift_comm_get_unique_id(char id[CUSIFT_UNIQUE_ID_BYTES]) {
  if (!id) return cusift_fail(CUSIFT_ERR_INVALID, "id is NULL");
  TRY(load_rccl());
  ncclUniqueId u;
  NCCL_TRY(rccl_state().GetUniqueId(&u));
  memcpy(id, u.internal, CUSIFT_UNIQUE_ID_BYTES);
  return CUSIFT_OK;
}

extern "C" int cusift_comm_create(cusift_comm **out, cusift_ctx *ctx, const char id[CUSIFT_UNIQUE_ID_BYTES], int rank,
                                  int world) {
  if (!out) return cusift_fail(CUSIFT_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (!ctx || !id) return cusift_fail(CUSIFT_ERR_INVALID, "ctx / id is NULL");
  if (world < 1 || rank < 0 || rank >= world) return cusift_fail(CUSIFT_ERR_INVALID, "bad rank %d of %d", rank, world);
  TRY(load_rccl());
  const int device = cusift_ctx_device(ctx);
  HIP_TRY(hipSetDevice(device));
  cusift_comm *c = new cusift_comm();
  c->ctx = ctx;
  c->stream = (hipStream_t)cusift_ctx_stream(ctx);
  c->device = device;
  c->rank = rank;
  c->world = world;
  if (const char *e = getenv("CUSIFT_COMM_SELF_P2P")) c->self_p2p = atoi(e) != 0;
  ncclUniqueId u;
  memcpy(u.internal, id, CUSIFT_UNIQUE_ID_BYTES);
  ncclResult_t r = rccl_state().CommInitRank(&c->nccl, world, u, rank);
  if (r != ncclSuccess) {
    delete c;
    return cusift_fail(CUSIFT_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world,
                       rccl_state().GetErrorString(r));
  }
  hipError_t e = hipEventCreateWithFlags(&c->counts_ready, hipEventDisableTiming);
  if (e != hipSuccess) {
    (void)rccl_state().CommDestroy(c->nccl);
    delete c;
    return cusift_fail(CUSIFT_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(e));
  }
  *out = c;
  return CUSIFT_OK;
}

extern "C" int cusift_comm_destroy(cusift_comm *c) {
  if (!c) return CUSIFT_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->nccl) (void)rccl_state().CommDestroy(c->nccl);
  if (c->counts_ready) (void)hipEventDestroy(c->counts_ready);
  if (c->d_local) (void)hipFree(c->d_local);
  if (c->d_all) (void)hipFree(c->d_all);
  if (c->h_all) (void)hipHostFree(c->h_all);
  if (c->d_stage) (void)hipFree(c->d_stage);
  delete c;
  return CUSIFT_OK;
}

extern "C" int cusift_comm_rank(cusift_comm *c, int *rank, int *world) {
  if (!c) return cusift_fail(CUSIFT_ERR_INVALID, "comm is NULL");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  return CUSIFT_OK;
}

extern "C" int cusift_comm_set_self_p2p(cusift_comm *c, int on) {
  if (!c) return cusift_fail(CUSIFT_ERR_INVALID, "comm is NULL");
  c->self_p2p = on != 0;
  return CUSIFT_OK;
}

extern "C" const char *cusift_comm_library(void) { return rccl_state().path.c_str(); }

// ------------------------------------------------------------------------------------------------
// all-gatherv of SiftData
// ------------------------------------------------------------------------------------------------
extern "C" int cusift_allgatherv_begin(cusift_comm *c, const cusift_point *d_points, const unsigned int *d_counters,
                                       int n_images, int max_pts, int n_images_max) {
  TRY(comm_enter(c));
  if (!d_points || !d_counters) return cusift_fail(CUSIFT_ERR_INVALID, "allgatherv: missing data");
  if (n_images < 0 || n_images > cusift::kMaxFlatImages || max_pts < 1 || n_images_max < std::max(1, n_images))
    return cusift_fail(CUSIFT_ERR_INVALID, "allgatherv: need 0 <= n_images <= %d, n_images <= n_images_max, max_pts >= 1",
                       cusift::kMaxFlatImages);
  if (c->pending) return cusift_fail(CUSIFT_ERR_INVALID, "allgatherv: the previous begin() has not been finished");
  TRY(ensure_slots(c, n_images_max));
  hipLaunchKernelGGL(clamp_counts_kernel, dim3((n_images_max + 255) / 256), dim3(256), 0, c->stream, d_counters,
                     n_images, max_pts, c->d_local, n_images_max);
  HIP_TRY(hipGetLastError());
  NCCL_TRY(rccl_state().AllGather(c->d_local, c->d_all, (size_t)n_images_max, ncclUint32, c->nccl, c->stream));
  HIP_TRY(hipMemcpyAsync(c->h_all, c->d_all, sizeof(unsigned int) * (size_t)n_images_max * c->world,
                         hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipEventRecord(c->counts_ready, c->stream));
  c->pending = true;
  c->p_points = d_points;
  c->p_counters = d_counters;
  c->p_images = n_images;
  c->p_max_pts = max_pts;
  c->p_slots = n_images_max;
  return CUSIFT_OK;
}

extern "C" int cusift_allgatherv_finish(cusift_comm *c, cusift_point *d_gathered, size_t capacity,
                                        unsigned int *h_counts, size_t *h_offsets) {
  TRY(comm_enter(c));
  if (!c->pending) return cusift_fail(CUSIFT_ERR_INVALID, "allgatherv: finish() without begin()");
  if (!d_gathered) return cusift_fail(CUSIFT_ERR_INVALID, "allgatherv: d_gathered is NULL");
  // The one host wait of the exchange: the sizes of ncclSend/ncclRecv are host arguments.  A pipelined caller enqueues
  // its next extraction between begin() and finish(), by which time this event has long fired.
  HIP_TRY(hipEventSynchronize(c->counts_ready));
  c->pending = false;
  const int W = c->world, S = c->p_slots;
  std::vector<size_t> off((size_t)W + 1, 0);
  for (int r = 0; r < W; ++r) {
    size_t t = 0;
    for (int i = 0; i < S; ++i) t += c->h_all[(size_t)r * S + i];
    off[r + 1] = off[r] + t;
  }
  if (h_counts) memcpy(h_counts, c->h_all, sizeof(unsigned int) * (size_t)W * S);
  if (h_offsets) memcpy(h_offsets, off.data(), sizeof(size_t) * ((size_t)W + 1));
  if (off[W] > capacity)
    return cusift_fail(CUSIFT_ERR_NOMEM, "allgatherv: %zu records gathered but d_gathered holds %zu", off[W], capacity);
  const size_t mine = off[c->rank + 1] - off[c->rank];
  // the local shard is packed straight into its place in the gathered buffer and sent from there
  cusift_point *dst_mine = d_gathered + off[c->rank];
  cusift_point *pack_to = dst_mine;
  if (c->self_p2p && mine > 0) {
    if (mine > c->stage_cap) {
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (c->d_stage) HIP_TRY(hipFree(c->d_stage));
      c->d_stage = nullptr;
      c->stage_cap = 0;
      HIP_TRY(hipMalloc((void **)&c->d_stage, sizeof(cusift_point) * mine));
      c->stage_cap = mine;
    }
    pack_to = c->d_stage;
  }
  if (mine > 0 && c->p_images > 0)
    TRY(cusift_pack_points(c->ctx, c->p_points, c->p_counters, c->p_images, c->p_max_pts, pack_to, mine, nullptr));
  // ONE group: every shard goes straight to every peer (and arrives straight from it) over the direct xGMI link
  Rccl &R = rccl_state();
  bool any = false;
  for (int step = c->self_p2p ? 0 : 1; step < W; ++step) any = true;
  if (any) {
    NCCL_TRY(R.GroupStart());
    for (int step = c->self_p2p ? 0 : 1; step < W; ++step) {
      const int to = (c->rank + step) % W, from = (c->rank - step + W) % W;
      const size_t n_from = off[from + 1] - off[from];
      if (mine > 0) NCCL_TRY(R.Send(pack_to, mine * kWordsPerPoint, ncclUint32, to, c->nccl, c->stream));
      if (n_from > 0)
        NCCL_TRY(R.Recv(d_gathered + off[from], n_from * kWordsPerPoint, ncclUint32, from, c->nccl, c->stream));
    }
    NCCL_TRY(R.GroupEnd());
  }
  return CUSIFT_OK;
}

extern "C" int cusift_allgatherv(cusift_comm *c, const cusift_point *d_points, const unsigned int *d_counters,
                                 int n_images, int max_pts, int n_images_max, cusift_point *d_gathered, size_t capacity,
                                 unsigned int *h_counts, size_t *h_offsets) {
  TRY(cusift_allgatherv_begin(c, d_points, d_counters, n_images, max_pts, n_images_max));
  return cusift_allgatherv_finish(c, d_gathered, capacity, h_counts, h_offsets);
}

// ------------------------------------------------------------------------------------------------
// rows of a pitched float image between ranks: halo exchange of the strip tiling, and gather/scatter of bands
// ------------------------------------------------------------------------------------------------
extern "C" int cusift_exchange_rows(cusift_comm *c, float *d_band, int pitch, int n_ops, const int *peers,
                                    const int *send_row, const int *send_rows, const int *recv_row,
                                    const int *recv_rows) {
  TRY(comm_enter(c));
  if (n_ops < 0 || (n_ops > 0 && (!d_band || !peers || !send_row || !send_rows || !recv_row || !recv_rows)) || pitch < 1)
    return cusift_fail(CUSIFT_ERR_INVALID, "exchange_rows: bad argument");
  bool any = false;
  for (int i = 0; i < n_ops; ++i) {
    if (peers[i] < 0 || peers[i] >= c->world || send_rows[i] < 0 || recv_rows[i] < 0 || send_row[i] < 0 ||
        recv_row[i] < 0)
      return cusift_fail(CUSIFT_ERR_INVALID, "exchange_rows: op %d: bad peer or rows", i);
    if (peers[i] == c->rank && !c->self_p2p)
      return cusift_fail(CUSIFT_ERR_INVALID, "exchange_rows: op %d addresses this rank", i);
    any = any || send_rows[i] > 0 || recv_rows[i] > 0;
  }
  if (!any) return CUSIFT_OK;
  Rccl &R = rccl_state();
  NCCL_TRY(R.GroupStart());
  for (int i = 0; i < n_ops; ++i) {
    if (send_rows[i] > 0)
      NCCL_TRY(R.Send(d_band + (size_t)send_row[i] * pitch, (size_t)send_rows[i] * pitch, ncclFloat32, peers[i], c->nccl,
                      c->stream));
    if (recv_rows[i] > 0)
      NCCL_TRY(R.Recv(d_band + (size_t)recv_row[i] * pitch, (size_t)recv_rows[i] * pitch, ncclFloat32, peers[i], c->nccl,
                      c->stream));
  }
  NCCL_TRY(R.GroupEnd());
  return CUSIFT_OK;
}

extern "C" int cusift_exchange_halos(cusift_comm *c, float *d_band, int pitch, int top_halo, int own_rows,
                                     int bottom_halo, int send_rows) {
  if (!c) return cusift_fail(CUSIFT_ERR_INVALID, "comm is NULL");
  if (top_halo < 0 || bottom_halo < 0 || own_rows < 1 || send_rows < 0 || send_rows > own_rows)
    return cusift_fail(CUSIFT_ERR_INVALID, "exchange_halos: bad row geometry");
  int peers[2], srow[2], srows[2], rrow[2], rrows[2], n = 0;
  if (c->rank > 0 && (top_halo > 0 || send_rows > 0)) {  // neighbour above: my first owned rows go up
    peers[n] = c->rank - 1;
    srow[n] = top_halo;
    srows[n] = send_rows;
    rrow[n] = 0;
    rrows[n] = top_halo;
    ++n;
  }
  if (c->rank < c->world - 1 && (bottom_halo > 0 || send_rows > 0)) {  // neighbour below: my last owned rows go down
    peers[n] = c->rank + 1;
    srow[n] = top_halo + own_rows - send_rows;
    srows[n] = send_rows;
    rrow[n] = top_halo + own_rows;
    rrows[n] = bottom_halo;
    ++n;
  }
  return cusift_exchange_rows(c, d_band, pitch, n, peers, srow, srows, rrow, rrows);
}
