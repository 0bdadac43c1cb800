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
// never creates a communicator never loads it.  cusift_comm_use_library() names another library with the same nine
// entry points for the communicators created after it -- the tests bind tests/fake_rccl (W ranks as W threads of one
// process on one GPU) next to the real RCCL in one process.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
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

// ---- RCCL (or a library with the same entry points), bound at run time ------------------------------------------
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
  // optional (cusift_comm_info): what the LIBRARY says about a communicator -- absent from a minimal stand-in
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  bool complete() const {
    return handle && GetUniqueId && CommInitRank && CommDestroy && AllGather && Send && Recv && GroupStart && GroupEnd &&
           GetErrorString;
  }
};

std::mutex g_lib_mu;
std::map<std::string, std::unique_ptr<Rccl>> g_libs;  // by the name it was asked for ("" = default search)
std::string g_lib_choice;                             // what cusift_comm_use_library() named last
std::string g_lib_path;                               // path of the library bound last (cusift_comm_library)

bool try_open(Rccl &r, const std::string &name) {
  void *h = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    r.error += name + ": " + dlerror() + "; ";
    return false;
  }
  r.handle = h;
  r.path = name;
  return true;
}

// The library the next communicator binds: cusift_comm_use_library()'s, else $CUSIFT_RCCL_LIB, else the librccl next
// to the HIP runtime of this process.
int load_rccl(Rccl **out) {
  std::lock_guard<std::mutex> lock(g_lib_mu);
  std::string key = g_lib_choice;
  if (key.empty())
    if (const char *e = getenv("CUSIFT_RCCL_LIB")) key = e;
  auto it = g_libs.find(key);
  if (it == g_libs.end()) {
    std::unique_ptr<Rccl> r(new Rccl());
    std::vector<std::string> cands;
    if (!key.empty()) {
      cands.push_back(key);
    } else {
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
    }
    for (const auto &c : cands)
      if (try_open(*r, c)) break;
    if (r->handle) {
#define BIND(field, sym)                                        \
  r->field = (decltype(r->field))dlsym(r->handle, sym);         \
  if (!r->field) r->error += std::string("missing symbol ") + sym + "; ";
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
      r->CommCount = (decltype(r->CommCount))dlsym(r->handle, "ncclCommCount");
      r->CommUserRank = (decltype(r->CommUserRank))dlsym(r->handle, "ncclCommUserRank");
      r->GetVersion = (decltype(r->GetVersion))dlsym(r->handle, "ncclGetVersion");
    }
    it = g_libs.emplace(key, std::move(r)).first;
  }
  Rccl *r = it->second.get();
  if (!r->handle)
    return cusift_fail(CUSIFT_ERR_INVALID, "RCCL not found (%s); set CUSIFT_RCCL_LIB or call cusift_comm_use_library",
                       r->error.c_str());
  if (!r->complete()) return cusift_fail(CUSIFT_ERR_INVALID, "RCCL at %s is incomplete: %s", r->path.c_str(), r->error.c_str());
  g_lib_path = r->path;
  *out = r;
  return CUSIFT_OK;
}

static_assert(CUSIFT_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "cusift_comm ids are ncclUniqueId");
static_assert(sizeof(cusift_point) % 4 == 0 && sizeof(cusift_compact_point) % 4 == 0 && sizeof(cusift_trimmed_point) % 4 == 0,
              "records travel as 32-bit words");
constexpr size_t kWordsPerPoint = sizeof(cusift_point) / 4;  // 147
constexpr int kSeqWords = 32;  // a ticket's arrival flag sits on a 128-byte line of its own

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

// The gathered counts go to pinned host memory by a kernel, followed by a sequence number the host can poll: no event,
// no copy engine, no HIP call on the reading side.  One workgroup; `n` is at most world * 256.
__global__ void __launch_bounds__(256) publish_counts_kernel(const unsigned int *__restrict__ d_all, int n,
                                                             unsigned int *__restrict__ h_all,
                                                             unsigned int *__restrict__ h_seq, unsigned int seq) {
  for (int i = threadIdx.x; i < n; i += 256) __builtin_nontemporal_store(d_all[i], h_all + i);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(h_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

// One all-gatherv in flight.
struct GatherTicket {
  unsigned int *d_local = nullptr;  // [n_slots]          this rank's clamped counts
  unsigned int *d_all = nullptr;    // [world * n_slots]  everybody's
  unsigned int *h_all = nullptr;    // pinned twin of d_all, written by publish_counts_kernel
  unsigned int *h_seq = nullptr;    // pinned; == seq once h_all is complete
  unsigned int *dev_h_all = nullptr, *dev_h_seq = nullptr;  // the device's addresses of the two pinned blocks
  unsigned int seq = 0;
  char *d_gathered = nullptr;
  size_t region_cap = 0;   // records per region
  size_t rec_bytes = 0;    // 588 (cusift_point) or 160 (cusift_compact_point)
  int slots = 0;  // count slots per rank of THIS exchange (<= comm n_slots)
  const char *stage = nullptr;  // self_p2p: this ticket's slice of the staging buffer (its packed local shard)
};

struct cusift_comm {
  cusift_ctx *ctx = nullptr;  // device + stream every exchange is enqueued on (not owned)
  hipStream_t stream = nullptr;
  int device = 0, rank = 0, world = 1;
  Rccl *lib = nullptr;
  ncclComm_t nccl = nullptr;
  int self_p2p = 0;  // world 1 / tests: route the local shard through ncclSend/ncclRecv to self as well
  int fixed = 0;     // 1: whole regions travel (region_cap records per peer), posted by begin(); no host read at all
  int compact = 0;   // wire format: 0 cusift_point (588 B), 1 cusift_compact_point (160 B), 2 cusift_trimmed_point (540 B)
  // ring of tickets: [head, head + pending) are in flight, oldest first
  int n_slots = 0, depth = 0, head = 0, pending = 0;
  unsigned int next_seq = 1;
  std::vector<GatherTicket> tickets;
  unsigned int *d_block = nullptr, *h_block = nullptr;
  // self_p2p only: the packed local shards before they are "sent" into their regions -- ONE SLICE PER TICKET
  // (stage_depth slices of stage_cap records): begin(i + 1) packs while the shard of ticket i is still unsent
  cusift_point *d_stage = nullptr;
  size_t stage_cap = 0;
  int stage_depth = 0;
  bool failed = false;  // an exchange timed out: device work of unknown state may still reference the tickets
  double host_wait_ms = 0.0;          // time finish() spent waiting for counts (diagnostic)
  unsigned long long host_waits = 0;  // finish() calls that found their counts not yet arrived (diagnostic)
  unsigned long long hip_syncs = 0;   // synchronising HIP calls this object has made (only ever while (re)sizing)
};

namespace {

#define NCCL_TRY(c, expr)                                                                                            \
  do {                                                                                                               \
    ncclResult_t r_ = (expr);                                                                                        \
    if (r_ != ncclSuccess)                                                                                           \
      return cusift_fail(CUSIFT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, (c)->lib->GetErrorString(r_), __FILE__,     \
                         __LINE__);                                                                                  \
  } while (0)

// ncclGroupStart .. ncclGroupEnd with the end guaranteed: an error return between the two must not leave the thread's
// group open (every later RCCL call of the thread would be queued into it and never run).
struct GroupScope {
  Rccl *lib;
  bool open = false;
  explicit GroupScope(Rccl *l) : lib(l) {}
  ncclResult_t start() {
    ncclResult_t r = lib->GroupStart();
    open = r == ncclSuccess;
    return r;
  }
  ncclResult_t end() {
    open = false;
    return lib->GroupEnd();
  }
  ~GroupScope() {
    if (open) (void)lib->GroupEnd();
  }
};

int comm_enter(cusift_comm *c) {
  if (!c) return cusift_fail(CUSIFT_ERR_INVALID, "comm is NULL");
  HIP_TRY(hipSetDevice(c->device));
  return CUSIFT_OK;
}

void free_tickets(cusift_comm *c) {
  if (c->d_block) (void)hipFree(c->d_block);
  if (c->h_block) (void)hipHostFree(c->h_block);
  c->d_block = c->h_block = nullptr;
  c->tickets.clear();
  c->n_slots = c->depth = c->head = c->pending = 0;
}

// (Re)builds the ticket ring: `depth` tickets of `n_slots` count slots per rank.  Synchronises; call before the loop.
int reserve_tickets(cusift_comm *c, int n_slots, int depth) {
  if (n_slots <= c->n_slots && depth <= c->depth) return CUSIFT_OK;
  if (c->pending) return cusift_fail(CUSIFT_ERR_INVALID, "comm: cannot re-size the ticket ring with exchanges in flight");
  n_slots = std::max(n_slots, c->n_slots);
  depth = std::max(depth, c->depth);
  c->hip_syncs++;
  HIP_TRY(hipStreamSynchronize(c->stream));
  free_tickets(c);
  const size_t W = (size_t)c->world;
  const size_t d_words = (size_t)n_slots * (1 + W);
  const size_t h_words = (size_t)n_slots * W + kSeqWords;
  HIP_TRY(hipMalloc((void **)&c->d_block, sizeof(unsigned int) * d_words * depth));
  HIP_TRY(hipHostMalloc((void **)&c->h_block, sizeof(unsigned int) * h_words * depth,
                        hipHostMallocMapped | hipHostMallocCoherent));
  memset(c->h_block, 0, sizeof(unsigned int) * h_words * depth);
  c->tickets.resize(depth);
  for (int t = 0; t < depth; ++t) {
    GatherTicket &k = c->tickets[t];
    k.d_local = c->d_block + d_words * t;
    k.d_all = k.d_local + n_slots;
    k.h_seq = c->h_block + h_words * t;
    k.h_all = k.h_seq + kSeqWords;
    HIP_TRY(hipHostGetDevicePointer((void **)&k.dev_h_seq, k.h_seq, 0));
    k.dev_h_all = k.dev_h_seq + kSeqWords;
  }
  c->n_slots = n_slots;
  c->depth = depth;
  return CUSIFT_OK;
}

// `records` per ticket, one slice for each ticket of the ring (call after reserve_tickets)
int reserve_stage(cusift_comm *c, size_t records) {
  if (records <= c->stage_cap && c->depth <= c->stage_depth) return CUSIFT_OK;
  if (c->pending)
    return cusift_fail(CUSIFT_ERR_INVALID, "comm: cannot grow the self-send staging (%zu -> %zu records per ticket) with "
                                           "exchanges in flight: an unsent shard lives there (cusift_comm_reserve first)",
                       c->stage_cap, records);
  records = std::max(records, c->stage_cap);
  c->hip_syncs++;
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->d_stage) HIP_TRY(hipFree(c->d_stage));
  c->d_stage = nullptr;
  c->stage_cap = 0;
  c->stage_depth = 0;
  HIP_TRY(hipMalloc((void **)&c->d_stage, sizeof(cusift_point) * records * (size_t)c->depth));
  c->stage_cap = records;
  c->stage_depth = c->depth;
  return CUSIFT_OK;
}

// ONE group: every shard goes straight to every peer (and arrives straight from it) over the direct xGMI link.
// n_records[r] = records rank r contributes (exact mode: its valid total; fixed mode: region_cap for everybody).
int post_shards(cusift_comm *c, const GatherTicket &k, const size_t *n_records) {
  const int W = c->world;
  const int first = c->self_p2p ? 0 : 1;
  if (first >= W) return CUSIFT_OK;
  const size_t mine = n_records[c->rank];
  const size_t words = k.rec_bytes / 4, region_bytes = k.region_cap * k.rec_bytes;
  const char *src = c->self_p2p ? k.stage : k.d_gathered + (size_t)c->rank * region_bytes;
  GroupScope g(c->lib);
  NCCL_TRY(c, g.start());
  for (int step = first; step < W; ++step) {
    const int to = (c->rank + step) % W, from = (c->rank - step + W) % W;
    if (mine > 0) NCCL_TRY(c, c->lib->Send(src, mine * words, ncclUint32, to, c->nccl, c->stream));
    if (n_records[from] > 0)
      NCCL_TRY(c, c->lib->Recv(k.d_gathered + (size_t)from * region_bytes, n_records[from] * words, ncclUint32, from,
                               c->nccl, c->stream));
  }
  NCCL_TRY(c, g.end());
  return CUSIFT_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// communicator
// ------------------------------------------------------------------------------------------------
extern "C" int cusift_comm_use_library(const char *path) {
  std::lock_guard<std::mutex> lock(g_lib_mu);
  g_lib_choice = path ? path : "";
  return CUSIFT_OK;
}

extern "C" int cusift_comm_get_unique_id(char id[CUSIFT_UNIQUE_ID_BYTES]) {
  if (!id) return cusift_fail(CUSIFT_ERR_INVALID, "id is NULL");
  Rccl *lib = nullptr;
  TRY(load_rccl(&lib));
  ncclUniqueId u;
  ncclResult_t r = lib->GetUniqueId(&u);
  if (r != ncclSuccess) return cusift_fail(CUSIFT_ERR_HIP, "ncclGetUniqueId failed: %s", lib->GetErrorString(r));
  memcpy(id, u.internal, CUSIFT_UNIQUE_ID_BYTES);
  return CUSIFT_OK;
}

extern "C" int cusift_comm_create(cusift_comm **out, cusift_ctx *ctx, const char id[CUSIFT_UNIQUE_ID_BYTES], int rank,
                                  int world) {
  if (!out) return cusift_fail(CUSIFT_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (!ctx || !id) return cusift_fail(CUSIFT_ERR_INVALID, "ctx / id is NULL");
  if (world < 1 || rank < 0 || rank >= world) return cusift_fail(CUSIFT_ERR_INVALID, "bad rank %d of %d", rank, world);
  Rccl *lib = nullptr;
  TRY(load_rccl(&lib));
  const int device = cusift_ctx_device(ctx);
  HIP_TRY(hipSetDevice(device));
  std::unique_ptr<cusift_comm> c(new cusift_comm());
  c->ctx = ctx;
  c->stream = (hipStream_t)cusift_ctx_stream(ctx);
  c->device = device;
  c->rank = rank;
  c->world = world;
  c->lib = lib;
  ncclUniqueId u;
  memcpy(u.internal, id, CUSIFT_UNIQUE_ID_BYTES);
  ncclResult_t r = lib->CommInitRank(&c->nccl, world, u, rank);
  if (r != ncclSuccess)
    return cusift_fail(CUSIFT_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, lib->GetErrorString(r));
  *out = c.release();
  return CUSIFT_OK;
}

extern "C" int cusift_comm_destroy(cusift_comm *c) {
  if (!c) return CUSIFT_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->nccl) (void)c->lib->CommDestroy(c->nccl);
  free_tickets(c);
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

extern "C" cusift_ctx *cusift_comm_ctx(cusift_comm *c) { return c ? c->ctx : nullptr; }

extern "C" int cusift_comm_info(cusift_comm *c, int *lib_ranks, int *lib_rank, int *lib_version) {
  if (!c) return cusift_fail(CUSIFT_ERR_INVALID, "comm is NULL");
  int n = -1, r = -1, v = -1;
  if (c->lib->CommCount && c->lib->CommCount(c->nccl, &n) != ncclSuccess) n = -1;
  if (c->lib->CommUserRank && c->lib->CommUserRank(c->nccl, &r) != ncclSuccess) r = -1;
  if (c->lib->GetVersion && c->lib->GetVersion(&v) != ncclSuccess) v = -1;
  if (lib_ranks) *lib_ranks = n;
  if (lib_rank) *lib_rank = r;
  if (lib_version) *lib_version = v;
  return CUSIFT_OK;
}

extern "C" int cusift_comm_set_self_p2p(cusift_comm *c, int on) {
  if (!c) return cusift_fail(CUSIFT_ERR_INVALID, "comm is NULL");
  if (c->pending) return cusift_fail(CUSIFT_ERR_INVALID, "comm: exchanges in flight");
  c->self_p2p = on != 0;
  return CUSIFT_OK;
}

extern "C" int cusift_comm_set_fixed_size(cusift_comm *c, int on) {
  if (!c) return cusift_fail(CUSIFT_ERR_INVALID, "comm is NULL");
  if (c->pending) return cusift_fail(CUSIFT_ERR_INVALID, "comm: exchanges in flight");
  c->fixed = on != 0;
  return CUSIFT_OK;
}

extern "C" int cusift_comm_set_wire_format(cusift_comm *c, int format) {
  if (!c) return cusift_fail(CUSIFT_ERR_INVALID, "comm is NULL");
  if (c->pending) return cusift_fail(CUSIFT_ERR_INVALID, "comm: exchanges in flight");
  if (format < 0 || format > 2) return cusift_fail(CUSIFT_ERR_INVALID, "wire format: 0 exact, 1 compact, 2 trimmed");
  c->compact = format;
  return CUSIFT_OK;
}

extern "C" int cusift_comm_reserve(cusift_comm *c, int n_images_max, int tickets, size_t stage_records) {
  TRY(comm_enter(c));
  if (n_images_max < 1 || n_images_max > cusift::kMaxFlatImages || tickets < 1 || tickets > 64)
    return cusift_fail(CUSIFT_ERR_INVALID, "comm_reserve: need 1 <= n_images_max <= %d and 1 <= tickets <= 64",
                       cusift::kMaxFlatImages);
  TRY(reserve_tickets(c, n_images_max, tickets));
  if (c->self_p2p && stage_records) TRY(reserve_stage(c, stage_records));
  return CUSIFT_OK;
}

extern "C" unsigned long long cusift_comm_host_waits(cusift_comm *c) { return c ? c->host_waits : 0; }
extern "C" double cusift_comm_host_wait_ms(cusift_comm *c) { return c ? c->host_wait_ms : 0.0; }
extern "C" unsigned long long cusift_comm_hip_syncs(cusift_comm *c) { return c ? c->hip_syncs : 0; }

extern "C" const char *cusift_comm_library(void) {
  std::lock_guard<std::mutex> lock(g_lib_mu);
  static thread_local std::string copy;
  copy = g_lib_path;
  return copy.c_str();
}

// ------------------------------------------------------------------------------------------------
// all-gatherv of SiftData
// ------------------------------------------------------------------------------------------------
extern "C" int cusift_allgatherv_begin(cusift_comm *c, cusift_ctx *producer, const cusift_point *d_points,
                                       const unsigned int *d_counters, int n_images, int max_pts, int n_images_max,
                                       void *d_gathered, size_t region_cap) {
  TRY(comm_enter(c));
  if (c->failed)
    return cusift_fail(CUSIFT_ERR_HIP, "allgatherv: an earlier exchange of this communicator timed out; destroy it");
  if ((n_images > 0 && (!d_points || !d_counters)) || !d_gathered)
    return cusift_fail(CUSIFT_ERR_INVALID, "allgatherv: missing data");
  if (n_images < 0 || n_images > cusift::kMaxFlatImages || max_pts < 1 || n_images_max < std::max(1, n_images) ||
      n_images_max > cusift::kMaxFlatImages || region_cap < 1)
    return cusift_fail(CUSIFT_ERR_INVALID,
                       "allgatherv: need 0 <= n_images <= n_images_max <= %d, max_pts >= 1, region_cap >= 1",
                       cusift::kMaxFlatImages);
  if (region_cap * kWordsPerPoint > (size_t)0x7fffffff * 4)
    return cusift_fail(CUSIFT_ERR_INVALID, "allgatherv: region_cap too large");
  // everything that can allocate (and therefore synchronise) happens here, before anything is enqueued; a caller that
  // called cusift_comm_reserve() never gets past these two ifs
  if (n_images_max > c->n_slots || c->depth == 0) TRY(reserve_tickets(c, n_images_max, std::max(c->depth, 4)));
  if (c->self_p2p) TRY(reserve_stage(c, region_cap));
  if (c->pending == c->depth)
    return cusift_fail(CUSIFT_ERR_INVALID, "allgatherv: %d exchanges in flight already (cusift_comm_reserve sets the depth)",
                       c->depth);
  // the records are produced on another stream: order this exchange after everything enqueued there so far
  if (producer) TRY(cusift_ctx_wait(c->ctx, producer));
  const int slot = (c->head + c->pending) % c->depth;
  GatherTicket &k = c->tickets[slot];
  k.stage = c->self_p2p ? (const char *)(c->d_stage + (size_t)slot * c->stage_cap) : nullptr;
  k.seq = c->next_seq++;
  k.rec_bytes = c->compact == 1 ? sizeof(cusift_compact_point)
                                : (c->compact == 2 ? sizeof(cusift_trimmed_point) : sizeof(cusift_point));
  k.d_gathered = (char *)d_gathered;
  k.region_cap = region_cap;
  k.slots = n_images_max;
  hipLaunchKernelGGL(clamp_counts_kernel, dim3((n_images_max + 255) / 256), dim3(256), 0, c->stream, d_counters,
                     n_images, max_pts, k.d_local, n_images_max);
  HIP_TRY(hipGetLastError());
  // the local shard is packed straight into its region of the gathered buffer and sent from there: once this has run
  // the caller's d_points / d_counters are free again
  if (n_images > 0) {
    char *dst = c->self_p2p ? const_cast<char *>(k.stage) : k.d_gathered + (size_t)c->rank * region_cap * k.rec_bytes;
    if (c->compact == 1)
      TRY(cusift_pack_points_compact(c->ctx, d_points, d_counters, n_images, max_pts, (cusift_compact_point *)dst,
                                     region_cap, nullptr));
    else if (c->compact == 2)
      TRY(cusift_pack_points_trimmed(c->ctx, d_points, d_counters, n_images, max_pts, (cusift_trimmed_point *)dst,
                                     region_cap, nullptr));
    else
      TRY(cusift_pack_points(c->ctx, d_points, d_counters, n_images, max_pts, (cusift_point *)dst, region_cap, nullptr));
  }
  NCCL_TRY(c, c->lib->AllGather(k.d_local, k.d_all, (size_t)n_images_max, ncclUint32, c->nccl, c->stream));
  hipLaunchKernelGGL(publish_counts_kernel, dim3(1), dim3(256), 0, c->stream, k.d_all, n_images_max * c->world,
                     k.dev_h_all, k.dev_h_seq, k.seq);
  HIP_TRY(hipGetLastError());
  c->pending++;
  if (c->fixed) {
    std::vector<size_t> n((size_t)c->world, region_cap);
    TRY(post_shards(c, k, n.data()));
  }
  return CUSIFT_OK;
}

extern "C" int cusift_allgatherv_finish(cusift_comm *c, unsigned int *h_counts, size_t *h_totals) {
  TRY(comm_enter(c));
  if (c->failed)
    return cusift_fail(CUSIFT_ERR_HIP, "allgatherv: an earlier exchange of this communicator timed out; destroy it");
  if (!c->pending) return cusift_fail(CUSIFT_ERR_INVALID, "allgatherv: finish() without begin()");
  GatherTicket &k = c->tickets[c->head];
  // The sizes of ncclSend/ncclRecv are host arguments, so the counts are READ here; they are not WAITED for when the
  // caller keeps a step or more of other work between begin() and finish() (bench.py: as many as it has streams) --
  // the flag was set long ago.  No HIP call is involved either way.
  const volatile unsigned int *seq = k.h_seq;
  if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) != k.seq) {
    c->host_waits++;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long spins = 0;
    while (__atomic_load_n(seq, __ATOMIC_ACQUIRE) != k.seq) {
      if ((++spins & 1023) == 0) {
        sched_yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
          // The ticket is NOT recycled: its clamp / AllGather / publish kernels may still be queued and would write
          // under a later begin().  The communicator is dead from here on (every later begin / finish fails); the
          // caller destroys it, which waits for the stream.
          c->failed = true;
          return cusift_fail(CUSIFT_ERR_HIP, "allgatherv: the gathered counts never arrived (a rank missing from the "
                                             "exchange, or the device is hung); the communicator must be destroyed");
        }
      }
    }
    c->host_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  c->head = (c->head + 1) % c->depth;
  c->pending--;
  const int W = c->world, S = k.slots;
  std::vector<size_t> n((size_t)W, 0);
  bool fits = true;
  for (int r = 0; r < W; ++r) {
    for (int i = 0; i < S; ++i) n[r] += k.h_all[(size_t)r * S + i];
    fits = fits && n[r] <= k.region_cap;
  }
  if (h_counts) memcpy(h_counts, k.h_all, sizeof(unsigned int) * (size_t)W * S);
  if (h_totals) memcpy(h_totals, n.data(), sizeof(size_t) * (size_t)W);
  // every rank sees the same counts, so every rank takes this exit together: nothing is left half posted
  if (!fits)
    return cusift_fail(CUSIFT_ERR_NOMEM, "allgatherv: a rank gathered more records than a region of d_gathered holds (%zu)",
                       k.region_cap);
  if (c->fixed) return CUSIFT_OK;  // the regions travelled whole, posted by begin()
  return post_shards(c, k, n.data());
}

extern "C" int cusift_allgatherv(cusift_comm *c, cusift_ctx *producer, const cusift_point *d_points,
                                 const unsigned int *d_counters, int n_images, int max_pts, int n_images_max,
                                 void *d_gathered, size_t region_cap, unsigned int *h_counts, size_t *h_totals) {
  TRY(cusift_allgatherv_begin(c, producer, d_points, d_counters, n_images, max_pts, n_images_max, d_gathered,
                              region_cap));
  return cusift_allgatherv_finish(c, h_counts, h_totals);
}

// trimmed regions -> SiftPoint regions, all ranks in one launch: block (x, r) walks region r's records x, x + gridDim.x, ...
struct RegionTotals {
  unsigned int n[64];
};
__global__ void __launch_bounds__(64) expand_regions_kernel(const cusift_trimmed_point *__restrict__ trimmed,
                                                           size_t region_cap, RegionTotals totals, int first_rank,
                                                           cusift_point *__restrict__ points) {
  const int lane = threadIdx.x;
  const int rk = blockIdx.y;
  const size_t n = totals.n[rk];
  constexpr int kData = (int)(offsetof(cusift_point, data) / 4), kSub = (int)(offsetof(cusift_point, subsampling) / 4);
  constexpr int kDwords = sizeof(cusift_point) / 4;
  const size_t base = (size_t)(first_rank + rk) * region_cap;
  for (size_t g = blockIdx.x; g < n; g += gridDim.x) {
    const unsigned int *src = reinterpret_cast<const unsigned int *>(trimmed + base + g);
    unsigned int *dst = reinterpret_cast<unsigned int *>(points + base + g);
    dst[kData + lane] = src[7 + lane];
    dst[kData + 64 + lane] = src[7 + 64 + lane];
    if (lane < kData) dst[lane] = lane < 6 ? src[lane] : (lane == kSub ? src[6] : 0u);
    if (lane < kDwords - kData - 128) dst[kData + 128 + lane] = 0u;  // coords3D
  }
}

extern "C" int cusift_expand_gathered(cusift_comm *c, const cusift_trimmed_point *d_gathered, size_t region_cap,
                                      const size_t *h_totals, cusift_point *d_points) {
  if (!c) return cusift_fail(CUSIFT_ERR_INVALID, "comm is NULL");
  if (!d_gathered || !h_totals || !d_points) return cusift_fail(CUSIFT_ERR_INVALID, "expand (gathered): NULL argument");
  HIP_TRY(hipSetDevice(c->device));
  for (int first = 0; first < c->world; first += 64) {
    RegionTotals t;
    memset(&t, 0, sizeof(t));
    const int nr = std::min(64, c->world - first);
    size_t most = 0;
    for (int r = 0; r < nr; ++r) {
      if (h_totals[first + r] > region_cap)
        return cusift_fail(CUSIFT_ERR_INVALID, "expand (gathered): rank %d holds %zu records, regions hold %zu", first + r,
                           h_totals[first + r], region_cap);
      t.n[r] = (unsigned int)h_totals[first + r];
      most = std::max(most, h_totals[first + r]);
    }
    if (most == 0) continue;
    dim3 grid((unsigned int)std::min<size_t>(most, 2048), nr);
    hipLaunchKernelGGL(expand_regions_kernel, grid, dim3(64), 0, c->stream, d_gathered, region_cap, t, first, d_points);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cusift_fail(CUSIFT_ERR_HIP, "expand (gathered) launch failed: %s", hipGetErrorString(e));
  }
  return CUSIFT_OK;
}

extern "C" int cusift_compact_gathered(cusift_ctx *ctx, const cusift_point *d_gathered, size_t region_cap, int world,
                                       const size_t *h_totals, cusift_point *d_out, size_t capacity) {
  if (!ctx || !d_gathered || !h_totals || !d_out || world < 1)
    return cusift_fail(CUSIFT_ERR_INVALID, "compact_gathered: bad argument");
  HIP_TRY(hipSetDevice(cusift_ctx_device(ctx)));
  size_t at = 0;
  for (int r = 0; r < world; ++r) {
    if (h_totals[r] > region_cap || at + h_totals[r] > capacity)
      return cusift_fail(CUSIFT_ERR_NOMEM, "compact_gathered: rank %d holds %zu records (region %zu, room %zu)", r,
                         h_totals[r], region_cap, capacity - at);
    if (h_totals[r])
      HIP_TRY(hipMemcpyAsync(d_out + at, d_gathered + (size_t)r * region_cap, sizeof(cusift_point) * h_totals[r],
                             hipMemcpyDeviceToDevice, (hipStream_t)cusift_ctx_stream(ctx)));
    at += h_totals[r];
  }
  return CUSIFT_OK;
}

// ------------------------------------------------------------------------------------------------
// rows of a pitched float image between ranks: halo exchange of the strip tiling, and gather/scatter of bands
// ------------------------------------------------------------------------------------------------
extern "C" int cusift_exchange_rows(cusift_comm *c, float *d_band, int pitch, int band_rows, int n_ops, const int *peers,
                                    const int *send_row, const int *send_rows, const int *recv_row,
                                    const int *recv_rows) {
  TRY(comm_enter(c));
  if (n_ops < 0 || (n_ops > 0 && (!d_band || !peers || !send_row || !send_rows || !recv_row || !recv_rows)) || pitch < 1 ||
      band_rows < 0)
    return cusift_fail(CUSIFT_ERR_INVALID, "exchange_rows: bad argument");
  bool any = false;
  for (int i = 0; i < n_ops; ++i) {
    if (peers[i] < 0 || peers[i] >= c->world || send_rows[i] < 0 || recv_rows[i] < 0 || send_row[i] < 0 ||
        recv_row[i] < 0)
      return cusift_fail(CUSIFT_ERR_INVALID, "exchange_rows: op %d: bad peer or rows", i);
    // the rows are read / written by RCCL on the device: a range outside the band is a memory fault, not an error code
    if ((long)send_row[i] + send_rows[i] > band_rows || (long)recv_row[i] + recv_rows[i] > band_rows)
      return cusift_fail(CUSIFT_ERR_INVALID, "exchange_rows: op %d: rows [%d, %d) sent / [%d, %d) received lie outside the "
                                             "band's %d rows", i, send_row[i], send_row[i] + send_rows[i], recv_row[i],
                         recv_row[i] + recv_rows[i], band_rows);
    if (peers[i] == c->rank && !c->self_p2p)
      return cusift_fail(CUSIFT_ERR_INVALID, "exchange_rows: op %d addresses this rank", i);
    any = any || send_rows[i] > 0 || recv_rows[i] > 0;
  }
  if (!any) return CUSIFT_OK;
  GroupScope g(c->lib);
  NCCL_TRY(c, g.start());
  for (int i = 0; i < n_ops; ++i) {
    if (send_rows[i] > 0)
      NCCL_TRY(c, c->lib->Send(d_band + (size_t)send_row[i] * pitch, (size_t)send_rows[i] * pitch, ncclFloat32, peers[i],
                               c->nccl, c->stream));
    if (recv_rows[i] > 0)
      NCCL_TRY(c, c->lib->Recv(d_band + (size_t)recv_row[i] * pitch, (size_t)recv_rows[i] * pitch, ncclFloat32, peers[i],
                               c->nccl, c->stream));
  }
  NCCL_TRY(c, g.end());
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
  return cusift_exchange_rows(c, d_band, pitch, top_halo + own_rows + bottom_halo, n, peers, srow, srows, rrow, rrows);
}
