// fake_rccl.cpp -- TEST-ONLY stand-in for librccl: the nine entry points cusift_amd/csrc/sift_comm.hip binds, for W ranks
// that are W THREADS of one process sharing one GPU (real RCCL refuses two ranks per device, so the world > 1 code of
// sift_comm.hip / sift_tiled.hip could otherwise only ever run on an 8-GPU node).  Selected per communicator with
// cusift_comm_use_library("tests/fake_rccl/libfake_rccl.so") or $CUSIFT_RCCL_LIB.  Not part of the product: nothing
// under cusift_amd/ or include/ refers to it.
//
// Semantics kept from NCCL: communicators of one unique id rendezvous in ncclCommInitRank; ncclSend/ncclRecv are matched
// per (source, destination) pair in posting order and must agree in size; operations between ncclGroupStart and
// ncclGroupEnd are issued together, so a rank may post its sends and receives in any order without deadlock; all data
// movement is asynchronous on the stream given to the call and ordered against both ranks' streams with events.
// Differences (all on the safe side for a test): ncclGroupEnd BLOCKS the calling thread until every peer has posted the
// matching operation (120 s, then ncclSystemError), and size mismatches are reported as ncclInvalidArgument instead of
// hanging or corrupting memory.  fake_rccl_stats() exposes a call log for assertions.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <stdio.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Msg {
  const void *ptr;
  size_t bytes;
  int device;
  hipEvent_t ready = nullptr;  // recorded on the sender's stream when the message was posted
  hipEvent_t done = nullptr;   // recorded on the receiver's stream after its copy
  bool consumed = false;
  bool mismatch = false;
};

struct World {
  int n = 0, joined = 0, left = 0;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::deque<std::shared_ptr<Msg>>> box;  // [src * n + dst], FIFO
};

std::mutex g_mu;
std::map<std::string, std::shared_ptr<World>> g_worlds;
std::atomic<unsigned long long> g_ids{1};
std::atomic<unsigned long long> g_stats[8];  // groups, sends, recvs, allgathers, bytes sent, mismatches, timeouts, comms

enum { kGroups, kSends, kRecvs, kAllGathers, kBytes, kMismatch, kTimeouts, kComms };

struct Op {
  bool send;
  const void *src;
  void *dst;
  size_t bytes;
  int peer;
  ncclComm *comm;
  hipStream_t stream;
};

thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

const auto kPatience = std::chrono::seconds(120);

size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

}  // namespace

struct ncclComm {
  std::shared_ptr<World> w;
  int rank = 0;
  std::string key;
};

namespace {

ncclResult_t run_ops(std::vector<Op> &ops) {
  if (ops.empty()) return ncclSuccess;
  g_stats[kGroups]++;
  ncclResult_t rc = ncclSuccess;
  struct Sent {
    std::shared_ptr<Msg> m;
    hipStream_t stream;
    World *w;
  };
  std::vector<Sent> mine;
  // 1. post every send
  for (Op &o : ops) {
    if (!o.send) continue;
    World &w = *o.comm->w;
    auto m = std::make_shared<Msg>();
    m->ptr = o.src;
    m->bytes = o.bytes;
    (void)hipGetDevice(&m->device);
    if (hipEventCreateWithFlags(&m->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->done, hipEventDisableTiming) != hipSuccess ||
        hipEventRecord(m->ready, o.stream) != hipSuccess) {
      ops.clear();
      return ncclUnhandledCudaError;
    }
    {
      std::lock_guard<std::mutex> lock(w.mu);
      w.box[(size_t)o.comm->rank * w.n + o.peer].push_back(m);
    }
    w.cv.notify_all();
    mine.push_back(Sent{m, o.stream, &w});
    g_stats[kSends]++;
    g_stats[kBytes] += o.bytes;
  }
  // 2. take every receive, in posting order per source
  for (Op &o : ops) {
    if (o.send) continue;
    World &w = *o.comm->w;
    std::shared_ptr<Msg> m;
    {
      std::unique_lock<std::mutex> lock(w.mu);
      auto &q = w.box[(size_t)o.peer * w.n + o.comm->rank];
      if (!w.cv.wait_for(lock, kPatience, [&] { return !q.empty(); })) {
        g_stats[kTimeouts]++;
        fprintf(stderr, "fake_rccl: rank %d waited 120 s for a send from rank %d (%zu bytes)\n", o.comm->rank, o.peer,
                o.bytes);
        rc = ncclSystemError;
        continue;
      }
      m = q.front();
      q.pop_front();
    }
    g_stats[kRecvs]++;
    if (m->bytes != o.bytes) {
      g_stats[kMismatch]++;
      fprintf(stderr, "fake_rccl: rank %d expects %zu bytes from rank %d, which sent %zu\n", o.comm->rank, o.bytes,
              o.peer, m->bytes);
      m->mismatch = true;
      rc = ncclInvalidArgument;
    } else if (hipStreamWaitEvent(o.stream, m->ready, 0) != hipSuccess ||
               hipMemcpyAsync(o.dst, m->ptr, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess) {
      rc = ncclUnhandledCudaError;
    }
    (void)hipEventRecord(m->done, o.stream);
    {
      std::lock_guard<std::mutex> lock(w.mu);
      m->consumed = true;
    }
    w.cv.notify_all();
  }
  // 3. my send buffers may be rewritten once the receivers' copies have run
  for (Sent &p : mine) {
    std::shared_ptr<Msg> &m = p.m;
    World *w = p.w;
    std::unique_lock<std::mutex> lock(w->mu);
    if (!w->cv.wait_for(lock, kPatience, [&] { return m->consumed; })) {
      g_stats[kTimeouts]++;
      fprintf(stderr, "fake_rccl: a send of %zu bytes was never received\n", m->bytes);
      rc = ncclSystemError;
      continue;  // the events leak: the message may still be in a mailbox
    }
    lock.unlock();
    if (m->mismatch) rc = ncclInvalidArgument;
    (void)hipStreamWaitEvent(p.stream, m->done, 0);
    (void)hipEventDestroy(m->ready);  // destruction is deferred until the event has completed
    (void)hipEventDestroy(m->done);
  }
  ops.clear();
  return rc;
}

ncclResult_t submit(Op op) {
  if (!op.comm || op.peer < 0 || op.peer >= op.comm->w->n) return ncclInvalidArgument;
  t_ops.push_back(op);
  if (t_depth > 0) return ncclSuccess;
  return run_ops(t_ops);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  if (!id) return ncclInvalidArgument;
  memset(id->internal, 0, NCCL_UNIQUE_ID_BYTES);
  snprintf(id->internal, NCCL_UNIQUE_ID_BYTES, "fake-rccl-%llu-%p", (unsigned long long)g_ids++, (void *)id);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  const std::string key(id.internal, NCCL_UNIQUE_ID_BYTES);
  std::shared_ptr<World> w;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    auto &slot = g_worlds[key];
    if (!slot) {
      slot = std::make_shared<World>();
      slot->n = nranks;
      slot->box.resize((size_t)nranks * nranks);
    }
    w = slot;
  }
  if (w->n != nranks) return ncclInvalidArgument;
  {
    std::unique_lock<std::mutex> lock(w->mu);
    w->joined++;
    w->cv.notify_all();
    if (!w->cv.wait_for(lock, kPatience, [&] { return w->joined >= w->n; })) {
      g_stats[kTimeouts]++;
      fprintf(stderr, "fake_rccl: only %d of %d ranks joined the communicator\n", w->joined, w->n);
      return ncclSystemError;
    }
  }
  ncclComm *c = new ncclComm();
  c->w = w;
  c->rank = rank;
  c->key = key;
  *comm = c;
  g_stats[kComms]++;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  if (!comm) return ncclSuccess;
  bool last;
  {
    std::lock_guard<std::mutex> lock(comm->w->mu);
    last = ++comm->w->left == comm->w->n;
  }
  if (last) {
    std::lock_guard<std::mutex> lock(g_mu);
    g_worlds.erase(comm->key);
  }
  delete comm;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
  ++t_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (t_depth <= 0) return ncclInvalidUsage;
  if (--t_depth > 0) return ncclSuccess;
  return run_ops(t_ops);
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm,
                      hipStream_t stream) {
  const size_t b = type_bytes(datatype);
  if (!b || !sendbuff) return ncclInvalidArgument;
  return submit(Op{true, sendbuff, nullptr, count * b, peer, comm, stream});
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm,
                      hipStream_t stream) {
  const size_t b = type_bytes(datatype);
  if (!b || !recvbuff) return ncclInvalidArgument;
  return submit(Op{false, nullptr, recvbuff, count * b, peer, comm, stream});
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype,
                           ncclComm_t comm, hipStream_t stream) {
  const size_t b = type_bytes(datatype);
  if (!b || !sendbuff || !recvbuff || !comm) return ncclInvalidArgument;
  g_stats[kAllGathers]++;
  ++t_depth;
  for (int r = 0; r < comm->w->n; ++r) {
    t_ops.push_back(Op{true, sendbuff, nullptr, sendcount * b, r, comm, stream});
    t_ops.push_back(Op{false, nullptr, (char *)recvbuff + (size_t)r * sendcount * b, sendcount * b, r, comm, stream});
  }
  return ncclGroupEnd();
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
  if (!comm || !count) return ncclInvalidArgument;
  *count = comm->w->n;
  return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank) {
  if (!comm || !rank) return ncclInvalidArgument;
  *rank = comm->rank;
  return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "success";
    case ncclUnhandledCudaError: return "fake_rccl: a HIP call failed";
    case ncclSystemError: return "fake_rccl: a peer never arrived (see stderr)";
    case ncclInvalidArgument: return "fake_rccl: invalid argument / send and receive sizes differ (see stderr)";
    case ncclInvalidUsage: return "fake_rccl: invalid usage";
    default: return "fake_rccl: error";
  }
}

// groups, sends, recvs, allgathers, bytes sent, size mismatches, timeouts, communicators -- since the library was loaded
void fake_rccl_stats(unsigned long long out[8]) {
  for (int i = 0; i < 8; ++i) out[i] = g_stats[i].load();
}

}  // extern "C"
