// One rank of the multi-GPU benchmark step, timed, with no Python and no torch anywhere: BASELINE configs[3] (512 x 1080p
// sharded 64 per GPU over 8 x MI355X, all-gatherv of SiftData every step) written against include/cusift_amd.h only --
// plain g++, no HIP / RCCL headers.  The scaling curve then does not depend on bench.py's plumbing: same step, same
// exchange (cusift_allgatherv_begin / _finish: counts all-gather + ONE ncclGroup of ncclSend/ncclRecv, trimmed 540-byte
// records expanded on arrival to the 588-byte SiftPoint records of the reference, cuSIFT.h:10-30), same JSON line.
//
//   scaling_bench <rank> <world> <id-file> <gray1.pgm> [steps=20] [warmup=5] [images=64] [width=1920] [height=1080] [streams=3]
//
// Launch one process per GPU (a shell loop, mpirun, a job scheduler ...): rank 0 writes the communicator id to
// <id-file>, the others wait for it; rank r runs on GPU r % device_count.  W untimed warm-up steps, then exactly K timed
// steps between two barriers (an empty all-gatherv + a stream synchronisation on every rank), the MAX over the ranks'
// times (through <id-file>.t<rank>), and rank 0 prints ONE JSON line on stdout.  With world == 1 the shard still
// travels through ncclSend / ncclRecv (self p2p): that is how tests/test_cpp_dropin.py runs it on one GPU.
// (No pre-flight here: the device leaves its idle clocks during the first ~30 steps -- bench.py's docstring -- so pass a
// warm-up of 40 or more to time the steady state.)
// Images: the fixture mirror-tiled with a per-image shift, low-passed to sigma 1.0 and re-quantised -- the SHAPE of
// cusift_amd.synth.tile, not bit for bit (this program measures; tests/ verify).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

#include "cusift_amd.h"
#include "cusift_amd_multigpu.h"

#define CHECK(call)                                                                 \
  do {                                                                              \
    int rc_ = (call);                                                               \
    if (rc_ != CUSIFT_OK) {                                                         \
      std::fprintf(stderr, "rank %d: %s failed: %s\n", g_rank, #call, cusift_last_error()); \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

static int g_rank = 0;

static bool read_pgm(const char *path, std::vector<float> &img, int &w, int &h) {
  FILE *fp = std::fopen(path, "rb");
  if (!fp) return false;
  int maxv = 0;
  if (std::fscanf(fp, "P5 %d %d %d", &w, &h, &maxv) != 3 || maxv != 255) return false;
  std::fgetc(fp);
  std::vector<unsigned char> raw((size_t)w * h);
  if (std::fread(raw.data(), 1, raw.size(), fp) != raw.size()) return false;
  std::fclose(fp);
  img.assign(raw.begin(), raw.end());
  return true;
}

// image g of the job: mirror-tiled fixture, gain 255/144, shifted, blurred to sigma 1.0 (radius 4, replicated borders), 8 bit
static void make_image(const std::vector<float> &base, int bw, int bh, int g, int W, int H, float *out) {
  float k[9], ksum = 0.0f;
  for (int i = 0; i < 9; ++i) ksum += (k[i] = std::exp(-(float)((i - 4) * (i - 4)) / 2.0f));
  for (int i = 0; i < 9; ++i) k[i] /= ksum;
  std::vector<float> raw((size_t)W * H), tmp((size_t)W * H);
  auto cl = [](int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); };
  const int sx = (g * 37) % bw, sy = (g * 91) % bh;
  for (int y = 0; y < H; ++y) {
    int yy = (y + sy) % (2 * bh);
    if (yy >= bh) yy = 2 * bh - 1 - yy;
    for (int x = 0; x < W; ++x) {
      int xx = (x + sx) % (2 * bw);
      if (xx >= bw) xx = 2 * bw - 1 - xx;
      raw[(size_t)y * W + x] = base[(size_t)yy * bw + xx] * (255.0f / 144.0f);
    }
  }
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      float a = 0.0f;
      for (int t = -4; t <= 4; ++t) a += k[t + 4] * raw[(size_t)y * W + cl(x + t, W)];
      tmp[(size_t)y * W + x] = a;
    }
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      float a = 0.0f;
      for (int t = -4; t <= 4; ++t) a += k[t + 4] * tmp[(size_t)cl(y + t, H) * W + x];
      a = std::nearbyint(a);
      out[(size_t)y * W + x] = a < 0.0f ? 0.0f : (a > 255.0f ? 255.0f : a);
    }
}

int main(int argc, char **argv) {
  if (argc < 5) {
    std::printf("usage: %s rank world id-file gray1.pgm [steps] [warmup] [images] [width] [height] [streams]\n", argv[0]);
    return 2;
  }
  const int rank = g_rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
  // Rendezvous files: <id-file> (rank 0 writes the communicator id) and <id-file>.t<rank> (every rank's time).  A crashed
  // run leaves them behind, so (a) a launcher may set CUSIFT_RUN_NONCE (any string, the same for all ranks of one run): it
  // becomes part of the names; (b) rank 0 removes whatever is there before it writes; (c) the other ranks only accept an id
  // file written within the last 30 s before their own start, or after it -- the ranks of one run start together.
  const char *nonce = std::getenv("CUSIFT_RUN_NONCE");
  const std::string id_file = std::string(argv[3]) + (nonce && *nonce ? std::string(".") + nonce : std::string());
  const time_t started = time(nullptr);
  const int K = argc > 5 ? std::atoi(argv[5]) : 20, WARM = argc > 6 ? std::atoi(argv[6]) : 5;
  const int B = argc > 7 ? std::atoi(argv[7]) : 64, W = argc > 8 ? std::atoi(argv[8]) : 1920,
            H = argc > 9 ? std::atoi(argv[9]) : 1080;
  // three extraction streams + the exchange stream = four busy streams, one per pipe of the command processor (bench.py)
  const int E = argc > 10 ? std::atoi(argv[10]) : 3;
  std::vector<float> base;
  int bw = 0, bh = 0;
  if (!read_pgm(argv[4], base, bw, bh) || world < 1 || rank < 0 || rank >= world || K < 1 || B < 1 || E < 1 || W % 4) return 2;

  // Rank 0 prints exactly ONE line on stdout.  Libraries write there too (RCCL prints a banner when a communicator is
  // created), so from here on file descriptor 1 is stderr and the JSON line goes to a duplicate of the original stdout.
  std::fflush(stdout);
  const int json_fd = dup(1);
  dup2(2, 1);

  int n_dev = 0;
  CHECK(cusift_device_count(&n_dev));
  if (n_dev < 1) return 1;
  const int device = rank % n_dev;
  CHECK(cusift_init(device));

  // ---- the communicator on a stream of its own: rank 0 makes the id, everybody reads it ----
  char id[CUSIFT_UNIQUE_ID_BYTES];
  if (rank == 0) {
    std::remove(id_file.c_str());
    for (int r = 0; r < world; ++r) std::remove((id_file + ".t" + std::to_string(r)).c_str());
    CHECK(cusift_comm_get_unique_id(id));
    const std::string tmp = id_file + ".tmp";
    FILE *fp = std::fopen(tmp.c_str(), "wb");
    if (!fp || std::fwrite(id, 1, sizeof(id), fp) != sizeof(id)) return 1;
    std::fclose(fp);
    std::rename(tmp.c_str(), id_file.c_str());
  } else {
    FILE *fp = nullptr;
    for (int tries = 0; tries < 1200 && !fp; ++tries) {
      struct stat st;
      if (stat(id_file.c_str(), &st) == 0 && st.st_mtime >= started - 30) fp = std::fopen(id_file.c_str(), "rb");
      if (!fp) std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    if (!fp || std::fread(id, 1, sizeof(id), fp) != sizeof(id)) return 1;
    std::fclose(fp);
  }
  cusift_ctx *cctx = nullptr;
  CHECK(cusift_ctx_create(&cctx, device, nullptr));
  cusift_comm *comm = nullptr;
  CHECK(cusift_comm_create(&comm, cctx, id, rank, world));
  if (world == 1) CHECK(cusift_comm_set_self_p2p(comm, 1));
  CHECK(cusift_comm_set_wire_format(comm, 2));  // records travel as the 135 floats extraction writes (540 B, exact)

  cusift_params prm;
  cusift_default_params(&prm);
  prm.num_octaves = 5;
  prm.init_blur = 1.0f;
  prm.peak_thresh = 3.0f;
  prm.edge_thresh = 10.0f;
  prm.lowest_scale = 0.0f;
  prm.subsampling = 1.0f;
  prm.max_pts = 32768;
  prm.concurrent_batches = E;

  // ---- this rank's shard: images [rank * B, (rank + 1) * B) of the job (contiguous blocks: configs[3]) ----
  const size_t img_floats = (size_t)W * H;
  std::vector<float> host((size_t)B * img_floats);
  for (int i = 0; i < B; ++i) make_image(base, bw, bh, rank * B + i, W, H, &host[(size_t)i * img_floats]);
  float *d_imgs = nullptr;
  CHECK(cusift_malloc((void **)&d_imgs, host.size() * sizeof(float)));
  std::vector<cusift_ctx *> ctx(E, nullptr);
  std::vector<cusift_point *> pts(E, nullptr);
  std::vector<unsigned int *> cnt(E, nullptr);
  for (int e = 0; e < E; ++e) {
    CHECK(cusift_ctx_create(&ctx[e], device, nullptr));
    CHECK(cusift_ctx_reserve(ctx[e], B, W, H, &prm));
    CHECK(cusift_malloc((void **)&pts[e], (size_t)B * prm.max_pts * sizeof(cusift_point)));
    CHECK(cusift_malloc((void **)&cnt[e], (size_t)B * sizeof(unsigned int)));
  }
  CHECK(cusift_memcpy_h2d(ctx[0], d_imgs, host.data(), host.size() * sizeof(float)));
  CHECK(cusift_ctx_synchronize(ctx[0]));

  // ---- gathered SiftData: `world` regions of region_cap records, LAG + 2 buffers deep, trimmed (as they arrive) and
  // expanded (what the step leaves: SiftPoint records of every rank's images) ----
  const int LAG = E, n_out = LAG + 2;
  const size_t region_cap = (size_t)B * 8192;
  std::vector<cusift_trimmed_point *> wire(n_out, nullptr);
  std::vector<cusift_point *> full(n_out, nullptr);
  for (int i = 0; i < n_out; ++i) {
    CHECK(cusift_malloc((void **)&wire[i], sizeof(cusift_trimmed_point) * region_cap * world));
    CHECK(cusift_malloc((void **)&full[i], sizeof(cusift_point) * region_cap * world));
  }
  CHECK(cusift_comm_reserve(comm, B, LAG + 1, region_cap));  // nothing below allocates
  std::vector<unsigned int> all_counts((size_t)world * B);
  std::vector<size_t> totals(world, 0);
  unsigned int *d_zero = nullptr;
  CHECK(cusift_malloc((void **)&d_zero, sizeof(unsigned int) * B));
  CHECK(cusift_memset(cctx, d_zero, 0, sizeof(unsigned int) * B));

  // per extraction slot: an event behind the begin() that packed its records -- the slot's NEXT extraction waits for exactly
  // that (cusift_ctx_wait(ctx[e], cctx) would wait for everything the exchange stream has queued since, i.e. for the pack
  // of the step before, i.e. for that step's extraction: the streams would run one after the other -- 2.06 ms per step
  // measured that way against 1.1 with the events)
  std::vector<cusift_event *> packed(E, nullptr);
  std::vector<char> packed_valid(E, 0);
  for (int e = 0; e < E; ++e) CHECK(cusift_event_create(cctx, &packed[e]));
  long begun = 0, finished = 0;
  size_t gathered_last = 0;
  auto finish_one = [&]() -> int {
    const int slot = (int)(finished % n_out);
    CHECK(cusift_allgatherv_finish(comm, all_counts.data(), totals.data()));
    CHECK(cusift_expand_gathered(comm, wire[slot], region_cap, totals.data(), full[slot]));
    gathered_last = 0;
    for (int r = 0; r < world; ++r) gathered_last += totals[r];
    ++finished;
    return 0;
  };
  auto step = [&](long i) -> int {
    const int e = (int)(i % E);
    if (packed_valid[e]) CHECK(cusift_event_wait(packed[e], ctx[e]));  // the pack of this slot's previous batch has read pts[e] / cnt[e]
    CHECK(cusift_extract_batch(ctx[e], d_imgs, B, W, H, W, img_floats, &prm, pts[e], cnt[e]));
    CHECK(cusift_allgatherv_begin(comm, ctx[e], pts[e], cnt[e], B, prm.max_pts, B, wire[begun % n_out], region_cap));
    CHECK(cusift_event_record(packed[e], cctx));
    packed_valid[e] = 1;
    ++begun;
    if (begun - finished > LAG) return finish_one();
    return 0;
  };
  auto drain = [&]() -> int {
    while (finished < begun)
      if (finish_one()) return 1;
    CHECK(cusift_ctx_synchronize(cctx));
    for (int e = 0; e < E; ++e) CHECK(cusift_ctx_synchronize(ctx[e]));
    return 0;
  };
  // barrier: an empty exchange -- finish() returns once every rank's counts have arrived
  auto barrier = [&]() -> int {
    if (drain()) return 1;
    CHECK(cusift_allgatherv_begin(comm, nullptr, pts[0], d_zero, B, prm.max_pts, B, wire[begun % n_out], region_cap));
    ++begun;
    CHECK(cusift_allgatherv_finish(comm, all_counts.data(), totals.data()));
    ++finished;
    CHECK(cusift_ctx_synchronize(cctx));
    return 0;
  };

  for (long i = 0; i < WARM; ++i)
    if (step(i)) return 1;
  if (barrier()) return 1;
  const auto t0 = std::chrono::steady_clock::now();
  for (long i = 0; i < K; ++i)
    if (step(WARM + i)) return 1;
  if (drain()) return 1;
  const size_t gathered = gathered_last;
  if (barrier()) return 1;  // (its cost -- one empty exchange -- is inside the timed region on every rank alike)
  const double mine_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

  // ---- the ranks' times through the file system; rank 0 takes the maximum ----
  {
    const std::string f = id_file + ".t" + std::to_string(rank), tmp = f + ".tmp";
    FILE *fp = std::fopen(tmp.c_str(), "w");
    if (!fp) return 1;
    std::fprintf(fp, "%.6f\n", mine_ms);
    std::fclose(fp);
    std::rename(tmp.c_str(), f.c_str());
  }
  std::vector<unsigned int> h_cnt(B);
  CHECK(cusift_memcpy_d2h(ctx[0], h_cnt.data(), cnt[0], B * sizeof(unsigned int)));
  CHECK(cusift_ctx_synchronize(ctx[0]));
  long kp_local = 0;
  for (int i = 0; i < B; ++i) kp_local += h_cnt[i] < (unsigned int)prm.max_pts ? h_cnt[i] : prm.max_pts;
  int lib_ranks = -1, lib_rank = -1, lib_version = -1;
  CHECK(cusift_comm_info(comm, &lib_ranks, &lib_rank, &lib_version));

  int rc = 0;
  if (rank == 0) {
    std::vector<double> ms(world, 0.0);
    double worst = 0.0;
    for (int r = 0; r < world; ++r) {
      const std::string f = id_file + ".t" + std::to_string(r);
      FILE *fp = nullptr;
      for (int tries = 0; tries < 600 && !(fp = std::fopen(f.c_str(), "r")); ++tries)
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
      if (!fp || std::fscanf(fp, "%lf", &ms[r]) != 1) return 1;
      std::fclose(fp);
      std::remove(f.c_str());
      worst = ms[r] > worst ? ms[r] : worst;
    }
    const double ms_per_step = worst / K;
    const double total_pix = (double)world * B * W * H;
    std::string by_rank;
    for (int r = 0; r < world; ++r) {
      char b[32];
      std::snprintf(b, sizeof(b), "%s%.4f", r ? ", " : "", ms[r] / K);
      by_rank += b;
    }
    std::fflush(stdout);
    dup2(json_fd, 1);
    std::printf(
        "{\"metric\": \"Mpix/s pyramid + keypoints/s end-to-end, 1920x1080 batch\", \"value\": %.2f, \"unit\": \"Mpix/s\", "
        "\"n_gpus\": %d, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.4f, \"higher_is_better\": true, \"scaling\": "
        "\"weak\", \"vs_baseline\": null, \"dtype\": \"f32\", \"data\": \"synthetic\", \"config\": {\"workload\": \"batch of "
        "%d x %dx%d images per GPU, 5 octaves, initBlur=1.0, thresh=3.0, edge=10, maxPts=32768; full SIFT extraction + "
        "all-gatherv of SiftData every step (C ABI over RCCL: counts all-gather + grouped send/recv, 540-byte trimmed "
        "records expanded on arrival)\", \"program\": \"tests/cpp/scaling_bench.cpp (C++ over the C ABI, no torch)\", "
        "\"images_per_gpu\": %d, \"parallelism\": \"image-sharded x%d\", \"streams_per_gpu\": %d, \"rccl_library\": \"%s\", "
        "\"rccl_ranks\": %d, \"rccl_version\": %d, \"gather_record_bytes\": 540, \"ms_per_step_by_rank\": [%s]}, "
        "\"keypoints_per_step_rank0\": %ld, \"records_gathered_per_step\": %zu}\n",
        total_pix / (ms_per_step * 1e-3) / 1e6, world, K, WARM, ms_per_step, B, W, H, B, world, E, cusift_comm_library(),
        lib_ranks, lib_version, by_rank.c_str(), kp_local, gathered);
    std::fflush(stdout);
    if (gathered < (size_t)kp_local * world / 2 || kp_local < 10) rc = 1;
    std::remove(id_file.c_str());
  }
  std::fprintf(stderr, "rank %d of %d on GPU %d: %.4f ms per step, %ld keypoints per step here, %zu gathered\n", rank, world,
               device, mine_ms / K, kp_local, gathered);

  for (int i = 0; i < n_out; ++i) {
    cusift_free(wire[i]);
    cusift_free(full[i]);
  }
  for (int e = 0; e < E; ++e) {
    cusift_free(pts[e]);
    cusift_free(cnt[e]);
    cusift_ctx_destroy(ctx[e]);
  }
  for (int e = 0; e < E; ++e) cusift_event_destroy(packed[e]);
  cusift_free(d_zero);
  cusift_free(d_imgs);
  cusift_comm_destroy(comm);
  cusift_ctx_destroy(cctx);
  return rc;
}
