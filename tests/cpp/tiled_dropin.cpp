// ONE large image strip-tiled over W ranks through the C ABI alone (BASELINE configs[4]; include/cusift_amd.h
// "one large image strip-tiled over the ranks"): cusift_tiled_create / cusift_tiled_extract / cusift_tiled_check, then
// the all-gatherv of the ranks' SiftData -- and the merged result must be the whole-image extraction, bit for bit.
// Plain g++, no HIP / RCCL headers.
//
//   tiled_dropin <gray1.pgm> <W> <H> <octaves> <world> [transport.so]
//
// world == 1 needs no communicator.  world > 1: the ranks are THREADS of this process, one context each, joined by the
// library named last (tests/fake_rccl/libfake_rccl.so: real RCCL wants one process and one GPU per rank -- there the
// same calls run in `world` processes).  The image is the fixture mirror-tiled to W x H.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "cusift_amd.h"
#include "cusift_amd_multigpu.h"

#define CHECK(call)                                                                    \
  do {                                                                                 \
    int rc_ = (call);                                                                  \
    if (rc_ != CUSIFT_OK) {                                                            \
      std::fprintf(stderr, "%s failed: %s\n", #call, cusift_last_error());             \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

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

struct Shared {
  int W, H, world;
  cusift_params prm;
  const float *d_image;  // the whole W x H image, dense rows, on the device
  char id[CUSIFT_UNIQUE_ID_BYTES];
  size_t region_cap;
  std::vector<std::vector<cusift_point>> merged;  // per rank: the all-gathered SiftData
  std::vector<int> rc;
};

// what ONE rank of a tiled job does
static int rank_main(Shared &S, int rank) {
  cusift_ctx *ctx = nullptr;
  CHECK(cusift_ctx_create(&ctx, 0, nullptr));
  cusift_comm *comm = nullptr;
  if (S.world > 1) CHECK(cusift_comm_create(&comm, ctx, S.id, rank, S.world));
  cusift_tiled *t = nullptr;
  CHECK(cusift_tiled_create(&t, ctx, comm, rank, S.world, S.W, S.H, &S.prm, 0));
  int own0 = 0, own1 = 0;
  CHECK(cusift_tiled_band(t, 0, nullptr, nullptr, nullptr, nullptr, &own0, &own1, nullptr, nullptr));
  cusift_point *d_points = nullptr, *d_gathered = nullptr;
  unsigned int *d_count = nullptr;
  CHECK(cusift_malloc((void **)&d_points, sizeof(cusift_point) * S.prm.max_pts));
  CHECK(cusift_malloc((void **)&d_count, sizeof(unsigned int)));
  CHECK(cusift_malloc((void **)&d_gathered, sizeof(cusift_point) * S.region_cap * S.world));
  // my strip = my owned base rows of the image (any device buffer with `strip_pitch` floats per row)
  for (int rep = 0; rep < 2; ++rep) {  // twice: the second extraction must not be disturbed by the first
    CHECK(cusift_tiled_extract(t, S.d_image + (size_t)own0 * S.W, S.W, d_points, d_count));
    unsigned int flagged = 0;
    CHECK(cusift_tiled_check(t, &flagged));
  }
  std::vector<size_t> totals(S.world);
  if (comm) {
    CHECK(cusift_allgatherv(comm, nullptr, d_points, d_count, 1, S.prm.max_pts, 1, d_gathered, S.region_cap, nullptr,
                            totals.data()));
    CHECK(cusift_ctx_synchronize(ctx));
  } else {
    unsigned int n = 0;
    CHECK(cusift_memcpy_d2h(ctx, &n, d_count, sizeof(n)));
    totals[0] = std::min<size_t>(n, (size_t)S.prm.max_pts);
    if (totals[0]) CHECK(cusift_memcpy_d2d(ctx, d_gathered, d_points, sizeof(cusift_point) * totals[0]));
  }
  std::vector<cusift_point> &out = S.merged[rank];
  for (int r = 0; r < S.world; ++r) {
    const size_t at = out.size();
    out.resize(at + totals[r]);
    if (totals[r])
      CHECK(cusift_memcpy_d2h(ctx, out.data() + at, d_gathered + (size_t)r * S.region_cap, sizeof(cusift_point) * totals[r]));
  }
  std::printf("rank %d of %d: rows [%d, %d), %zu keypoints here, %zu merged\n", rank, S.world, own0, own1, totals[rank],
              out.size());
  cusift_free(d_gathered);
  cusift_free(d_points);
  cusift_free(d_count);
  CHECK(cusift_tiled_destroy(t));
  if (comm) CHECK(cusift_comm_destroy(comm));
  CHECK(cusift_ctx_destroy(ctx));
  return 0;
}

static bool same_extracted(const cusift_point &a, const cusift_point &b) {
  // the fields extraction writes (the others are left as they were -- uninitialised in the reference, cuSIFT.cu:24,29)
  return std::memcmp(a.coords2D, b.coords2D, 6 * sizeof(float)) == 0 && a.subsampling == b.subsampling &&
         std::memcmp(a.data, b.data, sizeof(a.data)) == 0;
}

int main(int argc, char **argv) {
  if (argc < 6) {
    std::printf("usage: %s gray1.pgm W H octaves world [transport.so]\n", argv[0]);
    return 2;
  }
  std::vector<float> base;
  int bw = 0, bh = 0;
  if (!read_pgm(argv[1], base, bw, bh)) return 2;
  Shared S;
  S.W = std::atoi(argv[2]);
  S.H = std::atoi(argv[3]);
  S.world = std::atoi(argv[5]);
  cusift_default_params(&S.prm);
  S.prm.num_octaves = std::atoi(argv[4]);
  S.prm.init_blur = 0.0;
  S.prm.peak_thresh = 2.0f;
  S.prm.max_pts = 1 << 17;
  S.region_cap = (size_t)S.prm.max_pts;
  CHECK(cusift_init(0));
  if (S.world > 1) {
    if (argc < 7) {
      std::fprintf(stderr, "world > 1 in one process needs the in-process transport library\n");
      return 2;
    }
    CHECK(cusift_comm_use_library(argv[6]));
    CHECK(cusift_comm_get_unique_id(S.id));
  }
  // the image: the fixture mirror-tiled to W x H, gain 255/144 (SURVEY.md section 8d's generator without the shift)
  std::vector<float> img((size_t)S.W * S.H);
  for (int y = 0; y < S.H; ++y) {
    const int ty = y / bh, yy = (ty & 1) ? bh - 1 - y % bh : y % bh;
    for (int x = 0; x < S.W; ++x) {
      const int tx = x / bw, xx = (tx & 1) ? bw - 1 - x % bw : x % bw;
      img[(size_t)y * S.W + x] = (float)(int)(base[(size_t)yy * bw + xx] * (255.0f / 144.0f) + 0.5f);
    }
  }
  cusift_ctx *ctx = nullptr;
  CHECK(cusift_ctx_create(&ctx, 0, nullptr));
  float *d_image = nullptr;
  CHECK(cusift_malloc((void **)&d_image, sizeof(float) * img.size()));
  CHECK(cusift_memcpy_h2d(ctx, d_image, img.data(), sizeof(float) * img.size()));
  S.d_image = d_image;

  // the whole image on one GPU: the equality target (W is its own pitch here; the fast kernels want W % 4 == 0)
  std::vector<cusift_point> want((size_t)S.prm.max_pts);
  cusift_point *d_whole = nullptr;
  CHECK(cusift_malloc((void **)&d_whole, sizeof(cusift_point) * S.prm.max_pts));
  int n_whole = 0;
  CHECK(cusift_extract(ctx, d_image, S.W, S.H, S.W, &S.prm, d_whole, want.data(), &n_whole));
  want.resize(n_whole);
  CHECK(cusift_sort_points_host(want.data(), n_whole));

  S.merged.resize(S.world);
  S.rc.assign(S.world, 0);
  std::vector<std::thread> threads;
  for (int r = 0; r < S.world; ++r) threads.emplace_back([&S, r] { S.rc[r] = rank_main(S, r); });
  for (auto &th : threads) th.join();
  int failures = 0;
  for (int r = 0; r < S.world; ++r) failures += S.rc[r] != 0;
  for (int r = 0; r < S.world && !failures; ++r) {
    std::vector<cusift_point> &got = S.merged[r];
    if ((int)got.size() != n_whole) {
      std::printf("rank %d merged %zu keypoints, the whole image has %d\n", r, got.size(), n_whole);
      ++failures;
      continue;
    }
    CHECK(cusift_sort_points_host(got.data(), n_whole));
    int bad = 0;
    for (int i = 0; i < n_whole; ++i) bad += !same_extracted(got[i], want[i]);
    if (bad) {
      std::printf("rank %d: %d of %d merged keypoints differ from the whole image\n", r, bad, n_whole);
      ++failures;
    }
  }
  int n_oct = 0, collapse = 0;
  CHECK(cusift_tiled_plan(S.W, S.H, S.world, S.prm.num_octaves, 0, 0, 0, &n_oct, &collapse, nullptr, nullptr, nullptr,
                          nullptr, nullptr, nullptr, nullptr));
  std::printf("tiled: %dx%d, %d octaves (collapse from %d), %d ranks: whole image %d keypoints, every rank's merged "
              "SiftData %s\n", S.W, S.H, n_oct, collapse, S.world, n_whole, failures ? "DIFFERS" : "identical");
  cusift_free(d_whole);
  cusift_free(d_image);
  cusift_ctx_destroy(ctx);
  if (failures || n_whole < 1000) {
    std::printf("FAILED (%d)\n", failures);
    return 1;
  }
  std::printf("PASSED\n");
  return 0;
}
