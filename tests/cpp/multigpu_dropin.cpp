// One rank of an image-sharded extraction with the all-gatherv of SiftData -- the C++ side of BASELINE configs[3]
// (512 x 1080p over 8 GPUs, 64 per GPU), written against the drop-in headers only: plain g++, no HIP / RCCL headers.
//
//   multigpu_dropin <rank> <world> <id-file> <gray1.pgm> [images-per-rank]
//
// Launch one process per GPU (mpirun, a shell loop, a job scheduler ...): rank 0 writes the communicator id to
// <id-file>, the others wait for it.  Every rank extracts its images on GPU `rank % device_count` through the
// reference's own API (SiftData / ExtractSift), then all ranks exchange their SiftData so that each ends up with
// every image's keypoints.  With world == 1 the local shard still travels through ncclSend/ncclRecv (self p2p), which
// is how tests/test_cpp_dropin.py runs it on a single GPU.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "cuImage.h"
#include "cuSIFT.h"
#include "cusift_amd.h"
#include "cusift_amd_multigpu.h"

#define CHECK(call)                                                              \
  do {                                                                           \
    int rc_ = (call);                                                            \
    if (rc_ != CUSIFT_OK) {                                                      \
      std::fprintf(stderr, "%s failed: %s\n", #call, cusift_last_error());       \
      return 1;                                                                  \
    }                                                                            \
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

int main(int argc, char **argv) {
  if (argc < 5) {
    std::printf("usage: %s rank world id-file gray1.pgm [images-per-rank]\n", argv[0]);
    return 2;
  }
  const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
  const char *id_file = argv[3];
  const int n_local = argc > 5 ? std::atoi(argv[5]) : 3;
  std::vector<float> base;
  int w = 0, h = 0;
  if (!read_pgm(argv[4], base, w, h)) return 2;

  int n_dev = 0;
  CHECK(cusift_device_count(&n_dev));
  if (n_dev < 1) return 1;
  const int device = rank % n_dev;
  InitCuda(device);

  // ---- the communicator: rank 0 makes the id, everybody reads it ----
  char id[CUSIFT_UNIQUE_ID_BYTES];
  if (rank == 0) {
    CHECK(cusift_comm_get_unique_id(id));
    std::string tmp = std::string(id_file) + ".tmp";
    FILE *fp = std::fopen(tmp.c_str(), "wb");
    if (!fp || std::fwrite(id, 1, sizeof(id), fp) != sizeof(id)) return 1;
    std::fclose(fp);
    std::rename(tmp.c_str(), id_file);
  } else {
    FILE *fp = nullptr;
    for (int tries = 0; tries < 600 && !(fp = std::fopen(id_file, "rb")); ++tries)
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    if (!fp || std::fread(id, 1, sizeof(id), fp) != sizeof(id)) return 1;
    std::fclose(fp);
  }
  cusift_ctx *ctx = nullptr;
  CHECK(cusift_ctx_create(&ctx, device, nullptr));
  cusift_comm *comm = nullptr;
  CHECK(cusift_comm_create(&comm, ctx, id, rank, world));
  if (world == 1) CHECK(cusift_comm_set_self_p2p(comm, 1));

  // ---- this rank's images through the reference's API: image i of rank r is the fixture shifted by (7 g, 13 g),
  // g = r * n_local + i.  One SiftData per image (legacy trio, main.cpp:99-103), results kept on the device. ----
  const int max_pts = 4096;
  std::vector<unsigned int> counts(n_local);
  cusift_point *d_points = nullptr;
  unsigned int *d_counts = nullptr;
  CHECK(cusift_malloc((void **)&d_points, sizeof(cusift_point) * (size_t)n_local * max_pts));
  CHECK(cusift_malloc((void **)&d_counts, sizeof(unsigned int) * n_local));
  for (int i = 0; i < n_local; ++i) {
    const int g = rank * n_local + i;
    std::vector<float> img((size_t)w * h);
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) img[(size_t)y * w + x] = base[(size_t)((y + 7 * g) % h) * w + (x + 13 * g) % w];
    cuImage im(w, h, img.data());
    SiftData sift;
    InitSiftData(sift, max_pts, true, true);
    ExtractSift(sift, im, 4, 0.0, 1.0f, 0.0f, 1.0f);
    counts[i] = (unsigned int)sift.numPts;
    CHECK(cusift_memcpy_d2d(ctx, d_points + (size_t)i * max_pts, sift.d_data, sizeof(cusift_point) * sift.numPts));
    FreeSiftData(sift);
  }
  CHECK(cusift_memcpy_h2d(ctx, d_counts, counts.data(), sizeof(unsigned int) * n_local));

  // ---- all-gatherv of SiftData (two phases; a pipelined caller enqueues its next batches in between) ----
  // d_gathered is `world` regions of region_cap records: region r = rank r's records, packed in image order
  const size_t region_cap = (size_t)n_local * max_pts;
  cusift_point *d_gathered = nullptr;
  CHECK(cusift_malloc((void **)&d_gathered, sizeof(cusift_point) * region_cap * world));
  std::vector<unsigned int> all_counts((size_t)world * n_local);
  std::vector<size_t> totals(world);
  CHECK(cusift_comm_reserve(comm, n_local, 2, region_cap));  // nothing below allocates
  CHECK(cusift_allgatherv_begin(comm, /*producer=*/ctx, d_points, d_counts, n_local, max_pts, n_local, d_gathered,
                                region_cap));
  CHECK(cusift_allgatherv_finish(comm, all_counts.data(), totals.data()));
  CHECK(cusift_ctx_synchronize(ctx));

  // ---- checks: my counts came back in my slot, my shard sits in my region, bit for bit ----
  int failures = 0;
  size_t mine = 0, everybody = 0;
  for (int i = 0; i < n_local; ++i) {
    if (all_counts[(size_t)rank * n_local + i] != counts[i]) ++failures;
    mine += counts[i];
  }
  for (int r = 0; r < world; ++r) everybody += totals[r];
  if (totals[rank] != mine) ++failures;
  std::vector<cusift_point> got(mine), want(mine);
  if (mine) CHECK(cusift_memcpy_d2h(ctx, got.data(), d_gathered + (size_t)rank * region_cap, sizeof(cusift_point) * mine));
  size_t pos = 0;
  for (int i = 0; i < n_local; ++i) {
    if (counts[i])
      CHECK(cusift_memcpy_d2h(ctx, want.data() + pos, d_points + (size_t)i * max_pts, sizeof(cusift_point) * counts[i]));
    pos += counts[i];
  }
  if (mine && std::memcmp(got.data(), want.data(), sizeof(cusift_point) * mine) != 0) ++failures;
  // every peer's region starts with a record of that peer's first image: finite coordinates inside the image
  for (int r = 0; r < world; ++r) {
    if (!totals[r]) continue;
    cusift_point first;
    CHECK(cusift_memcpy_d2h(ctx, &first, d_gathered + (size_t)r * region_cap, sizeof(first)));
    if (!(first.coords2D[0] >= 0.f && first.coords2D[0] < (float)w && first.coords2D[1] >= 0.f &&
          first.coords2D[1] < (float)h && first.subsampling >= 1.f))
      ++failures;
  }
  std::printf("rank %d of %d on GPU %d (%s): %d images, %zu keypoints here, %zu gathered\n", rank, world, device,
              cusift_comm_library(), n_local, mine, everybody);
  if (mine < 100 || everybody < mine * (size_t)world / 2) ++failures;

  cusift_free(d_gathered);
  cusift_free(d_points);
  cusift_free(d_counts);
  cusift_comm_destroy(comm);
  cusift_ctx_destroy(ctx);
  if (failures) {
    std::printf("FAILED (%d)\n", failures);
    return 1;
  }
  std::printf("PASSED\n");
  return 0;
}
