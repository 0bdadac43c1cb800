// Host frames in, SiftData on the host out, from plain C++ over the C ABI (include/cusift_amd.h; no HIP headers, no
// Python): the pipelined form of what the reference's callers do one blocking image at a time -- decode an 8-bit frame,
// hand it to SiftData::Extract, read h_data (test/detector.cpp:19-56, cuSIFT.cu:61-120).
//
//   hostpipe_dropin <gray1.pgm> [batches=12] [images=8] [depth=3]
//
// Every batch is a different set of shifted copies of the fixture (8-bit pixels).  The program (1) extracts every frame
// ALONE through the blocking entry point (cusift_extract_host), (2) runs all batches through cusift_pipe_* with `depth`
// batches in flight, and demands that every image comes back with the same keypoints (same count, and the same
// multiset of x, y, scale, orientation and descriptor checksum); then prints one line:
//   hostpipe: N batches of B images WxH, depth D: T ms per batch, G Gpix/s host to host, K keypoints per batch, all equal
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cusift_amd.h"

#define CHECK(call)                                                        \
  do {                                                                     \
    int rc_ = (call);                                                      \
    if (rc_ != CUSIFT_OK) {                                                \
      std::fprintf(stderr, "%s failed: %s\n", #call, cusift_last_error()); \
      return 1;                                                            \
    }                                                                      \
  } while (0)

static bool read_pgm(const char *path, std::vector<unsigned char> &img, int &w, int &h) {
  FILE *fp = std::fopen(path, "rb");
  if (!fp) return false;
  int maxv = 0;
  if (std::fscanf(fp, "P5 %d %d %d", &w, &h, &maxv) != 3 || maxv != 255) return false;
  std::fgetc(fp);
  img.resize((size_t)w * h);
  const bool ok = std::fread(img.data(), 1, img.size(), fp) == img.size();
  std::fclose(fp);
  return ok;
}

struct Key {  // what identifies a keypoint, bit for bit
  float x, y, scale, ori;
  double desc;
  bool operator<(const Key &o) const { return std::memcmp(this, &o, sizeof(Key)) < 0; }
  bool operator==(const Key &o) const { return std::memcmp(this, &o, sizeof(Key)) == 0; }
};
static std::vector<Key> keys(const cusift_point *p, size_t n) {
  std::vector<Key> k(n);
  for (size_t i = 0; i < n; ++i) {
    std::memset(&k[i], 0, sizeof(Key));
    k[i].x = p[i].coords2D[0];
    k[i].y = p[i].coords2D[1];
    k[i].scale = p[i].scale;
    k[i].ori = p[i].orientation;
    double s = 0.0;
    for (int j = 0; j < 128; ++j) s += (j + 1) * (double)p[i].data[j];
    k[i].desc = s;
  }
  std::sort(k.begin(), k.end());
  return k;
}

int main(int argc, char **argv) {
  if (argc < 2) {
    std::printf("usage: %s gray1.pgm [batches] [images] [depth]\n", argv[0]);
    return 2;
  }
  const int N = argc > 2 ? std::atoi(argv[2]) : 12, B = argc > 3 ? std::atoi(argv[3]) : 8, D = argc > 4 ? std::atoi(argv[4]) : 3;
  std::vector<unsigned char> base;
  int w = 0, h = 0;
  if (!read_pgm(argv[1], base, w, h) || N < 1 || B < 1) return 2;
  CHECK(cusift_init(0));
  cusift_params prm;
  cusift_default_params(&prm);
  prm.num_octaves = 4;
  prm.init_blur = 0.0f;
  prm.peak_thresh = 2.0f;
  prm.max_pts = 4096;

  // the frames: pinned host memory (cusift_malloc_host), [N][B][h][w] bytes
  const size_t frame = (size_t)w * h, batch = frame * B;
  unsigned char *frames = nullptr;
  CHECK(cusift_malloc_host((void **)&frames, batch * N));
  for (int k = 0; k < N; ++k)
    for (int i = 0; i < B; ++i) {
      unsigned char *dst = frames + batch * k + frame * i;
      const int sx = (37 * k + 11 * i) % w, sy = (23 * k + 7 * i) % h;
      for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) dst[(size_t)y * w + x] = base[(size_t)((y + sy) % h) * w + (x + sx) % w];
    }

  // (1) every frame alone, blocking: float conversion on the host + cusift_extract_host, as the reference's callers do
  cusift_ctx *ctx = nullptr;
  CHECK(cusift_ctx_create(&ctx, 0, nullptr));
  cusift_point *d_pts = nullptr;
  CHECK(cusift_malloc((void **)&d_pts, sizeof(cusift_point) * prm.max_pts));
  std::vector<cusift_point> h_pts(prm.max_pts);
  std::vector<float> fimg(frame);
  std::vector<std::vector<Key>> want((size_t)N * B);
  size_t want_total = 0;
  for (int k = 0; k < N; ++k)
    for (int i = 0; i < B; ++i) {
      const unsigned char *src = frames + batch * k + frame * i;
      for (size_t t = 0; t < frame; ++t) fimg[t] = (float)src[t];
      int n = 0;
      CHECK(cusift_extract_host(ctx, fimg.data(), w, h, &prm, d_pts, h_pts.data(), &n));
      want[(size_t)k * B + i] = keys(h_pts.data(), (size_t)n);
      want_total += (size_t)n;
    }
  CHECK(cusift_free(d_pts));
  CHECK(cusift_ctx_destroy(ctx));

  // (2) the pipeline: D batches in flight
  cusift_pipe *pipe = nullptr;
  CHECK(cusift_pipe_create(&pipe, 0, B, w, h, &prm, CUSIFT_PIPE_U8, D, 0));
  size_t got_total = 0;
  int collected = 0, bad = 0;
  auto collect = [&]() -> int {
    const cusift_point *rec = nullptr;
    const unsigned int *off = nullptr;
    int n_img = 0;
    size_t total = 0;
    CHECK(cusift_pipe_collect(pipe, &rec, &off, &n_img, &total));
    if (n_img != B || off[B] != total) ++bad;
    for (int i = 0; i < n_img; ++i) {
      const std::vector<Key> g = keys(rec + off[i], off[i + 1] - off[i]);
      if (!(g == want[(size_t)collected * B + i])) ++bad;
    }
    got_total += total;
    ++collected;
    return 0;
  };
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < N; ++k) {
    if (cusift_pipe_in_flight(pipe) == D && collect()) return 1;
    CHECK(cusift_pipe_submit(pipe, frames + batch * k, B));
  }
  while (cusift_pipe_in_flight(pipe))
    if (collect()) return 1;
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  CHECK(cusift_pipe_destroy(pipe));
  CHECK(cusift_free_host(frames));
  if (bad || got_total != want_total || collected != N) {
    std::printf("hostpipe: MISMATCH (%d images differ; %zu keypoints, the blocking path found %zu)\n", bad, got_total, want_total);
    return 1;
  }
  std::printf("hostpipe: %d batches of %d images %dx%d, depth %d: %.3f ms per batch, %.2f Gpix/s host to host, %zu keypoints per "
              "batch, all equal\n", N, B, w, h, D, ms / N, (double)frame * B * N / ms / 1e6, got_total / (size_t)N);
  return 0;
}
