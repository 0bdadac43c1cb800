// The throughput mode of bench.py from C++: consecutive 64 x 1080p batches rotated over E contexts (one stream, one
// scratch arena, one output buffer each) of one GPU, written against include/cusift_amd.h only -- plain g++, no HIP
// headers, no Python.  This is what a C++ caller of the reference (host code stays C++: BASELINE north_star) does to
// keep the GPU busy; every batch is still one complete cusift_extract_batch (cuSIFT.cu:61-120 per image).
//
//   pipeline_dropin <gray1.pgm> [contexts=4] [batches=40] [images=64] [width=1920] [height=1080]
//
// Images: the fixture mirror-tiled to width x height with a per-image cyclic shift, pre-blurred to sigma 1.0 and
// re-quantised (the shape of cusift_amd.synth.tile, not bit for bit -- this program measures, tests/ verify).  Prints one line:
//   pipeline: E contexts, N batches of B images WxH: T ms per batch, G Gpix/s, K keypoints per batch
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
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
  if (argc < 2) {
    std::printf("usage: %s gray1.pgm [contexts] [batches] [images] [width] [height]\n", argv[0]);
    return 2;
  }
  const int E = argc > 2 ? std::atoi(argv[2]) : 4, N = argc > 3 ? std::atoi(argv[3]) : 40;
  const int B = argc > 4 ? std::atoi(argv[4]) : 64, W = argc > 5 ? std::atoi(argv[5]) : 1920,
            H = argc > 6 ? std::atoi(argv[6]) : 1080;
  std::vector<float> base;
  int bw = 0, bh = 0;
  if (!read_pgm(argv[1], base, bw, bh) || E < 1 || N < 1 || B < 1) return 2;
  CHECK(cusift_init(0));

  cusift_params prm;
  cusift_default_params(&prm);
  prm.num_octaves = 5;
  prm.init_blur = 1.0f;
  prm.peak_thresh = 3.0f;
  prm.edge_thresh = 10.0f;
  prm.lowest_scale = 0.0f;
  prm.subsampling = 1.0f;
  prm.max_pts = 32768;
  prm.concurrent_batches = E;  // the scheduling hint: taller detection chunks when several batches are in flight

  // B images: mirror-tiled fixture with a gain (the fixture is dark) and a per-image shift, low-passed to sigma = 1.0
  // (the blur that initBlur = 1.0 declares; separable, radius 4, replicated borders) and re-quantised to 8 bit
  const size_t img_floats = (size_t)W * H;
  std::vector<float> host((size_t)B * img_floats);
  {
    float k[9], ksum = 0.0f;
    for (int i = 0; i < 9; ++i) ksum += (k[i] = std::exp(-(float)((i - 4) * (i - 4)) / 2.0f));
    for (int i = 0; i < 9; ++i) k[i] /= ksum;
    std::vector<float> raw(img_floats), tmp(img_floats);
    auto cl = [](int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); };
    for (int i = 0; i < B; ++i) {
      const int sx = (i * 37) % bw, sy = (i * 91) % bh;
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
      float *out = &host[(size_t)i * img_floats];
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
          float a = 0.0f;
          for (int t = -4; t <= 4; ++t) a += k[t + 4] * tmp[(size_t)cl(y + t, H) * W + x];
          a = std::nearbyint(a);
          out[(size_t)y * W + x] = a < 0.0f ? 0.0f : (a > 255.0f ? 255.0f : a);
        }
    }
  }

  std::vector<cusift_ctx *> ctx(E, nullptr);
  std::vector<cusift_point *> pts(E, nullptr);
  std::vector<unsigned int *> cnt(E, nullptr);
  float *d_imgs = nullptr;
  CHECK(cusift_malloc((void **)&d_imgs, host.size() * sizeof(float)));
  for (int e = 0; e < E; ++e) {
    CHECK(cusift_ctx_create(&ctx[e], 0, nullptr));  // owns a non-blocking stream
    CHECK(cusift_ctx_reserve(ctx[e], B, W, H, &prm));
    CHECK(cusift_malloc((void **)&pts[e], (size_t)B * prm.max_pts * sizeof(cusift_point)));
    CHECK(cusift_malloc((void **)&cnt[e], (size_t)B * sizeof(unsigned int)));
  }
  CHECK(cusift_memcpy_h2d(ctx[0], d_imgs, host.data(), host.size() * sizeof(float)));
  CHECK(cusift_ctx_synchronize(ctx[0]));

  auto run = [&](int batches) -> int {
    for (int i = 0; i < batches; ++i) {
      const int e = i % E;
      CHECK(cusift_extract_batch(ctx[e], d_imgs, B, W, H, W, img_floats, &prm, pts[e], cnt[e]));
    }
    for (int e = 0; e < E; ++e) CHECK(cusift_ctx_synchronize(ctx[e]));
    return 0;
  };
  if (run(2 * E)) return 1;  // warm-up
  const auto t0 = std::chrono::steady_clock::now();
  if (run(N)) return 1;
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / N;

  std::vector<unsigned int> h_cnt(B);
  CHECK(cusift_memcpy_d2h(ctx[0], h_cnt.data(), cnt[0], B * sizeof(unsigned int)));
  CHECK(cusift_ctx_synchronize(ctx[0]));
  long kp = 0;
  for (int i = 0; i < B; ++i) kp += h_cnt[i] < (unsigned int)prm.max_pts ? h_cnt[i] : prm.max_pts;
  std::printf("pipeline: %d contexts, %d batches of %d images %dx%d: %.4f ms per batch, %.1f Gpix/s, %ld keypoints per batch\n",
              E, N, B, W, H, ms, (double)B * W * H / (ms * 1e-3) / 1e9, kp);

  for (int e = 0; e < E; ++e) {
    cusift_free(pts[e]);
    cusift_free(cnt[e]);
    cusift_ctx_destroy(ctx[e]);
  }
  cusift_free(d_imgs);
  return kp > 0 ? 0 : 1;
}
