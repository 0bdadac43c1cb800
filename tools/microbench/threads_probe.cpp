// Where does a multi-threaded caller of the blocking single-image entry point lose its time?  T host threads, each its own
// context, each calling cusift_extract on one device-resident 1080p frame in a loop -- with and without the records'
// read-back -- against one thread.  g++ -O2 -std=c++14 -I include tools/microbench/threads_probe.cpp -Lcusift_amd -lcusift_amd -lpthread
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "cusift_amd.h"

int main(int argc, char **argv) {
  const int W = 1920, H = 1080, P = 1920, N = argc > 1 ? std::atoi(argv[1]) : 300;
  cusift_init(0);
  // the fixture mirror-tiled to 1080p with a gain, low-passed to sigma 1.0, re-quantised (the frames of tests/cpp/threads_dropin.cpp)
  std::vector<float> host((size_t)P * H), base;
  {
    FILE *fp = std::fopen(argc > 2 ? argv[2] : "tests/golden/gray1.pgm", "rb");
    int bw = 0, bh = 0, maxv = 0;
    if (!fp || std::fscanf(fp, "P5 %d %d %d", &bw, &bh, &maxv) != 3) return 2;
    std::fgetc(fp);
    std::vector<unsigned char> raw((size_t)bw * bh);
    if (std::fread(raw.data(), 1, raw.size(), fp) != raw.size()) return 2;
    std::fclose(fp);
    float k[9], ksum = 0.0f;
    for (int i = 0; i < 9; ++i) ksum += (k[i] = std::exp(-(float)((i - 4) * (i - 4)) / 2.0f));
    for (int i = 0; i < 9; ++i) k[i] /= ksum;
    std::vector<float> a((size_t)W * H), b((size_t)W * H);
    auto cl = [](int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); };
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        int yy = y % (2 * bh), xx = x % (2 * bw);
        if (yy >= bh) yy = 2 * bh - 1 - yy;
        if (xx >= bw) xx = 2 * bw - 1 - xx;
        a[(size_t)y * W + x] = raw[(size_t)yy * bw + xx] * (255.0f / 144.0f);
      }
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        float v = 0.0f;
        for (int t = -4; t <= 4; ++t) v += k[t + 4] * a[(size_t)y * W + cl(x + t, W)];
        b[(size_t)y * W + x] = v;
      }
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        float v = 0.0f;
        for (int t = -4; t <= 4; ++t) v += k[t + 4] * b[(size_t)cl(y + t, H) * W + x];
        host[(size_t)y * P + x] = std::nearbyint(v);
      }
  }
  float *d_img = nullptr;
  cusift_malloc((void **)&d_img, host.size() * 4);
  cusift_ctx *c0 = nullptr;
  cusift_ctx_create(&c0, 0, nullptr);
  cusift_memcpy_h2d(c0, d_img, host.data(), host.size() * 4);
  cusift_ctx_synchronize(c0);
  cusift_params prm;
  cusift_default_params(&prm);
  prm.num_octaves = 5; prm.init_blur = 1.0f; prm.peak_thresh = 3.0f; prm.max_pts = 32768;
  const int NB = argc > 3 ? std::atoi(argv[3]) : 1;  // frames per cusift_extract_batch call in mode 2 (the same frame NB times)
  float *d_batch = nullptr;
  if (NB > 1) {
    cusift_malloc((void **)&d_batch, host.size() * 4 * NB);
    for (int i = 0; i < NB; ++i) cusift_memcpy_h2d(c0, d_batch + (size_t)i * host.size(), host.data(), host.size() * 4);
    cusift_ctx_synchronize(c0);
  }
  for (int mode = 0; mode < 4; ++mode)      // 0: count only, 1: records to pinned host, 2: batch entry point, sync per call,
                                            // 3: the same launch sequence replayed as a recorded hipGraph, sync per call
    for (int T : {1, 2, 4, 8, 16}) {
      std::vector<std::thread> th;
      std::vector<int> kp(T, 0);
      const auto t0 = std::chrono::steady_clock::now();
      for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
          cusift_ctx *c = nullptr;
          cusift_ctx_create(&c, 0, nullptr);
          cusift_point *d_pts = nullptr, *h_pts = nullptr;
          unsigned int *d_cnt = nullptr;
          cusift_malloc((void **)&d_pts, (size_t)prm.max_pts * sizeof(cusift_point) * (mode >= 2 ? NB : 1));
          cusift_malloc((void **)&d_cnt, 256);
          cusift_malloc_host((void **)&h_pts, (size_t)prm.max_pts * sizeof(cusift_point));
          int n = 0;
          cusift_graph *graph = nullptr;
          if (mode == 3)
            cusift_graph_create(c, &graph, NB > 1 ? d_batch : d_img, NB, W, H, P, (size_t)P * H, &prm, d_pts, d_cnt);
          for (int i = 0; i < N + 20; ++i) {
            if (mode == 3) {
              cusift_graph_launch(graph);
              cusift_ctx_synchronize(c);
            } else if (mode == 2) {
              cusift_extract_batch(c, NB > 1 ? d_batch : d_img, NB, W, H, P, (size_t)P * H, &prm, d_pts, d_cnt);
              cusift_ctx_synchronize(c);
            } else {
              cusift_extract(c, d_img, W, H, P, &prm, d_pts, mode == 1 ? h_pts : nullptr, &n);
            }
          }
          kp[t] = n;
          if (graph) cusift_graph_destroy(graph);
          cusift_free(d_pts); cusift_free(d_cnt); cusift_free_host(h_pts);
          cusift_ctx_destroy(c);
        });
      for (auto &x : th) x.join();
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      std::printf("mode %d (%s) threads %d: %.4f ms per frame overall (%.1f Gpix/s), %d keypoints\n", mode,
                  mode == 0 ? "count only" : mode == 1 ? "records to pinned host" : mode == 2 ? "extract_batch + sync" : "graph replay + sync", T,
                  ms / ((double)T * (N + 20) * (mode >= 2 ? NB : 1)),
                  (double)W * H * T * (N + 20) * (mode >= 2 ? NB : 1) / (ms * 1e-3) / 1e9, kp[0]);
    }
  return 0;
}
