// An UNCHANGED cuSIFT program run from several host threads: every thread owns its SiftData and cuImage objects and
// calls nothing but the reference's API (InitCuda, cuImage::Allocate / HostToDevice, InitSiftData, ExtractSift,
// FreeSiftData -- main.cpp:99-103,313-328,348-349).  include/cuSIFT.h gives each calling thread its own implicit
// context (stream + scratch arena), so the threads' extractions overlap on the device; the reference's single global
// state (cuSIFT_D.cu:13-20) allows one at a time.
//
//   threads_dropin <gray1.pgm> [threads=4] [frames_per_thread=16] [rounds=6] [width=1920] [height=1080]
//
// 1. one thread extracts all threads x frames images, results sorted canonically and kept;
// 2. the threads extract their own frames concurrently: every image's SiftData must equal pass 1's bit for bit;
// 3. timed, between two barriers: `rounds` more passes of 2. without the comparison; then the same on one thread.
// Prints   threads: T threads x F frames WxH: one thread A ms per frame (a Gpix/s), T threads B ms per frame (b Gpix/s), all equal
#define CUSIFT_QUIET
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "cuSIFT.h"

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

struct Frame {
  std::vector<float> host;
  cuImage img;
};

static const int kOctaves = 5, kMaxPts = 32768;
static const double kInitBlur = 1.0;
static const float kThresh = 3.0f;

int main(int argc, char **argv) {
  if (argc < 2) {
    std::printf("usage: %s gray1.pgm [threads] [frames_per_thread] [rounds] [width] [height]\n", argv[0]);
    return 2;
  }
  const int T = argc > 2 ? std::atoi(argv[2]) : 4, F = argc > 3 ? std::atoi(argv[3]) : 16;
  const int R = argc > 4 ? std::atoi(argv[4]) : 6, W = argc > 5 ? std::atoi(argv[5]) : 1920,
            H = argc > 6 ? std::atoi(argv[6]) : 1080;
  std::vector<float> base;
  int bw = 0, bh = 0;
  if (!read_pgm(argv[1], base, bw, bh) || T < 1 || F < 1 || R < 1) return 2;
  InitCuda(0);

  // T x F frames: the fixture mirror-tiled with a gain and a per-frame shift, low-passed to sigma = 1.0 (what initBlur
  // = 1.0 declares) and re-quantised -- the images of pipeline_dropin.cpp
  const int N = T * F;
  std::vector<std::unique_ptr<Frame>> frames;
  {
    float k[9], ksum = 0.0f;
    for (int i = 0; i < 9; ++i) ksum += (k[i] = std::exp(-(float)((i - 4) * (i - 4)) / 2.0f));
    for (int i = 0; i < 9; ++i) k[i] /= ksum;
    const size_t px = (size_t)W * H;
    std::vector<float> raw(px), tmp(px);
    auto cl = [](int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); };
    for (int i = 0; i < N; ++i) {
      std::unique_ptr<Frame> f(new Frame);
      f->host.resize(px);
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
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
          float a = 0.0f;
          for (int t = -4; t <= 4; ++t) a += k[t + 4] * tmp[(size_t)cl(y + t, H) * W + x];
          a = std::nearbyint(a);
          f->host[(size_t)y * W + x] = a < 0.0f ? 0.0f : (a > 255.0f ? 255.0f : a);
        }
      // main.cpp:313-318: wrap the caller's pixels, upload
      f->img.Allocate(W, H, iAlignUp(W, 128), false, NULL, f->host.data());
      f->img.HostToDevice();
      frames.push_back(std::move(f));
    }
  }

  // pass 1: one thread, every frame; canonical order (the append order inside an octave is an atomic counter's)
  std::vector<std::vector<SiftPoint>> want(N);
  {
    SiftData data;
    InitSiftData(data, kMaxPts, true, true);
    for (int i = 0; i < N; ++i) {
      ExtractSift(data, frames[i]->img, kOctaves, kInitBlur, kThresh, 0.0f);
      data.SortCanonical();
      want[i].assign(data.h_data, data.h_data + data.numPts);
    }
    FreeSiftData(data);
  }

  // passes 2 and 3: T threads, each its own SiftData and its own F frames.  A thread first extracts its frames once and
  // compares them with pass 1 bit for bit (pass 2), then waits at a barrier; between that barrier and the next one every
  // thread runs R more rounds over its frames, untouched by any comparison (pass 3, timed by the main thread).  The
  // objects a thread needs -- its implicit context, its SiftData -- exist before the first barrier: what is timed is
  // ExtractSift, not cudaMalloc.
  std::vector<int> bad(T, 0);
  std::vector<long> kp(T, 0);
  struct Barrier {
    std::mutex m;
    std::condition_variable cv;
    int waiting = 0, generation = 0, parties;
    explicit Barrier(int n) : parties(n) {}
    void wait() {
      std::unique_lock<std::mutex> lk(m);
      const int gen = generation;
      if (++waiting == parties) {
        waiting = 0;
        ++generation;
        cv.notify_all();
      } else {
        cv.wait(lk, [&] { return gen != generation; });
      }
    }
  };
  auto run_threads = [&](int threads, bool check) {  // returns ms per frame of the timed rounds
    const int per = N / threads;  // frames per thread (threads == 1: all of them)
    Barrier ready(threads + 1), done(threads + 1);
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
      th.emplace_back([&, t] {
        SiftData data;
        InitSiftData(data, kMaxPts, true, true);
        for (int j = 0; j < per; ++j) {
          const int i = t * per + j;
          ExtractSift(data, frames[i]->img, kOctaves, kInitBlur, kThresh, 0.0f);
          if (!check) continue;
          data.SortCanonical();
          kp[t] += data.numPts;
          // every field extraction writes (the 12 it does not are uninitialised in the reference too: cuSIFT.cu:24,29)
          bool same = data.numPts == (int)want[i].size();
          for (int p = 0; same && p < data.numPts; ++p) {
            const SiftPoint &a = data.h_data[p], &b = want[i][p];
            same = std::memcmp(a.coords2D, b.coords2D, 6 * sizeof(float)) == 0 && a.subsampling == b.subsampling &&
                   std::memcmp(a.data, b.data, sizeof(a.data)) == 0;
          }
          bad[t] += !same;
        }
        ready.wait();
        for (int r = 0; r < R; ++r)
          for (int j = 0; j < per; ++j) ExtractSift(data, frames[t * per + j]->img, kOctaves, kInitBlur, kThresh, 0.0f);
        done.wait();
        FreeSiftData(data);
      });
    ready.wait();
    const auto t0 = std::chrono::steady_clock::now();
    done.wait();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (auto &x : th) x.join();
    return ms / ((double)R * per * threads);
  };
  const double many_ms = run_threads(T, true);
  int n_bad = 0;
  long n_kp = 0;
  for (int t = 0; t < T; ++t) n_bad += bad[t], n_kp += kp[t];
  const double one_ms = run_threads(1, false);
  const double gp = (double)W * H / 1e6;  // Mpix per frame; / ms = Gpix/s
  std::printf("threads: %d threads x %d frames %dx%d: one thread %.4f ms per frame (%.1f Gpix/s), %d threads %.4f ms per frame "
              "(%.1f Gpix/s), %ld keypoints, %s\n",
              T, F, W, H, one_ms, gp / one_ms, T, many_ms, gp / many_ms, n_kp, n_bad ? "MISMATCH" : "all equal");
  frames.clear();
  return n_bad == 0 && n_kp > 0 ? 0 : 1;
}
