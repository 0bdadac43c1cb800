// extras/homography.h surface against the drop-in headers: matched SiftData with a planted homography and gross
// outliers -> FindHomography (RANSAC, GPU) -> ImproveHomography (host refinement), the sequence of the reference's
// demo (main.cpp:330-340).  Plain C++ (g++), no HIP headers.  The reference holds no fixture for this path.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "cuSIFT.h"
#include "homography.h"

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static double uniform01() {  // splitmix64
  uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (double)(z >> 11) / 9007199254740992.0;
}

int main() {
  InitCuda(0);
  int failures = 0;
  {
    const double H[9] = {0.92, -0.11, 37.0, 0.08, 1.05, -21.0, 2.1e-5, -3.4e-5, 1.0};
    const int nIn = 600, nOut = 400, n = nIn + nOut;
    SiftData data;
    InitSiftData(data, n, true, true);
    std::memset(data.h_data, 0, sizeof(SiftPoint) * n);
    for (int i = 0; i < n; i++) {
      SiftPoint &p = data.h_data[i];
      const double x = 1280.0 * uniform01(), y = 960.0 * uniform01();
      p.coords2D[0] = (float)x;
      p.coords2D[1] = (float)y;
      const bool inlier = (i % 5) != 1 && (i % 5) != 3;  // 60 % inliers, interleaved
      if (inlier) {
        const double den = H[6] * x + H[7] * y + 1.0;
        p.match_xpos = (float)((H[0] * x + H[1] * y + H[2]) / den + 0.6 * (uniform01() - 0.5));
        p.match_ypos = (float)((H[3] * x + H[4] * y + H[5]) / den + 0.6 * (uniform01() - 0.5));
      } else {
        p.match_xpos = (float)(1280.0 * uniform01());
        p.match_ypos = (float)(960.0 * uniform01());
      }
      // dot-product scores as MatchSiftData leaves them: a fifth of the outliers fail the score filter
      p.score = (!inlier && i % 4 == 0) ? 0.5f : 0.93f;
      p.ambiguity = 0.6f;
    }
    data.numPts = n;
    safeCall(cusift_memcpy_h2d(cusift_dropin::ctx(), data.d_data, data.h_data, sizeof(SiftPoint) * n));

    float homography[9];
    int numMatches = 0;
    FindHomography(data, homography, &numMatches, 1000, 0.85f, 0.95f, 5.0f);
    std::printf("FindHomography: %d inliers of %d points (%d planted)\n", numMatches, n, nIn);
    if (numMatches < (int)(0.9 * nIn) || numMatches > nIn + 40) ++failures;
    if (homography[8] != 1.0f) ++failures;
    const int numFit = ImproveHomography(data, homography, 5, 0.85f, 0.95f, 3.0f);
    std::printf("ImproveHomography: %d points within 3 px\n", numFit);
    if (numFit < (int)(0.97 * nIn) || numFit > nIn + 25) ++failures;
    double worst = 0.0;
    const double corners[4][2] = {{0, 0}, {1280, 0}, {0, 960}, {1280, 960}};
    for (const auto &c : corners) {
      const double d0 = H[6] * c[0] + H[7] * c[1] + 1.0, d1 = homography[6] * c[0] + homography[7] * c[1] + 1.0;
      const double ex = (H[0] * c[0] + H[1] * c[1] + H[2]) / d0 - (homography[0] * c[0] + homography[1] * c[1] + homography[2]) / d1;
      const double ey = (H[3] * c[0] + H[4] * c[1] + H[5]) / d0 - (homography[3] * c[0] + homography[4] * c[1] + homography[5]) / d1;
      worst = std::fmax(worst, std::sqrt(ex * ex + ey * ey));
    }
    std::printf("refined homography: worst corner error %.3f px\n", worst);
    if (!(worst < 0.5)) ++failures;
    int errors_set = 0;
    for (int i = 0; i < n; i++) errors_set += data.h_data[i].match_error > 0.0f;
    if (errors_set < n - 2) ++failures;  // match_error written for every record

    // fewer than 8 points / no device data: identity, zero matches, no crash (extras/homography.cu:184-203)
    SiftData tiny;
    InitSiftData(tiny, 4, true, true);
    tiny.numPts = 4;
    float h2[9];
    int m2 = -1;
    FindHomography(tiny, h2, &m2);
    if (m2 != 0 || h2[0] != 1.0f || h2[4] != 1.0f || h2[1] != 0.0f) ++failures;
  }
  cusift_dropin::shutdown();
  std::printf(failures ? "FAILED (%d)\n" : "PASSED\n", failures);
  return failures ? 1 : 0;
}
