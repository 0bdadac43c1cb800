// homography.h -- drop-in for the reference's extras/homography.h on top of the C ABI: FindHomography (RANSAC on
// the GPU) and ImproveHomography (iteratively re-weighted least squares on the host).
//
// FindHomography keeps the reference's division of labour (extras/homography.cu:182-269): the host filters the
// matched points by score/ambiguity and draws four distinct samples per hypothesis with rand(); the device solves
// the num_loops 8x8 systems and counts every hypothesis' inliers (cusift_find_homography); the host takes the first
// hypothesis with the most inliers.  ImproveHomography (extras/homography.cu:271-336) is host arithmetic; the
// reference solves its 8x8 normal equations with cv::solve(..., DECOMP_CHOLESKY) -- here a plain Cholesky in
// double, so the header needs no OpenCV.
#ifndef CUSIFT_AMD_HOMOGRAPHY_H
#define CUSIFT_AMD_HOMOGRAPHY_H

#include <cmath>
#include <cstdlib>
#include <vector>

#include "cuSIFT.h"
#include "cusift_amd_extras.h"

// Returns the elapsed milliseconds like the reference.  homography: 9 floats (row-major 3x3, h[8] = 1);
// *numMatches: inliers of the winning hypothesis among ALL numPts points of `data`.
inline double FindHomography(SiftData &data, float *homography, int *numMatches, int numLoops = 1000,
                             float minScore = 0.85f, float maxAmbiguity = 0.95f, float thresh = 5.0f) {
  *numMatches = 0;
  homography[0] = homography[4] = homography[8] = 1.0f;
  homography[1] = homography[2] = homography[3] = 0.0f;
  homography[5] = homography[6] = homography[7] = 0.0f;
  if (data.d_data == nullptr) return 0.0;
  TimerGPU timer;
  numLoops = iDivUp(numLoops, 16) * 16;  // :200
  const int numPts = data.numPts;
  if (numPts < 8) return 0.0;
  cusift_ctx *ctx = cusift_dropin::ctx();
  // score / ambiguity of every record (two strided device-to-host copies, :214-215)
  std::vector<float> scores((size_t)numPts), ambiguities((size_t)numPts);
  safeCall(cusift_memcpy2d_d2h(ctx, scores.data(), sizeof(float), &data.d_data[0].score, sizeof(SiftPoint),
                               sizeof(float), (size_t)numPts));
  safeCall(cusift_memcpy2d_d2h(ctx, ambiguities.data(), sizeof(float), &data.d_data[0].ambiguity, sizeof(SiftPoint),
                               sizeof(float), (size_t)numPts));
  std::vector<int> valid;
  for (int i = 0; i < numPts; i++)
    if (scores[i] > minScore && ambiguities[i] < maxAmbiguity) valid.push_back(i);  // :218-219
  const int numValid = (int)valid.size();
  if (numValid >= 8) {
    std::vector<int> randPts(4 * (size_t)numLoops);
    for (int i = 0; i < numLoops; i++) {  // :222-235: four distinct valid points, host rand()
      int p1 = std::rand() % numValid;
      int p2 = std::rand() % numValid;
      int p3 = std::rand() % numValid;
      int p4 = std::rand() % numValid;
      while (p2 == p1) p2 = std::rand() % numValid;
      while (p3 == p1 || p3 == p2) p3 = std::rand() % numValid;
      while (p4 == p1 || p4 == p2 || p4 == p3) p4 = std::rand() % numValid;
      randPts[i + 0 * (size_t)numLoops] = valid[p1];
      randPts[i + 1 * (size_t)numLoops] = valid[p2];
      randPts[i + 2 * (size_t)numLoops] = valid[p3];
      randPts[i + 3 * (size_t)numLoops] = valid[p4];
    }
    safeCall(cusift_find_homography(ctx, reinterpret_cast<const cusift_point *>(data.d_data), numPts, randPts.data(),
                                    numLoops, thresh, homography, numMatches, nullptr, nullptr));
  }
  const double gpuTime = timer.read();
#ifdef VERBOSE
  std::printf("FindHomography time =         %.2f ms\n", gpuTime);
#endif
  return gpuTime;
}

namespace cusift_dropin {
// Solves the symmetric positive definite 8x8 system M a = x in place (a <- solution); false if M is not positive
// definite (a is left untouched).
inline bool cholesky_solve8(const double M[8][8], const double x[8], double a[8]) {
  double L[8][8] = {{0}};
  for (int i = 0; i < 8; i++)
    for (int j = 0; j <= i; j++) {
      double s = M[i][j];
      for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
      if (i == j) {
        if (!(s > 0.0)) return false;
        L[i][i] = std::sqrt(s);
      } else {
        L[i][j] = s / L[j][j];
      }
    }
  double y[8];
  for (int i = 0; i < 8; i++) {
    double s = x[i];
    for (int k = 0; k < i; k++) s -= L[i][k] * y[k];
    y[i] = s / L[i][i];
  }
  for (int i = 7; i >= 0; i--) {
    double s = y[i];
    for (int k = i + 1; k < 8; k++) s -= L[k][i] * a[k];
    a[i] = s / L[i][i];
  }
  return true;
}
}  // namespace cusift_dropin

// extras/homography.cu:271-336: numLoops rounds of weighted least squares over the points that pass
// minScore/maxAmbiguity (weight limit/(err+limit), limit = thresh^2), then the number of points of `data` whose
// reprojection error is below thresh; writes match_error of every host record.  Works on data.h_data.
inline int ImproveHomography(SiftData &data, float *homography, int numLoops, float minScore, float maxAmbiguity,
                             float thresh) {
  if (data.h_data == nullptr) return 0;
  SiftPoint *mpts = data.h_data;
  const float limit = thresh * thresh;
  const int numPts = data.numPts;
  double A[8];
  for (int i = 0; i < 8; i++) A[i] = homography[i] / homography[8];
  for (int loop = 0; loop < numLoops; loop++) {
    double M[8][8] = {{0}}, X[8] = {0};
    for (int i = 0; i < numPts; i++) {
      const SiftPoint &pt = mpts[i];
      if (pt.score < minScore || pt.ambiguity > maxAmbiguity) continue;
      const float px = pt.coords2D[0], py = pt.coords2D[1];
      const float den = (float)(A[6] * px + A[7] * py + 1.0f);
      const float dx = (float)((A[0] * px + A[1] * py + A[2]) / den - pt.match_xpos);
      const float dy = (float)((A[3] * px + A[4] * py + A[5]) / den - pt.match_ypos);
      const float err = dx * dx + dy * dy;
      const float wei = limit / (err + limit);
      const double Yx[8] = {px, py, 1.0, 0.0, 0.0, 0.0, -(double)(px * pt.match_xpos), -(double)(py * pt.match_xpos)};
      const double Yy[8] = {0.0, 0.0, 0.0, px, py, 1.0, -(double)(px * pt.match_ypos), -(double)(py * pt.match_ypos)};
      for (int c = 0; c < 8; c++) {
        for (int r = 0; r < 8; r++) M[r][c] += Yx[c] * Yx[r] * wei + Yy[c] * Yy[r] * wei;
        X[c] += Yx[c] * pt.match_xpos * wei + Yy[c] * pt.match_ypos * wei;
      }
    }
    cusift_dropin::cholesky_solve8(M, X, A);
  }
  int numfit = 0;
  for (int i = 0; i < numPts; i++) {
    SiftPoint &pt = mpts[i];
    const float den = (float)(A[6] * pt.coords2D[0] + A[7] * pt.coords2D[1] + 1.0);
    const float dx = (float)((A[0] * pt.coords2D[0] + A[1] * pt.coords2D[1] + A[2]) / den - pt.match_xpos);
    const float dy = (float)((A[3] * pt.coords2D[0] + A[4] * pt.coords2D[1] + A[5]) / den - pt.match_ypos);
    const float err = dx * dx + dy * dy;
    if (err < limit) numfit++;
    pt.match_error = std::sqrt(err);
  }
  for (int i = 0; i < 8; i++) homography[i] = (float)A[i];
  homography[8] = 1.0f;
  return numfit;
}

#endif  // CUSIFT_AMD_HOMOGRAPHY_H
