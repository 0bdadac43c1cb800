// matching.h -- drop-in for the reference's extras/matching.h (MatchSiftData) on top of the C ABI.
// Same enums, SiftMatch record and call signature (extras/matching.h:10-39); the N x M search runs in
// cusift_match() (one MFMA kernel, no score matrix in memory), the threshold filter runs here like in the
// reference (extras/matching.cu:318-349).
#ifndef CUSIFT_AMD_MATCHING_H
#define CUSIFT_AMD_MATCHING_H

#include <vector>

#include "cuSIFT.h"
#include "cusift_amd_extras.h"

typedef enum { MatchSiftDistanceDotProduct, MatchSiftDistanceL2 } MatchSiftDistance;
typedef enum { MatchType2D, MatchType3D } MatchType;

typedef struct {
  SiftPoint *pt1;
  SiftPoint *pt2;
  float score;      // distance metric (dot product or L2 = 2 - 2 x.y)
  float ambiguity;  // ratio of best and second best
  float error;
} SiftMatch;

// Exhaustive search between all SIFT keypoints of two images (the caller owns the returned SiftMatch objects,
// as in the reference).  Both SiftData need host AND device buffers, as in the reference.
inline std::vector<SiftMatch *> MatchSiftData(SiftData &data1, SiftData &data2,
                                              MatchSiftDistance distance = MatchSiftDistanceL2,
                                              float scoreThreshold = 999.0, float ambiguityThreshold = 1.0,
                                              MatchType type = MatchType2D) {
  std::vector<SiftMatch *> matches;
  if (!data1.numPts || !data2.numPts) return matches;
  if (data1.d_data == nullptr || data2.d_data == nullptr) return matches;
  cusift_ctx *ctx = cusift_dropin::ctx();
  safeCall(cusift_match(ctx, reinterpret_cast<cusift_point *>(data1.d_data), data1.numPts,
                        reinterpret_cast<const cusift_point *>(data2.d_data), data2.numPts,
                        distance == MatchSiftDistanceL2 ? 1 : 0));
  if (data1.h_data != nullptr)  // the 5 match fields of every record, extras/matching.cu:311-315
    safeCall(cusift_memcpy2d_d2h(ctx, &data1.h_data[0].score, sizeof(SiftPoint), &data1.d_data[0].score,
                                 sizeof(SiftPoint), 5 * sizeof(float), (size_t)data1.numPts));
  else
    safeCall(cusift_ctx_synchronize(ctx));
  if (data1.h_data == nullptr) return matches;
  const float thresh2 = scoreThreshold * scoreThreshold;
  const float athresh2 = ambiguityThreshold * ambiguityThreshold;
  for (int i = 0; i < data1.numPts; i++) {
    SiftPoint &p = data1.h_data[i];
    if (!(p.score < thresh2 && p.ambiguity < athresh2)) continue;
    if (p.match < 0 || p.match >= data2.numPts || data2.h_data == nullptr) continue;
    if (type == MatchType2D || (p.coords3D[2] != 0 && data2.h_data[p.match].coords3D[2] != 0)) {
      SiftMatch *m = new SiftMatch();
      m->pt1 = &p;
      m->pt2 = &data2.h_data[p.match];
      m->score = p.score;
      m->ambiguity = p.ambiguity;
      m->error = 0.0f;
      matches.push_back(m);
    }
  }
  return matches;
}

#endif  // CUSIFT_AMD_MATCHING_H
