/*
 * cusift_amd_extras.h -- the next rows of SURVEY 8f behind the C ABI: the brute-force matcher and the RANSAC homography.
 * Part of the C ABI of libcusift_amd.so; conventions and the map of the four headers: cusift_amd.h.
 */
#ifndef CUSIFT_AMD_EXTRAS_H
#define CUSIFT_AMD_EXTRAS_H

#include "cusift_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- matcher (first consumer of SiftData) ----------------------------- */
/* MatchSiftData(data1, data2, distance, ...), extras/matching.cu:232-362: for every point of d_sift1 the best
 * and second-best point of d_sift2 under `distance` (0 = MatchSiftDistanceDotProduct, 1 = MatchSiftDistanceL2,
 * extras/matching.h:10-13); writes score, ambiguity, match, match_xpos, match_ypos of d_sift1 (extras/matching.cu:
 * 140-150,219-229).  The score/ambiguity thresholds of the reference are a host-side filter over those fields
 * (:318-349) and stay on the caller's side (include/matching.h does it).  Asynchronous on the context's stream. */
int cusift_match(cusift_ctx *ctx, cusift_point *d_sift1, int num_pts1, const cusift_point *d_sift2, int num_pts2,
                 int distance);
/* cudaMemcpy2D device->host (extras/matching.cu:311-315 copies the 5 match fields of every record); blocking. */
int cusift_memcpy2d_d2h(cusift_ctx *ctx, void *h_dst, size_t dst_pitch, const void *d_src, size_t src_pitch,
                        size_t width_bytes, size_t rows);

/* ---- RANSAC homography from matched SiftData ------------------------------ */
/* The device part and the final selection of FindHomography(data, homography, numMatches, numLoops, minScore,
 * maxAmbiguity, thresh), extras/homography.cu:182-269: for num_loops hypotheses -- h_rand_pts[i*num_loops + l] is
 * the i-th (of 4) sample of hypothesis l, an index into d_sift; the reference draws them on the host with rand()
 * from the points that pass minScore/maxAmbiguity (:208-235), and so does include/homography.h -- solve the 8x8
 * system (ComputeHomographies :89-130), count the points with reprojection error < thresh (TestHomographies
 * :135-178, over coords2D -> match_xpos/ypos of ALL num_pts records) and return the first hypothesis with the
 * most inliers: h_homography[0..7], h_homography[8] = 1, *num_matches = its count.  h_all_homo ([8][num_loops])
 * and h_all_counts ([num_loops]) may be NULL.  Blocking. */
int cusift_find_homography(cusift_ctx *ctx, const cusift_point *d_sift, int num_pts, const int *h_rand_pts,
                           int num_loops, float thresh, float h_homography[9], int *num_matches, float *h_all_homo,
                           int *h_all_counts);


#ifdef __cplusplus
}
#endif
#endif /* CUSIFT_AMD_EXTRAS_H */
