/*
 * sift_oracle.h -- CPU restatement of the cuSIFT extraction hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (cusift_amd/, include/) never links,
 * imports or falls back to anything in oracle/.
 *
 * Parity status: PINNED for keypoint location/scale/orientation against the reference's own
 * golden pair test/data/gray1 -> test/data/cusift1_check (tests/test_oracle_golden.py);
 * "parity unpinned" for descriptors, sharpness, edgeness and RootSIFT (the reference holds no
 * golden vector for them: test/descriptor.cpp is empty).
 *
 * Every function cites the reference file:line it restates (paths relative to the reference
 * root: cuSIFT.cu = host drivers, cuSIFT_D.cu = device kernels).
 */
#ifndef SIFT_ORACLE_H
#define SIFT_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* cuSIFT.h:10-30 -- 147 x 4 B = 588 B, no padding. */
typedef struct oracle_sift_point {
  float coords2D[2];
  float scale;
  float sharpness;
  float edgeness;
  float orientation;
  float score;
  float ambiguity;
  int match;
  float match_xpos;
  float match_ypos;
  float match_error;
  float subsampling;
  float empty[3];
  float data[128];
  float coords3D[3];
} oracle_sift_point;

/* Public fields of SiftData (cuSIFT.h:44-51) + the two documented rules of this build. */
typedef struct oracle_params {
  int num_octaves;
  double init_blur;
  float peak_thresh;
  float edge_thresh;
  float lowest_scale;
  float subsampling;   /* Extract()'s 4th argument, default 1.0f */
  int max_pts;
  int tex_frac_bits;   /* 8 = CUDA texture unit (1/256 fractions); 0 = exact fp32 fractions */
} oracle_params;

/* cuSIFT.cu:313-353 + cuSIFT_D.cu:37-182.  dst is (w/2) x (h/2). */
void oracle_scale_down(const float *src, int w, int h, int src_pitch, float *dst, int dst_pitch);

/* cuSIFT.cu:399-412: the 8 x 9 tap table (row stride 16 floats, taps at [16*i + j + 4]).
 * Rule of this build (the reference yields NaN / an inverted kernel there): var <= 1e-6 => identity. */
void oracle_laplace_taps(float init_blur, float taps[8 * 16]);

/* cuSIFT.cu:399-422 + cuSIFT_D.cu:525-553: dog = 7 planes [7][h][pitch]; pad columns untouched. */
void oracle_laplace_multi(const float *img, int w, int h, int pitch, float init_blur, float *dog);

/* cuSIFT.cu:424-455 + cuSIFT_D.cu:402-523.  Appends header fields at points[*counter],
 * raster order (y, x, scale); slots >= max_pts are dropped but *counter keeps counting. */
void oracle_find_points_multi(const float *dog, int w, int h, int pitch, float peak_thresh, float edge_thresh,
                              float subsampling, oracle_sift_point *points, int max_pts, int *counter);

/* cuSIFT_D.cu:319-396 over points [first, last). */
void oracle_compute_orientations(const float *img, int w, int h, int pitch, oracle_sift_point *points, int first,
                                 int last, int tex_frac_bits);

/* Diagnostic tap of oracle_compute_orientations (per calling thread; NULL switches it off): diag[2*bx] = second /
 * first smoothed-histogram peak, diag[2*bx+1] = the orientation of the second peak in degrees (NaN if none). */
void oracle_set_orientation_diag(float *buf, int n_points);

/* cuSIFT_D.cu:184-297 over points [first, last); scales coords2D/scale by subsampling at the end. */
void oracle_extract_descriptors(const float *img, int w, int h, int pitch, oracle_sift_point *points, int first,
                                int last, float subsampling, int tex_frac_bits);

/* cuSIFT_D.cu:299-317 */
void oracle_rootsift(oracle_sift_point *points, int n);

/* cuSIFT.cu:61-120,175-270: full driver on a dense host image (row stride w). Returns numPts. */
int oracle_extract(const float *img, int w, int h, const oracle_params *prm, oracle_sift_point *points);

/* MatchSiftData (extras/matching.cu:232-362) -- the first "next" row after the extraction path.
 * distance: 0 = MatchSiftDistanceDotProduct, 1 = MatchSiftDistanceL2 (extras/matching.h:10-13).
 * Writes score, ambiguity, match, match_xpos, match_ypos of every point of sift1 (ComputeDistance :12-58 with its
 * rotated summation order, ComputeL2Distance :63-74, FindMaxCorr :76-152 / FindMinCorr :154-230).
 * Pinned by the reference's own fixtures: sift/sift{1,2} + match_indices1_2 and the 340-match ratio test
 * (test/test.cpp:25-56). */
void oracle_match_sift_data(oracle_sift_point *sift1, int n1, const oracle_sift_point *sift2, int n2, int distance);
/* The host-side filter of MatchSiftData (extras/matching.cu:318-349, MatchType2D): number of points of sift1
 * with score < scoreThreshold^2 && ambiguity < ambiguityThreshold^2; their indices go to `idx` if not NULL. */
int oracle_match_filter(const oracle_sift_point *sift1, int n1, float score_threshold, float ambiguity_threshold,
                        int *idx);

/* RANSAC homography (PARITY UNPINNED -- the reference has no test or fixture for it):
 * ComputeHomographies (extras/homography.cu:89-130, InvertMatrix<8> :3-87), TestHomographies (:135-178) and the
 * selection at the end of FindHomography (:237-258).  Layouts: coord [4][num_pts], rand_pts [4][num_loops],
 * homo [8][num_loops].  The random draws (host rand(), :222-235) are the caller's. */
void oracle_compute_homographies(const float *coord, int num_pts, const int *rand_pts, int num_loops, float *homo);
void oracle_test_homographies(const float *coord, int num_pts, const float *homo, int num_loops, float thresh2,
                              int *counts);
int oracle_find_homography(const oracle_sift_point *pts, int num_pts, const int *rand_pts, int num_loops, float thresh,
                           float homography[9], int *num_matches, float *all_homo, int *all_counts);

/* Caller-side front-end (main.cpp:300-318, test/detector.cpp:19-27): convertTo(CV_32FC1) and
 * cv::GaussianBlur(Size(3,3), sigma).  The blur is PARITY UNPINNED (OpenCV is absent here; see the .c file). */
void oracle_u8_to_f32(const unsigned char *src, int w, int h, int src_pitch, float *dst, int dst_pitch);
void oracle_gaussian3x3(const float *src, int w, int h, int src_pitch, float *dst, int dst_pitch, float sigma);

/* Software model of tex2D<float>(x, y) with cudaFilterModeLinear / clamp / unnormalised coords. */
float oracle_tex2d(const float *img, int w, int h, int pitch, float x, float y, int frac_bits);

/* The written-out transcendental functions shared with the HIP kernels (cusift_amd/csrc/sift_math.h), exported so
 * that tests can bound them against float64 (tests/test_math.py) and compare the device's results bit for bit. */
float oracle_math_expf(float x);
float oracle_math_exp2f(float x);
float oracle_math_atan2f(float y, float x);
void oracle_math_sincosf(float x, float *s, float *c);
void oracle_math_eval(int op, const float *a, const float *b, float *out, float *out2, int n);

#ifdef __cplusplus
}
#endif
#endif
