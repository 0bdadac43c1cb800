// sift_match.hip -- brute-force descriptor matcher (the first consumer of SiftData).
// Reference: MatchSiftData, extras/matching.cu:232-362 = ComputeDistance (:12-58) + ComputeL2Distance (:63-74) +
// FindMaxCorr (:76-152) / FindMinCorr (:154-230).
//
// The reference materialises the numPts1 x numPts2 score matrix in memory and scans it a second time.  Here one
// kernel does both: a wave owns 16 descriptors of image 1 (their 128 floats stay in registers as MFMA A fragments),
// the workgroup streams image 2 through LDS in tiles of 32 descriptors, the 16x16 dot products come from
// v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fmaf chain, no reduced-precision path), and every lane keeps the
// running (best, second, index) of its rows for the columns it sees -- exactly what reference thread `tx` sees
// (columns tx, tx+16, ...), followed by the reference's tree over tx.  N1*N2*4 bytes of HBM traffic never exist.
//
// Numerics: the reference sums pt1[k]*pt2[k] starting at k = (p2 mod 16) and wrapping; the MFMA chain visits k in
// the order 16u+j, 16u+4+j, 16u+8+j, 16u+12+j (u = 0..7, j = 0..3).  Scores agree to ~1e-7; indices agree except
// for exact-score near-ties (tests bound both).
#include "sift_device.h"

namespace cusift {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMatchRowsPerBlock = 64;  // 4 waves x 16 descriptors of image 1
constexpr int kMatchTileCols = 32;      // descriptors of image 2 per LDS tile
constexpr int kBStride = 132;           // floats per LDS row: 16-byte aligned and conflict-free for ds_read_b128
constexpr float kMatchFltMax = 999.0f;  // extras/matching.cu:3

// (val, idx) beats (best) under the reference's strict comparison; L2 looks for minima, dot product for maxima
__device__ __forceinline__ bool beats(float val, float cur, bool l2) { return l2 ? (val < cur) : (val > cur); }

// FindMinCorr/FindMaxCorr inner update (extras/matching.cu:104-113,182-191)
__device__ __forceinline__ void top2_scan(float &best, float &second, int &idx, float val, int i, bool l2) {
  if (beats(val, best, l2)) {
    second = best;
    best = val;
    idx = i;
  } else if (beats(val, second, l2)) {
    second = val;
  }
}

__global__ void __launch_bounds__(256) match_kernel(cusift_point *__restrict__ sift1, int n1,
                                                   const cusift_point *__restrict__ sift2, int n2, int l2_mode) {
  __shared__ float sB[kMatchTileCols * kBStride];
  const bool l2 = l2_mode != 0;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int p1_base = blockIdx.x * kMatchRowsPerBlock + wv * 16;
  const int corr_width = ((n2 + 15) / 16) * 16;  // extras/matching.cu:254

  // A fragments: lane (r, g) holds elements 16u + 4g + j of descriptor p1_base + r (u = 0..7, j = 0..3)
  float a[8][4];
  {
    const float *d1 = sift1[min(p1_base + r, n1 - 1)].data;
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) a[u][j] = d1[16 * u + 4 * g + j];
  }
  // running top-2 of rows 4g + reg for the columns this lane sees (p2 = r mod 16): reference thread tx = r
  float best[4], second[4];
  int bidx[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    best[q] = second[q] = l2 ? kMatchFltMax : -1.0f;
    bidx[q] = -1;
  }

  for (int c0 = 0; c0 < corr_width; c0 += kMatchTileCols) {
    __syncthreads();  // the previous tile has been consumed
    for (int e = threadIdx.x; e < kMatchTileCols * 128; e += 256) {
      const int row = e >> 7, k = e & 127;
      sB[row * kBStride + k] = (c0 + row < n2) ? sift2[c0 + row].data[k] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < kMatchTileCols / 16; ++t) {
      if (c0 + 16 * t >= corr_width) break;  // wave-uniform
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const float *brow = sB + (16 * t + r) * kBStride + 4 * g;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const f4 b = *reinterpret_cast<const f4 *>(brow + 16 * u);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][j], b[j], acc, 0, 0, 0);
      }
      // acc[q] = <descriptor p1_base + 4g + q, descriptor c0 + 16t + r>
      const int p2 = c0 + 16 * t + r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float val;
        if (p2 < n2) {
          const float dot = acc[q];
          val = l2 ? (dot > -1.0f ? 2 - 2 * dot : kMatchFltMax) : dot;  // ComputeL2Distance :71-72
        } else {
          val = l2 ? kMatchFltMax : -1.0f;  // padded columns, ComputeDistance :57
        }
        top2_scan(best[q], second[q], bidx[q], val, p2, l2);
      }
    }
  }
  // tree over tx = r (extras/matching.cu:122-138,201-218): lane r < len takes lane r + len; ties keep the lower r
#pragma unroll
  for (int len = 8; len > 0; len >>= 1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float ob = __shfl_down(best[q], len, 16);
      const float os = __shfl_down(second[q], len, 16);
      const int oi = __shfl_down(bidx[q], len, 16);
      if (r < len) {
        top2_scan(best[q], second[q], bidx[q], ob, oi, l2);
        if (beats(os, second[q], l2)) second[q] = os;
      }
    }
  }
  if (r == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p1 = p1_base + 4 * g + q;
      if (p1 < n1) {
        cusift_point *pt = sift1 + p1;
        pt->score = best[q];
        // the 1e-6 is a double constant in the reference (:143,:222): evaluate in double, store float
        pt->ambiguity = l2 ? (float)(best[q] / (second[q] + 1e-6)) : (float)((1 - best[q]) / (1 - second[q] + 1e-6));
        pt->match = bidx[q];
        const int m = (bidx[q] >= 0 && bidx[q] < n2) ? bidx[q] : 0;  // the reference reads sift2[-1] here
        pt->match_xpos = sift2[m].coords2D[0];
        pt->match_ypos = sift2[m].coords2D[1];
      }
    }
  }
}

}  // namespace cusift
