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
// The next tile's global loads are in flight while the current one is multiplied (register staging), and image 2's
// columns are split over blockIdx.y so that a few thousand keypoints still fill the chip; per-split results are
// folded in column order by match_merge_kernel (equal to the single scan except for exact score ties between
// columns of different splits).
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

// the tail of FindMinCorr/FindMaxCorr (extras/matching.cu:140-150,220-229): the five fields MatchSiftData fills
__device__ __forceinline__ void write_match(cusift_point *pt, const cusift_point *__restrict__ sift2, int n2,
                                            float best, float second, int idx, bool l2) {
  pt->score = best;
  // the 1e-6 is a double constant in the reference (:143,:222): evaluate in double, store float
  pt->ambiguity = l2 ? (float)(best / (second + 1e-6)) : (float)((1 - best) / (1 - second + 1e-6));
  pt->match = idx;
  const int m = (idx >= 0 && idx < n2) ? idx : 0;  // the reference reads sift2[-1] here
  pt->match_xpos = sift2[m].coords2D[0];
  pt->match_ypos = sift2[m].coords2D[1];
}

// 16-byte loads from the 588-byte records: descriptors are only 4-byte aligned
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

// The same update without branches: two compares and four selects per score (a NaN score changes nothing, as in the
// reference: both comparisons are false).
template <bool kL2>
__device__ __forceinline__ void top2_update(float &best, float &second, int &idx, float val, int i) {
  const bool win = kL2 ? (val < best) : (val > best);
  const bool place = kL2 ? (val < second) : (val > second);
  second = win ? best : (place ? val : second);
  idx = win ? i : idx;
  best = win ? val : best;
}

// One workgroup = 64 descriptors of image 1 (16 per wave) x one contiguous range of image 2's columns
// (blockIdx.y = column split; the host picks the number of splits so that the grid fills 256 CUs even for a few
// thousand keypoints).  With one split the kernel writes the final fields; otherwise the (best, second, index)
// partial of every row goes to `partials[split][row]` and match_merge_kernel folds the splits in column order.
template <bool kL2>
__global__ void __launch_bounds__(256) match_kernel(cusift_point *__restrict__ sift1, int n1,
                                                   const cusift_point *__restrict__ sift2, int n2,
                                                   int cols_per_split, MatchPartial *__restrict__ partials,
                                                   int n1_pad) {
  __shared__ float sB[kMatchTileCols * kBStride];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int p1_base = blockIdx.x * kMatchRowsPerBlock + wv * 16;
  const int col_begin = blockIdx.y * cols_per_split;
  const int col_end = min(col_begin + cols_per_split, n2);  // padded columns (:57) can never win: skip them
  constexpr float kInit = kL2 ? kMatchFltMax : -1.0f;       // also what a masked column scores: it changes nothing

  // A fragments: lane (r, g) holds elements 16u + 4g + j of descriptor p1_base + r (u = 0..7, j = 0..3)
  float a[8][4];
  {
    const float *d1 = sift1[min(p1_base + r, n1 - 1)].data;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const f4u v = *reinterpret_cast<const f4u *>(d1 + 16 * u + 4 * g);
#pragma unroll
      for (int j = 0; j < 4; ++j) a[u][j] = v[j];
    }
  }
  // running top-2 of rows 4g + q for the columns this lane sees (p2 = r mod 16): reference thread tx = r
  float best[4], second[4];
  int bidx[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    best[q] = second[q] = kInit;
    bidx[q] = -1;
  }

  // staging: thread t moves four 16-byte chunks per tile; chunk c = t + 256 i -> descriptor c >> 5, floats 4 (c & 31).
  // Raw buffer loads from a descriptor based at this split's first column: the lane offset is computed once, the tile
  // and the chunk row advance in the scalar offset, and columns past col_end read as 0 (their scores are masked below).
  const __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(sift2 + col_begin), 0, (int)((col_end > col_begin ? col_end - col_begin : 0) * sizeof(cusift_point)),
      kBufFlags);
  constexpr int kRec = (int)sizeof(cusift_point);
  const int voff = (int)(threadIdx.x >> 5) * kRec + (int)offsetof(cusift_point, data) + 16 * (int)(threadIdx.x & 31);
  u4 stage[4];
  auto fetch = [&](int c0) {
    const int soff = __builtin_amdgcn_readfirstlane((c0 - col_begin) * kRec);
#pragma unroll
    for (int i = 0; i < 4; ++i) stage[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc2, voff, soff + 8 * i * kRec, 0);
  };
  if (col_begin < col_end) fetch(col_begin);
  const float *brow = sB + r * kBStride + 4 * g;
  for (int c0 = col_begin; c0 < col_end; c0 += kMatchTileCols) {
    __syncthreads();  // the previous tile has been consumed
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = threadIdx.x + 256 * i;
      *reinterpret_cast<u4 *>(sB + (c >> 5) * kBStride + 4 * (c & 31)) = stage[i];
    }
    __syncthreads();
    if (c0 + kMatchTileCols < col_end) fetch(c0 + kMatchTileCols);  // in flight while this tile is multiplied

    // two independent 16x16 accumulators (columns c0 + r and c0 + 16 + r), each a k-ordered chain; the B fragments
    // of step u + 1 are read from LDS before the MFMAs of step u are issued
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_setprio(1);  // a wave with MFMAs to issue goes before its SIMD's waves that are in the update
    f4 b0 = *reinterpret_cast<const f4 *>(brow);
    f4 b1 = *reinterpret_cast<const f4 *>(brow + 16 * kBStride);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      f4 n0 = b0, n1v = b1;
      if (u < 7) {
        n0 = *reinterpret_cast<const f4 *>(brow + 16 * (u + 1));
        n1v = *reinterpret_cast<const f4 *>(brow + 16 * kBStride + 16 * (u + 1));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][j], b0[j], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][j], b1[j], acc1, 0, 0, 0);
      }
      b0 = n0;
      b1 = n1v;
    }
    __builtin_amdgcn_s_setprio(0);
    // acc[q] = <descriptor p1_base + 4g + q, descriptor p2>.  (Deferring this update into the next tile's MFMA gaps
    // -- one score per step u -- was built and measured: no change, 110-112 TFLOP/s at 16k either way.)
    {
      const bool full = c0 + kMatchTileCols <= col_end;  // wave-uniform: only a split's last tile can be partial
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int p2 = c0 + 16 * t + r;
        const bool live = full || p2 < col_end;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float dot = t ? acc1[q] : acc0[q];
          // ComputeL2Distance :71-72, 2 - 2*dot (2*dot is exact, so the fused form has the same bits)
          float val = kL2 ? (dot > -1.0f ? __builtin_fmaf(-2.0f, dot, 2.0f) : kMatchFltMax) : dot;
          val = live ? val : kInit;
          top2_update<kL2>(best[q], second[q], bidx[q], val, p2);
        }
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
        top2_scan(best[q], second[q], bidx[q], ob, oi, kL2);
        if (beats(os, second[q], kL2)) second[q] = os;
      }
    }
  }
  if (r == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p1 = p1_base + 4 * g + q;
      if (p1 >= n1) continue;
      if (partials) {
        MatchPartial mp;
        mp.best = best[q];
        mp.second = second[q];
        mp.idx = bidx[q];
        partials[(size_t)blockIdx.y * n1_pad + p1] = mp;
      } else {
        write_match(sift1 + p1, sift2, n2, best[q], second[q], bidx[q], kL2);
      }
    }
  }
}

template __global__ void match_kernel<false>(cusift_point *, int, const cusift_point *, int, int, MatchPartial *, int);
template __global__ void match_kernel<true>(cusift_point *, int, const cusift_point *, int, int, MatchPartial *, int);

// Folds the column splits of one row in column order: the same update the tree applies between lanes.
__global__ void __launch_bounds__(256) match_merge_kernel(cusift_point *__restrict__ sift1, int n1,
                                                         const cusift_point *__restrict__ sift2, int n2, int l2_mode,
                                                         const MatchPartial *__restrict__ partials, int n1_pad,
                                                         int n_splits) {
  const bool l2 = l2_mode != 0;
  const int p1 = blockIdx.x * 256 + threadIdx.x;
  if (p1 >= n1) return;
  MatchPartial m = partials[p1];
  for (int s = 1; s < n_splits; ++s) {
    const MatchPartial o = partials[(size_t)s * n1_pad + p1];
    top2_scan(m.best, m.second, m.idx, o.best, o.idx, l2);
    if (beats(o.second, m.second, l2)) m.second = o.second;
  }
  write_match(sift1 + p1, sift2, n2, m.best, m.second, m.idx, l2);
}

}  // namespace cusift
