// sift_homography.hip -- RANSAC homography from matched SiftData.
// Reference: FindHomography, extras/homography.cu:182-269, with ComputeHomographies (:89-130, one 8x8 solve per
// hypothesis through InvertMatrix<8>, :3-87) and TestHomographies (:135-178, inlier count per hypothesis).
//
// Arithmetic follows oracle/sift_oracle.c operation by operation (fused multiply-adds only where written out,
// -ffp-contract=off; the inlier test multiplies with round-toward-zero like the reference's __fmul_rz).
//
// The work is tiny (1000 hypotheses x a few thousand points), so the design goal is only "no scratch memory":
// the reference keeps two 8x8 matrices per thread in local memory with data-dependent row swaps; here they live
// in LDS, entry-major / thread-minor (conflict-free), one 64-thread workgroup per 64 hypotheses.
#include "sift_device.h"

namespace cusift {

// coords2D and match_xpos/ypos of every record -> SoA rows [x1 | y1 | x2 | y2], extras/homography.cu:237-240
__global__ void __launch_bounds__(256) homography_gather_kernel(const cusift_point *__restrict__ pts, int num_pts,
                                                               float *__restrict__ coord) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num_pts) return;
  coord[i] = pts[i].coords2D[0];
  coord[i + num_pts] = pts[i].coords2D[1];
  coord[i + 2 * num_pts] = pts[i].match_xpos;
  coord[i + 3 * num_pts] = pts[i].match_ypos;
}

constexpr int kHomoThreads = 64;

__global__ void __launch_bounds__(kHomoThreads) homography_solve_kernel(const float *__restrict__ coord, int num_pts,
                                                                       const int *__restrict__ rand_pts,
                                                                       int num_loops, float *__restrict__ homo) {
  __shared__ float s_m[64 * kHomoThreads];    // the system matrix, LU-decomposed in place
  __shared__ float s_inv[64 * kHomoThreads];  // its inverse
  __shared__ float s_scale[8 * kHomoThreads];
  __shared__ float s_rhs[8 * kHomoThreads];
  __shared__ int s_perm[8 * kHomoThreads];
  const int tx = threadIdx.x;
  const int idx = blockIdx.x * kHomoThreads + tx;
  if (idx >= num_loops) return;  // no barrier below: every thread touches only its own LDS column
#define M(i, j) s_m[((i) * 8 + (j)) * kHomoThreads + tx]
#define INV(i, j) s_inv[((i) * 8 + (j)) * kHomoThreads + tx]
#define SCALE(i) s_scale[(i) * kHomoThreads + tx]
#define RHS(i) s_rhs[(i) * kHomoThreads + tx]
#define PERM(i) s_perm[(i) * kHomoThreads + tx]
  float b[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int pt = rand_pts[i * num_loops + idx];
    pt = clampi(pt, 0, num_pts - 1);  // memory safety only: the caller's indices are in range
    const float x1 = coord[pt], y1 = coord[pt + num_pts];
    const float x2 = coord[pt + 2 * num_pts], y2 = coord[pt + 3 * num_pts];
    M(2 * i, 0) = x1, M(2 * i, 1) = y1, M(2 * i, 2) = 1.0f;
    M(2 * i, 3) = 0.0f, M(2 * i, 4) = 0.0f, M(2 * i, 5) = 0.0f;
    M(2 * i, 6) = -x2 * x1, M(2 * i, 7) = -x2 * y1;
    M(2 * i + 1, 0) = 0.0f, M(2 * i + 1, 1) = 0.0f, M(2 * i + 1, 2) = 0.0f;
    M(2 * i + 1, 3) = x1, M(2 * i + 1, 4) = y1, M(2 * i + 1, 5) = 1.0f;
    M(2 * i + 1, 6) = -y2 * x1, M(2 * i + 1, 7) = -y2 * y1;
    b[2 * i] = x2;
    b[2 * i + 1] = y2;
  }
  // ---- InvertMatrix<8>: implicit row scaling (:17-26) ----
  int imax = 0;
  for (int i = 0; i < 8; ++i) {
    PERM(i) = 0;
    float big = 0.0f;
    for (int j = 0; j < 8; ++j) {
      const float t = fabsf(M(i, j));
      if (t > big) big = t;
    }
    SCALE(i) = big > 0.0f ? (float)(1.0 / (double)big) : 1e16f;
  }
  // ---- Crout LU with partial pivoting (:27-65) ----
  for (int j = 0; j < 8; ++j) {
    for (int i = 0; i < j; ++i) {
      float sum = M(i, j);
      for (int k = 0; k < i; ++k) sum = fmaf(-M(i, k), M(k, j), sum);
      M(i, j) = sum;
    }
    float big = 0.0f;
    for (int i = j; i < 8; ++i) {
      float sum = M(i, j);
      for (int k = 0; k < j; ++k) sum = fmaf(-M(i, k), M(k, j), sum);
      M(i, j) = sum;
      const float dum = SCALE(i) * fabsf(sum);
      if (dum >= big) {
        big = dum;
        imax = i;
      }
    }
    if (j != imax) {
      for (int k = 0; k < 8; ++k) {
        const float t = M(imax, k);
        M(imax, k) = M(j, k);
        M(j, k) = t;
      }
      SCALE(imax) = SCALE(j);
    }
    PERM(j) = imax;
    if (M(j, j) == 0.0f) M(j, j) = 1e-16f;
    if (j != 7) {
      const float dum = (float)(1.0 / (double)M(j, j));
      for (int i = j + 1; i < 8; ++i) M(i, j) *= dum;
    }
  }
  // ---- solve for the 8 columns of the identity (:66-86) ----
  for (int c = 0; c < 8; ++c) {
    for (int k = 0; k < 8; ++k) RHS(k) = 0.0f;
    RHS(c) = 1.0f;
    int first = -1;
    for (int i = 0; i < 8; ++i) {
      const int ip = PERM(i);
      float sum = RHS(ip);
      RHS(ip) = RHS(i);
      if (first != -1) {
        for (int k = first; k < i; ++k) sum = fmaf(-M(i, k), RHS(k), sum);
      } else if (sum != 0.0f) {
        first = i;
      }
      RHS(i) = sum;
    }
    for (int i = 7; i >= 0; --i) {
      float sum = RHS(i);
      for (int k = i + 1; k < 8; ++k) sum = fmaf(-M(i, k), RHS(k), sum);
      RHS(i) = sum / M(i, i);
    }
    for (int i = 0; i < 8; ++i) INV(i, c) = RHS(i);
  }
  // ---- homography = inverse * b (:123-128) ----
  for (int j = 0; j < 8; ++j) {
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) sum = fmaf(INV(j, i), b[i], sum);
    homo[j * num_loops + idx] = sum;
  }
#undef M
#undef INV
#undef SCALE
#undef RHS
#undef PERM
}

// a*b rounded toward zero (__fmul_rz): the double product of two floats is exact; round it to nearest, then step
// back towards zero if that overshot.  Same formulation as the oracle's mul_rz.
__device__ __forceinline__ float mul_rz(float a, float b) {
  const double p = (double)a * (double)b;
  float f = (float)p;
  if (fabs((double)f) > fabs(p)) f = nextafterf(f, 0.0f);
  return f;
}

// One wave per hypothesis; lanes stride over the points (coalesced SoA reads), counts are integers so the order of
// the reduction is immaterial.  TestHomographies, extras/homography.cu:135-178 (only the num_pts real points are
// tested: the reference also walks up to 15 uninitialised padding entries, :152).
__global__ void __launch_bounds__(64) homography_test_kernel(const float *__restrict__ coord, int num_pts,
                                                            const float *__restrict__ homo, int num_loops,
                                                            float thresh2, int *__restrict__ counts) {
  const int idx = blockIdx.x;
  if (idx >= num_loops) return;
  const int lane = threadIdx.x;
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = homo[i * num_loops + idx];
  int cnt = 0;
  for (int i = lane; i < num_pts; i += 64) {
    const float x1 = coord[i], y1 = coord[i + num_pts], x2 = coord[i + 2 * num_pts], y2 = coord[i + 3 * num_pts];
    const float nomx = mul_rz(a[0], x1) + mul_rz(a[1], y1) + a[2];
    const float nomy = mul_rz(a[3], x1) + mul_rz(a[4], y1) + a[5];
    const float deno = mul_rz(a[6], x1) + mul_rz(a[7], y1) + 1.0f;
    const float errx = mul_rz(x2, deno) - nomx;
    const float erry = mul_rz(y2, deno) - nomy;
    const float err2 = mul_rz(errx, errx) + mul_rz(erry, erry);
    if (err2 < mul_rz(thresh2, mul_rz(deno, deno))) ++cnt;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
  if (lane == 0) counts[idx] = cnt;
}

}  // namespace cusift
