// sift_device.h -- device-side helpers shared by the stencil kernels (sift_stencils.hip) and the keypoint
// kernels (sift_keypoints.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "sift_math.h"  // expf / exp2f / atan2f / sincosf written out: the same bits as the CPU oracle
#include "sift_types.h"

namespace cusift {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

constexpr unsigned int kBufFlags = 0x00020000u;  // raw buffer, 32-bit data format (gfx9/CDNA dword 3)
constexpr int kOobOffset = 0x7fffffff;           // lane offset beyond any num_records: store dropped

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// global row -> row of the local band: clamp to the global image (the reference's border rule), translate,
// then clamp into the band (memory safety only: with enough halo the second clamp never acts)
__device__ __forceinline__ int local_row(int y_global, int h_local, RowWindow rw) {
  return clampi(clampi(y_global, 0, rw.hg - 1) - rw.row0, 0, h_local - 1);
}

// lane i receives the value of lane i-1 (lane 0 receives 0): DPP wave_shr:1.  bound_ctrl makes the hardware
// write 0 for the lane without a source, so no "old" value has to be materialised in front of every DPP move.
__device__ __forceinline__ float from_prev_lane(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
// lane i receives the value of lane i+1 (lane 63 receives 0): DPP wave_shl:1
__device__ __forceinline__ float from_next_lane(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// Right/left image border for lanes that own GROUPS of 4 (or 2) adjacent columns of a 16-byte (8-byte) aligned row,
// for ANY width w >= 4 (>= 2): the lane's vector load is clamped to the last column group that starts inside the
// image -- always aligned and inside the pitch, because pitch % 4 == 0 (% 2) -- and the components that lie outside
// the image are overwritten with the border column: clamp addressing, exactly what the reference's texture unit and
// the generic kernels' per-column clampi() do (cuSIFT_D.cu:534-548).  Lanes whose group lies inside the image are
// untouched; the masks are loop-invariant per lane.
struct EdgeFix4 {
  int voff;    // byte offset of the lane's float4 inside a row, clamped into the image
  int src;     // component that holds the border column for this lane
  bool m0, m1, m2, m3;  // component j is replaced by the border column
  __device__ __forceinline__ EdgeFix4(int c0, int w) {
    const int cA = (w - 1) & ~3;  // first column of the last, possibly partial, group
    const bool left = c0 < 0;
    voff = clampi(c0, 0, cA) * 4;
    src = left ? 0 : ((w - 1) & 3);
    m0 = left || c0 + 0 > w - 1;
    m1 = left || c0 + 1 > w - 1;
    m2 = left || c0 + 2 > w - 1;
    m3 = left || c0 + 3 > w - 1;
  }
  __device__ __forceinline__ f4 operator()(f4 v) const {
    const float s = src == 0 ? v.x : (src == 1 ? v.y : (src == 2 ? v.z : v.w));
    return f4{m0 ? s : v.x, m1 ? s : v.y, m2 ? s : v.z, m3 ? s : v.w};
  }
};
struct EdgeFix2 {
  int voff;
  int src;
  bool m0, m1;
  __device__ __forceinline__ EdgeFix2(int c0, int w) {
    const int cA = (w - 1) & ~1;
    const bool left = c0 < 0;
    voff = clampi(c0, 0, cA) * 4;
    src = left ? 0 : ((w - 1) & 1);
    m0 = left || c0 + 0 > w - 1;
    m1 = left || c0 + 1 > w - 1;
  }
  __device__ __forceinline__ f2 operator()(f2 v) const {
    const float s = src == 0 ? v.x : v.y;
    return f2{m0 ? s : v.x, m1 ? s : v.y};
  }
};

// XCD-aware work mapping.  Hardware deals consecutive workgroup ids round-robin over the 8 XCDs (each
// with its own L2), so spatially adjacent tiles would land on different L2s and fetch their shared
// halo lines twice.  Remapping id -> (id % 8) * (total / 8) + id / 8 gives every XCD a contiguous
// run of tiles, dispatched in order, so neighbours meet in one L2.  Placement only affects speed.
__device__ __forceinline__ void xcd_remap(int &bx, int &by, int &bz) {
  const int nx = gridDim.x, ny = gridDim.y, nz = gridDim.z;
  const int total = nx * ny * nz;
  if ((total & 7) != 0) return;
  int id = (bz * ny + by) * nx + bx;
  id = (id & 7) * (total >> 3) + (id >> 3);
  bx = id % nx;
  const int t = id / nx;
  by = t % ny;
  bz = t / ny;
}

}  // namespace cusift
