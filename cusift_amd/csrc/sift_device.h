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
