// sift_types.h -- shared host/device declarations of the MI355X SIFT extraction path.
//
// Arithmetic convention (identical to oracle/sift_oracle.c so the filter stages are bit-exact):
// filter sums are "first product, then a left-to-right fmaf chain"; everything else is evaluated
// operation by operation (the translation units are built with -ffp-contract=off).
#pragma once

#include <stddef.h>
#include <stdint.h>

#include "../../include/cusift_amd.h"

namespace cusift {

constexpr int kNumScales = 5;                 // cuSIFT_D.h:8   NUM_SCALES
constexpr int kNumLevels = kNumScales + 3;    // cuSIFT_D.h:23  LAPLACE_S: blurred levels per octave
constexpr int kNumDog = kNumLevels - 1;       // DoG planes per octave
constexpr int kBlurRadius = 4;                // cuSIFT_D.h:26  LAPLACE_R
constexpr int kMaxOctaves = 16;

// Column-strip geometry of the wave-autonomous stencil kernels:
// a 64-lane wave owns 64*kBlurCols consecutive columns, of which the outer lane on each side is halo.
constexpr int kBlurCols = 4;                              // columns per lane in the blur kernel (float4)
constexpr int kBlurStrip = 64 * kBlurCols - 2 * kBlurCols;  // 248 valid output columns per wave
constexpr int kFindCols = 2;                              // columns per lane in the extrema kernel (float2)
constexpr int kFindStrip = 64 * kFindCols - 2 * kFindCols;  // 124 valid output columns per wave
constexpr int kWavesPerBlock = 4;

struct LaplaceTaps {
  // k[s][0..4] = reference kernel[16*s + 0..4]; k[s][4] is the centre tap, k[s][4-j] the tap at +-j
  // (cuSIFT.cu:400-412; only the lower half of each 9-tap row is read by the kernel, cuSIFT_D.cu:536-548)
  float k[kNumLevels][5];
};

struct ScaleDownTaps {
  float k[3];  // k0, k1, k2 of cuSIFT.cu:330-341 (k[2] centre)
};

struct FindParams {
  float thr_pos;            // d_Threshold[0] = +peakThresh   (cuSIFT.cu:432)
  float thr_neg;            // d_Threshold[1] = -peakThresh
  float edge_limit;         // d_EdgeLimit                    (cuSIFT.cu:442)
  float factor;             // d_Factor = 1/NUM_SCALES        (cuSIFT.cu:444)
  float scales[kNumScales]; // d_Scales                       (cuSIFT.cu:433-443)
  float subsampling;
};

static_assert(sizeof(cusift_point) == 588, "SiftPoint is a 588-byte ABI record (cuSIFT.h:10-30)");

}  // namespace cusift
