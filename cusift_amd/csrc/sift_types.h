// sift_types.h -- shared host/device declarations of the MI355X SIFT extraction path.
//
// Arithmetic convention (identical to oracle/sift_oracle.c so the filter stages are bit-exact):
// filter sums are "first product, then a left-to-right fmaf chain"; everything else is evaluated
// operation by operation (the translation units are built with -ffp-contract=off).
#pragma once

#include <stddef.h>
#include <stdint.h>

#include "../../include/cusift_amd_all.h"

namespace cusift {

constexpr int kNumScales = 5;                 // cuSIFT_D.h:8   NUM_SCALES
constexpr int kNumLevels = kNumScales + 3;    // cuSIFT_D.h:23  LAPLACE_S: blurred levels per octave
constexpr int kNumDog = kNumLevels - 1;       // DoG planes per octave
constexpr int kBlurRadius = 4;                // cuSIFT_D.h:26  LAPLACE_R
constexpr int kMaxOctaves = 16;

// Column-strip geometry of the wave-autonomous stencil kernels:
// a 64-lane wave owns 64*kBlurCols consecutive columns.  The blur needs +-4 columns of halo (one lane per
// side), but the wave keeps FOUR halo lanes per side so that the 224 columns it writes start and end on
// 128-byte lines: measured on MI355X (tools/microbench/stream_mix.hip), 7-plane stores in 992-byte
// unaligned row segments sustain 3.4 TB/s, in 896-byte line-aligned segments 4.3 TB/s.
constexpr int kBlurCols = 4;        // columns per lane in the blur kernel (float4)
constexpr int kBlurHaloLanes = 4;   // halo lanes on each side of a wave
constexpr int kBlurStrip = (64 - 2 * kBlurHaloLanes) * kBlurCols;  // 224 valid output columns per wave
constexpr int kFindCols = 2;                              // columns per lane in the extrema kernel (float2)
constexpr int kFindStrip = 64 * kFindCols - 2 * kFindCols;  // 124 valid output columns per wave
constexpr int kWavesPerBlock = 4;

struct LaplaceTaps {
  // k[s][0..4] = reference kernel[16*s + 0..4]; k[s][4] is the centre tap, k[s][4-j] the tap at +-j
  // (cuSIFT.cu:400-412; only the lower half of each 9-tap row is read by the kernel, cuSIFT_D.cu:536-548)
  float k[kNumLevels][5];
};

struct LaplaceTapsPk {
  // the same taps arranged for packed fp32: k[q][t] = (k[2q][t], k[2q+1][t]) -- one SGPR pair per
  // (scale pair, tap), consumed by v_pk_fma_f32 in laplace_multi_fast_kernel
  float __attribute__((ext_vector_type(2))) k[kNumLevels / 2][5];
};

struct ScaleDownTaps {
  float k[3];  // k0, k1, k2 of cuSIFT.cu:330-341 (k[2] centre)
};

// Where detect_fused_kernel<.., kDown = true> writes the next octave's image while it searches this one (sift_stencils.hip)
struct DownOut {
  float *dst;       // image 0 of the next octave (w/2 x h/2); unused by the kDown = false instantiations
  int pitch;        // floats (even: float2 stores)
  long stride;      // floats between images (even)
  ScaleDownTaps T;  // cuSIFT.cu:330-341
};

// A device image may be a horizontal band of a larger ("global") image: its local row 0 is global row
// `row0` and the global image has `hg` rows.  Whole images use {0, h}.  Row addressing everywhere is
// "clamp to the global image, then translate" so that a band with enough halo rows behaves exactly like
// the whole image (used by the multi-GPU strip tiling).
struct RowWindow {
  int row0;
  int hg;
};

// Per-octave base images of a batch, passed by value to describe_all_kernel (driver path).
#ifdef CUSIFT_STAMPS
constexpr int kMaxFlatImages = 128;  // (the phase-stamp build's counters take LDS of the prefix table)
#else
constexpr int kMaxFlatImages = 256;  // batch size up to which the flattened keypoint kernel is used
#endif
constexpr int kDetectWaveLdsFloats = 128 * 21;  // detect_fused_kernel: candidate list per wave (128 entries of 21 words)
constexpr int kStagedRecBytes = 64;   // a staged keypoint: the first 16 floats of a cusift_point (coords2D .. subsampling)
// Where the keypoints of a batch wait before describe_all_kernel puts them in place (cusift_extract_batch).  A batch's
// SiftData is a sequence of SEGMENTS in list order -- coarsest octave first.  Segment r of image i holds
// seg_end[i * n_seg + r] - seg_end[i * n_seg + r - 1] keypoints (join_counts_kernel clamps the sequence at max_pts, so
// the coarser segments survive whole); its keypoints wait as record heads at base[r] + (i * max_pts + j) * kStagedRecBytes,
// or -- base[r] == NULL, segment 0 only -- already sit in the caller's records.
struct SegmentTable {
  int n_seg;                          // 0: no table, everything is in place
  const char *base[kMaxOctaves];      // staging list of segment r ([image][max_pts] heads), or NULL
  const unsigned int *count[kMaxOctaves];  // per-image counters the detection of segment r incremented (unclamped)
};
// Row windows of the octaves of describe_bands_kernel (one strip-tiled image: every octave is a band of its own).
struct BandWindows {
  int row0[kMaxOctaves], hg[kMaxOctaves];
};
constexpr int kQueueShards = 64;      // work cursors of describe_all_kernel (a power of two)
struct OctaveTable {
  const float *base[kMaxOctaves];  // image 0 of octave o
  long stride[kMaxOctaves];        // floats between images
  int w[kMaxOctaves], h[kMaxOctaves], pitch[kMaxOctaves];
  float sub[kMaxOctaves];          // subsampling of octave o (sub[0] * 2^o)
  int n_oct;
};

struct FindParams {
  float thr_pos;            // d_Threshold[0] = +peakThresh   (cuSIFT.cu:432)
  float thr_neg;            // d_Threshold[1] = -peakThresh
  float edge_limit;         // d_EdgeLimit                    (cuSIFT.cu:442)
  float factor;             // d_Factor = 1/NUM_SCALES        (cuSIFT.cu:444)
  float scales[kNumScales]; // d_Scales                       (cuSIFT.cu:433-443)
  float subsampling;
};

// The ScaleDown chain of a SMALL call in one launch (pyramid_small_kernel): level k+1 from level k, k = 0 .. n-1.
constexpr int kMaxPyramidLevels = 4;
struct PyramidLevels {
  int n;                                  // levels produced (1..kMaxPyramidLevels)
  int tile;                               // side of a workgroup's square of level n
  int w[kMaxPyramidLevels + 1], h[kMaxPyramidLevels + 1], pitch[kMaxPyramidLevels + 1];
  long stride[kMaxPyramidLevels + 1];     // floats between images
  float *base[kMaxPyramidLevels + 1];     // image 0 of level k (base[0]: the source, read only)
};

// One launch for several octaves of the fused detection (detect_multi_kernel): per octave what detect_fused_kernel takes
// as arguments, plus the octave's range of workgroups.  Passed by value (kernarg: <= 4 KB).
struct DetectOctave {
  const float *img;        // image 0 of the octave
  long img_stride;         // floats between images
  char *lists;             // image 0's keypoint list of this octave (records or heads, the kernel's kRecBytes apart)
  unsigned int *counters;  // [n_images]
  int w, h, pitch;
  int rows_per_wave;       // centre rows per workgroup (one wave each)
  int strips, chunks;      // workgroups of this octave: strips x chunks x images, strip fastest
  int first_block;         // its first workgroup in the launch
  int ident;               // levels 0 and 1 of T are the identity (initBlur >= their sigma): the pass-through body
  // the image may be a horizontal band of a larger one (RowWindow): local row 0 = global row row0 of hg rows; extremum
  // centres are global rows [cy_begin, cy_end).  Whole images: 0, h, 0, h.
  int row0, hg, cy_begin, cy_end;
  LaplaceTapsPk T;
  FindParams P;
};
constexpr int kMaxMultiOctaves = 8;
struct DetectTable {
  int n;
  DetectOctave o[kMaxMultiOctaves];
};
static_assert(sizeof(DetectTable) <= 3072, "DetectTable travels as a kernel argument");

// per-(column split, row) result of the matcher, folded by match_merge_kernel
struct MatchPartial {
  float best, second;
  int idx;
};

static_assert(sizeof(cusift_point) == 588, "SiftPoint is a 588-byte ABI record (cuSIFT.h:10-30)");

}  // namespace cusift
