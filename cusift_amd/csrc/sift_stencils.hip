// sift_stencils.hip -- hand-written gfx950 (CDNA4) kernels of the SIFT extraction path: the image-sized
// stencils (ScaleDown, blur+DoG, extrema, and their fusion).  Keypoint kernels: sift_keypoints.hip.
//
// Design (DESIGN.md has the full account):
//  * The two HBM-bound stencils (blur+DoG, extrema) are "wave-autonomous column strips": a 64-lane
//    wave owns a strip of columns (4 or 2 per lane, read as float4/float2 = 1 KiB/512 B per wave
//    instruction), marches down its rows with a register sliding window and gets its horizontal
//    neighbours from the adjacent lanes with DPP wave shifts.  No LDS, no barriers, no re-reads
//    except the row/column halo; every wave is independent so occupancy hides HBM latency.
//  * Keypoint kernels (orientation, descriptor) run one wave per keypoint on a persistent grid that
//    reads the point count from device memory -- no host read-back between stages.
//  * Arithmetic follows oracle/sift_oracle.c operation by operation (explicit fmaf chains in the
//    filters, nothing else fused: this file is built with -ffp-contract=off); transcendental functions are
//    the written-out ones of sift_math.h, which the oracle compiles too.
#include <type_traits>

#include "sift_device.h"

namespace cusift {

// ------------------------------------------------------------------------------------------------
// ScaleDown: 5x5 separable low-pass + 2x decimation.  Reference: ScaleDown_D, cuSIFT_D.cu:37-182.
// One lane per output column, marching down output rows; the five horizontally filtered source rows
// (2r-1 .. 2r+3) live in registers and slide by two per output row.
//   horizontal (cuSIFT_D.cu:111-113): k0*(S[2c-2]+S[2c+2]) + k1*(S[2c-1]+S[2c+1]) + k2*S[2c]
//   vertical   (cuSIFT_D.cu:123-125): k2*B[2r] + k0*(B[2r+3]+B[2r+2]) + k1*(B[2r-1]+B[2r+1])
// (the asymmetric vertical support is the reference's: yRead = yStart + tx - 1, cuSIFT_D.cu:75).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) scale_down_kernel(float *__restrict__ dst, int dst_pitch, long dst_stride,
                                                        const float *__restrict__ src, int w, int h, int src_pitch,
                                                        long src_stride, int rows_per_wave, ScaleDownTaps T) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave id: uniform, say so
  const int ow = w >> 1, oh = h >> 1;
  const int c = blockIdx.x * 64 + lane;
  const int r0 = (blockIdx.y * kWavesPerBlock + wv) * rows_per_wave;
  if (r0 >= oh) return;
  const int r1 = min(r0 + rows_per_wave, oh);
  src += (long)blockIdx.z * src_stride;
  dst += (long)blockIdx.z * dst_stride;
  const int x0 = clampi(2 * c - 2, 0, w - 1), x1 = clampi(2 * c - 1, 0, w - 1), x2 = clampi(2 * c, 0, w - 1),
            x3 = clampi(2 * c + 1, 0, w - 1), x4 = clampi(2 * c + 2, 0, w - 1);
  const float k0 = T.k[0], k1 = T.k[1], k2 = T.k[2];
  auto hrow = [&](int y) -> float {
    const float *s = src + (long)clampi(y, 0, h - 1) * src_pitch;
    float v = k0 * (s[x0] + s[x4]);
    v = fmaf(k1, s[x1] + s[x3], v);
    v = fmaf(k2, s[x2], v);
    return v;
  };
  float bm1 = hrow(2 * r0 - 1), b0 = hrow(2 * r0), bp1 = hrow(2 * r0 + 1);
  for (int r = r0; r < r1; ++r) {
    const float bp2 = hrow(2 * r + 2), bp3 = hrow(2 * r + 3);
    float v = k2 * b0;
    v = fmaf(k0, bp3 + bp2, v);
    v = fmaf(k1, bm1 + bp1, v);
    if (c < ow) dst[(long)r * dst_pitch + c] = v;
    bm1 = bp1;
    b0 = bp2;
    bp1 = bp3;
  }
}

// ------------------------------------------------------------------------------------------------
// LaplaceMulti: 8 Gaussian blurs (9-tap separable, vertical then horizontal, clamp borders) of the
// octave base image and the 7 differences, fused.  Reference: LaplaceMulti_D, cuSIFT_D.cu:525-553.
//
// Each lane owns kBlurCols=4 adjacent columns; a wave owns 256 columns of which lanes 4..59 (224
// columns, 128-byte aligned) produce output and the outer lanes are halo (sift_types.h).  Per row: one float4 load per lane
// (1 KiB per wave), a 9-row register window, the scale-independent pair sums S[y-k]+S[y+k], then per
// level the vertical 9-tap (5 multiplies), the +-4 column exchange with the neighbouring lanes by
// DPP, the horizontal 9-tap and the DoG against the previous level; 7 float4 stores per row.
// Algorithmic HBM bytes: 4 read + 28 written per pixel.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) laplace_multi_kernel(const float *__restrict__ img, float *__restrict__ dog,
                                                           int w, int h, int pitch, long img_stride, long dog_stride,
                                                           int rows_per_wave, int vec_ok, LaplaceTaps T) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave id: uniform, say so
  const int y0 = (blockIdx.y * kWavesPerBlock + wv) * rows_per_wave;
  if (y0 >= h) return;  // wave-uniform
  const int y1 = min(y0 + rows_per_wave, h);
  img += (long)blockIdx.z * img_stride;
  dog += (long)blockIdx.z * dog_stride;

  const int c0 = blockIdx.x * kBlurStrip - kBlurHaloLanes * kBlurCols + lane * kBlurCols;  // first of this lane's 4 columns
  const bool interior = vec_ok && c0 >= 0 && c0 + 3 < w;
  const int cc0 = clampi(c0, 0, w - 1), cc1 = clampi(c0 + 1, 0, w - 1), cc2 = clampi(c0 + 2, 0, w - 1),
            cc3 = clampi(c0 + 3, 0, w - 1);

  float win[9][4];
  auto load_row = [&](int y, float (&o)[4]) {
    const float *r = img + (long)clampi(y, 0, h - 1) * pitch;
    if (interior) {
      const float4 v = *reinterpret_cast<const float4 *>(r + c0);
      o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    } else {
      o[0] = r[cc0]; o[1] = r[cc1]; o[2] = r[cc2]; o[3] = r[cc3];
    }
  };
#pragma unroll
  for (int i = 0; i < 8; ++i) load_row(y0 - 4 + i, win[i]);

  const bool writer = lane >= kBlurHaloLanes && lane < 64 - kBlurHaloLanes && c0 < w;
  const bool vec_store = vec_ok && (c0 + 3 < w);
  const long plane = (long)h * pitch;

  for (int y = y0; y < y1; ++y) {
    load_row(y + 4, win[8]);
    float ctr[4], p1[4], p2[4], p3[4], p4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ctr[j] = win[4][j];
      p1[j] = win[3][j] + win[5][j];
      p2[j] = win[2][j] + win[6][j];
      p3[j] = win[1][j] + win[7][j];
      p4[j] = win[0][j] + win[8][j];
    }
    float prevL[4];
    float *out = dog + (long)y * pitch + c0;
#pragma unroll
    for (int s = 0; s < kNumLevels; ++s) {
      const float k0 = T.k[s][0], k1 = T.k[s][1], k2 = T.k[s][2], k3 = T.k[s][3], k4 = T.k[s][4];
      float e[12];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = k4 * ctr[j];
        v = fmaf(k3, p1[j], v);
        v = fmaf(k2, p2[j], v);
        v = fmaf(k1, p3[j], v);
        v = fmaf(k0, p4[j], v);
        e[4 + j] = v;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        e[j] = from_prev_lane(e[4 + j]);
        e[8 + j] = from_next_lane(e[4 + j]);
      }
      float L[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = 4 + j;
        float v = k4 * e[m];
        v = fmaf(k3, e[m - 1] + e[m + 1], v);
        v = fmaf(k2, e[m - 2] + e[m + 2], v);
        v = fmaf(k1, e[m - 3] + e[m + 3], v);
        v = fmaf(k0, e[m - 4] + e[m + 4], v);
        L[j] = v;
      }
      if (s > 0 && writer) {
        float *o = out + (long)(s - 1) * plane;
        if (vec_store) {
          *reinterpret_cast<float4 *>(o) =
              make_float4(prevL[0] - L[0], prevL[1] - L[1], prevL[2] - L[2], prevL[3] - L[3]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (c0 + j < w) o[j] = prevL[j] - L[j];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) prevL[j] = L[j];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) win[i][j] = win[i + 1][j];
  }
}

// ------------------------------------------------------------------------------------------------
// LaplaceMulti, fast path (16-byte aligned rows: pitch % 4 == 0, any w >= 4, image and DoG plane < 2 GiB).
// Same strip geometry and the same arithmetic as laplace_multi_kernel above, restructured for the
// CDNA4 VALU and memory pipes:
//  * packed fp32 (v_pk_fma_f32 / v_pk_add_f32) across SCALE PAIRS: a register pair holds levels
//    (2q, 2q+1) of one column, the tap pair (k_2q, k_2q+1) is an SGPR pair and the scale-independent
//    operand is splat by op_sel -- no repacking moves (pairs across columns would be misaligned in the
//    horizontal pass).  ~330 VALU instructions per 4x8 outputs instead of ~600.
//  * buffer_load/store_dwordx4 with a wave-uniform row base (SGPR) and a 32-bit lane offset: no 64-bit
//    per-lane address arithmetic, and the halo lanes / columns >= w are dropped by the hardware range
//    check of a per-row buffer descriptor (num_records = w*4) -- no divergent branches in the loop.
//  * the next source row is requested one full iteration before it is needed.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
#ifdef CUSIFT_BLUR_BPERMUTE
// experiment (tools/ab_libs.sh): the blur's neighbour exchange on the LDS crossbar (ds_bpermute_b32: no VALU issue slot)
// instead of DPP moves (slow-class VALU instructions, 18 % of the blur's)
__device__ __forceinline__ float bperm(int addr, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ f2 dpp_prev2(f2 v) {
  const int a = (((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) - 1) & 63) * 4;
  return f2{bperm(a, v.x), bperm(a, v.y)};
}
__device__ __forceinline__ f2 dpp_next2(f2 v) {
  const int a = (((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) + 1) & 63) * 4;
  return f2{bperm(a, v.x), bperm(a, v.y)};
}
#else
__device__ __forceinline__ f2 dpp_prev2(f2 v) { return f2{from_prev_lane(v.x), from_prev_lane(v.y)}; }
__device__ __forceinline__ f2 dpp_next2(f2 v) { return f2{from_next_lane(v.x), from_next_lane(v.y)}; }
#endif

template <int kStoreAux>
__global__ void __launch_bounds__(256) laplace_multi_fast_kernel(const float *__restrict__ img,
                                                                float *__restrict__ dog, int w, int h, int pitch,
                                                                long img_stride, long dog_stride, int rows_per_wave,
                                                                LaplaceTapsPk T) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave id: uniform, say so
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  xcd_remap(bx, by, bz);
  const int y0 = (by * (int)(blockDim.x >> 6) + wv) * rows_per_wave;  // 1..4 waves per workgroup, stacked vertically
  if (y0 >= h) return;  // wave-uniform
  const int y1 = min(y0 + rows_per_wave, h);
  img += (long)bz * img_stride;
  dog += (long)bz * dog_stride;

  const int c0 = bx * kBlurStrip - kBlurHaloLanes * kBlurCols + lane * kBlurCols;
  const EdgeFix4 edge(c0, w);  // any w >= 4: border groups are clamped + replicated (sift_device.h)
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc((void *)img, 0, (int)((unsigned int)h * (unsigned int)pitch * 4u), kBufFlags);
  // columns >= w are dropped by num_records = w*4: dwordx4 stores are range-checked per component, so the last
  // group of a ragged width (w % 4 != 0) stores exactly its valid columns and the pad columns stay untouched
  const int voff_out = (lane >= kBlurHaloLanes && lane < 64 - kBlurHaloLanes) ? c0 * 4 : kOobOffset;
  const long plane = (long)h * pitch;

  auto load_row = [&](int y) -> f4 {
    const int yc = clampi(y, 0, h - 1);
    const u4 raw = __builtin_amdgcn_raw_buffer_load_b128(rin, edge.voff, yc * pitch * 4, 0);
    return edge(__builtin_bit_cast(f4, raw));
  };

  f4 win[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) win[i] = load_row(y0 - 4 + i);

  for (int y = y0; y < y1; ++y) {
    const f4 nxt = load_row(y + 5);  // consumed at the end of this iteration
    const f4 ctr = win[4];
    const f4 p1 = win[3] + win[5];
    const f4 p2 = win[2] + win[6];
    const f4 p3 = win[1] + win[7];
    const f4 p4 = win[0] + win[8];
    float *row = dog + (long)y * pitch;
    float prev_hi[4];
#pragma unroll
    for (int q = 0; q < kNumLevels / 2; ++q) {
      const f2 k0 = T.k[q][0], k1 = T.k[q][1], k2 = T.k[q][2], k3 = T.k[q][3], k4 = T.k[q][4];
      f2 e[12];  // columns c0-4 .. c0+7, each a (level 2q, level 2q+1) pair
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f2 v = k4 * splat(ctr[j]);
        v = pk_fma(k3, splat(p1[j]), v);
        v = pk_fma(k2, splat(p2[j]), v);
        v = pk_fma(k1, splat(p3[j]), v);
        v = pk_fma(k0, splat(p4[j]), v);
        e[4 + j] = v;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        e[j] = dpp_prev2(e[4 + j]);
        e[8 + j] = dpp_next2(e[4 + j]);
      }
      f2 L[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = 4 + j;
        f2 v = k4 * e[m];
        v = pk_fma(k3, e[m - 1] + e[m + 1], v);
        v = pk_fma(k2, e[m - 2] + e[m + 2], v);
        v = pk_fma(k1, e[m - 3] + e[m + 3], v);
        v = pk_fma(k0, e[m - 4] + e[m + 4], v);
        L[j] = v;
      }
      if (q > 0) {  // DoG plane 2q-1 = level 2q-1 (previous pair, high half) - level 2q
        const f4 d = f4{prev_hi[0] - L[0].x, prev_hi[1] - L[1].x, prev_hi[2] - L[2].x, prev_hi[3] - L[3].x};
        const __amdgpu_buffer_rsrc_t ro =
            __builtin_amdgcn_make_buffer_rsrc((void *)(row + (long)(2 * q - 1) * plane), 0, w * 4, kBufFlags);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, d), ro, voff_out, 0, kStoreAux);
      }
      {  // DoG plane 2q = level 2q - level 2q+1 (both halves of this pair)
        const f4 d = f4{L[0].x - L[0].y, L[1].x - L[1].y, L[2].x - L[2].y, L[3].x - L[3].y};
        if (2 * q < kNumDog) {
          const __amdgpu_buffer_rsrc_t ro =
              __builtin_amdgcn_make_buffer_rsrc((void *)(row + (long)(2 * q) * plane), 0, w * 4, kBufFlags);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, d), ro, voff_out, 0, kStoreAux);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) prev_hi[j] = L[j].y;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) win[i] = win[i + 1];
    win[8] = nxt;
  }
}
template __global__ void laplace_multi_fast_kernel<0>(const float *, float *, int, int, int, long, long, int, LaplaceTapsPk);
template __global__ void laplace_multi_fast_kernel<2>(const float *, float *, int, int, int, long, long, int, LaplaceTapsPk);
template __global__ void laplace_multi_fast_kernel<16>(const float *, float *, int, int, int, long, long, int, LaplaceTapsPk);
template __global__ void laplace_multi_fast_kernel<18>(const float *, float *, int, int, int, long, long, int, LaplaceTapsPk);

// ------------------------------------------------------------------------------------------------
// ScaleDown, fast path (16-byte aligned source rows, any w >= 4, 8-byte aligned destination rows).
// Same arithmetic as scale_down_kernel.  A lane loads source columns 4l..4l+3 as one float4 (1 KiB per wave
// row), takes columns 4l-2, 4l-1 and 4l+4 from its neighbours by DPP and produces two output columns
// (one float2 store); the five horizontally filtered rows 2r-1..2r+3 slide through registers.
// Lanes 0 and 63 are halo: 62 lanes x 2 = 124 output columns per wave.
// ------------------------------------------------------------------------------------------------
constexpr int kDownStrip = 62 * 2;  // output columns per wave

// Horizontal 5-tap of ScaleDown (cuSIFT_D.cu:111-113) for a lane that holds source columns c..c+3 (c % 4 == 0) of one row:
// output columns c/2 (centre c) and c/2+1 (centre c+2); columns c-2, c-1 and c+4 come from the neighbouring lanes by DPP.
// Shared by scale_down_fast_kernel and by the fused detection when it emits the next octave (detect_chunk.inc): the
// same operations in the same order on the same values, so the same bits whichever kernel produces a pixel.
__device__ __forceinline__ f2 down_hrow(const f4 v, float k0, float k1, float k2) {
  const float m2 = from_prev_lane(v.z), m1 = from_prev_lane(v.w), p4 = from_next_lane(v.x);
  float a = k0 * (m2 + v.z);
  a = fmaf(k1, m1 + v.y, a);
  a = fmaf(k2, v.x, a);
  float b = k0 * (v.x + p4);
  b = fmaf(k1, v.y + v.w, b);
  b = fmaf(k2, v.z, b);
  return f2{a, b};
}
// Vertical pass (cuSIFT_D.cu:123-125): k2*B[2r] + k0*(B[2r+3]+B[2r+2]) + k1*(B[2r-1]+B[2r+1]) -- the reference's
// asymmetric support (yRead = yStart + tx - 1, cuSIFT_D.cu:75).
__device__ __forceinline__ f2 down_vcol(f2 bm1, f2 b0, f2 bp1, f2 bp2, f2 bp3, float k0, float k1, float k2) {
  f2 v;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float t = k2 * b0[j];
    t = fmaf(k0, bp3[j] + bp2[j], t);
    t = fmaf(k1, bm1[j] + bp1[j], t);
    v[j] = t;
  }
  return v;
}

__global__ void __launch_bounds__(256) scale_down_fast_kernel(float *__restrict__ dst, int dst_pitch, long dst_stride,
                                                             const float *__restrict__ src, int w, int h,
                                                             int src_pitch, long src_stride, int rows_per_wave,
                                                             ScaleDownTaps T, RowWindow src_rw, int dst_row0,
                                                             int r_begin, int r_end) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave id: uniform, say so
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  xcd_remap(bx, by, bz);
  const int ow = w >> 1;
  // the 4 waves of a block take 4 horizontally adjacent strips (their shared 128-byte lines meet in L1/L2)
  const int strip = bx * kWavesPerBlock + wv;
  // output rows are GLOBAL rows [r_begin, r_end) of the half-size image; the destination band starts at
  // global row dst_row0, the source band is described by src_rw (whole images: {0, h}, 0, 0, h/2)
  const int r0 = r_begin + by * rows_per_wave;
  if (strip * kDownStrip >= ow || r0 >= r_end) return;  // wave-uniform
  const int r1 = min(r0 + rows_per_wave, r_end);
  src += (long)bz * src_stride;
  dst += (long)bz * dst_stride;

  const int cs = strip * (2 * kDownStrip) - 4 + lane * 4;  // first source column of this lane's float4
  const EdgeFix4 edge(cs, w);  // any w >= 4: source columns beyond the image replicate column w-1 (clamp addressing)
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
      (void *)src, 0, (int)((unsigned int)h * (unsigned int)src_pitch * 4u), kBufFlags);
  const int o0 = cs >> 1;  // first output column (even)
  // output columns >= ow are dropped by num_records = ow*4, per component (odd ow: the pair's first column only)
  const int voff_out = (lane >= 1 && lane <= 62 && cs >= 0) ? o0 * 4 : kOobOffset;
  const float k0 = T.k[0], k1 = T.k[1], k2 = T.k[2];

  auto load_row = [&](int y) -> f4 {
    const u4 raw = __builtin_amdgcn_raw_buffer_load_b128(rin, edge.voff, local_row(y, h, src_rw) * src_pitch * 4, 0);
    return edge(__builtin_bit_cast(f4, raw));
  };
  // horizontal 5-tap of one source row for output columns o0 (centre 4l) and o0+1 (centre 4l+2)
  auto hrow = [&](const f4 v) -> f2 { return down_hrow(v, k0, k1, k2); };

  f2 bm1 = hrow(load_row(2 * r0 - 1)), b0 = hrow(load_row(2 * r0)), bp1 = hrow(load_row(2 * r0 + 1));
  f4 n2 = load_row(2 * r0 + 2), n3 = load_row(2 * r0 + 3);
  for (int r = r0; r < r1; ++r) {
    const f4 s2 = n2, s3 = n3;
    n2 = load_row(2 * r + 4);  // next iteration's rows, requested now
    n3 = load_row(2 * r + 5);
    const f2 bp2 = hrow(s2), bp3 = hrow(s3);
    const f2 v = down_vcol(bm1, b0, bp1, bp2, bp3, k0, k1, k2);
    const __amdgpu_buffer_rsrc_t ro =
        __builtin_amdgcn_make_buffer_rsrc((void *)(dst + (long)(r - dst_row0) * dst_pitch), 0, ow * 4, kBufFlags);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), ro, voff_out, 0, 0);
    bm1 = bp1;
    b0 = bp2;
    bp1 = bp3;
  }
}

// ------------------------------------------------------------------------------------------------
// FindPointsMulti: 26-neighbour DoG extrema for the 5 searchable scales, edge test, 3-D quadratic
// refinement and append.  Reference: FindPointsMulti_D, cuSIFT_D.cu:402-523.
//
// Each lane owns kFindCols=2 columns (float2 loads, 512 B per wave instruction) of all 7 planes with a
// 3-row register window; column minima/maxima are exchanged with the neighbouring lanes by DPP.
// Candidates (rare) are refined by the detecting lane from L2-resident data and appended with one
// atomic per candidate (the compiler aggregates them per wave).  Algorithmic HBM bytes: 28 per pixel.
// Not reproduced from the reference: the per-tile candidate list that wraps at 32,
// and the overflow path that overwrites slot maxPts-1 -- overflow is dropped here.
// ------------------------------------------------------------------------------------------------
struct RefinedPoint {
  float x, y, scale, sharpness, edgeness;
};

// THE refinement of a candidate (cuSIFT_D.cu:478-521), from the 19 DoG values it reads: c[3*(dy+1) + (dx+1)] = centre
// plane, lo[] / hi[] = planes below / above at {(0,0), (-1,0), (+1,0), (0,-1), (0,+1)} as (dx,dy).  Evaluated operation
// by operation exactly as oracle_find_points_multi.  Returns whether the candidate passes the edge test; the record
// fields go to `r`.  Every detection kernel ends here: the fused one hands over values it holds in registers, the
// two-stage ones load them from the DoG planes first (load_candidate).
__device__ __forceinline__ bool refine_from_values(const float (&c)[9], const float (&lo)[5], const float (&hi)[5], int x,
                                                   int y, int s, const FindParams &P, RefinedPoint &r) {
  const float val = c[4];
  const float dxx = 2.0f * val - c[3] - c[5];
  const float dyy = 2.0f * val - c[1] - c[7];
  const float dxy = 0.25f * (c[8] + c[0] - c[2] - c[6]);
  const float tra = dxx + dyy;
  const float det = dxx * dyy - dxy * dxy;
  if (!(tra * tra < P.edge_limit * det)) return false;
  const float edge = (tra * tra) / det;
  const float dx = 0.5f * (c[5] - c[3]);
  const float dy = 0.5f * (c[7] - c[1]);
  const float ds = 0.5f * (lo[0] - hi[0]);
  const float dss = 2.0f * val - hi[0] - lo[0];
  const float dxs = 0.25f * (hi[2] + lo[1] - lo[2] - hi[1]);
  const float dys = 0.25f * (hi[4] + lo[3] - hi[3] - lo[4]);
  const float idxx = dyy * dss - dys * dys;
  const float idxy = dys * dxs - dxy * dss;
  const float idxs = dxy * dys - dyy * dxs;
  const float idet = 1.0f / (idxx * dxx + idxy * dxy + idxs * dxs);
  const float idyy = dxx * dss - dxs * dxs;
  const float idys = dxy * dxs - dxx * dys;
  const float idss = dxx * dyy - dxy * dxy;
  float pdx = idet * (idxx * dx + idxy * dy + idxs * ds);
  float pdy = idet * (idxy * dx + idyy * dy + idys * ds);
  float pds = idet * (idxs * dx + idys * dy + idss * ds);
  if (pdx < -0.5f || pdx > 0.5f || pdy < -0.5f || pdy > 0.5f || pds < -0.5f || pds > 0.5f) {
    pdx = dx / dxx;
    pdy = dy / dyy;
    pds = ds / dss;
  }
  const float dval = 0.5f * (dx * pdx + dy * pdy + ds * pds);
  r.x = (float)x + pdx;
  r.y = (float)y + pdy;
  r.scale = P.scales[s] * sm_exp2f(pds * P.factor);
  r.sharpness = val + dval;
  r.edgeness = edge;
  return true;
}

// the 19 values of a candidate from the DoG planes in global memory (L2-hot)
__device__ __forceinline__ void load_candidate(const float *__restrict__ dog, long plane, int pitch, int x, int y, int s,
                                               float (&c)[9], float (&lo)[5], float (&hi)[5]) {
  const float *d1 = dog + (long)(s + 1) * plane + (long)y * pitch + x;
  const float *d0 = d1 - plane;
  const float *d2 = d1 + plane;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int k = 0; k < 3; ++k) c[3 * r + k] = d1[(r - 1) * pitch + (k - 1)];
  lo[0] = d0[0], lo[1] = d0[-1], lo[2] = d0[1], lo[3] = d0[-pitch], lo[4] = d0[pitch];
  hi[0] = d2[0], hi[1] = d2[-1], hi[2] = d2[1], hi[3] = d2[-pitch], hi[4] = d2[pitch];
}

__device__ __forceinline__ bool refine_from_planes(const float *__restrict__ dog, long plane, int pitch, int x, int y,
                                                   int s, const FindParams &P, RefinedPoint &r) {
  float c[9], lo[5], hi[5];
  load_candidate(dog, plane, pitch, x, y, s, c, lo, hi);
  return refine_from_values(c, lo, hi, x, y, s, P, r);
}

// Per-wave list of candidate POSITIONS in LDS for the two-stage path (the DoG planes are in memory, L2-hot, so a
// candidate is its position): refined 64 at a time like CandList below -- every lane walks one candidate's ~26 loads
// and ~150 dependent instructions instead of one or two lanes doing so while the wave waits -- and appended to the
// image's SiftData with one atomic per batch.
constexpr int kPosCap = 128;  // < 64 waiting + at most 64 pushed at a time
struct PosList {
  unsigned int *buf;  // [kPosCap][2]: x | s << 28, y
  int n;              // wave-uniform
  __device__ __forceinline__ void push(bool mine, int x, int y, int s) {  // called by ALL lanes (convergent)
    const unsigned long long m = __builtin_amdgcn_ballot_w64(mine);
    if (m == 0) return;
    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
    if (mine) {
      buf[2 * (n + rank)] = (unsigned int)x | ((unsigned int)s << 28);
      buf[2 * (n + rank) + 1] = (unsigned int)y;
    }
    n += __builtin_popcountll(m);
  }
  // refines the last min(n, 64) entries and appends the accepted ones to the image's SiftData
  __device__ __forceinline__ void refine_batch(const float *__restrict__ dog, long plane, int pitch,
                                               cusift_point *__restrict__ pts, int max_pts, unsigned int *counter,
                                               const FindParams &P, int lane) {
    const int cnt = n < 64 ? n : 64;  // wave-uniform
    if (cnt == 0) return;
    n -= cnt;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's list writes have landed (one wave, in-order LDS)
    RefinedPoint r;
    bool accept = false;
    if (lane < cnt) {
      const unsigned int xs = buf[2 * (n + lane)], y = buf[2 * (n + lane) + 1];
      accept = refine_from_planes(dog, plane, pitch, (int)(xs & 0x0fffffffu), (int)y, (int)(xs >> 28), P, r);
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(accept);
    if (m == 0) return;
    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
    unsigned int base = 0;
    if (lane == 0) base = atomicAdd(counter, (unsigned int)__builtin_popcountll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    const unsigned int idx = base + (unsigned int)rank;
    if (accept && idx < (unsigned int)max_pts) {
      cusift_point *pt = pts + idx;
      pt->coords2D[0] = r.x;
      pt->coords2D[1] = r.y;
      pt->scale = r.scale;
      pt->sharpness = r.sharpness;
      pt->edgeness = r.edgeness;
      pt->subsampling = P.subsampling;
    }
  }
};

__device__ __forceinline__ void refine_and_append(const float *__restrict__ dog, long plane, int pitch, int x, int y,
                                                   int s, const FindParams &P, cusift_point *__restrict__ pts,
                                                   int max_pts, unsigned int *counter) {
  RefinedPoint r;
  if (!refine_from_planes(dog, plane, pitch, x, y, s, P, r)) return;
  const unsigned int idx = atomicAdd(counter, 1u);
  if (idx >= (unsigned int)max_pts) return;
  cusift_point *pt = pts + idx;
  pt->coords2D[0] = r.x;
  pt->coords2D[1] = r.y;
  pt->scale = r.scale;
  pt->sharpness = r.sharpness;
  pt->edgeness = r.edgeness;
  pt->subsampling = P.subsampling;
}

__global__ void __launch_bounds__(256) find_points_kernel(const float *__restrict__ dog, int w, int h, int pitch,
                                                         long dog_stride, cusift_point *__restrict__ points,
                                                         int max_pts, unsigned int *__restrict__ counters,
                                                         int rows_per_wave, int vec_ok, FindParams P) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave id: uniform, say so
  const int y0 = (blockIdx.y * kWavesPerBlock + wv) * rows_per_wave;
  if (y0 >= h) return;
  const int y1 = min(y0 + rows_per_wave, h);
  dog += (long)blockIdx.z * dog_stride;
  points += (long)blockIdx.z * max_pts;
  unsigned int *counter = counters + blockIdx.z;

  const int c0 = blockIdx.x * kFindStrip - kFindCols + lane * kFindCols;
  const bool interior = vec_ok && c0 >= 0 && c0 + 1 < w;
  const int cc0 = clampi(c0, 0, w - 1), cc1 = clampi(c0 + 1, 0, w - 1);
  const long plane = (long)h * pitch;

  float d[kNumDog][3][2];  // [plane][row y-1, y, y+1][column]
  auto load_rows = [&](int y, int slot) {
    const float *r = dog + (long)clampi(y, 0, h - 1) * pitch;
#pragma unroll
    for (int p = 0; p < kNumDog; ++p) {
      if (interior) {
        const float2 v = *reinterpret_cast<const float2 *>(r + (long)p * plane + c0);
        d[p][slot][0] = v.x;
        d[p][slot][1] = v.y;
      } else {
        d[p][slot][0] = r[(long)p * plane + cc0];
        d[p][slot][1] = r[(long)p * plane + cc1];
      }
    }
  };
  // slots are rotated by copying (the compiler renames registers in the unrolled body)
  load_rows(y0 - 1, 0);
  load_rows(y0, 1);
  const bool lane_valid = lane >= 1 && lane <= 62;

  for (int y = y0; y < y1; ++y) {
    load_rows(y + 1, 2);
    float cmin[kNumDog][2], cmax[kNumDog][2], nmin[kNumDog][2][2], nmax[kNumDog][2][2];  // n*[p][j][0=left,1=right]
#pragma unroll
    for (int p = 0; p < kNumDog; ++p) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        cmin[p][j] = fminf(fminf(d[p][0][j], d[p][1][j]), d[p][2][j]);
        cmax[p][j] = fmaxf(fmaxf(d[p][0][j], d[p][1][j]), d[p][2][j]);
      }
      nmin[p][0][0] = from_prev_lane(cmin[p][1]);
      nmin[p][0][1] = cmin[p][1];
      nmin[p][1][0] = cmin[p][0];
      nmin[p][1][1] = from_next_lane(cmin[p][0]);
      nmax[p][0][0] = from_prev_lane(cmax[p][1]);
      nmax[p][0][1] = cmax[p][1];
      nmax[p][1][0] = cmax[p][0];
      nmax[p][1][1] = from_next_lane(cmax[p][0]);
    }
    unsigned int cand = 0;  // bit (2*s + j)
#pragma unroll
    for (int s = 0; s < kNumScales; ++s) {
      const int c = s + 1;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float v = d[c][1][j];
        // plane below / above: full 3x3; centre plane: 3x3 without the centre
        float mn = fminf(fminf(nmin[c - 1][j][0], cmin[c - 1][j]), nmin[c - 1][j][1]);
        mn = fminf(mn, fminf(fminf(nmin[c + 1][j][0], cmin[c + 1][j]), nmin[c + 1][j][1]));
        mn = fminf(mn, fminf(fminf(nmin[c][j][0], nmin[c][j][1]), fminf(d[c][0][j], d[c][2][j])));
        float mx = fmaxf(fmaxf(nmax[c - 1][j][0], cmax[c - 1][j]), nmax[c - 1][j][1]);
        mx = fmaxf(mx, fmaxf(fmaxf(nmax[c + 1][j][0], cmax[c + 1][j]), nmax[c + 1][j][1]));
        mx = fmaxf(mx, fmaxf(fmaxf(nmax[c][j][0], nmax[c][j][1]), fmaxf(d[c][0][j], d[c][2][j])));
        const bool hit = (v < P.thr_neg && v < mn) || (v > P.thr_pos && v > mx);
        cand |= (hit ? 1u : 0u) << (2 * s + j);
      }
    }
    if (!lane_valid) cand = 0;
    if (cand) {
      // strict extrema cannot sit on the image border (a clamped neighbour equals the centre);
      // the explicit range test keeps the refinement's reads in bounds for any input (NaN/Inf).
      for (int b = 0; b < 2 * kNumScales; ++b) {
        if (cand & (1u << b)) {
          const int x = c0 + (b & 1), s = b >> 1;
          if (x >= 1 && x <= w - 2 && y >= 1 && y <= h - 2)
            refine_and_append(dog, plane, pitch, x, y, s, P, points, max_pts, counter);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < kNumDog; ++p)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        d[p][0][j] = d[p][1][j];
        d[p][1][j] = d[p][2][j];
      }
  }
}

// 3-input min/max: these map to one v_min3_f32 / v_max3_f32 each.  (A 2-input IEEE minnum on loaded
// values costs two extra sNaN-quieting v_max_f32 x,x, so the extremum test below only uses 3-input trees.)
__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// ------------------------------------------------------------------------------------------------
// FindPointsMulti, fast path (8-byte aligned DoG rows: pitch % 2 == 0, any w >= 2, block < 2 GiB): the same test and the
// same refinement as find_points_kernel, with buffer_load_dwordx2 on a wave-uniform row base, the next
// row of all 7 planes requested one iteration ahead, and v_min3/v_max3 trees.  (Non-temporal loads were tried and
// lose: 4.2 -> 3.6 TB/s, tools/ab_findpoints.sh -- the 3-row window's re-reads want the L2.)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) find_points_fast_kernel(const float *__restrict__ dog, int w, int h, int pitch,
                                                              long dog_stride, cusift_point *__restrict__ points,
                                                              int max_pts, unsigned int *__restrict__ counters,
                                                              int rows_per_wave, FindParams P) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave id: uniform, say so
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  xcd_remap(bx, by, bz);
  // the 4 waves of a block take 4 horizontally adjacent strips of the same rows: the 128-byte lines that
  // straddle their strip borders are fetched once (L1/L2 hit) instead of once per strip
  const int strip = bx * kWavesPerBlock + wv;
  const int y0 = by * rows_per_wave;
  if (strip * kFindStrip >= w || y0 >= h) return;  // wave-uniform
  const int y1 = min(y0 + rows_per_wave, h);
  dog += (long)bz * dog_stride;
  points += (long)bz * max_pts;
  unsigned int *counter = counters + bz;

  const int c0 = strip * kFindStrip - kFindCols + lane * kFindCols;
  const EdgeFix2 edge(c0, w);  // any w >= 2
  const long plane = (long)h * pitch;
  const int plane_bytes = (int)((unsigned int)h * (unsigned int)pitch * 4u);
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc((void *)dog, 0, (int)((unsigned int)plane_bytes * (unsigned int)kNumDog), kBufFlags);

  auto load_row = [&](int y, f2 (&o)[kNumDog]) {
    const int row_off = clampi(y, 0, h - 1) * pitch * 4;
#pragma unroll
    for (int p = 0; p < kNumDog; ++p) {
      const u2 raw = __builtin_amdgcn_raw_buffer_load_b64(rin, edge.voff, p * plane_bytes + row_off, 0);
      o[p] = edge(__builtin_bit_cast(f2, raw));
    }
  };

  __shared__ unsigned int s_cands[kWavesPerBlock][2 * kPosCap];
  PosList cands{s_cands[wv], 0};
  f2 r0[kNumDog], r1[kNumDog], r2[kNumDog], nxt[kNumDog];
  load_row(y0 - 1, r0);
  load_row(y0, r1);
  load_row(y0 + 1, r2);
  const bool lane_valid = lane >= 1 && lane <= 62;

  for (int y = y0; y < y1; ++y) {
    load_row(y + 2, nxt);
    // per plane: 3-row column min/max for this lane's two columns and for the columns left/right of them
    float cmn[kNumDog][2], cmx[kNumDog][2], hmn[kNumDog][2], hmx[kNumDog][2], lmn[kNumDog], rmn[kNumDog],
        lmx[kNumDog], rmx[kNumDog];
#pragma unroll
    for (int p = 0; p < kNumDog; ++p) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        cmn[p][j] = min3f(r0[p][j], r1[p][j], r2[p][j]);
        cmx[p][j] = max3f(r0[p][j], r1[p][j], r2[p][j]);
      }
      lmn[p] = from_prev_lane(cmn[p][1]);  // column c0-1
      rmn[p] = from_next_lane(cmn[p][0]);  // column c0+2
      lmx[p] = from_prev_lane(cmx[p][1]);
      rmx[p] = from_next_lane(cmx[p][0]);
      hmn[p][0] = min3f(lmn[p], cmn[p][0], cmn[p][1]);
      hmn[p][1] = min3f(cmn[p][0], cmn[p][1], rmn[p]);
      hmx[p][0] = max3f(lmx[p], cmx[p][0], cmx[p][1]);
      hmx[p][1] = max3f(cmx[p][0], cmx[p][1], rmx[p]);
    }
    unsigned int cand = 0;  // bit (2*s + j)
#pragma unroll
    for (int s = 0; s < kNumScales; ++s) {
      const int c = s + 1;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float v = r1[c][j];
        const float nl_mn = j == 0 ? lmn[c] : cmn[c][0], nr_mn = j == 0 ? cmn[c][1] : rmn[c];
        const float nl_mx = j == 0 ? lmx[c] : cmx[c][0], nr_mx = j == 0 ? cmx[c][1] : rmx[c];
        // 26 neighbours as two 3-input trees per direction; "v < min(a, b)" is evaluated as v < a && v < b
        // (a 2-input IEEE min would cost two extra quieting instructions)
        const float mn_a = min3f(nl_mn, nr_mn, r0[c][j]);
        const float mn_b = min3f(r2[c][j], hmn[c - 1][j], hmn[c + 1][j]);
        const float mx_a = max3f(nl_mx, nr_mx, r0[c][j]);
        const float mx_b = max3f(r2[c][j], hmx[c - 1][j], hmx[c + 1][j]);
        const bool hit = (v < min3f(mn_a, mn_b, P.thr_neg)) || (v > max3f(mx_a, mx_b, P.thr_pos));
        cand |= (hit ? 1u : 0u) << (2 * s + j);
      }
    }
    if (!lane_valid) cand = 0;
    if (__builtin_amdgcn_ballot_w64(cand != 0) != 0) {  // wave-uniform, rare
      for (int b = 0; b < 2 * kNumScales; ++b) {
        const int x = c0 + (b & 1), s = b >> 1;
        const bool mine = (cand & (1u << b)) && x >= 1 && x <= w - 2 && y >= 1 && y <= h - 2;
        cands.push(mine, x, y, s);  // at most 64 more entries
        if (cands.n >= 64) cands.refine_batch(dog, plane, pitch, points, max_pts, counter, P, lane);
      }
    }
#pragma unroll
    for (int p = 0; p < kNumDog; ++p) {
      r0[p] = r1[p];
      r1[p] = r2[p];
      r2[p] = nxt[p];
    }
  }
  cands.refine_batch(dog, plane, pitch, points, max_pts, counter, P, lane);  // what is left (fewer than 64)
}

// ------------------------------------------------------------------------------------------------
// Fused detection: LaplaceMulti + FindPointsMulti in one pass, the DoG planes never leave the chip.
// (Reference: LaplaceMulti_D cuSIFT_D.cu:525-553 followed by FindPointsMulti_D cuSIFT_D.cu:402-523; the
// reference round-trips 7 DoG planes through memory between them: 28 B/px written + 28 B/px read.)
//
// Same column strips and the same arithmetic as laplace_multi_fast_kernel and find_points_fast_kernel --
// the DoG values tested and refined here are bit for bit the ones those kernels store/load -- but each
// wave keeps the DoG rows y-1, y, y+1 of all 7 planes in registers (float4 per lane) and runs the
// 26-neighbour test on them as soon as row y+1 is blurred.  HBM traffic drops from 60 B/px to the 4 B/px
// of the source image; the kernel is VALU-bound (~2.8 wave instructions per pixel).
//   * lanes 0-1 / 62-63 are halo (blur needs +-4 columns = 1 lane, the extremum test 1 more column);
//     60 lanes x 4 = 240 columns per wave produce keypoints.  Nothing is stored, so no alignment rule.
//   * a wave owns rows [y0,y1) as extremum CENTRES: it blurs rows y0-1 .. y1 (2 extra rows per chunk).
//     Image-border pixels are never extrema in the reference (a clamped neighbour equals the centre), so
//     centres are restricted to 1..h-2 / 1..w-2 and no clamped DoG row is ever needed.
//   * refinement (rare): the detecting lane parks the 19 DoG values its refinement reads -- all in its own registers
//     but the column beyond its float4 (five DPP moves) -- in a per-wave LDS list, and the wave refines 64 candidates
//     at a time (CandList below).
// ------------------------------------------------------------------------------------------------
constexpr int kDetHaloLanes = 2;
constexpr int kDetStrip = (64 - 2 * kDetHaloLanes) * kBlurCols;  // 240 columns of extremum centres per wave
constexpr int kCandWords = 21;                                   // 19 DoG values, x, (y << 3) | scale index
constexpr int kCandCap = 128;                                    // < 64 waiting + at most 64 pushed at a time
static_assert(kCandCap * kCandWords == kDetectWaveLdsFloats, "host and kernel agree on the LDS size");

// kIdent0: levels 0 and 1 have identity taps (initBlur >= their sigma: the "var <= 1e-6 => identity" rule, e.g.
// octave 0 of the initBlur = 1.0 configuration).  1*c and fma(0, x, c) are exact for finite x, so the pair is
// passed through instead of being filtered: L0 = L1 = S, DoG plane 0 = 0 -- bit-identical, 24 % less arithmetic.
template <bool kIdent0>
__device__ __forceinline__ void blur_dog_row(const f4 (&win)[9], const LaplaceTapsPk &T, f4 (&D)[kNumDog]) {
  const f4 ctr = win[4];
  const f4 p1 = win[3] + win[5];
  const f4 p2 = win[2] + win[6];
  const f4 p3 = win[1] + win[7];
  const f4 p4 = win[0] + win[8];
  float prev_hi[4];
  if (kIdent0) {
    D[0] = f4{0.f, 0.f, 0.f, 0.f};  // S - S
#pragma unroll
    for (int j = 0; j < 4; ++j) prev_hi[j] = ctr[j];
  }
#pragma unroll
  for (int q = kIdent0 ? 1 : 0; q < kNumLevels / 2; ++q) {
    const f2 k0 = T.k[q][0], k1 = T.k[q][1], k2 = T.k[q][2], k3 = T.k[q][3], k4 = T.k[q][4];
    f2 e[12];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f2 v = k4 * splat(ctr[j]);
      v = pk_fma(k3, splat(p1[j]), v);
      v = pk_fma(k2, splat(p2[j]), v);
      v = pk_fma(k1, splat(p3[j]), v);
      v = pk_fma(k0, splat(p4[j]), v);
      e[4 + j] = v;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      e[j] = dpp_prev2(e[4 + j]);
      e[8 + j] = dpp_next2(e[4 + j]);
    }
    f2 L[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = 4 + j;
      f2 v = k4 * e[m];
      v = pk_fma(k3, e[m - 1] + e[m + 1], v);
      v = pk_fma(k2, e[m - 2] + e[m + 2], v);
      v = pk_fma(k1, e[m - 3] + e[m + 3], v);
      v = pk_fma(k0, e[m - 4] + e[m + 4], v);
      L[j] = v;
    }
    if (q > 0) D[2 * q - 1] = f4{prev_hi[0] - L[0].x, prev_hi[1] - L[1].x, prev_hi[2] - L[2].x, prev_hi[3] - L[3].x};
    if (2 * q < kNumDog) D[2 * q] = f4{L[0].x - L[0].y, L[1].x - L[1].y, L[2].x - L[2].y, L[3].x - L[3].y};
#pragma unroll
    for (int j = 0; j < 4; ++j) prev_hi[j] = L[j].y;
  }
}

// Per-wave list of CANDIDATES in LDS, refined 64 at a time.  Measured (tools/exp_detect_parts.sh, 64 x 1080p): refining
// each candidate where it is found -- one or two active lanes walking ~150 dependent instructions with three IEEE
// divisions and an exp2, ~0.25 times per wave-row -- cost 21 % of the fused kernel although it is 1 % of its
// arithmetic.  Now the detecting lane only copies the 19 DoG values its refinement reads (from the wave's cube) and its
// (x, y, scale index) into the list; when 64 candidates have gathered, or at the end of the chunk, every lane refines
// one of them -- the same instructions on the same values, so the same bits -- and the accepted ones are appended to
// the image's SiftData with ONE atomic per batch (ballot + mbcnt for the slots; the reference takes one atomicInc per
// candidate, cuSIFT_D.cu:512).  The order inside an octave, unspecified before, is unspecified still; overflow beyond
// max_pts is dropped as before while the counter keeps counting.
struct CandList {
  float *buf;  // [kCandCap][kCandWords] in LDS; an entry's stride is odd: lane i reading word k of entry i is conflict-free
  int n;       // wave-uniform
  // Called by ALL lanes (convergent): the entry of a lane with `mine` set (its rank among them, after the `n` entries
  // already waiting), nullptr for the others; the caller fills the kCandWords words.
  __device__ __forceinline__ float *reserve(bool mine) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(mine);
    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
    float *e = mine ? buf + (n + rank) * kCandWords : nullptr;
    n += __builtin_popcountll(m);
    return e;
  }
  // refines the last min(n, 64) entries and appends the accepted ones to the image's SiftData
  // kRecBytes: the stride of the destination list -- sizeof(cusift_point), or kStagedRecBytes when the points go to the
  // context's staging list (heads only, see cusift_extract_batch) -- a record's head has the same layout in both
  template <int kRecBytes>
  __device__ __forceinline__ void refine_batch(char *__restrict__ pts, int max_pts, unsigned int *counter,
                                               const FindParams &P, int lane) {
    const int cnt = n < 64 ? n : 64;  // wave-uniform
    if (cnt == 0) return;
    n -= cnt;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's list writes have landed (one wave, in-order LDS)
    RefinedPoint r;
    bool accept = false;
    if (lane < cnt) {
      const float *e = buf + (n + lane) * kCandWords;
      float c[9], lo[5], hi[5];
#pragma unroll
      for (int k = 0; k < 9; ++k) c[k] = e[k];
#pragma unroll
      for (int k = 0; k < 5; ++k) lo[k] = e[9 + k], hi[k] = e[14 + k];
      const int x = __builtin_bit_cast(int, e[19]);
      const int ys = __builtin_bit_cast(int, e[20]);
      accept = refine_from_values(c, lo, hi, x, ys >> 3, ys & 7, P, r);
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(accept);
    if (m == 0) return;
    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
    unsigned int base = 0;
    if (lane == 0) base = atomicAdd(counter, (unsigned int)__builtin_popcountll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    const unsigned int idx = base + (unsigned int)rank;
    if (accept && idx < (unsigned int)max_pts) {
      cusift_point *pt = reinterpret_cast<cusift_point *>(pts + (size_t)idx * kRecBytes);
      pt->coords2D[0] = r.x;
      pt->coords2D[1] = r.y;
      pt->scale = r.scale;
      pt->sharpness = r.sharpness;
      pt->edgeness = r.edgeness;
      pt->subsampling = P.subsampling;
    }
  }
};

// kDown: the kernel also EMITS THE NEXT OCTAVE'S IMAGE (ScaleDown, cuSIFT.cu:185 / cuSIFT_D.cu:37-182) as a by-product:
// the source rows 2r-1 .. 2r+3 and columns 2c-2 .. 2c+2 an output pixel reads are inside the 9-row window (and its
// neighbouring lanes) that the blur streams through anyway, so the pyramid costs no second read of the image and no
// launch of its own.  The wave whose chunk holds source row 2r (as an unclipped chunk row) owns output row r; a lane
// owns the two output columns under its four source columns.  Whole images only (rw = {0, h}); same operations as
// scale_down_fast_kernel (down_hrow / down_vcol), so the same bits.
template <bool kIdent0, int kRecBytes, bool kDown>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) detect_fused_kernel(const float *__restrict__ img, int w, int h, int pitch,
                                                          long img_stride, cusift_point *__restrict__ points,
                                                          int max_pts, unsigned int *__restrict__ counters,
                                                          int rows_per_wave, LaplaceTapsPk T, FindParams P,
                                                          RowWindow rw, int cy_begin, int cy_end, DownOut down) {
  extern __shared__ float s_cands[];  // [waves per workgroup][kDetectWaveLdsFloats]: the wave's candidate list
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave id: uniform, say so
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  xcd_remap(bx, by, bz);
  // extremum centres are GLOBAL rows [cy_begin, cy_end) minus the global border rows; this wave takes a chunk
  // of them and works in band-local row indices (global - rw.row0).  Whole image: rw = {0, h}, [0, h).
  const int gy0 = cy_begin + (by * (int)(blockDim.x >> 6) + wv) * rows_per_wave;
  const int ya = max(max(gy0, 1), cy_begin) - rw.row0;
  const int yb = min(min(gy0 + rows_per_wave, rw.hg - 1), cy_end) - rw.row0;  // local centres [ya, yb)
  if (ya >= yb) return;                                                         // wave-uniform
  img += (long)bz * img_stride;
  char *const list = reinterpret_cast<char *>(points) + (size_t)bz * max_pts * kRecBytes;
  unsigned int *counter = counters + bz;
  CandList cands{s_cands + wv * kDetectWaveLdsFloats, 0};
  // (kDown) this wave owns the output rows r with own_lo <= 2r < own_hi: its chunk before the clipping to [1, h-1)
  const int own_lo = gy0, own_hi = gy0 + rows_per_wave;
  float *const down_img = kDown ? down.dst + (long)bz * down.stride : nullptr;
  const int down_pitch = down.pitch;
  const ScaleDownTaps DT = down.T;

#include "detect_chunk.inc"
}


// ------------------------------------------------------------------------------------------------
// The ScaleDown chain of a small call in ONE launch.  For one 1080p frame the four ScaleDowns are 6-10 us launches each,
// nearly all of it dispatch; they depend on each other (level k+1 reads level k), so one launch has to carry the
// dependency inside: a workgroup owns a square of the LAST level, works out which pixels of every level it needs for
// it (out <- source columns 2c-2 .. 2c+2 and rows 2r-1 .. 2r+3, clamped: the reference's windows, cuSIFT_D.cu:75-177),
// and computes them level by level in LDS -- horizontal pass into H, vertical pass into the level -- recomputing what
// its neighbours also compute (for four levels a 109 x 109 source region per 4 x 4 pixels of level 4: 2.9x the reads,
// which is why only small calls take this kernel).  Every level's pixels are written to HBM by the one workgroup that
// OWNS them (its square scaled up; the last square of a row / column also takes the odd remainder).  Same operations
// in the same order as scale_down_fast_kernel on the same values: the same bits whoever computes a pixel.
// ------------------------------------------------------------------------------------------------
struct PyrRange {
  int x0, x1, y0, y1;  // [x0, x1) x [y0, y1)
  __device__ __forceinline__ int w() const { return x1 - x0; }
  __device__ __forceinline__ int h() const { return y1 - y0; }
};

__global__ void __launch_bounds__(256) pyramid_small_kernel(PyramidLevels P, ScaleDownTaps T, unsigned int *zero,
                                                            int n_zero) {
  extern __shared__ float s_pyr[];
  const int tid = threadIdx.x;
  // (the driver's keypoint counters for the detection that follows: saves a small call the memset dispatch)
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
    for (int i = tid; i < n_zero; i += 256) zero[i] = 0u;
  const long img = blockIdx.z;
  // what this workgroup owns (writes) and needs (computes) of every level, from the last level down
  PyrRange own[kMaxPyramidLevels + 2], need[kMaxPyramidLevels + 2];
#pragma unroll
  for (int k = kMaxPyramidLevels; k >= 1; --k) {  // (compile-time indices: the ranges stay in registers)
    if (k > P.n) continue;
    if (k == P.n) {
      own[k].x0 = blockIdx.x * P.tile;
      own[k].y0 = blockIdx.y * P.tile;
      own[k].x1 = min(own[k].x0 + P.tile, P.w[k]);
      own[k].y1 = min(own[k].y0 + P.tile, P.h[k]);
      need[k] = own[k];
    } else {
      // level k: owned = the owner's square scaled up (+ the odd remainder at the far edges) ...
      own[k].x0 = 2 * own[k + 1].x0;
      own[k].y0 = 2 * own[k + 1].y0;
      own[k].x1 = own[k + 1].x1 == P.w[k + 1] ? P.w[k] : 2 * own[k + 1].x1;
      own[k].y1 = own[k + 1].y1 == P.h[k + 1] ? P.h[k] : 2 * own[k + 1].y1;
      // ... needed = owned + what level k+1's needed pixels read (clamped to the image)
      need[k].x0 = min(own[k].x0, max(2 * need[k + 1].x0 - 2, 0));
      need[k].x1 = max(own[k].x1, min(2 * (need[k + 1].x1 - 1) + 2, P.w[k] - 1) + 1);
      need[k].y0 = min(own[k].y0, max(2 * need[k + 1].y0 - 1, 0));
      need[k].y1 = max(own[k].y1, min(2 * (need[k + 1].y1 - 1) + 3, P.h[k] - 1) + 1);
    }
  }
  need[0] = own[0] = PyrRange{0, 0, 0, 0};
  const float k0 = T.k[0], k1 = T.k[1], k2 = T.k[2];
  // LDS: [ H of the current level | level 1 | level 2 | ... ]; the levels stay (level k+1 reads level k)
  float *lvl[kMaxPyramidLevels + 1];
  float *cursor = s_pyr;
  lvl[0] = nullptr;
#pragma unroll
  for (int k = 1; k <= kMaxPyramidLevels; ++k) {
    lvl[k] = cursor;
    if (k <= P.n) cursor += need[k].w() * need[k].h();
  }
  float *const H = cursor;  // the largest H (level 1's) fits behind the levels: the host sizes the allocation for it

#pragma unroll
  for (int k = 1; k <= kMaxPyramidLevels; ++k) {
    if (k > P.n) break;
    const int sw = P.w[k - 1], sh = P.h[k - 1];  // source level
    const PyrRange N = need[k];
    const int nw = N.w();
    // source rows the vertical pass reads, clamped: [ry0, ry1]
    const int ry0 = max(2 * N.y0 - 1, 0), ry1 = min(2 * (N.y1 - 1) + 3, sh - 1);
    const int hrows = ry1 - ry0 + 1;
    // horizontal pass: H[y - ry0][c - N.x0] for the source rows y and the needed columns c
    const float *src0 = P.base[0] + img * P.stride[0];
    const PyrRange S = need[k - 1];  // (k >= 2) where level k-1 sits in LDS
    const int spw = k >= 2 ? S.w() : 0;
    // (tried and dropped, same-box A/B on one 1080p frame: unrolling this loop by 8, +1.5 % per frame; a thread per
    // column that walks the rows -- a third of the instructions -- +4 %: the kernel is latency-, not issue-bound)
    for (int e = tid; e < hrows * nw; e += 256) {
      const int yy = e / nw, cc = e - yy * nw;
      const int y = ry0 + yy, c = N.x0 + cc;
      const int xa = max(2 * c - 2, 0), xb = max(2 * c - 1, 0), xc = 2 * c, xd = min(2 * c + 1, sw - 1), xe = min(2 * c + 2, sw - 1);
      float a0, a1, a2, a3, a4;
      if (k == 1) {
        const float *r = src0 + (long)y * P.pitch[0];
        a0 = r[xa], a1 = r[xb], a2 = r[min(xc, sw - 1)], a3 = r[xd], a4 = r[xe];
      } else {
        const float *r = lvl[k - 1] + (y - S.y0) * spw - S.x0;
        a0 = r[xa], a1 = r[xb], a2 = r[min(xc, sw - 1)], a3 = r[xd], a4 = r[xe];
      }
      float v = k0 * (a0 + a4);
      v = fmaf(k1, a1 + a3, v);
      v = fmaf(k2, a2, v);
      H[e] = v;
    }
    __syncthreads();
    // vertical pass: level k over the needed range; the owned part also goes to HBM
    float *dst = P.base[k] + img * P.stride[k];
    const PyrRange O = own[k];
    for (int e = tid; e < N.h() * nw; e += 256) {
      const int rr = e / nw, cc = e - rr * nw;
      const int r = N.y0 + rr, c = N.x0 + cc;
      const float *h = H + cc;
      auto row = [&](int y) { return h[(min(max(y, 0), sh - 1) - ry0) * nw]; };
      float v = k2 * row(2 * r);
      v = fmaf(k0, row(2 * r + 3) + row(2 * r + 2), v);
      v = fmaf(k1, row(2 * r - 1) + row(2 * r + 1), v);
      lvl[k][e] = v;
      if (r >= O.y0 && r < O.y1 && c >= O.x0 && c < O.x1) dst[(long)r * P.pitch[k] + c] = v;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// The fused detection of SEVERAL octaves in one launch.  The coarser octaves of an image are small -- for one 1080p frame
// octaves 1..4 are 14-16 us launches each, most of it the launch itself and the latency of a chunk's window fill -- and
// they do not depend on each other, only on the ScaleDown chain.  One launch, the workgroups of octave 1 first and the
// smaller ones behind them, costs what the largest costs.  Their keypoints cannot go to one list then (SiftData is
// coarsest octave first): every octave appends to a list of its own (sift_types.h: SegmentTable) and
// describe_all_kernel joins them.  Same chunk body as detect_fused_kernel (detect_chunk.inc), in both instantiations.
// ------------------------------------------------------------------------------------------------
template <int kRecBytes>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) detect_multi_kernel(DetectTable tab, int max_pts,
                                                                                          unsigned int *queue) {
  extern __shared__ float s_cands[];  // the wave's candidate list
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  // (describe_all_kernel's work cursors for the launch that follows: saves a small call the memset dispatch)
  if (b == 0 && queue)
    for (int i = lane; i < kQueueShards * 32; i += 64) queue[i] = 0u;
  int oi = 0;
  while (oi + 1 < tab.n && b >= tab.o[oi + 1].first_block) ++oi;  // wave-uniform
  const DetectOctave &O = tab.o[oi];
  const int local = b - O.first_block;
  const int bx = local % O.strips;
  const int by = (local / O.strips) % O.chunks;
  const int bz = local / (O.strips * O.chunks);
  const int w = O.w, h = O.h, pitch = O.pitch;
  const RowWindow rw{O.row0, O.hg};
  // as in detect_fused_kernel: centres are global rows [cy_begin, cy_end) minus the global border rows, in local indices
  const int gy0 = O.cy_begin + by * O.rows_per_wave;
  const int ya = max(max(gy0, 1), O.cy_begin) - rw.row0;
  const int yb = min(min(gy0 + O.rows_per_wave, rw.hg - 1), O.cy_end) - rw.row0;  // local centres [ya, yb)
  if (ya >= yb) return;                                                            // wave-uniform
  const float *img = O.img + (long)bz * O.img_stride;
  char *const list = O.lists + (size_t)bz * max_pts * kRecBytes;
  unsigned int *counter = O.counters + bz;
  CandList cands{s_cands, 0};
  const LaplaceTapsPk &T = O.T;
  const FindParams &P = O.P;
  // (this launch searches octaves whose images exist already: nothing is emitted)
  constexpr bool kDown = false;
  constexpr int own_lo = 0, own_hi = 0, down_pitch = 0;
  float *const down_img = nullptr;
  const ScaleDownTaps DT{};
  if (O.ident) {  // wave-uniform; both bodies live in this kernel, a wave runs one
    constexpr bool kIdent0 = true;
#include "detect_chunk.inc"
  } else {
    constexpr bool kIdent0 = false;
#include "detect_chunk.inc"
  }
}

template __global__ void detect_multi_kernel<kStagedRecBytes>(DetectTable, int, unsigned int *);

#define CUSIFT_DETECT_INSTANCE(IDENT, REC, DOWN)                                                                     \
  template __global__ void detect_fused_kernel<IDENT, REC, DOWN>(const float *, int, int, int, long, cusift_point *,  \
                                                                 int, unsigned int *, int, LaplaceTapsPk, FindParams, \
                                                                 RowWindow, int, int, DownOut);
CUSIFT_DETECT_INSTANCE(false, (int)sizeof(cusift_point), false)
CUSIFT_DETECT_INSTANCE(true, (int)sizeof(cusift_point), false)
CUSIFT_DETECT_INSTANCE(false, kStagedRecBytes, false)
CUSIFT_DETECT_INSTANCE(true, kStagedRecBytes, false)
// (the next octave can only be emitted when the octaves are searched finest first, i.e. into lists of their own)
CUSIFT_DETECT_INSTANCE(false, kStagedRecBytes, true)
CUSIFT_DETECT_INSTANCE(true, kStagedRecBytes, true)
#undef CUSIFT_DETECT_INSTANCE

}  // namespace cusift
