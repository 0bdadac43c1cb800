// sift_keypoints.hip -- gfx950 keypoint kernels: dominant orientation, 128-D descriptor, their fusion over all
// octaves of a batch, RootSIFT.  One 64-lane wave per keypoint; the pixels a keypoint samples are staged in LDS
// and the reference's texture fetches are modelled in software (gfx950 has no image unit).
//
// Arithmetic follows oracle/sift_oracle.c operation by operation (nothing fused: -ffp-contract=off); expf, atan2f,
// sincosf are the written-out functions of sift_math.h, which the oracle compiles too -- orientations are therefore
// bit-identical to the oracle's, descriptors differ only by the summation order of the histogram (~1e-7).
#include "sift_device.h"

namespace cusift {


// ------------------------------------------------------------------------------------------------
// Software model of the CUDA texture fetch the reference relies on:
// tex2D<float>(x, y), cudaFilterModeLinear, clamp, unnormalised coordinates (cuSIFT.cu:227-233).
// xB = x - 0.5, i = floor(xB), alpha = frac(xB) rounded to `frac_bits` bits.  Same operation order
// as oracle_tex2d (first product, then three fused multiply-adds).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float tex2d(const float *__restrict__ img, int w, int h, int pitch, RowWindow rw, float x,
                                       float y, float q, float inv_q) {
  const float xb = x - 0.5f, yb = y - 0.5f;
  float fx = floorf(xb), fy = floorf(yb);
  float a = xb - fx, b = yb - fy;
  if (q > 0.0f) {
    a = floorf(fmaf(a, q, 0.5f)) * inv_q;  // q is a power of two: a*q is exact, * inv_q == / q exactly
    b = floorf(fmaf(b, q, 0.5f)) * inv_q;
  }
  fx = fminf(fmaxf(fx, -1.0f), (float)w);
  fy = fminf(fmaxf(fy, -1.0f), (float)rw.hg);
  const int i = (int)fx, j = (int)fy;
  const int i0 = clampi(i, 0, w - 1), i1 = clampi(i + 1, 0, w - 1);
  const int j0 = local_row(j, h, rw), j1 = local_row(j + 1, h, rw);
  const float s00 = img[(long)j0 * pitch + i0], s10 = img[(long)j0 * pitch + i1];
  const float s01 = img[(long)j1 * pitch + i0], s11 = img[(long)j1 * pitch + i1];
  // The four weights: the products of exact fractions, or (quantised) bilinear_weights<true>'s fixed-point ones -- there the
  // products are exact too (multiples of 2^-16) and W11 = round(a b) differs from a b by d, |d| <= 2^-9, which the other
  // three take up: a (1-b) + d == a - W11 and so on, every operation exact.  With exact fractions d is +0.
  const float ia = 1.0f - a, ib = 1.0f - b, ab = a * b;
  const float w11 = q > 0.0f ? floorf(fmaf(ab, q, 0.5f)) * inv_q : ab;
  const float d = ab - w11;
  float t = (ia * ib - d) * s00;
  t = fmaf(a * ib + d, s10, t);
  t = fmaf(ia * b + d, s01, t);
  t = fmaf(w11, s11, t);
  return t;
}

// ------------------------------------------------------------------------------------------------
// LDS-staged image patches.  A patch holds S[clamp(y0+r)][clamp(x0+c)] for the UNCLAMPED coordinates
// (x0+c, y0+r), so a bilinear fetch whose 2x2 footprint lies inside the patch reads exactly the pixels the
// clamping texture model would (a footprint clamped at the border interpolates between equal values in both
// cases).  tex2d_patch() is tex2d() operation for operation, reading LDS.
// ------------------------------------------------------------------------------------------------
constexpr int kDescPatch = 40;  // LDS patch edge: covers descriptor windows up to scale ~2.1 at 45 degrees

struct PatchGeom {
  int x0, y0;  // image coordinate of patch element (0,0)
  int stride;  // floats per patch row
};

// kWait = false: the loads are left in flight -- the caller has work that does not read the patch and waits for
// vmcnt(0) itself (describe_all_kernel).
template <bool kWait = true>
__device__ __forceinline__ void stage_patch(const float *__restrict__ img, int w, int h, int pitch, RowWindow rw,
                                            float *lds, const PatchGeom &g, int pw, int ph, int lane) {
  // Everything about the patch is wave-uniform (it derives from the keypoint's fields).  One instruction per patch
  // row, lane = column, written by the memory pipeline straight into LDS (`buffer_load_dword ... lds`: LDS address =
  // M0 + 4 * lane, lanes at or beyond the patch width masked off): no staging registers, so ALL rows of the patch are
  // in flight at once and nothing but one vmcnt(0) stands between the last load and the first tap.
  // History (phase stamps, tools/exp_describe_stamps.sh; 64 x 1080p, cycles per keypoint in this function):
  //   row arithmetic on the scalar unit, 8 loads in flight, register staging + ds_write   5,450
  //   row offsets by lane + v_readlane, 16 in flight                                      4,380
  //   LDS-DMA                                                                             3,580   (wide patches: 9,700 -> 4,300)
  const int x0 = __builtin_amdgcn_readfirstlane(g.x0), y0 = __builtin_amdgcn_readfirstlane(g.y0);
  pw = __builtin_amdgcn_readfirstlane(pw);
  ph = __builtin_amdgcn_readfirstlane(ph);
  // Round 4: a patch that touches no border of the image (nearly all of them) needs no clamp, so its rows are a linear
  // walk -- and gfx950's LDS-DMA moves 16 bytes per lane: lane l = (row l / 10, columns 4 (l % 10) .. + 3) of a group
  // of SIX patch rows per instruction (10 lanes x 4 floats = the 40-float row stride: LDS address M0 + 16 l is exactly
  // the row-major patch), the row group chosen by the SCALAR offset.  4-7 load instructions per patch instead of 21-40,
  // no per-row v_readlane.  Patches at a border keep the row-per-instruction form below (per-column clamp).
  if constexpr (true) {
    const int gx = (pw + 3) & ~3;  // columns loaded: whole groups of four
    const int lo_row = rw.row0 > 0 ? rw.row0 : 0;
    const int hi_row = (rw.row0 + h < rw.hg ? rw.row0 + h : rw.hg) - 1;  // last global row held locally
    const bool interior = g.stride == kDescPatch && x0 >= 0 && x0 + gx <= w && y0 >= lo_row && y0 + ph - 1 <= hi_row;
    if (interior) {  // wave-uniform
      const int row_first = y0 - rw.row0;
      const long left = ((long)(h - row_first) * pitch) * 4;
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          (void *)(img + (long)row_first * pitch), 0, (int)(left < 0x7fffffffL ? left : 0x7fffffffL), kBufFlags);
      const int lr = (lane * 205) >> 11;  // lane / 10 for lane < 64
      const int lc = lane - 10 * lr;
      const int voff = (lr * pitch + x0 + 4 * lc) * 4;
      const bool mine = lane < 60 && 4 * lc < pw;
      const int group_bytes = 6 * pitch * 4;
      for (int k = 0, r0 = 0; r0 < ph; ++k, r0 += 6) {  // wave-uniform trip count
        if (mine && lr + r0 < ph) {
          auto *dst = (__attribute__((address_space(3))) void *)(lds + r0 * kDescPatch);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, k * group_bytes, 0, 0);
        }
      }
      if (kWait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      return;
    }
  }
  const int voff = clampi(x0 + lane, 0, w - 1) * 4;
  // buffer descriptor re-based at the patch's first row: offsets stay small whatever the image size
  const int row_first = local_row(y0, h, rw);
  const long left = ((long)(h - row_first) * pitch) * 4;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(img + (long)row_first * pitch), 0, (int)(left < 0x7fffffffL ? left : 0x7fffffffL), kBufFlags);
  // byte offset of patch row i, computed by lane i for all rows at once (ph <= 64) and handed to the loads with
  // v_readlane (on the scalar unit the two clamps and the multiply are ~22 dependent instructions per row)
  const int row_off = (local_row(y0 + lane, h, rw) - row_first) * (pitch * 4);
  if (lane < pw) {
    for (int r = 0; r < ph; ++r) {
      const int off = __builtin_amdgcn_readlane(row_off, r);
      auto *dst = (__attribute__((address_space(3))) void *)(lds + r * g.stride);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 4, voff, off, 0, 0);
    }
  }
  if (kWait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the patch is in LDS (this also waits for the wave's older stores)
}

// The four weights of the texture model (oracle_tex2d).  Quantised (the reference's texture unit): a = A / q and
// b = B / q are multiples of 1/q = 2^-8 in [0, 1], and so are the WEIGHTS -- the unit rounds the product to the
// fractions' precision, W11 = floor(A B / q + 1/2), and takes the other three by subtraction (W10 = A - W11,
// W01 = B - W11, W00 = q - A - B + W11): round 6, pinned on the reference's golden orientations (4,095 of 4,095 rows
// within 6.1e-5 degree with this rule, 35 % within 1e-3 with the exact 16-bit products the kernels formed until then).
// Every operation below is exact in fp32 (a b is a multiple of 2^-16 below 1; the rest are multiples of 2^-8 below 2),
// so these are the bits of the oracle's W / q whatever the route.
template <bool kQuant>
__device__ __forceinline__ void bilinear_weights(float a, float b, float q, float inv_q, float &w00, float &w10,
                                                 float &w01, float &w11) {
  if (kQuant) {
    w11 = floorf(fmaf(a * b, q, 0.5f)) * inv_q;
    w10 = a - w11;
    w01 = b - w11;
    w00 = (1.0f - a) - w01;
  } else {
    const float ia = 1.0f - a, ib = 1.0f - b;
    w00 = ia * ib, w10 = a * ib, w01 = ia * b, w11 = a * b;
  }
}

template <int kStride, bool kQuant>
__device__ __forceinline__ float tex2d_patch(const float *lds, int x0, int y0, float x, float y, float q, float inv_q) {
  const float xb = x - 0.5f, yb = y - 0.5f;
  const float fx = floorf(xb), fy = floorf(yb);
  float a = xb - fx, b = yb - fy;
  if (kQuant) {
    a = floorf(fmaf(a, q, 0.5f)) * inv_q;
    b = floorf(fmaf(b, q, 0.5f)) * inv_q;
  }
  // element index (fy - y0) * kStride + (fx - x0): fy * kStride + fx is formed in floating point -- both terms are
  // integers below 2^23 (patch_for_reach bounds |coordinates| by 1e5), so the fma is exact -- and the patch origin,
  // which is wave-uniform, comes off as one integer: v_fma + v_cvt + a subtraction that folds into the scalar part of
  // the address, instead of two conversions, two integer subtractions and an integer multiply
  const int e = (int)fmaf(fy, (float)kStride, fx) - (y0 * kStride + x0);
  const float *p0 = lds + e;
  const float *p1 = p0 + kStride;
  const float s00 = p0[0], s10 = p0[1], s01 = p1[0], s11 = p1[1];
  float w00, w10, w01, w11;
  bilinear_weights<kQuant>(a, b, q, inv_q, w00, w10, w01, w11);
  float t = w00 * s00;
  t = fmaf(w10, s10, t);
  t = fmaf(w01, s01, t);
  t = fmaf(w11, s11, t);
  return t;
}

// The DESCRIPTOR's tap (round 4): tex2d_patch operation for operation -- same index, same 8-bit fractions, same four
// products in the same chain -- with x and y travelling as one register pair through packed fp32 operations (one v_pk_*
// at 4.2 cycles for two fast-class operations at 2.65 each, and for two SGPR-operand ones at 4.2 each).
// (Tried and dropped: the interpolation as three lerps -- two packed operations on the row pairs + one fma, five
// instructions fewer per tap.  A flat neighbourhood then interpolates to EXACTLY its value, while the reference's
// four-product form leaves +-1 ulp of the pixel value: for low-contrast keypoints -- gradients of ~0.05 grey levels on
// pixels of ~100 -- that ulp is 2e-4 of the gradient and the descriptors left the 1e-4 bar.)
template <int kStride, bool kQuant>
__device__ __forceinline__ float tex2d_patch_desc(const float *lds, int x0, int y0, f2 p, float q, float inv_q) {
  const f2 pb = p - f2{0.5f, 0.5f};
  const f2 fl = f2{floorf(pb.x), floorf(pb.y)};
  f2 ab = pb - fl;
  if (kQuant) {  // A, B: the fractions as integers in [0, q] -- the weights below stay integers too (see the return)
    const f2 t = __builtin_elementwise_fma(ab, f2{q, q}, f2{0.5f, 0.5f});
    ab = f2{floorf(t.x), floorf(t.y)};
  }
  const int e = (int)fmaf(fl.y, (float)kStride, fl.x) - (y0 * kStride + x0);
  const float *p0 = lds + e;
  const float *p1 = p0 + kStride;
  const float s00 = p0[0], s10 = p0[1], s01 = p1[0], s11 = p1[1];
  float w00, w10, w01, w11;
  if (kQuant) {
    // bilinear_weights<true> times q: W11 = floor(A B / q + 1/2), W10 = A - W11, W01 = B - W11, W00 = q - A - B + W11
    // (every step exact), the two middle subtractions as one packed operation.  The tap then comes out as q TIMES the
    // texture value -- the same four operations on operands scaled by a power of two, so exactly q times the oracle's
    // bits -- and the caller folds the 1/q into a factor it multiplies with anyway (PatchSampler::desc_scale): one
    // instruction more per tap than the exact-product weights of rounds 1-5 instead of three.
    w11 = floorf(fmaf(ab.x * ab.y, inv_q, 0.5f));
    const f2 wm = ab - f2{w11, w11};
    w10 = wm.x;
    w01 = wm.y;
    w00 = (q - ab.x) - w01;
  } else {
    bilinear_weights<false>(ab.x, ab.y, q, inv_q, w00, w10, w01, w11);
  }
  float t = w00 * s00;
  t = fmaf(w10, s10, t);
  t = fmaf(w01, s01, t);
  t = fmaf(w11, s11, t);
  return t;
}

// The keypoint kernels run ONE wave per workgroup, so "all threads of the block have written LDS" only needs
// this wave's LDS operations to have completed (they execute in order): wait for lgkmcnt, not for outstanding
// global loads/stores as __syncthreads() would (its vmcnt(0) drain cost ~1-2 us per keypoint).
__device__ __forceinline__ void wave_sync() {
  // compiler barrier for memory + "my LDS operations are done"; no s_barrier (one wave), no vmcnt wait
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// Phase stamps (tools/describe_stamps.py; a build with -DCUSIFT_STAMPS, never the product): lane 0 adds the shader-clock
// time since the previous stamp to counter i of the wave, kept in the `hist` words of the wave's LDS block (unused by the
// stages since the orientation's tail moved into registers); describe_all_kernel adds them to g_describe_stamps at its end.
#ifdef CUSIFT_STAMPS
__device__ unsigned long long g_describe_stamps[16];
#define STAMP(S, i)                                                                          \
  do {                                                                                       \
    const unsigned int t_ = (unsigned int)__builtin_amdgcn_s_memtime();                      \
    if (threadIdx.x == 0) {                                                                  \
      unsigned int *h_ = (S).stamps;                                                         \
      h_[i] += t_ - h_[15];                                                                  \
      h_[15] = t_;                                                                           \
    }                                                                                        \
  } while (0)
#else
#define STAMP(S, i) do { } while (0)
#endif

// Where a keypoint's taps come from: an LDS patch (usual) or global memory (footprint larger than the patch).
// The choice is wave-uniform and made ONCE per keypoint -- the whole keypoint body is instantiated per sampler
// type, so there is no branch, no runtime stride multiply and no vmcnt wait around the individual taps (with one
// run-time Sampler the compiler guarded each of the 24 taps per lane with two branches).
template <int kStride, bool kQuant>
struct PatchSampler {
  static constexpr bool kIsPatch = true;
  static constexpr int kPatchStride = kStride;
  static constexpr bool kQuantised = kQuant;
  const float *patch;
  int x0, y0;  // image coordinate of patch element (0,0)
  float q, inv_q;
  __device__ __forceinline__ float operator()(float x, float y) const {
    return tex2d_patch<kStride, kQuant>(patch, x0, y0, x, y, q, inv_q);
  }
  // a descriptor tap at p = (x, y): the same operations with x and y as a register pair (tex2d_patch_desc) -- quantised:
  // q times the texture value (the weights are left as integers; desc_lane_consts folds the 1 / q)
  __device__ __forceinline__ float desc(f2 p) const { return tex2d_patch_desc<kStride, kQuant>(patch, x0, y0, p, q, inv_q); }
};
struct GlobalSampler {
  static constexpr bool kIsPatch = false;
  static constexpr int kPatchStride = 1;
  static constexpr bool kQuantised = false;
  const float *img;
  int w, h, pitch;
  RowWindow rw;
  float q, inv_q;
  __device__ __forceinline__ float operator()(float x, float y) const {
    return tex2d(img, w, h, pitch, rw, x, y, q, inv_q);
  }
  // (the descriptor's taps are q times the value for a quantised model, as the patch sampler's: desc_lane_consts)
  __device__ __forceinline__ float desc(f2 p) const {
    const float t = tex2d(img, w, h, pitch, rw, p.x, p.y, q, inv_q);
    return q > 0.0f ? t * q : t;
  }
};

// A value every lane loaded from the same address, declared wave-uniform: it then lives in an SGPR and whatever
// is derived from it alone (patch origin, loop bounds, the patch-or-global decision) is scalar work.
__device__ __forceinline__ float uniform(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// Where descriptor sample (row y, column tx) of the 16x16 grid lives in grad() / angfrac().  Two access patterns must
// both be conflict-free: phase 1 writes rows {q + 4 step} x 16 columns (q = lane / 16) in one instruction, the vertical
// pass of the gather reads rows {4 vi - 2 + r} x 16 columns (vi = lane / 16) in one instruction -- four rows 4 apart.  With
// the plain index 16 y + tx rows 4 apart are 64 floats apart: the same banks.  The skew 16 (y >> 2) moves row y to bank
// 16 (y + (y >> 2)) mod 64: four consecutive rows of one group of four AND four rows 4 apart each cover the 64 banks once.
constexpr int kDescSlots = 16 * 16 + 16 * 3;  // 304
constexpr int kAngOffset = 320;               // >= kDescSlots, a multiple of 64
__device__ __forceinline__ int desc_slot(int y, int tx) { return 16 * y + tx + 16 * (y >> 2); }

// The lane-private angle histograms of the gather's vertical pass: kHistSlots rows of 64 floats, [slot][position], lane
// (x = lane % 16, vi = lane / 16) at position 16 vi + (x + 2) % 16 -- a bijection of the lanes onto the 64 banks whatever
// slots they pick; the rotation by 2 makes the 8-column window of a cell, columns 4 hi - 2 .. 4 hi + 5, start on a
// 16-byte boundary (two ds_read_b128 in the horizontal pass).
//   slots 0..7  angle bins;   slot 8  bin 0 again (the upper neighbour of bin 7: one address register + an immediate for
//   both adds);   slot 9  the reference's angle-index 8 (atan2f == +pi: its index 8 + ... is bin 0 of the NEXT linear
//   cell, cuSIFT_D.cu:233-255) and slot 10 its upper neighbour (bin 0 of the sample's own cell);   slot 11 padding
constexpr int kHistSlots = 12;
__device__ __forceinline__ int hist_pos(int x, int vi) { return 16 * vi + ((x + 2) & 15); }

// LDS of one keypoint wave (9.0 KB => 16 waves per CU together with the 1 KB prefix table of describe_all).
// The descriptor's histogram buffers are only needed after its sampling phase, when the patch is dead, so they
// live inside the patch storage.
struct alignas(16) KpShared {
  // first, i.e. at LDS address 0 (the compiler places this 16-byte aligned object before the prefix table): a tap's four
  // pixels are then offset0:0/1 and offset0:40/41 of ONE address register, with no base to add (-2 instructions per tap)
  float patch[kDescPatch * kDescPatch];
  float hist[64];  // (unused since the orientation's tail moved into registers; kept: the layout below is tuned)
#ifdef CUSIFT_STAMPS
  unsigned int stamps[16];  // (the stamps build halves kMaxFlatImages to make room: same 16 waves per CU)
#endif
  float gauss[11];      // orientation window; gauss[0] also carries the finished orientation to all lanes
  float pad_[1];        // keeps scratch 16-byte aligned (float4 stores)
  // shared by the two stages (they never overlap in time):
  //   orientation: wmat() = [2 halves][64 samples] (bin, weight) list of the samples being summed
  //   descriptor : grad() / angfrac() = weighted gradient magnitude (its four lowest mantissa bits carry the sample's
  //                angle code, see kp_descriptor) and the angle fraction of the 16x16 samples, stored at desc_slot(y, tx);
  //                the two arrays lie kAngOffset = 5 x 64 floats apart, so one ds_read2st64_b32 / ds_write2st64_b32
  //                moves a sample's pair
  float scratch[kAngOffset + kDescSlots];
  __device__ __forceinline__ float *wmat() { return scratch; }
  __device__ __forceinline__ float *grad() { return scratch; }
  __device__ __forceinline__ float *angfrac() { return scratch + kAngOffset; }
  __device__ __forceinline__ float *cellhist() { return patch; }                     // [kHistSlots][64 positions]
  __device__ __forceinline__ float *fin() { return patch + 64 * kHistSlots; }        // 128
};
static_assert(kAngOffset + kDescSlots >= 256 && kAngOffset % 64 == 0 && kAngOffset >= kDescSlots,
              "the orientation stage's sample list lives in the same storage");
static_assert(64 * kHistSlots + 128 <= kDescPatch * kDescPatch, "histogram buffers must fit in the patch storage");

// LDS of the orientation-only stage kernel
struct alignas(16) OriShared {
  float patch[16 * 16];
  float hist[64];
  float gauss[11];
  float pad_[1];
#ifdef CUSIFT_STAMPS
  unsigned int stamps[16];
#endif
  float scratch[256];  // the (bin, weight) list of the 128 sample slots (kp_orientation)
  __device__ __forceinline__ float *wmat() { return scratch; }
};

// ------------------------------------------------------------------------------------------------
// Dominant orientation.  Reference: ComputeOrientations_D, cuSIFT_D.cu:319-396 (second peak compiled out, :380).
// 121 samples (11x11) -> 32-bin histogram: bins are lanes, each half-wave walks half of the samples in index
// order, so the sums are deterministic and in the oracle's order.  Every lane returns the orientation (degrees).
// ------------------------------------------------------------------------------------------------
// max over the 16 lanes of each DPP row, then over the row pairs (0,1) and (2,3): lanes 0..31 all return the
// maximum of lanes 0..31.  max() is exact, so the order of the reduction does not matter.
template <int kCtrl>
__device__ __forceinline__ float dpp_perm(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), kCtrl, 0xf, 0xf, true));
}
__device__ __forceinline__ float max_over_32(float v) {
  v = fmaxf(v, dpp_perm<0xB1>(v));   // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_perm<0x4E>(v));   // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_perm<0x141>(v));  // row_half_mirror
  v = fmaxf(v, dpp_perm<0x140>(v));  // row_mirror
  return fmaxf(v, __shfl_xor(v, 16));
}
// The maximum of lanes 0..31 as a wave-uniform value, without the LDS round trip of the cross-row shuffle above: the
// four in-row steps, then row 0's lane 15 into row 1 (DPP row_bcast:15) and lane 31 read back.  max() is exact and
// order-free.
__device__ __forceinline__ float max_of_lanes_0_31(float v) {
  v = fmaxf(v, dpp_perm<0xB1>(v));
  v = fmaxf(v, dpp_perm<0x4E>(v));
  v = fmaxf(v, dpp_perm<0x141>(v));
  v = fmaxf(v, dpp_perm<0x140>(v));
  asm("s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 0"
      : "+v"(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
}

// What the orientation stage computes from the keypoint's fields alone -- nothing that reads the patch: the Gaussian
// window's table (into S.gauss) and the lattice shortcut's test and weights.  A function of its own so that
// describe_all_kernel can run it while the patch's loads are in flight.
struct OriPrep {
  bool lattice;
  float w00, w10, w01, w11;
  int e_c;  // patch element of the pixel that holds floor(k - 0.5) of the window's centre sample
};

// Sum over the wave, every lane returns it, in whatever association is cheapest: four in-row DPP steps, then the two
// row broadcasts gfx9 has for exactly this (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) -- 6 vector
// instructions and a v_readlane (the reference's own tree, cuSIFT_D.cu:262-280, on lane swaps and DPP shifts: 17).  For the
// descriptor's two norms, whose summation order is free.
__device__ __forceinline__ float wave_sum64(float v) {
  v += dpp_perm<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_perm<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_perm<0x141>(v);  // row_half_mirror
  v += dpp_perm<0x140>(v);  // row_mirror: every lane holds its row's sum
  // (inline assembly: from the builtin the compiler makes a zeroed register, a DPP move and an addition of each step; the
  // rows outside row_mask keep their value.  s_nop 1: the two wait states between a VALU write and a DPP read of it.)
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 0"
      : "+v"(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// The orientation histogram's bin of a gradient, cuSIFT_D.cu:349: bin = (int)(16 atan2f(dy, dx) / 3.1416f + 16.5f), 32 -> 0.
// The bin is a DISCRETE function of the gradient's direction, so it needs no angle: with a = atan(min / max) of the two
// magnitudes in [0, pi/4], every octant holds four bin edges, at a = (j - 1/2) 3.1416 / 16 up to the octant's share of
// (3.1416 - pi) -- the same four edges in all eight octants to within 5.4e-6 rad (2.8e-5 of a bin) -- and the bin is the
// octant's first bin plus or minus the number of edges below the direction: cnt = #{j : mn > T_j mx}, T_j the tangent of
// the middle of the eight octants' edges.  No division, no polynomial, no pi: four fused multiply-adds, four compares.
// EXACTNESS: the oracle evaluates the reference's formula in fp32 (sm_atan2f <= 2 ulp, one division, one rounding at
// magnitude 32: within 1e-5 of a bin of the real value), the shared edges are within 2.8e-5 of a bin of the true ones;
// a sample whose direction lies within kOriEdgeEps * mx of an edge in d = mn - T mx -- at least 1.5e-5 rad = 7.6e-5 of a
// bin on either side of every edge, twice what the two can differ by -- is reported `near`, and the caller sends the
// whole wave through the reference's formula behind a ballot (3 % of the keypoints hold such a sample).  Outside that
// margin the two bins are equal by the bound above; tests/test_gpu_parity.py::test_orientation_bin_shortcut checks it
// on 2^22 random directions and on samples dense at every edge (cusift_math_eval op 5), test_orientations_match_oracle
// and the fuzz sweep on whole orientations (bit-identical).  Zero gradients (weight 0) are never `near`; NaN and
// infinite ones are.
constexpr float kOriEdgeT1 = 0.09848980424854088f, kOriEdgeT2 = 0.3033454482644209f, kOriEdgeT3 = 0.5345102632831631f,
                kOriEdgeT4 = 0.820678412197299f, kOriEdgeEps = 2.5e-5f;
__device__ __forceinline__ int ori_bin_shortcut(float dy, float dx, bool &near) {
  const float ax = sm_abs(dx), ay = sm_abs(dy);
  const bool swap = ay > ax;
  const float mx = swap ? ay : ax, mn = swap ? ax : ay;
  const float d1 = fmaf(-kOriEdgeT1, mx, mn), d2 = fmaf(-kOriEdgeT2, mx, mn), d3 = fmaf(-kOriEdgeT3, mx, mn),
              d4 = fmaf(-kOriEdgeT4, mx, mn);
  const int cnt = (int)(d1 > 0.0f) + (int)(d2 > 0.0f) + (int)(d3 > 0.0f) + (int)(d4 > 0.0f);
  const float m = fminf(fminf(sm_abs(d1), sm_abs(d2)), fminf(sm_abs(d3), sm_abs(d4)));
  near = !(m > kOriEdgeEps * mx) && !(mx == 0.0f);  // a NaN anywhere: near
  const bool negx = (sm_bits(dx) & 0x80000000u) != 0u, negy = (sm_bits(dy) & 0x80000000u) != 0u;
  // position in bins from the positive x axis, as if dy >= 0: 8 h from the quadrant boundary, +- the edges passed
  const int h = negx ? 2 - (int)swap : (int)swap;
  const int p = 8 * h + ((swap != negx) ? -cnt : cnt);
  const int bin = negy ? 16 - p : 16 + p;
  return bin & 31;  // 32 -> 0 (cuSIFT_D.cu:350-351)
}

template <typename SH, typename TEX>
__device__ __forceinline__ OriPrep kp_orientation_prep(SH &S, const TEX &tex, float kx, float ky, float scale, int tx) {
  const float i2sigma2 = -1.0f / (4.5f * scale * scale);
  if (tx < 11) S.gauss[tx] = sm_expf(i2sigma2 * (tx - 5) * (tx - 5));
  // Lattice shortcut.  The 484 taps of a keypoint sit at kx + n/2, ky + m/2 (n, m integers, |n|,|m| <= 13).  Let
  // u = ulp(kx) (<= 1/2 for kx < 2^23).  Every intermediate of the reference's arithmetic ((kx - 5) + i) +- 1 - 0.5 is
  // a multiple of u; it is exactly representable -- so no operation rounds -- as long as it does not climb into the
  // binade ABOVE kx's, where the spacing doubles (below kx's binade the spacing only gets finer).  If kx + 8 is still
  // in kx's binade, all taps therefore share ONE pair of bilinear fractions (frac(kx - 0.5), frac(ky - 0.5)) and
  // their footprints are integer shifts of each other: the weights are formed once per keypoint and a tap costs one
  // product and three fused multiply-adds -- the same operations on the same operands as the general path, so the
  // same bits.  Keypoints within 8 px below a power of two (or with a coordinate < 16) take the general path.
  bool lattice = false;
  float w00 = 0.f, w10 = 0.f, w01 = 0.f, w11 = 0.f;
  int e_c = 0;
  if constexpr (TEX::kIsPatch) {
    const float xhi = kx + 8.0f, yhi = ky + 8.0f;
    lattice = kx >= 1.0f && ky >= 1.0f && xhi < 4.0e6f && yhi < 4.0e6f &&
              (__float_as_int(kx) >> 23) == (__float_as_int(xhi) >> 23) &&
              (__float_as_int(ky) >> 23) == (__float_as_int(yhi) >> 23);
    if (lattice) {  // wave-uniform
      const float xb = kx - 0.5f, yb = ky - 0.5f;  // the centre sample's T(x, y) coordinate: (xp + 5) == kx exactly
      const float fx = floorf(xb), fy = floorf(yb);
      float a = xb - fx, b = yb - fy;
      if (TEX::kQuantised) {
        a = floorf(fmaf(a, tex.q, 0.5f)) * tex.inv_q;
        b = floorf(fmaf(b, tex.q, 0.5f)) * tex.inv_q;
      }
      bilinear_weights<TEX::kQuantised>(a, b, tex.q, tex.inv_q, w00, w10, w01, w11);
      e_c = ((int)fy - tex.y0) * TEX::kPatchStride + ((int)fx - tex.x0);
    }
  }
  return OriPrep{lattice, w00, w10, w01, w11, e_c};
}

template <typename SH, typename TEX>
__device__ __forceinline__ float kp_orientation(SH &S, const TEX &tex, const OriPrep &P, float kx, float ky, float scale,
                                                int tx) {
  const float xp = kx - 5.0f;
  const float yp = ky - 5.0f;
  const bool lattice = P.lattice;
  const float w00 = P.w00, w10 = P.w10, w01 = P.w01, w11 = P.w11;
  const int e_c = P.e_c;
  wave_sync();
  int sbin[2] = {0, 0};            // this lane's two samples: histogram bin ...
  float swgt[2] = {0.0f, 0.0f};    // ... and weight (a second sample exists for tx < 57)
  // Both samples are computed without a branch between them (lanes 57..63 repeat a sample; their second weight is
  // never posted), so that the two gradient / atan2f / sqrtf chains -- each a long run of dependent operations --
  // interleave instead of running one after the other.
  constexpr int kReps = 2;
  int xd[2], yd[2];
  float dx[2], dy[2];
  // Which two of the 121 samples a lane computes is free -- only the LIST below is in sample order.  Round 4: the first
  // 64 lanes' samples are rows 0..7 x columns 0..7 of the window: in the 40-float patch rows those are 8 groups of 8
  // consecutive banks (40 yd mod 64 = 0, 40, 16, 56, 32, 8, 48, 24) -- every tap read of the first sample is
  // conflict-free (sample t = lane put row pairs on the same banks: ~2-way conflicts on all of them).  Second samples:
  // rows 8..10 complete (lanes 0..32), then columns 8..10 of rows 0..7 (lanes 33..56); lanes 57..63 repeat the last one
  // and post weight 0 at the list's padding slots 121..127.
  int tidx[2];  // sample index t = 11 yd + xd: where the sample goes in the list
  yd[0] = tx >> 3;
  xd[0] = tx & 7;
  {
    const int k = tx < 33 ? tx : min(tx, 56) - 33;
    const int q = tx < 33 ? k / 11 : k / 3;
    yd[1] = tx < 33 ? 8 + q : q;
    xd[1] = tx < 33 ? k - 11 * q : 8 + (k - 3 * q);
  }
  tidx[0] = 11 * yd[0] + xd[0];
  tidx[1] = tx < 57 ? 11 * yd[1] + xd[1] : 64 + tx;
  bool done = false;
  if constexpr (TEX::kIsPatch) {
    if (lattice) {  // wave-uniform
      constexpr int RS = TEX::kPatchStride;
#pragma unroll
      for (int rep = 0; rep < kReps; ++rep) {
        // P(r, c): pixel (floor(yb) + r, floor(xb) + c) of this sample's own T(x, y) footprint origin
        const float *P = tex.patch + e_c + (yd[rep] - 5) * RS + (xd[rep] - 5);
        auto T = [&](int r, int c) {  // T(x + c, y + r): footprint rows r, r+1 x columns c, c+1
          float v = w00 * P[r * RS + c];
          v = fmaf(w10, P[r * RS + c + 1], v);
          v = fmaf(w01, P[(r + 1) * RS + c], v);
          v = fmaf(w11, P[(r + 1) * RS + c + 1], v);
          return v;
        };
        dx[rep] = T(0, 1) - T(0, -1);
        dy[rep] = T(1, 0) - T(-1, 0);
      }
      done = true;
    }
  }
  if (!done) {
#pragma unroll
    for (int rep = 0; rep < kReps; ++rep) {
      const float xf = xp + xd[rep];
      const float yf = yp + yd[rep];
      dx[rep] = tex(xf + 1.0f, yf) - tex(xf - 1.0f, yf);
      dy[rep] = tex(xf, yf + 1.0f) - tex(xf, yf - 1.0f);
    }
  }
  bool near_edge = false;
#pragma unroll
  for (int rep = 0; rep < kReps; ++rep) {
    bool near;
#ifdef CUSIFT_AB_ORI_FORMULA  // (an A/B build: the reference's formula for every sample, as rounds 1-5)
    near = true;
#else
    sbin[rep] = ori_bin_shortcut(dy[rep], dx[rep], near);  // the bin without an angle (see ori_bin_shortcut)
#endif
    near_edge = near_edge || near;
    const float grad = sqrtf(dx[rep] * dx[rep] + dy[rep] * dy[rep]);
    swgt[rep] = grad * S.gauss[xd[rep]] * S.gauss[yd[rep]];
  }
  if (__ballot(near_edge) != 0ull) {  // wave-uniform, ~3 % of the keypoints: some sample sits at a bin edge (or is not finite)
#pragma unroll
    for (int rep = 0; rep < kReps; ++rep) {
      int bin = (int)(16.0f * sm_atan2f(dy[rep], dx[rep]) / 3.1416f + 16.5f);  // 0..32; v_cvt_i32_f32 turns a NaN into 0
      if ((unsigned int)bin > 31u) bin = 0;  // 32 -> 0 as in the reference (cuSIFT_D.cu:352); also memory safety
      sbin[rep] = bin;
    }
  }
  STAMP(S, 4);
  float hist_half;  // this lane's (half, bin) sum
  {
    // Histogram without LDS atomics.  Bins are lanes (tx & 31); the lower half-wave sums samples 0..63 in index order,
    // the upper half-wave samples 64..120, then hist = lower + upper -- the oracle accumulates in exactly this order (the
    // reference's LDS atomics have none).
    // (bin, weight) of every sample as a list in LDS, lower-half samples at [0, 64), upper-half at [64, 128); lane
    // (half, bin) then walks its half's list in index order -- 32 broadcast ds_read_b128 (two samples each, every
    // lane of a half-wave reads the same address: no conflicts), all issued back to back -- and adds the weights of
    // its own bin (+0 for the others: adding +0 to a non-negative sum is exact).  One LDS round trip and 140 LDS cycles
    // per keypoint; round 2's one-hot matrix (eight steps of post / read 8 rows / clear) was eight DEPENDENT round trips
    // and 256 LDS cycles for fewer vector instructions (80 against 192) -- in a kernel bound by its LDS pipe and by the
    // latency of its dependent chains the list is 3 % faster (same-box A/B, 64 x 1080p: 0.476 -> 0.461 ms).
    // The accesses are scalar on purpose: with f2 stores / f4 loads of the same words this compiler selected the wrong
    // vector components (seen in the ISA: one compare used for both samples of a ds_read_b128).
    unsigned int *list = reinterpret_cast<unsigned int *>(S.wmat());  // [128][2]: bin, weight bits
    list[2 * tidx[0]] = (unsigned int)sbin[0];
    list[2 * tidx[0] + 1] = __builtin_bit_cast(unsigned int, swgt[0]);
    list[2 * tidx[1]] = (unsigned int)sbin[1];
    list[2 * tidx[1] + 1] = tx < 57 ? __builtin_bit_cast(unsigned int, swgt[1]) : 0u;
    wave_sync();
    const unsigned int mybin = (unsigned int)(tx & 31);
    const unsigned int *src = list + (tx >> 5) * 128;
    float acc = 0.0f;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      const unsigned int b0 = src[4 * k], w0 = src[4 * k + 1], b1 = src[4 * k + 2], w1 = src[4 * k + 3];
      acc += (b0 == mybin) ? __builtin_bit_cast(float, w0) : 0.0f;
      acc += (b1 == mybin) ? __builtin_bit_cast(float, w1) : 0.0f;
    }
    wave_sync();  // the list has been read: the caller may reuse the storage
    hist_half = acc;
  }
  STAMP(S, 5);
  // Round 4: from here on the histogram lives in registers -- no LDS round trip.  (A lone wave issues one vector
  // instruction per ~8 cycles and an LDS round trip costs a wave ~120: the five dependent round trips this tail used to
  // make -- combine the halves, smooth, find the peaks, fetch the peak's neighbours -- were worth ~75 instructions.)
  // hist[b] = lower half's sum + upper half's, in that order (the oracle's), for b = lane % 32 in BOTH halves of the
  // wave: the swap leaves (lower, lower) in `lo` and (upper, upper) in `hi`.
  float lo = hist_half, hi = hist_half;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
  const float h0 = lo + hi;
  // With the 32 bins replicated in both halves, a rotation of the whole wave by one lane IS the circular neighbour
  // over the 32 bins: lane 0 receives lane 63 = bin 31, lane 32 receives lane 31 = bin 31.
  const float h_m1 = dpp_perm<0x13C>(h0), h_p1 = dpp_perm<0x134>(h0);  // wave_ror:1: lane i <- i - 1; wave_rol:1: i <- i + 1
  const float h_m2 = dpp_perm<0x13C>(h_m1), h_p2 = dpp_perm<0x134>(h_p1);
  const float sm = 6.0f * h0 + 4.0f * (h_m1 + h_p1) + (h_m2 + h_p2);  // cuSIFT_D.cu:357-361, the oracle's order
  const float sm_m1 = dpp_perm<0x13C>(sm), sm_p1 = dpp_perm<0x134>(sm);
  const float pk = (sm > sm_m1 && sm >= sm_p1) ? sm : 0.0f;
  // The reference's thread 0 scans the 32 peaks for the first strict maximum (cuSIFT_D.cu:369-379):
  // maxval1 = max(0, max pk), i1 = first index that attains it, -1 if no peak is positive.  Same result from a
  // wave reduction + ballot (pk is never NaN: it comes out of ordered comparisons).
  const float mv = max_of_lanes_0_31(pk);  // wave-uniform (both halves hold the same 32 values)
  const unsigned long long hit = __ballot(tx < 32 && pk == mv && mv > 0.0f);
  const int i1 = hit ? (int)__builtin_ctzll(hit) : -1;
  const int smi = __builtin_bit_cast(int, sm);
  const float val1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(smi, (i1 + 1) & 31));
  const float val2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(smi, (i1 + 31) & 31));
  const float peak = i1 + 0.5f * (val1 - val2) / (2.0f * mv - val1 - val2);
  const float ori = 11.25f * (peak < 0.0f ? peak + 32.0f : peak);
  STAMP(S, 6);
  return ori;
}

template <typename SH, typename TEX>
__device__ __forceinline__ float kp_orientation(SH &S, const TEX &tex, float kx, float ky, float scale, int tx) {
  return kp_orientation(S, tex, kp_orientation_prep(S, tex, kx, ky, scale, tx), kx, ky, scale, tx);
}

// ------------------------------------------------------------------------------------------------
// 128-D descriptor.  Reference: ExtractSiftDescriptors_D, cuSIFT_D.cu:184-297.  The reference adds every sample's eight
// trilinear shares into the 128 bins with LDS float atomics (:230-255) -- in whatever order the hardware retires them:
// NO summation order is prescribed for a bin, and the bar is 1e-4 L2 against the oracle (north_star).  What stays
// operation for operation is everything that DECIDES something: the sample coordinates and the texture model's index /
// 8-bit fraction (a coordinate on a 1/256 step must round the same way) and the angle index.  What is free -- and used
// since round 4 -- is the order and association of the sums and the last bit of sqrt / division / rsqrt, whose effect is
// continuous in the result (measured: largest L2 distance to the oracle ~1e-6 instead of 3e-7).
//   phase 1  the 16x16 rotated sample grid, 4 samples per lane: the Gaussian-weighted gradient magnitude, the angle
//            fraction and the angle index (carried in the magnitude's four lowest mantissa bits: 2^-20 relative) go to
//            LDS.
//   phase 2  SEPARABLE gather (round 4; until then lane (cell, row pair) walked its 18 samples: 18 dependent LDS
//            read-modify-writes and ~220 vector instructions).  With A[y][x][b] = the sample's share for angle bin b,
//            bin[vi][hi][b] = sum_y wy(vi, y) sum_x wx(hi, x) A[y][x][b], and the weights are the same eight numbers in
//            both directions (1/8, 3/8, 5/8, 7/8, 7/8, 5/8, 3/8, 1/8 over the eight rows / columns around a cell).
//            vertical   lane (x, vi) walks the 8 rows 4 vi - 2 .. 4 vi + 5 of column x and adds wy * grad * (1 - frac,
//                       frac) into a lane-private angle histogram in LDS -- the only data-dependent addressing, 8
//                       read-modify-writes, conflict-free, one ds_read2st64 + one ds_write2st64 each.
//            horizontal lane (b mod 4, vi, hi) forms bins b and b + 4 of cell (vi, hi): 8 columns x fixed weights read
//                       with two ds_read_b128 per bin, no data-dependent address, no read-modify-write.
//            The reference's column-14 quirk (guard `tx <= 14`, :243: the right-hand share of column 14 lands in the
//            NEXT row's first cell) is one more term, 1/8 of column 14 of the cell row above.  Its angle-index-8 quirk
//            (atan2f == +pi: index 8 is bin 0 of the next linear cell) is collected in slot 9 and folded in by a branch
//            that a ballot skips for the keypoints that have no such sample (nearly all).
//   phase 3  L2-normalise, clamp at 0.2, L2-normalise (v_rsq_f32; the reference calls rsqrtf, :272,:282).
// Every sum has a fixed order, so results are reproducible run to run.  Lane l returns elements desc_elem(l), + 4.
// ------------------------------------------------------------------------------------------------
struct DescLaneConsts {
  int tx1;
  float gx1;
  float gy1[4];
};

// The descriptor's taps come back as q TIMES the texture value when the texture model is quantised (q > 0: the samplers
// leave their weights as integers, tex2d_patch_desc) -- a power of two.  The gradient's direction does not see it
// (quotient, reciprocal and comparisons scale exactly) and its magnitude sqrt(dx^2 + dy^2) scales by exactly q, so the
// 1/q goes into the column's Gaussian weight here, once per lane and launch: gy * (gx / q) * (q m) is gy * gx * m bit for
// bit.
__device__ __forceinline__ DescLaneConsts desc_lane_consts(int lane, float q, float inv_q) {
  // sample idx = lane + 64*step -> column tx = lane%16, row y = lane/16 + 4*step
  DescLaneConsts c;
  c.tx1 = lane & 15;
  c.gx1 = sm_expf(-(c.tx1 - 7.5f) * (c.tx1 - 7.5f) / 128.0f) * (q > 0.0f ? inv_q : 1.0f);
#pragma unroll
  for (int step = 0; step < 4; ++step) {
    const int y = (lane >> 4) + 4 * step;
    c.gy1[step] = sm_expf(-(y - 7.5f) * (y - 7.5f) / 128.0f);
  }
  return c;
}

// element of the descriptor that lane l holds first (the second is 4 further): cell l % 16 = (vi, hi), bin l / 16.
// (Cells vary fastest so that the 16 lanes the LDS serves together in a ds_read_b128 read 16 different 16-byte groups of
// one histogram row; with the bins fastest, four lanes at a time read the same positions of four different rows -- the
// same banks -- and a quarter of this kernel's bank-conflict cycles came from these four reads.)
__device__ __forceinline__ int desc_elem(int lane) { return 8 * (lane & 15) + (lane >> 4); }

// sqrt and quotient of the DESCRIPTOR samples: one hardware instruction (1 ulp) instead of the IEEE-exact expansions
// (~11 instructions each) the orientation stage needs for its bit-identical histogram bins.
__device__ __forceinline__ float desc_sqrtf(float x) { return __builtin_amdgcn_sqrtf(x); }
// The descriptor's angle coordinate 4/3.1415f * atan2f(dy, dx) + 4 (cuSIFT_D.cu:231-233).  The share a sample hands to
// its two angle bins is continuous in this value, so it need not be sm_atan2f's to the ulp: atan t = t + t s Q(s) with Q
// of degree 4 in s = t^2 (Lawson fit as tools/fit_math_polys.py's: 2.4e-6 rad = 3.1e-6 of a bin; with v_rcp_f32's last
// ulp on t the value is within 4e-6 OF A BIN of the oracle's -- THE bound, the one include/cusift_amd_stages.h states and
// tests/test_gpu_parity.py::test_descriptor_angle_coordinate asserts (measured 3.8e-6) -- which reaches the descriptor as
// ~4e-6 of its norm against the 1e-4 bar) and no low-order word on pi/2 -- 30 instructions per sample where the form
// exact to the ulp took 38.  What is kept operation for operation is the ONE place where the value decides something: at
// 8.0 the reference's index becomes 8 and the share lands in the next cell (cuSIFT_D.cu:233-255) -- a jump, at |atan2f| =
// 3.1415, t = 9.3e-5 from the negative x axis.  There atan t == t in sm_atan2f and here alike (the cubic term is below
// half an ulp), and (pi_hi - t) + pi_lo, the product with 4/3.1415f and the + 4 are the reference's operations: the same
// bits -- but for the last ulp of t itself, which is v_rcp_f32's quotient here as it was before the fit (a pair of
// gradients whose exact quotient lies within 7e-12 of the threshold can land on the other side: ~1e-12 of an image's
// samples; tests/test_gpu_parity.py::test_descriptor_angle_coordinate counts them in a sample dense at the jump).  ((dy = +0, dx < 0) gives 8.000118, (dy = -0, dx < 0) -0.000118; a zero gradient's 0 * inf comes out finite, its
// weight is 0.)  (Measured and dropped: the octant fix-ups in units of bins, three instructions fewer, with the value
// near pi re-formed behind a ballot-skipped branch -- the branch splits the four samples' straight-line code and costs
// more than it saves.)
__device__ __forceinline__ float desc_angle_bins(float dy, float dx) {
  const float ax = sm_abs(dx), ay = sm_abs(dy);
  const bool swap = ay > ax;
  const float mx = swap ? ay : ax;
  const float mn = swap ? ax : ay;
  const float t = fminf(mn * __builtin_amdgcn_rcpf(mx), 1.0f);
  const float s = t * t;
  float q = -0.0128082866f;
  q = sm_fma(q, s, 0.055805929f);
  q = sm_fma(q, s, -0.119818673f);
  q = sm_fma(q, s, 0.195182785f);
  q = sm_fma(q, s, -0.33296597f);
  float r = sm_fma(q * s, t, t);
  if (swap) r = 1.5707963705062866f - r;
  if (sm_bits(dx) & 0x80000000u) r = (3.1415927410125732f - r) + -8.742277657347586e-08f;  // pi = hi + lo, as sm_atan2f
  const float theta = sm_float(sm_bits(r) | (sm_bits(dy) & 0x80000000u));
  return 4.0f / 3.1415f * theta + 4.0f;
}

// sum over the 8-column window of cell column `hi` in histogram row `row` (= cellhist() + 64 slot + 16 vi), plus the
// column-14 term from the cell row above (`above`: that row's position 0 = column 14; w14 = 1/8 for hi == 0, vi >= 1)
__device__ __forceinline__ float window_sum(const float *row, int hi, const f4 wa, const f4 wb, const float *above,
                                            float w14) {
  const f4 a = *reinterpret_cast<const f4 *>(row + 4 * hi);               // columns 4 hi - 2 .. 4 hi + 1
  const f4 b = *reinterpret_cast<const f4 *>(row + ((4 * hi + 4) & 15));  // columns 4 hi + 2 .. 4 hi + 5
  float s = wa.x * a.x;
  s = fmaf(wa.y, a.y, s);
  s = fmaf(wa.z, a.z, s);
  s = fmaf(wa.w, a.w, s);
  s = fmaf(wb.x, b.x, s);
  s = fmaf(wb.y, b.y, s);
  s = fmaf(wb.z, b.z, s);
  s = fmaf(wb.w, b.w, s);
  return fmaf(w14, above[0], s);
}
// the window's weights for cell column hi: columns outside 0..15 (hi == 0: the first two, hi == 3: the last two) weigh
// nothing -- the positions they would read hold other columns' (finite) sums
__device__ __forceinline__ void window_weights(int hi, f4 &wa, f4 &wb) {
  wa = f4{hi >= 1 ? 0.125f : 0.0f, hi >= 1 ? 0.375f : 0.0f, 0.625f, 0.875f};
  wb = f4{0.875f, 0.625f, hi <= 2 ? 0.375f : 0.0f, hi <= 2 ? 0.125f : 0.0f};
}

template <typename TEX>
__device__ __forceinline__ void kp_descriptor(KpShared &S, const TEX &tex, const DescLaneConsts &C, float px,
                                              float py, float kp_scale, float orientation, int lane, float &out0,
                                              float &out1) {
  const float theta = 2.0f * 3.1415f / 360.0f * orientation;
  float sina, cosa;
  sm_sincosf(theta, &sina, &cosa);  // sift_math.h
  const float scale = 12.0f / 16.0f * kp_scale;
  const float ssina = scale * sina;
  const float scosa = scale * cosa;

  // ---- phase 1: samples (the taps are q times the texture value; C.gx1 carries the 1 / q: desc_lane_consts) ----
  unsigned int codes = 0;
#pragma unroll
  for (int step = 0; step < 4; ++step) {
    const int idx = lane + 64 * step;
    const int y = idx >> 4, tx = C.tx1;
    const float gy = C.gy1[step], gx = C.gx1;
    // cuSIFT_D.cu:216-229, x and y as a pair: xpos = (px + (tx - 7.5) scosa) - (y - 7.5) ssina, ypos = (py + (tx - 7.5)
    // ssina) + (y - 7.5) scosa; the taps at pos +- (cosa, sina) and pos +- (-sina, cosa).  a - b == a + (-b) bit for bit.
    const f2 pos = (f2{px, py} + (tx - 7.5f) * f2{scosa, ssina}) + (y - 7.5f) * f2{-ssina, scosa};
    const f2 u = f2{cosa, sina}, v = f2{-sina, cosa};
    const float dx = tex.desc(pos + u) - tex.desc(pos - u);
    const float dy = tex.desc(pos + v) - tex.desc(pos - v);
    const float grad = gy * gx * desc_sqrtf(dx * dx + dy * dy);
    // cuSIFT_D.cu:231-236: angf = 4/pi atan2 + 4 in [0, 8.0001] for every finite gradient (v_cvt_i32_f32 turns a NaN
    // into 0); angi = (int)angf, fraction angf - angi.  The unsigned min is for memory safety only.  code = angi, or 9
    // for angi == 8: slot `code` receives the (1 - fraction) share, slot code + 1 the fraction's.
    const float angraw = desc_angle_bins(dy, dx);
    const int angc = (int)angraw;
    const float angf = angraw - (float)angc;
    unsigned int code = min((unsigned int)angc, 8u);
    code += code >> 3;
    codes |= code;
    const int slot = desc_slot(y, tx);
    S.grad()[slot] = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned int, grad) & ~15u) | code);
    S.angfrac()[slot] = angf;
  }
  const bool spill = __ballot((codes & 8u) != 0u) != 0ull;  // wave-uniform: some sample has angle index 8
  wave_sync();
  STAMP(S, 7);
  // the patch is dead from here on: its storage becomes the histogram rows

  // ---- phase 2a: vertical pass ----
  {
    const int x = lane & 15, vi = lane >> 4;
    float *mine = S.cellhist() + hist_pos(x, vi);
#pragma unroll
    for (int b = 0; b < kHistSlots; ++b) mine[b * 64] = 0.0f;
    // the row weights, compile-time constants per visit: wy(vi, 4 vi - 2 + r), r = 0..7
    constexpr float kW[8] = {0.125f, 0.375f, 0.625f, 0.875f, 0.875f, 0.625f, 0.375f, 0.125f};
#pragma unroll
    for (int half = 0; half < 2; ++half) {  // four samples are read first, then accumulated (the compiler cannot move a
      float sg[4], sf[4];                   // sample read above the previous visit's histogram write)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int y = clampi(4 * vi - 2 + 4 * half + c, 0, 15);  // rows outside the grid: a valid slot, the add is skipped
        const int slot = desc_slot(y, x);
        sg[c] = S.grad()[slot];
        sf[c] = S.angfrac()[slot];
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int r = 4 * half + c;
        const bool live = (r >= 2 || vi >= 1) && (r <= 5 || vi <= 2);  // row 4 vi - 2 + r in 0..15
        // (the visit as an exec-masked `if`: with weight 0 instead -- one basic block, every lane making all eight
        // read-modify-writes -- the launch is 2 % SLOWER, round 4: the LDS pipe's time is what the skipped lanes save)
        if (live) {
          const unsigned int code = __builtin_bit_cast(unsigned int, sg[c]) & 15u;
          float *p = mine + code * 64;
          const float m = kW[r] * sg[c];
          const float v2 = sf[c] * m;
          const float v1 = m - v2;
          const float h1 = p[0], h2 = p[64];
          p[0] = h1 + v1;
          p[64] = h2 + v2;
        }
      }
    }
    // bin 0 also received shares as "bin 8" (the upper neighbour of bin 7) and, for angle index 8, in slot 10
    float fold = mine[8 * 64];
    if (spill) fold += mine[10 * 64];
    mine[0] += fold;
  }
  wave_sync();
  STAMP(S, 8);

  // ---- phase 2b: horizontal pass ----
  float b0, b1;
  {
    const int bq = lane >> 4, vi = (lane >> 2) & 3, hi = lane & 3;  // desc_elem(lane) = 8 (4 vi + hi) + bq
    f4 wa, wb;
    window_weights(hi, wa, wb);
    const float w14 = (hi == 0 && vi >= 1) ? 0.125f : 0.0f;
    const float *row = S.cellhist() + bq * 64 + 16 * vi;
    const float *above = S.cellhist() + bq * 64 + 16 * (vi >= 1 ? vi - 1 : 0);  // position 0 = column 14
    b0 = window_sum(row, hi, wa, wb, above, w14);
    b1 = window_sum(row + 4 * 64, hi, wa, wb, above + 4 * 64, w14);
    if (spill && bq == 0 && (vi | hi) != 0) {
      // bin 0 of this cell also receives the slot-9 sums of the linear cell before it (which for hi == 0 is the last
      // cell of the row above, with ITS column-14 term from the row above that)
      const int pc = 4 * vi + hi - 1, pv = pc >> 2, ph = pc & 3;
      f4 pa, pb;
      window_weights(ph, pa, pb);
      const float p14 = (ph == 0 && pv >= 1) ? 0.125f : 0.0f;
      b0 += window_sum(S.cellhist() + 9 * 64 + 16 * pv, ph, pa, pb, S.cellhist() + 9 * 64 + 16 * (pv >= 1 ? pv - 1 : 0), p14);
    }
  }

  STAMP(S, 9);
  // ---- phase 3: normalisation ----
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const float tsum = wave_sum64(b0 * b0 + b1 * b1);
    const float r = __builtin_amdgcn_rsqf(tsum);
    b0 = b0 * r;
    b1 = b1 * r;
    if (pass == 0) {
      if (b0 > 0.2f) b0 = 0.2f;
      if (b1 > 0.2f) b1 = 0.2f;
    }
  }
  out0 = b0;
  out1 = b1;
  STAMP(S, 10);
}

// ------------------------------------------------------------------------------------------------
// RootSIFT of one descriptor held in LDS (v[0..127]): ConvertSiftToRootSift_D, cuSIFT_D.cu:299-317 -- sequential
// L1 sum in index order, then sqrtf(max(0.0, x) / sum) (the max promotes to double there).  Every lane forms the
// same sum from broadcast reads.  Used by the stand-alone kernel and, with cusift_params.root_sift, as the
// descriptor epilogue (the fusion the reference leaves as a TODO, cuSIFT.cu:376-379) -- same bits either way.
// The lane returns elements e0 and e1.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rootsift_lanes(const float *v, int e0, int e1, float &o0, float &o1) {
  float sum = 0.0f;
#pragma unroll 16
  for (int i = 0; i < 128; ++i) sum += v[i];
  const float x0 = v[e0], x1 = v[e1];
  const double m0 = x0 > 0.0 ? (double)x0 : 0.0;
  const double m1 = x1 > 0.0 ? (double)x1 : 0.0;
  o0 = sqrtf((float)(m0 / sum));
  o1 = sqrtf((float)(m1 / sum));
}

// patch that covers every tap within `reach` of (px, py); false if it does not fit kDescPatch^2
__device__ __forceinline__ bool patch_for_reach(float px, float py, float reach, PatchGeom &g, int &pw, int &ph) {
  g.stride = kDescPatch;
  g.x0 = (int)floorf(px - reach - 0.5f) - 1;
  g.y0 = (int)floorf(py - reach - 0.5f) - 1;
  pw = (int)floorf(px + reach - 0.5f) + 2 - g.x0 + 1;
  ph = (int)floorf(py + reach - 0.5f) + 2 - g.y0 + 1;
  return (reach < 0.5f * kDescPatch) && (fabsf(px) < 1e5f) && (fabsf(py) < 1e5f) && pw <= kDescPatch &&
         ph <= kDescPatch;
}

// ------------------------------------------------------------------------------------------------
// ComputeOrientations (stage entry point; persistent grid over [first, min(count,max_pts)) of each image)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) orientations_kernel(const float *__restrict__ img, int w, int h, int pitch,
                                                         long img_stride, cusift_point *__restrict__ points,
                                                         int max_pts, const unsigned int *__restrict__ first,
                                                         const unsigned int *__restrict__ counters, float q,
                                                         float inv_q, RowWindow rw) {
  __shared__ OriShared S;
  const int tx = threadIdx.x;
  img += (long)blockIdx.y * img_stride;
  points += (long)blockIdx.y * max_pts;
  const unsigned int fst = first ? first[blockIdx.y] : 0u;
  const unsigned int cnt = counters[blockIdx.y];
  const unsigned int last = cnt < (unsigned int)max_pts ? cnt : (unsigned int)max_pts;

  for (unsigned int bx = fst + blockIdx.x; bx < last; bx += gridDim.x) {
    cusift_point *pt = points + bx;
    const float scale = uniform(pt->scale);
    const float kx = uniform(pt->coords2D[0]), ky = uniform(pt->coords2D[1]);
    // every tap lies in [k-6, k+6]; its 2x2 footprint starts at floor(k-6.5) .. floor(k+5.5): a 16x16 patch
    const bool use_patch = (fabsf(kx) < 1e5f) && (fabsf(ky) < 1e5f);
    float ori;
    if (use_patch) {
      const PatchGeom g{(int)floorf(kx - 6.5f) - 1, (int)floorf(ky - 6.5f) - 1, 16};
      stage_patch(img, w, h, pitch, rw, S.patch, g, 16, 16, tx);
      if (q > 0.0f)
        ori = kp_orientation(S, PatchSampler<16, true>{S.patch, g.x0, g.y0, q, inv_q}, kx, ky, scale, tx);
      else
        ori = kp_orientation(S, PatchSampler<16, false>{S.patch, g.x0, g.y0, q, inv_q}, kx, ky, scale, tx);
    } else {
      ori = kp_orientation(S, GlobalSampler{img, w, h, pitch, rw, q, inv_q}, kx, ky, scale, tx);
    }
    if (tx == 0) pt->orientation = ori;
    wave_sync();
  }
}

// RootSIFT epilogue (wave-uniform flag) + the record stores of ExtractSiftDescriptors_D, cuSIFT_D.cu:288-296
__device__ __forceinline__ void finish_descriptor(KpShared &S, cusift_point *pt, float b0, float b1, float px,
                                                  float py, float kscale, float sub, int lane, int root_sift) {
  const int e0 = desc_elem(lane);  // kp_descriptor's element of this lane (and e0 + 4)
  if (root_sift) {
    float *v = S.fin();
    v[e0] = b0;
    v[e0 + 4] = b1;
    wave_sync();
    rootsift_lanes(v, e0, e0 + 4, b0, b1);
  }
  {
    pt->data[e0] = b0;
    pt->data[e0 + 4] = b1;
  }
  if (lane == 0) {
    pt->coords2D[0] = px * sub;
    pt->coords2D[1] = py * sub;
    pt->scale = kscale * sub;
  }
}

// ------------------------------------------------------------------------------------------------
// ExtractSiftDescriptors (stage entry point)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) descriptors_kernel(const float *__restrict__ img, int w, int h, int pitch,
                                                        long img_stride, cusift_point *__restrict__ points,
                                                        int max_pts, const unsigned int *__restrict__ first,
                                                        const unsigned int *__restrict__ counters, float subsampling,
                                                        float q, float inv_q, RowWindow rw, int root_sift,
                                                        unsigned int *__restrict__ flags) {
  __shared__ KpShared S;
  const int lane = threadIdx.x;
  img += (long)blockIdx.y * img_stride;
  points += (long)blockIdx.y * max_pts;
  const unsigned int fst = first ? first[blockIdx.y] : 0u;
  const unsigned int cnt = counters[blockIdx.y];
  const unsigned int last = cnt < (unsigned int)max_pts ? cnt : (unsigned int)max_pts;
  const DescLaneConsts C = desc_lane_consts(lane, q, inv_q);

  for (unsigned int bx = fst + blockIdx.x; bx < last; bx += gridDim.x) {
    cusift_point *pt = points + bx;
    const float px = uniform(pt->coords2D[0]), py = uniform(pt->coords2D[1]);
    const float kscale = uniform(pt->scale), ori = uniform(pt->orientation);
    if (flags) {
      // Band of a strip-tiled image: count the keypoints whose sampling footprint (orientation window +-6 px,
      // descriptor grid +-7.5*spacing*(|cos|+|sin|), +-1 px taps, bilinear 2x2) leaves the band where the band does
      // not end at the image border -- there the clamp would replace the neighbour's rows (cusift_describe_band).
      float sn, cs;
      sm_sincosf(2.0f * 3.1415f / 360.0f * ori, &sn, &cs);
      const float r = fmaxf(7.5f * (12.0f / 16.0f * kscale) * (fabsf(sn) + fabsf(cs)), 6.0f) + 2.5f;
      const bool cut = (rw.row0 > 0 && !(py - r >= (float)rw.row0)) ||
                       (rw.row0 + h < rw.hg && !(py + r <= (float)(rw.row0 + h - 1)));
      if (cut && lane == 0) atomicAdd(flags, 1u);
    }
    // every tap is within `reach` of the keypoint: 7.5*spacing*(|cos|+|sin|) for the grid + 1 for the tap
    const float reach = 7.5f * (12.0f / 16.0f * kscale) * 1.41422f + 1.0f + 0.01f;  // |cos| + |sin| <= sqrt 2
    PatchGeom g;
    int pw, ph;
    const bool use_patch = patch_for_reach(px, py, reach, g, pw, ph);
    float b0, b1;
    if (use_patch) {
      stage_patch(img, w, h, pitch, rw, S.patch, g, pw, ph, lane);
      wave_sync();
      if (q > 0.0f)
        kp_descriptor(S, PatchSampler<kDescPatch, true>{S.patch, g.x0, g.y0, q, inv_q}, C, px, py, kscale, ori, lane,
                      b0, b1);
      else
        kp_descriptor(S, PatchSampler<kDescPatch, false>{S.patch, g.x0, g.y0, q, inv_q}, C, px, py, kscale, ori, lane,
                      b0, b1);
    } else {
      kp_descriptor(S, GlobalSampler{img, w, h, pitch, rw, q, inv_q}, C, px, py, kscale, ori, lane, b0, b1);
    }
    finish_descriptor(S, pt, b0, b1, px, py, kscale, subsampling, lane, root_sift);
    wave_sync();
  }
}

// s_prefix[1 .. n] hold per-image counts: turns them into running sums (s_prefix[0] = 0) with a wave scan, 64 images per
// pass.  (Lane 0 alone, one LDS read-modify-write after the other, took ~8,000 cycles for 64 images -- once per wave, 1 %
// of describe_all_kernel's launch in the phase stamps.)  The caller syncs before and after.
__device__ __forceinline__ void running_sums_in_place(unsigned int *s_prefix, int n_images, int lane) {
  unsigned int carry = 0;
  for (int base = 0; base < n_images; base += 64) {  // wave-uniform
    const int i = base + lane;
    unsigned int x = i < n_images ? s_prefix[i + 1] : 0u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned int y = (unsigned int)__shfl_up((int)x, d);
      if (lane >= d) x += y;
    }
    if (i < n_images) s_prefix[i + 1] = carry + x;
    carry += (unsigned int)__builtin_amdgcn_readlane((int)x, 63);
  }
  if (lane == 0) s_prefix[0] = 0;
}

// ------------------------------------------------------------------------------------------------
// Orientation + descriptor of ALL keypoints of a batch in one launch (driver path).  The reference runs the
// two kernels once per octave (cuSIFT.cu:253-258); here detection runs for every octave first and this kernel
// then walks the flattened list of (image, keypoint) pairs -- the octave comes from the record's `subsampling`
// -- so images with more keypoints do not leave the rest of the grid idle, one patch load serves both stages,
// and 2 x octaves launches become one.  Same device functions as the two stage kernels => same results.
// ------------------------------------------------------------------------------------------------
// One keypoint, both stages, handing the orientation back (describe_bands_kernel's footprint check wants it);
// describe_all_kernel makes the same calls itself, with its work-queue steps between them.
template <typename TEX>
__device__ __forceinline__ float describe_keypoint_ori(KpShared &S, const TEX &tex, const DescLaneConsts &C,
                                                       cusift_point *pt, float px, float py, float kscale, float sub,
                                                       int lane, int root_sift) {
  const float ori = kp_orientation(S, tex, px, py, kscale, lane);
  float b0, b1;
  kp_descriptor(S, tex, C, px, py, kscale, ori, lane, b0, b1);
  if (lane == 0) pt->orientation = ori;
  finish_descriptor(S, pt, b0, b1, px, py, kscale, sub, lane, root_sift);
  return ori;
}

// 4 waves per SIMD (<= 128 VGPRs) is what this kernel needs: its LDS read-modify-write chains and dependent taps are
// latency that only other waves hide (forced to 3 / 2 waves the launch takes 1.25x / 1.8x as long)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) describe_all_kernel(OctaveTable T, cusift_point *__restrict__ points, int max_pts,
                                                         unsigned int *counters, int n_images,
                                                         float q, float inv_q, int root_sift,
                                                         unsigned int *__restrict__ queue,
                                                         SegmentTable G,
                                                         const unsigned int *__restrict__ seg_end) {
  // G.n_seg == 0: every keypoint was appended to `points` directly.  Otherwise the batch's keypoints wait in segments
  // (sift_types.h) -- octave 0's when cusift_extract_batch ran that octave's detection beside the coarser ones, every
  // octave's when the octaves were searched by one launch -- and are moved into place here, coarsest octave first:
  // keypoint k of image i belongs to the first segment r with k < seg_end[i * n_seg + r].  seg_end == NULL (with every
  // segment staged): no join_counts_kernel ran -- a small call saves that dispatch -- and this kernel forms the running
  // sums itself from the segments' counters, workgroup 0 also leaving the images' totals in `counters`.
  __shared__ KpShared S;
  __shared__ unsigned int s_prefix[kMaxFlatImages + 1];
  const int lane = threadIdx.x;
  // exclusive prefix sums of the per-image counts (every block computes them: n_images <= kMaxFlatImages)
  const bool self_join = G.n_seg > 0 && seg_end == nullptr;  // wave-uniform
  for (int i = lane; i < n_images; i += 64) {
    unsigned int c;
    if (self_join) {
      c = 0;
      for (int r = 0; r < G.n_seg; ++r) c += G.count[r][i];
      if (blockIdx.x == 0) counters[i] = c;  // what one stream would have counted (nobody reads it in this launch)
    } else {
      c = counters[i];
    }
    s_prefix[i + 1] = c < (unsigned int)max_pts ? c : (unsigned int)max_pts;
  }
  wave_sync();
  running_sums_in_place(s_prefix, n_images, lane);
  wave_sync();
  const unsigned int total = s_prefix[n_images];
  const DescLaneConsts C = desc_lane_consts(lane, q, inv_q);
  const int exp0 = (__float_as_int(T.sub[0]) >> 23) & 0xff;

  // Items are handed out dynamically: the first one is the workgroup's index, every further one comes from a global
  // cursor (*queue, zeroed before the launch) that is advanced when the previous item STARTS, so the atomic's
  // round trip hides behind that item's work.  Keypoints differ in cost (patch size, the global-memory path), and
  // with a static interleaving the launch waited for the unluckiest wave.
  // One cursor would serialise ~170 k same-address atomics (~12 ns each: measured 2.4x slower than no queue at all),
  // so the items are dealt into kQueueShards interleaved sub-sequences, each with a cursor on a cache line of its own.
  // The record of the NEXT item (position, scale, subsampling: four floats behind one memory round trip) is fetched
  // while the current one is described -- its index arrives with the cursor's atomic, long before the current
  // keypoint is finished -- so a keypoint starts with its patch loads instead of with a wait for its own record.
  int im = 0;  // image of the current item; a workgroup sees increasing items, so the cursor only moves forward
  const unsigned int shard = blockIdx.x & (kQueueShards - 1);
  unsigned int *cursor = queue + shard * 32;                 // 128-byte stride
  const unsigned int per_shard = gridDim.x / kQueueShards;   // workgroups per shard (host: gridDim.x % shards == 0)
  unsigned int g = blockIdx.x;                               // = shard + kQueueShards * (blockIdx.x / kQueueShards)
  // item < total: moves `im` forward to the item's image and returns the record the keypoint ends up in (locate_begin);
  // where its head waits -- the same record, or a staged one -- comes from locate_end.  Two steps because the second
  // needs memory: the image's segment ends.  locate_begin ISSUES their load (one vector load: lane r < n_seg holds
  // ends[r]) and locate_end reads it a whole orientation stage later, so nobody waits for it.  (Until round 4 the segment
  // was found by a loop of DEPENDENT scalar loads -- five round trips for a keypoint of octave 0, the last of five
  // segments: 17 % of a wave's time in the phase stamps, tools/describe_stamps.py.)
  unsigned int loc_idx = 0, loc_ends = 0xffffffffu;
  auto locate_begin = [&](unsigned int item) {
    while (__builtin_amdgcn_readfirstlane(s_prefix[im + 1]) <= item) ++im;  // ends: item < total = s_prefix[n_images]
    loc_idx = item - __builtin_amdgcn_readfirstlane(s_prefix[im]);
    if (G.n_seg > 0 && !self_join) {
      const unsigned int *ends = seg_end + im * G.n_seg;  // wave-uniform
      loc_ends = lane < G.n_seg ? ends[lane] : 0xffffffffu;
    }
    return points + (long)im * max_pts + loc_idx;
  };
  auto locate_end = [&](cusift_point *dst) {
    const cusift_point *from = dst;
    if (G.n_seg > 0) {
      const unsigned int idx = loc_idx;
      int r = 0;
      unsigned int first = 0;  // keypoints of this image in the segments before r
      if (self_join) {
        for (;; ++r) {  // ends: idx < this image's clamped total
          const unsigned int c = G.count[r][im], room = (unsigned int)max_pts - first;
          const unsigned int kept = c < room ? c : room;
          if (idx < first + kept) break;
          first += kept;
        }
      } else {
        // ends[] is non-decreasing and idx < ends[n_seg - 1] (this image's total): the segment is the number of ends <= idx
        r = __builtin_popcountll(__ballot(idx >= loc_ends));
        first = r > 0 ? (unsigned int)__builtin_amdgcn_readlane((int)loc_ends, r - 1) : 0u;
      }
      const char *base = G.base[r];
      if (base) from = reinterpret_cast<const cusift_point *>(base + ((size_t)im * max_pts + (idx - first)) * kStagedRecBytes);
    }
    return from;
  };
  // lane l < 16 holds float l of the record's head (coords2D, scale, ..., subsampling = float 12): one register
  constexpr int kSubIndex = (int)(offsetof(cusift_point, subsampling) / sizeof(float));
  static_assert(kSubIndex < 16 && offsetof(cusift_point, scale) == 8, "record head layout");
  cusift_point *pt = nullptr;
  const cusift_point *src = nullptr;
  float rec = 0.0f;
  auto fetch_head = [&](const cusift_point *p) { return reinterpret_cast<const float *>(p)[lane & 15]; };
  auto head = [&](int i) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rec), i)); };
  constexpr int kSharpIndex = (int)(offsetof(cusift_point, sharpness) / sizeof(float));
  constexpr int kEdgeIndex = (int)(offsetof(cusift_point, edgeness) / sizeof(float));
  if (g < total) {
    pt = locate_begin(g);
    src = locate_end(pt);
    rec = fetch_head(src);
    // (waited for here so that the loop's head holds no wait of the compiler's: see the wait before the stores below)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(rec) : : "memory");
  }
#ifdef CUSIFT_STAMPS
  if (lane < 16) S.stamps[lane] = lane == 15 ? (unsigned int)__builtin_amdgcn_s_memtime() : 0u;
  wave_sync();
#endif
  while (g < total) {
    STAMP(S, 12);  // loop back edge (+ the kernel's prologue, once)
    // a staged keypoint: the three fields of the head that detection wrote and nothing below rewrites
    if (src != pt && (lane == kSharpIndex || lane == kEdgeIndex || lane == kSubIndex))
      reinterpret_cast<float *>(pt)[lane] = rec;
    const float px = head(0), py = head(1), kscale = head(2), sub = head(kSubIndex);
    const int im_cur = im;
    // The next item's index: one atomic by lane 0, whose round trip (~1-2 us to the L2 and back) is meant to hide behind
    // the patch loads.  Written as atomicAdd() it did not: the compiler forms the result's arithmetic at once and put
    // `s_waitcnt vmcnt(0)` directly behind the atomic -- every keypoint began by waiting for it (seen in the ISA, round
    // 4).  As inline assembly the returning atomic is just an instruction with a vector result; nothing reads `raw`
    // until the wait below, behind the patch loads.  (Issued AFTER the record's fields have been read: the compiler's own
    // `vmcnt(0)` for that load -- it does not know this asm is a memory operation -- would wait for the atomic as well.)
    unsigned int raw = 0;
    {
      unsigned long long saved_exec;
      asm volatile("s_mov_b64 %[sv], exec\n\t"
                   "s_mov_b64 exec, 1\n\t"
                   "global_atomic_add %[r], %[off], %[one], %[base] sc0\n\t"
                   "s_mov_b64 exec, %[sv]"
                   : [r] "+v"(raw), [sv] "=&s"(saved_exec)
                   : [off] "v"(0u), [one] "v"(1u), [base] "s"(cursor)
                   : "memory");
    }
    int o = ((__float_as_int(sub) >> 23) & 0xff) - exp0;  // subsampling = sub0 * 2^octave
    o = clampi(o, 0, T.n_oct - 1);
    const float *img = T.base[o] + (long)im_cur * T.stride[o];
    const int w = T.w[o], h = T.h[o], pitch = T.pitch[o];
    const RowWindow rw{0, h};
    // one patch for both stages: orientation taps reach 6 px, descriptor taps 7.5*spacing*sqrt(2)+1 at most
    const float reach = fmaxf(7.5f * (12.0f / 16.0f * kscale) * 1.41422f + 1.0f + 0.01f, 6.0f);
    PatchGeom pg;
    int pw, ph;
    const bool use_patch = patch_for_reach(px, py, reach, pg, pw, ph);
    // The patch's loads are issued and NOT waited for: what the orientation stage derives from the keypoint's fields
    // alone (Gaussian table, lattice test and weights: ~70 instructions) runs while they are in flight.
    OriPrep prep;
    if (use_patch) {
      stage_patch<false>(img, w, h, pitch, rw, S.patch, pg, pw, ph, lane);
      STAMP(S, 0);
      if (q > 0.0f)
        prep = kp_orientation_prep(S, PatchSampler<kDescPatch, true>{S.patch, pg.x0, pg.y0, q, inv_q}, px, py, kscale, lane);
      else
        prep = kp_orientation_prep(S, PatchSampler<kDescPatch, false>{S.patch, pg.x0, pg.y0, q, inv_q}, px, py, kscale, lane);
      STAMP(S, 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the patch is in LDS
    } else {
      prep = kp_orientation_prep(S, GlobalSampler{img, w, h, pitch, rw, q, inv_q}, px, py, kscale, lane);
    }
    wave_sync();
    STAMP(S, 2);
    // the patch loads have been waited for, so has the atomic issued before them: the next item's index is here.
    // (Tried on top of this and dropped, tools/exp_describe_stamps.sh: forming the next keypoint's geometry between the
    // stages -- the ~600 cycles it saves here come back, and more, in the descriptor stage; touching the next patch's
    // lines with two LDS-DMA loads so that they are in L2 by the time they are staged -- no gain.)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw) : : "memory");  // (already satisfied after stage_patch)
    g = __builtin_amdgcn_readfirstlane(shard + kQueueShards * (raw + per_shard));
    cusift_point *pt_next = nullptr;
    if (g < total) pt_next = locate_begin(g);  // issues the load of its image's segment ends
    STAMP(S, 3);
    float ori;
    if (use_patch) {
      if (q > 0.0f)
        ori = kp_orientation(S, PatchSampler<kDescPatch, true>{S.patch, pg.x0, pg.y0, q, inv_q}, prep, px, py, kscale, lane);
      else
        ori = kp_orientation(S, PatchSampler<kDescPatch, false>{S.patch, pg.x0, pg.y0, q, inv_q}, prep, px, py, kscale, lane);
    } else {
      ori = kp_orientation(S, GlobalSampler{img, w, h, pitch, rw, q, inv_q}, prep, px, py, kscale, lane);
    }
    // Between the stages: where the next item's head waits, and the load of that head -- its round trip hides behind
    // the descriptor stage, as the segment ends' did behind the orientation stage.
    const cusift_point *src_next = nullptr;
    float rec_next = 0.0f;
    if (g < total) {
      src_next = locate_end(pt_next);
      rec_next = fetch_head(src_next);
    }
    STAMP(S, 14);
    float b0, b1;
    if (use_patch) {
      if (q > 0.0f)
        kp_descriptor(S, PatchSampler<kDescPatch, true>{S.patch, pg.x0, pg.y0, q, inv_q}, C, px, py, kscale, ori, lane, b0, b1);
      else
        kp_descriptor(S, PatchSampler<kDescPatch, false>{S.patch, pg.x0, pg.y0, q, inv_q}, C, px, py, kscale, ori, lane, b0, b1);
    } else {
      kp_descriptor(S, GlobalSampler{img, w, h, pitch, rw, q, inv_q}, C, px, py, kscale, ori, lane, b0, b1);
    }
    // The next head has arrived long ago; waiting for it HERE, before this keypoint's stores are issued, is free -- at
    // the top of the loop the same wait also waited for those stores' acknowledgements (the compiler's vmcnt(0) for the
    // head's first use: ~1-2 us of write round trip exposed per keypoint).
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(rec_next) : : "memory");
    if (lane == 0) pt->orientation = ori;
    finish_descriptor(S, pt, b0, b1, px, py, kscale, sub, lane, root_sift);
    wave_sync();
    STAMP(S, 11);
#ifdef CUSIFT_STAMPS
    if (lane == 0) S.stamps[13] += 1u;
#endif
    pt = pt_next;
    src = src_next;
    rec = rec_next;
  }
#ifdef CUSIFT_STAMPS
  wave_sync();
  if (lane < 15) atomicAdd(&g_describe_stamps[lane], (unsigned long long)S.stamps[lane]);
#endif
}

#ifdef CUSIFT_STAMPS
// the sums of all describe_all_kernel waves since the last call (counters 0..12: shader-clock cycles by phase, 13: keypoints)
extern "C" int cusift_stamps_read(unsigned long long *out16) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_describe_stamps), sizeof(g_describe_stamps)) != hipSuccess) return -1;
  unsigned long long zero[16] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_describe_stamps), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

// Joins the keypoint segments of a batch (sift_types.h: SegmentTable) before describe_all_kernel walks them: per image
// the running sum of the segments' counts in list order, clamped at max_pts -- so the coarser segments survive whole,
// as when one stream searches coarsest first (cuSIFT.cu:190-196) -- and the image's counter as one stream would have left
// it: the sum of what every detection counted (it keeps counting beyond max_pts, like the reference's).  Segment 0 may be
// the caller's own counter (keypoints appended in place).  Also clears the work cursors of describe_all_kernel.
__global__ void __launch_bounds__(256) join_counts_kernel(unsigned int *__restrict__ counters, SegmentTable G,
                                                          unsigned int *__restrict__ seg_end, int n_images, int max_pts,
                                                          unsigned int *__restrict__ queue, int clear_counts) {
  for (int i = threadIdx.x; i < kQueueShards * 32; i += 256) queue[i] = 0u;
  for (int i = threadIdx.x; i < n_images; i += 256) {
    unsigned int raw = 0, kept = 0;
    for (int r = 0; r < G.n_seg; ++r) {
      const unsigned int c = G.count[r][i];
      // clear_counts: the lists' counters are the context's own (never the caller's) and this is their one reader --
      // leaving them zero saves the NEXT extraction its memset dispatch (cusift_extract_batch keeps track)
      if (clear_counts) const_cast<unsigned int *>(G.count[r])[i] = 0u;
      raw += c;
      const unsigned int room = (unsigned int)max_pts - kept;
      kept += c < room ? c : room;
      seg_end[i * G.n_seg + r] = kept;
    }
    counters[i] = raw;
  }
}

// ------------------------------------------------------------------------------------------------
// Orientation + descriptor of the keypoints of ONE strip-tiled image whose octaves are bands (cusift_extract_bands): the
// band form of describe_all_kernel.  Segment 0 of G is the caller's list as it stands -- keypoints of coarser octaves,
// already described, left alone -- and segments 1.. are the bands' staging lists, coarsest first; a keypoint's head is
// read there, its record written at its place in list order.  A rank has a few thousand keypoints: a plain
// grid-stride walk, none of describe_all_kernel's queue and prefetch.  `flags`: the footprint check of
// descriptors_kernel (a keypoint whose samples leave the band where the band does not end at the image border).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) describe_bands_kernel(
    OctaveTable T, BandWindows BW, cusift_point *__restrict__ points, int max_pts, SegmentTable G,
    const unsigned int *__restrict__ seg_end, float q, float inv_q, int root_sift, unsigned int *__restrict__ flags) {
  __shared__ KpShared S;
  const int lane = threadIdx.x;
  const unsigned int total = seg_end[G.n_seg - 1];
  const DescLaneConsts C = desc_lane_consts(lane, q, inv_q);
  const int exp0 = (__float_as_int(T.sub[0]) >> 23) & 0xff;
  for (unsigned int k = seg_end[0] + blockIdx.x; k < total; k += gridDim.x) {  // wave-uniform
    int r = 1;
    unsigned int first = seg_end[0];
    while (k >= seg_end[r]) first = seg_end[r++];
    const float *src = reinterpret_cast<const float *>(G.base[r] + (size_t)(k - first) * kStagedRecBytes);
    cusift_point *pt = points + k;
    const float px = uniform(src[0]), py = uniform(src[1]), kscale = uniform(src[2]);
    const float sharp = uniform(src[offsetof(cusift_point, sharpness) / 4]);
    const float edge = uniform(src[offsetof(cusift_point, edgeness) / 4]);
    const float sub = uniform(src[offsetof(cusift_point, subsampling) / 4]);
    if (lane == 0) {
      pt->sharpness = sharp;
      pt->edgeness = edge;
      pt->subsampling = sub;
    }
    int o = ((__float_as_int(sub) >> 23) & 0xff) - exp0;  // subsampling = sub0 * 2^octave
    o = clampi(o, 0, T.n_oct - 1);
    const float *img = T.base[o];
    const int w = T.w[o], h = T.h[o], pitch = T.pitch[o];
    const RowWindow rw{BW.row0[o], BW.hg[o]};
    const float reach = fmaxf(7.5f * (12.0f / 16.0f * kscale) * 1.41422f + 1.0f + 0.01f, 6.0f);
    PatchGeom pg;
    int pw, ph;
    const bool use_patch = patch_for_reach(px, py, reach, pg, pw, ph);
    if (use_patch) stage_patch(img, w, h, pitch, rw, S.patch, pg, pw, ph, lane);
    wave_sync();
    float ori;
    if (use_patch) {
      if (q > 0.0f)
        ori = describe_keypoint_ori(S, PatchSampler<kDescPatch, true>{S.patch, pg.x0, pg.y0, q, inv_q}, C, pt, px, py, kscale,
                                sub, lane, root_sift);
      else
        ori = describe_keypoint_ori(S, PatchSampler<kDescPatch, false>{S.patch, pg.x0, pg.y0, q, inv_q}, C, pt, px, py, kscale,
                                sub, lane, root_sift);
    } else {
      ori = describe_keypoint_ori(S, GlobalSampler{img, w, h, pitch, rw, q, inv_q}, C, pt, px, py, kscale, sub, lane, root_sift);
    }
    if (flags) {  // as in descriptors_kernel
      float sn, cs;
      sm_sincosf(2.0f * 3.1415f / 360.0f * ori, &sn, &cs);
      const float rr = fmaxf(7.5f * (12.0f / 16.0f * kscale) * (fabsf(sn) + fabsf(cs)), 6.0f) + 2.5f;
      const bool cut = (rw.row0 > 0 && !(py - rr >= (float)rw.row0)) ||
                       (rw.row0 + h < rw.hg && !(py + rr <= (float)(rw.row0 + h - 1)));
      if (cut && lane == 0) atomicAdd(flags, 1u);
    }
    wave_sync();
  }
}

// ------------------------------------------------------------------------------------------------
// ConvertSiftToRootSift: reference cuSIFT_D.cu:299-317 (sequential L1 sum, sqrt(max(0,v)/sum)).
// One wave per point; the 128-term sum is kept sequential to match the reference order (rootsift_lanes).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) rootsift_kernel(cusift_point *__restrict__ points, int num_pts) {
  __shared__ float v[128];
  const int lane = threadIdx.x;
  for (int p = blockIdx.x; p < num_pts; p += gridDim.x) {
    cusift_point *pt = points + p;
    v[lane] = pt->data[lane];
    v[lane + 64] = pt->data[lane + 64];
    __syncthreads();
    float o0, o1;
    rootsift_lanes(v, lane, lane + 64, o0, o1);
    pt->data[lane] = o0;
    pt->data[lane + 64] = o1;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Packing of a batch's SiftData for the all-gatherv (new: the reference is single-image, single-GPU).
// [n][max_pts] records + per-image counters -> the valid records back to back in image order, plus the
// exclusive prefix sums of the valid counts (offsets[n] = total).  One wave per record, 147 dwords each.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) pack_points_kernel(const cusift_point *__restrict__ points,
                                                        const unsigned int *__restrict__ counters, int n_images,
                                                        int max_pts, cusift_point *__restrict__ packed,
                                                        unsigned int capacity, unsigned int *__restrict__ offsets) {
  __shared__ unsigned int s_prefix[kMaxFlatImages + 1];
  const int lane = threadIdx.x;
  for (int i = lane; i < n_images; i += 64) {
    const unsigned int c = counters[i];
    s_prefix[i + 1] = c < (unsigned int)max_pts ? c : (unsigned int)max_pts;
  }
  wave_sync();
  running_sums_in_place(s_prefix, n_images, lane);
  wave_sync();
  if (blockIdx.x == 0 && offsets)
    for (int i = lane; i <= n_images; i += 64) offsets[i] = s_prefix[i];
  const unsigned int total = min(s_prefix[n_images], capacity);
  constexpr int kDwords = sizeof(cusift_point) / 4;  // 147
  for (unsigned int g = blockIdx.x; g < total; g += gridDim.x) {
    int lo = 0, hi_ = n_images;
    while (hi_ - lo > 1) {
      const int mid = (lo + hi_) >> 1;
      if (s_prefix[mid] <= g) lo = mid; else hi_ = mid;
    }
    const unsigned int *src = reinterpret_cast<const unsigned int *>(points + (long)lo * max_pts + (g - s_prefix[lo]));
    unsigned int *dst = reinterpret_cast<unsigned int *>(packed + g);
#pragma unroll
    for (int k = 0; k < (kDwords + 63) / 64; ++k) {
      const int e = lane + 64 * k;
      if (e < kDwords) dst[e] = src[e];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Trimmed wire format (round 4): the 135 floats of a record that extraction writes -- its first six floats, subsampling
// (float 12) and data[128] (floats 16..143) -- as a 540-byte cusift_trimmed_point: EXACT, 8 % fewer bytes over a link
// than the 588-byte record, whose other 12 floats are whatever the caller's buffer held (cuSIFT.cu:24,29).
// Same walk as pack_points_kernel; one wave per record.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) pack_points_trimmed_kernel(const cusift_point *__restrict__ points,
                                                                const unsigned int *__restrict__ counters,
                                                                int n_images, int max_pts,
                                                                cusift_trimmed_point *__restrict__ packed,
                                                                unsigned int capacity,
                                                                unsigned int *__restrict__ offsets) {
  __shared__ unsigned int s_prefix[kMaxFlatImages + 1];
  const int lane = threadIdx.x;
  for (int i = lane; i < n_images; i += 64) {
    const unsigned int c = counters[i];
    s_prefix[i + 1] = c < (unsigned int)max_pts ? c : (unsigned int)max_pts;
  }
  wave_sync();
  running_sums_in_place(s_prefix, n_images, lane);
  wave_sync();
  if (blockIdx.x == 0 && offsets)
    for (int i = lane; i <= n_images; i += 64) offsets[i] = s_prefix[i];
  const unsigned int total = min(s_prefix[n_images], capacity);
  constexpr int kData = (int)(offsetof(cusift_point, data) / 4), kSub = (int)(offsetof(cusift_point, subsampling) / 4);
  for (unsigned int g = blockIdx.x; g < total; g += gridDim.x) {
    int lo = 0, hi_ = n_images;
    while (hi_ - lo > 1) {
      const int mid = (lo + hi_) >> 1;
      if (s_prefix[mid] <= g) lo = mid; else hi_ = mid;
    }
    const unsigned int *src = reinterpret_cast<const unsigned int *>(points + (long)lo * max_pts + (g - s_prefix[lo]));
    unsigned int *dst = reinterpret_cast<unsigned int *>(packed + g);
    // trimmed dword e <- record dword: 0..5 as they are, 6 <- subsampling, 7 + k <- data[k]
    dst[7 + lane] = src[kData + lane];
    dst[7 + 64 + lane] = src[kData + 64 + lane];
    if (lane < 7) dst[lane] = src[lane < 6 ? lane : kSub];
  }
}

// trimmed records -> full records on the device (the 12 floats extraction never writes are zeroed); one wave per record
__global__ void __launch_bounds__(64) expand_trimmed_kernel(const cusift_trimmed_point *__restrict__ trimmed, size_t n,
                                                           cusift_point *__restrict__ points) {
  const int lane = threadIdx.x;
  constexpr int kData = (int)(offsetof(cusift_point, data) / 4), kSub = (int)(offsetof(cusift_point, subsampling) / 4);
  constexpr int kDwords = sizeof(cusift_point) / 4;
  for (size_t g = blockIdx.x; g < n; g += gridDim.x) {
    const unsigned int *src = reinterpret_cast<const unsigned int *>(trimmed + g);
    unsigned int *dst = reinterpret_cast<unsigned int *>(points + g);
    dst[kData + lane] = src[7 + lane];
    dst[kData + 64 + lane] = src[7 + 64 + lane];
    if (lane < kData) dst[lane] = lane < 6 ? src[lane] : (lane == kSub ? src[6] : 0u);
    if (lane < kDwords - kData - 128) dst[kData + 128 + lane] = 0u;  // coords3D
  }
}

// ------------------------------------------------------------------------------------------------
// Compact wire format of extracted SiftData (new: the reference copies whole 588-byte records, cuSIFT.cu:52-59).
// Same walk as pack_points_kernel, but a record leaves as a cusift_compact_point (160 B): the seven fields extraction
// writes, exactly, and the descriptor as 128 bytes with one step per record -- q[i] = min(255, floor(data[i] / step +
// 0.5)), step = max(data) / 255 (IEEE operations, so a host restatement gives the same bytes).  A descriptor that is
// not finite and positive somewhere (flat patch: NaN) travels as step = its maximum as computed (NaN or 0), q = 0.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) pack_points_compact_kernel(const cusift_point *__restrict__ points,
                                                                const unsigned int *__restrict__ counters,
                                                                int n_images, int max_pts,
                                                                cusift_compact_point *__restrict__ packed,
                                                                unsigned int capacity,
                                                                unsigned int *__restrict__ offsets) {
  __shared__ unsigned int s_prefix[kMaxFlatImages + 1];
  const int lane = threadIdx.x;
  for (int i = lane; i < n_images; i += 64) {
    const unsigned int c = counters[i];
    s_prefix[i + 1] = c < (unsigned int)max_pts ? c : (unsigned int)max_pts;
  }
  wave_sync();
  running_sums_in_place(s_prefix, n_images, lane);
  wave_sync();
  if (blockIdx.x == 0 && offsets)
    for (int i = lane; i <= n_images; i += 64) offsets[i] = s_prefix[i];
  const unsigned int total = min(s_prefix[n_images], capacity);
  // TWO records per wave, one per half-wave: lane l of a half holds elements 4l .. 4l+3 of the descriptor (one 16-byte
  // load), so its four bytes are one 32-bit store and no value has to change lanes -- only the maximum does (DPP).
  const int half = lane >> 5, l = lane & 31;
  for (unsigned int g0 = 2u * blockIdx.x; g0 < total; g0 += 2u * gridDim.x) {
    const unsigned int g = g0 + (unsigned int)half;
    const bool live = g < total;
    int lo = 0, hi_ = n_images;
    const unsigned int gs = live ? g : g0;  // the idle half of the last pair repeats its neighbour's search
    while (hi_ - lo > 1) {
      const int mid = (lo + hi_) >> 1;
      if (s_prefix[mid] <= gs) lo = mid; else hi_ = mid;
    }
    const cusift_point *src = points + (long)lo * max_pts + (gs - s_prefix[lo]);
    cusift_compact_point *dst = packed + gs;
    const float *dp = src->data + 4 * l;
    const float d0 = dp[0], d1 = dp[1], d2 = dp[2], d3 = dp[3];
    // maximum over the record's 128 elements; a NaN anywhere makes it NaN (fmaxf would drop it)
    const bool has_nan = (d0 != d0) || (d1 != d1) || (d2 != d2) || (d3 != d3);
    const unsigned long long nan_lanes = __ballot(has_nan);
    const bool rec_nan = ((nan_lanes >> (32 * half)) & 0xffffffffull) != 0;
    float m = max_over_32(fmaxf(fmaxf(d0, d1), fmaxf(d2, d3)));  // every lane of a half gets its half's maximum
    if (rec_nan) m = __builtin_nanf("");
    const bool ok = m > 0.0f && m < __builtin_inff();
    const float step = ok ? m / 255.0f : m;
    unsigned int word = 0;
    if (ok) {
      const unsigned int q0 = (unsigned int)fminf(fmaxf(floorf(d0 / step + 0.5f), 0.0f), 255.0f);
      const unsigned int q1 = (unsigned int)fminf(fmaxf(floorf(d1 / step + 0.5f), 0.0f), 255.0f);
      const unsigned int q2 = (unsigned int)fminf(fmaxf(floorf(d2 / step + 0.5f), 0.0f), 255.0f);
      const unsigned int q3 = (unsigned int)fminf(fmaxf(floorf(d3 / step + 0.5f), 0.0f), 255.0f);
      word = q0 | (q1 << 8) | (q2 << 16) | (q3 << 24);
    }
    if (live) {
      reinterpret_cast<unsigned int *>(dst->q)[l] = word;
      if (l < 7) {  // coords2D[2], scale, sharpness, edgeness, orientation are the record's first six floats
        const float v = l < 6 ? reinterpret_cast<const float *>(src)[l] : src->subsampling;
        reinterpret_cast<float *>(dst)[l] = v;
      }
      if (l == 7) dst->desc_step = step;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The written-out transcendental functions of sift_math.h evaluated on the device, array form (cusift_math_eval):
// lets a test compare the device's results with the host's bit for bit.  op 0 expf(a), 1 exp2f(a), 2 atan2f(a, b),
// 3 sincosf(a) -> (out, out2); op 4: the DESCRIPTOR's angle coordinate desc_angle_bins(a = dy, b = dx), which is not
// sm_atan2f to the ulp (a test prices its distance and checks the one decision it carries).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) math_eval_kernel(int op, const float *__restrict__ a,
                                                       const float *__restrict__ b, float *__restrict__ out,
                                                       float *__restrict__ out2, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    if (op == 0) out[i] = sm_expf(a[i]);
    else if (op == 1) out[i] = sm_exp2f(a[i]);
    else if (op == 2) out[i] = sm_atan2f(a[i], b[i]);
    else if (op == 4) out[i] = desc_angle_bins(a[i], b[i]);
    else if (op == 5) {  // the orientation bin's shortcut and its `near` flag; out2 = the reference's formula
      bool near;
      const int bin = ori_bin_shortcut(a[i], b[i], near);
      out[i] = (float)(bin + (near ? 64 : 0));
      int ref = (int)(16.0f * sm_atan2f(a[i], b[i]) / 3.1416f + 16.5f);
      if ((unsigned int)ref > 31u) ref = 0;
      out2[i] = (float)ref;
    }
    else {
      float s, c;
      sm_sincosf(a[i], &s, &c);
      out[i] = s;
      out2[i] = c;
    }
  }
}

}  // namespace cusift
