// blur_lanes.hip -- would the fused detection gain from TWO pixels per lane at four waves per SIMD?
//
// detect_fused_kernel keeps four pixels per lane: three rotating sets of 7 DoG rows (84 registers), a 9-row window (36)
// and a scale pair in flight need 224 registers => two waves per SIMD, where a slow-class VALU instruction (packed, DPP:
// 90 % of the kernel) costs 4.6-5.7 cycles instead of the 4.2-4.5 it costs at four or more.  With two pixels per lane the
// same state is ~110 registers => four waves per SIMD -- but the +-4-column neighbourhood of the horizontal pass then
// spans TWO lanes per side (a second DPP hop, DPP has no wave_shr:2), and the halo is three lanes per side instead of two.
// This microbenchmark prices that trade on the part of the kernel every wave-row executes (73 % of the headline images'
// wave-rows execute nothing else): window slide + vertical pass + neighbour exchange + horizontal pass + DoG into
// rotating row sets + the threshold pre-test, same operations in the same order as blur_dog_row / detect_chunk.inc
// (general taps: four scale pairs), on a 64 x 1920 x 1080 batch.  Occupancy is pinned with dynamic LDS (20 KB per
// one-wave workgroup: 8 per CU = 2 per SIMD; 10 KB: 4 per SIMD) so that both forms can be run at both occupancies.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o /tmp/blur_lanes tools/microbench/blur_lanes.hip && /tmp/blur_lanes
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
constexpr unsigned int kBufFlags = 0x00020000u;
constexpr int kPairs = 4, kDog = 7;

struct Taps {
  f2 k[kPairs][5];
};

__device__ __forceinline__ float prev_lane(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float next_lane(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ f2 prev2(f2 v) { return f2{prev_lane(v.x), prev_lane(v.y)}; }
__device__ __forceinline__ f2 next2(f2 v) { return f2{next_lane(v.x), next_lane(v.y)}; }
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// one row of C pixels per lane: blur + DoG of the 8 levels into D[7][C]
template <int C>
__device__ __forceinline__ void blur_dog_row(const float (&win)[9][C], const Taps &T, float (&D)[kDog][C]) {
  float ctr[C], p1[C], p2[C], p3[C], p4[C];
#pragma unroll
  for (int j = 0; j < C; ++j) {
    ctr[j] = win[4][j];
    p1[j] = win[3][j] + win[5][j];
    p2[j] = win[2][j] + win[6][j];
    p3[j] = win[1][j] + win[7][j];
    p4[j] = win[0][j] + win[8][j];
  }
  float prev_hi[C];
#pragma unroll
  for (int q = 0; q < kPairs; ++q) {
    const f2 k0 = T.k[q][0], k1 = T.k[q][1], k2 = T.k[q][2], k3 = T.k[q][3], k4 = T.k[q][4];
    f2 e[C + 8];  // columns -4 .. C+3
#pragma unroll
    for (int j = 0; j < C; ++j) {
      f2 v = k4 * splat(ctr[j]);
      v = pk_fma(k3, splat(p1[j]), v);
      v = pk_fma(k2, splat(p2[j]), v);
      v = pk_fma(k1, splat(p3[j]), v);
      v = pk_fma(k0, splat(p4[j]), v);
      e[4 + j] = v;
    }
    if (C == 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        e[j] = prev2(e[4 + j]);
        e[8 + j] = next2(e[4 + j]);
      }
    } else {  // C == 2: columns -4..-1 live in lanes -2 and -1, columns 2..5 in lanes +1 and +2: two hops
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        e[2 + j] = prev2(e[4 + j]);   // lane - 1
        e[j] = prev2(e[2 + j]);       // lane - 2
        e[6 + j] = next2(e[4 + j]);   // lane + 1
        e[8 + j] = next2(e[6 + j]);   // lane + 2
      }
    }
    f2 L[C];
#pragma unroll
    for (int j = 0; j < C; ++j) {
      const int m = 4 + j;
      f2 v = k4 * e[m];
      v = pk_fma(k3, e[m - 1] + e[m + 1], v);
      v = pk_fma(k2, e[m - 2] + e[m + 2], v);
      v = pk_fma(k1, e[m - 3] + e[m + 3], v);
      v = pk_fma(k0, e[m - 4] + e[m + 4], v);
      L[j] = v;
    }
#pragma unroll
    for (int j = 0; j < C; ++j) {
      if (q > 0) D[2 * q - 1][j] = prev_hi[j] - L[j].x;
      if (2 * q < kDog) D[2 * q][j] = L[j].x - L[j].y;
      prev_hi[j] = L[j].y;
    }
  }
}

template <int C>
__global__ void __launch_bounds__(64) blur_kernel(const float *__restrict__ img, int w, int h, int pitch, long stride,
                                                  int rows_per_wave, Taps T, float thr, unsigned int *__restrict__ hits) {
  extern __shared__ float pin[];  // occupancy pin only
  const int lane = threadIdx.x;
  constexpr int kHalo = C == 4 ? 2 : 3;
  constexpr int kStrip = (64 - 2 * kHalo) * C;
  const int bx = blockIdx.x, by = blockIdx.y;
  img += (long)blockIdx.z * stride;
  const int ya = by * rows_per_wave, yb = min(ya + rows_per_wave, h);
  if (ya >= yb) return;
  const int c0 = bx * kStrip - kHalo * C + lane * C;
  const int cc = min(max(c0, 0), w - C);
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc((void *)img, 0, (int)((unsigned int)h * (unsigned int)pitch * 4u), kBufFlags);
  auto load = [&](int y, float (&o)[C]) {
    const int yc = min(max(__builtin_amdgcn_readfirstlane(y), 0), h - 1);
    if (C == 4) {
      const f4 v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rin, cc * 4, yc * pitch * 4, 0));
      o[0] = v.x, o[1] = v.y, o[2 % C] = v.z, o[3 % C] = v.w;
    } else {
      const f2 v = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rin, cc * 4, yc * pitch * 4, 0));
      o[0] = v.x, o[1] = v.y;
    }
  };
  float win[9][C];
#pragma unroll
  for (int i = 0; i < 9; ++i) load(ya - 1 - 4 + i, win[i]);
  float DA[kDog][C], DB[kDog][C], DC[kDog][C];
#pragma unroll
  for (int p = 0; p < kDog; ++p)
#pragma unroll
    for (int j = 0; j < C; ++j) DA[p][j] = DB[p][j] = DC[p][j] = 0.f;
  unsigned int n_hit = 0;
  float ahead[C];
  load(ya - 1 + 5, ahead);
  auto row_step = [&](int yy, float (&D0)[kDog][C], float (&D1)[kDog][C], float (&D2)[kDog][C]) {
    float nxt[C];
#pragma unroll
    for (int j = 0; j < C; ++j) nxt[j] = ahead[j];
    load(yy + 6, ahead);
    blur_dog_row<C>(win, T, D2);
    // the threshold pre-test over the five searchable scales of the centre row (D1), then a stand-in for the analysis
    // that keeps D0 and D2 alive: it only runs for wave-rows with a centre above the threshold (none with thr = inf)
    float vmax = 0.0f;
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
      for (int j = 0; j < C; j += 2) vmax = max3f(vmax, fabsf(D1[s + 1][j]), fabsf(D1[s + 1][j + 1]));
    if (__builtin_amdgcn_ballot_w64(vmax > thr) != 0) {
      float acc = 0.f;
#pragma unroll
      for (int p = 0; p < kDog; ++p)
#pragma unroll
        for (int j = 0; j < C; ++j) acc += fminf(fminf(D0[p][j], D1[p][j]), D2[p][j]);
      n_hit += acc > 1e30f ? 1u : 0u;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < C; ++j) win[i][j] = win[i + 1][j];
#pragma unroll
    for (int j = 0; j < C; ++j) win[8][j] = nxt[j];
  };
  for (int yy = ya - 1; yy <= yb; yy += 3) {
    row_step(yy, DA, DB, DC);
    if (yy + 1 > yb) break;
    row_step(yy + 1, DB, DC, DA);
    if (yy + 2 > yb) break;
    row_step(yy + 2, DC, DA, DB);
  }
  if (n_hit) atomicAdd(hits, n_hit);
  if (pin[0] == 12345.f && lane == 99) hits[1] = 1;  // (never: keeps the LDS allocation)
}

int main() {
  const int n = 64, w = 1920, h = 1080, pitch = 1920, rows = 24;
  std::vector<float> host((size_t)h * pitch);
  unsigned int seed = 12345u;
  for (auto &v : host) {
    seed = seed * 1664525u + 1013904223u;
    v = (float)((seed >> 24) & 0xff);
  }
  float *img;
  unsigned int *hits;
  hipMalloc(&img, sizeof(float) * n * h * pitch);
  hipMalloc(&hits, 64);
  for (int i = 0; i < n; ++i) hipMemcpy(img + (size_t)i * h * pitch, host.data(), host.size() * 4, hipMemcpyHostToDevice);
  hipMemset(hits, 0, 64);
  Taps T;
  for (int q = 0; q < kPairs; ++q) {
    const float s0 = 0.6f + 0.35f * (2 * q), s1 = 0.6f + 0.35f * (2 * q + 1);
    float a[5], b[5], sa = 0.f, sb = 0.f;
    for (int j = 0; j < 5; ++j) {
      a[j] = expf(-(float)((4 - j) * (4 - j)) / (2 * s0 * s0));
      b[j] = expf(-(float)((4 - j) * (4 - j)) / (2 * s1 * s1));
      sa += (j < 4 ? 2 : 1) * a[j];
      sb += (j < 4 ? 2 : 1) * b[j];
    }
    for (int j = 0; j < 5; ++j) T.k[q][j] = f2{a[j] / sa, b[j] / sb};
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipFuncSetAttribute((const void *)blur_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024);
  hipFuncSetAttribute((const void *)blur_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024);
  struct Case {
    int cols, lds_kb;
    const char *name;
  } cases[] = {{4, 20, "4 px per lane, 2 waves per SIMD (the shipped form)"},
               {4, 10, "4 px per lane, LDS allows 4 waves (registers decide)"},
               {2, 20, "2 px per lane, 2 waves per SIMD"},
               {2, 10, "2 px per lane, 4 waves per SIMD"},
               {2, 5, "2 px per lane, LDS allows 8 waves (registers decide)"}};
  for (int pass = 0; pass < 2; ++pass)
    for (auto &c : cases) {
      const int halo = c.cols == 4 ? 2 : 3, strip = (64 - 2 * halo) * c.cols;
      dim3 grid((w + strip - 1) / strip, (h + rows - 1) / rows, n);
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (c.cols == 4)
          hipLaunchKernelGGL(blur_kernel<4>, grid, dim3(64), c.lds_kb * 1024, 0, img, w, h, pitch, (long)h * pitch, rows, T, 1e30f, hits);
        else
          hipLaunchKernelGGL(blur_kernel<2>, grid, dim3(64), c.lds_kb * 1024, 0, img, w, h, pitch, (long)h * pitch, rows, T, 1e30f, hits);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      if (pass == 1)
        printf("%-56s %7.3f ms per 64 x 1080p  (%5.1f Gpix/s)  grid %u x %u x %u\n", c.name, best, n * (double)w * h / best / 1e6,
               grid.x, grid.y, grid.z);
    }
  hipFuncAttributes a4, a2;
  hipFuncGetAttributes(&a4, (const void *)blur_kernel<4>);
  hipFuncGetAttributes(&a2, (const void *)blur_kernel<2>);
  printf("registers: 4 px per lane %d, 2 px per lane %d (512 per SIMD lane: waves = 512 / registers)\n", a4.numRegs, a2.numRegs);
  return 0;
}
