// tap_reads.hip -- would a second, column-shifted copy of the patch pay for the descriptor's bilinear taps?
//
// describe_all_kernel's descriptor phase reads 16 taps per lane per keypoint, each a 2 x 2 footprint of the 40-float-row
// LDS patch at an arbitrary column: two ds_read2_b32 (offsets 0,1 and 40,41).  The footprints of the 64 lanes are a
// rotated lattice; their bank conflicts are 29 % of the kernel's LDS-active cycles (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
// = 0.30).  The idea on the table since round 2: keep a second copy of the patch shifted by one column, so that every
// (s00, s10) pair is an ALIGNED 8-byte pair in one of the two copies and a tap becomes two ds_read_b64 -- at the price of
// 6.4 KB more LDS per wave (10.2 -> 16.6 KB: 9 waves per CU instead of 16, i.e. 2 per SIMD instead of 4; a 32 x 32 patch
// would allow 3).  This microbenchmark prices the READ side of that trade alone: the same rotated-lattice footprints
// (random orientation and scale per wave, a uniform shift per iteration), both forms, at 16 / 12 / 8 waves per CU.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/tap_reads tools/microbench/tap_reads.hip && /tmp/tap_reads
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kStride = 40, kPatch = kStride * kStride, kTaps = 16;

// kForm 0: two ds_read2_b32 at an arbitrary column (the shipped form); 1: two aligned ds_read_b64 from the copy whose
// column parity makes the pair aligned; 2: form 0 reading 32-bit words one by one (what a naive tap would do)
template <int kForm>
__global__ void __launch_bounds__(64) taps_kernel(const int *__restrict__ tab, float *__restrict__ out, int iters) {
  extern __shared__ float lds[];  // [patch A | patch B (A shifted left by one column) | padding that pins the occupancy]
  const int lane = threadIdx.x;
  for (int i = lane; i < kPatch; i += 64) {
    lds[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;
    lds[kPatch + i] = (float)(((i + 1) * 2654435761u) >> 20) * 1e-3f;  // B[i] = A[i + 1]
  }
  __syncthreads();
  int e[kTaps];
#pragma unroll
  for (int t = 0; t < kTaps; ++t) e[t] = tab[(blockIdx.x % 64 * kTaps + t) * 64 + lane];
  float acc = 0.f;
  const unsigned int base = (unsigned int)(size_t)(__attribute__((address_space(3))) float *)lds;
  for (int it = 0; it < iters; ++it) {
    const int shift = (it * 7) & 7;  // wave-uniform: moves the whole lattice, changes every lane's parity on odd shifts
    f2 a[kTaps], b[kTaps];
    // all 32 reads of the 16 taps in flight behind ONE wait, as in the kernel (the instructions are written out: the
    // compiler turns an f2 load through a float pointer into ds_read2_b32 whatever the alignment)
#pragma unroll
    for (int t = 0; t < kTaps; ++t) {
      const int i0 = e[t] + shift;
      if (kForm == 1) {
        const unsigned int addr = base + 4u * (unsigned int)((i0 & 1) ? kPatch + i0 - 1 : i0);  // 8-byte aligned (kStride even)
        asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:160" : "=v"(a[t]), "=v"(b[t]) : "v"(addr));
      } else {
        const unsigned int addr = base + 4u * (unsigned int)i0;
        asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %2 offset0:40 offset1:41"
                     : "=v"(a[t]), "=v"(b[t])
                     : "v"(addr));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < kTaps; ++t) acc += (a[t].x + a[t].y) + (b[t].x + b[t].y);
  }
  out[blockIdx.x * 64 + lane] = acc;
}

int main() {
  // 64 lattices: sample (i, j) of a 16 x 4 slab of the rotated 16 x 16 grid, lane = (j % 4) * 16 + i; taps at +-u, +-v
  std::vector<int> tab(64 * kTaps * 64);
  unsigned int seed = 777u;
  auto rnd = [&]() {
    seed = seed * 1664525u + 1013904223u;
    return (seed >> 8) * (1.0f / 16777216.0f);
  };
  for (int w = 0; w < 64; ++w) {
    const float th = rnd() * 6.2831853f, sp = 0.75f * (1.0f + rnd()), cx = 19.5f + rnd(), cy = 19.5f + rnd();
    const float c = std::cos(th), s = std::sin(th);
    for (int step = 0; step < 4; ++step)
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, j = (lane >> 4) + 4 * step;
        const float px = cx + (i - 7.5f) * sp * c - (j - 7.5f) * sp * s, py = cy + (i - 7.5f) * sp * s + (j - 7.5f) * sp * c;
        const float ox[4] = {c, -c, -s, s}, oy[4] = {s, -s, c, -c};
        for (int k = 0; k < 4; ++k) {
          int x = (int)std::floor(px + ox[k] - 0.5f), y = (int)std::floor(py + oy[k] - 0.5f);
          x = x < 0 ? 0 : (x > kStride - 10 ? kStride - 10 : x);  // (+ shift <= 7 and the +1 column stay inside the row)
          y = y < 0 ? 0 : (y > kStride - 2 ? kStride - 2 : y);
          tab[(w * kTaps + step * 4 + k) * 64 + lane] = y * kStride + x;
        }
      }
  }
  // calibration: every lane its own consecutive pair (no bank conflict possible), same instructions
  std::vector<int> lin(64 * kTaps * 64);
  for (size_t i = 0; i < lin.size(); ++i) lin[i] = 2 * (int)(i % 64);
  int *d_lin;
  hipMalloc(&d_lin, lin.size() * 4);
  hipMemcpy(d_lin, lin.data(), lin.size() * 4, hipMemcpyHostToDevice);
  int *d_tab;
  float *out;
  const int waves = 256 * 16 * 2, iters = 400;
  hipMalloc(&d_tab, tab.size() * 4);
  hipMalloc(&out, sizeof(float) * waves * 64);
  hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipFuncSetAttribute((const void *)taps_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipFuncSetAttribute((const void *)taps_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  struct Case {
    int form, lds_bytes, waves_per_cu;
    const char *name;
  } cases[] = {{2, 10240, 16, "calibration: 2 x ds_read2_b32, one consecutive pair per lane, 16 waves per CU"},
               {3, 10240, 16, "calibration: 2 x ds_read_b64,   one consecutive pair per lane, 16 waves per CU"},
               {0, 10240, 16, "2 x ds_read2_b32, any column (shipped), 16 waves per CU"},
               {1, 13312, 12, "2 x ds_read_b64 from the aligned copy, 12 waves per CU"},
               {1, 17408, 9, "2 x ds_read_b64 from the aligned copy,  9 waves per CU"},
               {1, 10240, 16, "2 x ds_read_b64 from the aligned copy, 16 waves per CU (if the LDS were there)"},
               {0, 17408, 9, "2 x ds_read2_b32, any column,            9 waves per CU"}};
  for (auto &c : cases) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      const int *t = c.form >= 2 ? d_lin : d_tab;
      if ((c.form & 1) == 0) hipLaunchKernelGGL(taps_kernel<0>, dim3(waves), dim3(64), c.lds_bytes, 0, t, out, iters);
      else hipLaunchKernelGGL(taps_kernel<1>, dim3(waves), dim3(64), c.lds_bytes, 0, t, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    // per CU: (waves / 256) waves x iters x 16 taps
    const double taps_per_cu = (double)waves / 256.0 * iters * kTaps;
    printf("%-78s %8.3f ms  %6.1f LDS-pipe cycles per wave-tap (2.4 GHz; 4 dwords x 64 lanes: 4 cycles at no conflict)\n", c.name,
           best, best * 1e-3 * 2.4e9 / taps_per_cu);
  }
  return 0;
}
