// valu_rate.hip -- what one SIMD of gfx950 sustains in VALU wave-instructions per cycle, as a function of the waves
// resident on it and of the instruction kind (v_fma_f32, v_pk_fma_f32, DPP move, a dependent chain).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/microbench/valu_rate.hip && ./valu_rate
// Each wave runs `iters` iterations of 32 instructions on 16 independent accumulators (or ONE for the chain variant).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int kKind>
__global__ void __launch_bounds__(64) k(float *out, int iters, float a, float b) {
  float x[16];
  f2 p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    x[i] = threadIdx.x * 0.001f + i;
    p[i] = f2{x[i], x[i] + 1.0f};
  }
  const f2 a2 = f2{a, a}, b2 = f2{b, b};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (kKind == 0) x[i] = __builtin_fmaf(x[i], a, b);
        if (kKind == 1) p[i] = __builtin_elementwise_fma(p[i], a2, b2);
        if (kKind == 2) x[0] = __builtin_fmaf(x[0], a, b);  // one dependent chain
        if (kKind == 3)
          x[i] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x[(i + 1) & 15]), 0x138, 0xf, 0xf, true));
        if (kKind == 4) x[i] = fmaxf(fmaxf(x[i], a), x[(i + 3) & 15]);  // v_max3: two VGPR sources + a scalar
        // round 3: the classes the issue bound of bench.py prices (tools/isa_mix.py)
        if (kKind == 5) x[i] = x[i] * x[(i + 3) & 15];                              // v_mul_f32 v, v: two VGPR sources
        if (kKind == 6) x[i] = __builtin_fmaf(x[i], x[(i + 3) & 15], x[(i + 7) & 15]);  // v_fma_f32 v, v, v: three
        if (kKind == 7) x[i] = fmaxf(fmaxf(x[i], x[(i + 5) & 15]), x[(i + 3) & 15]);   // v_max3_f32 v, v, v
        if (kKind == 8) {                                                             // v_add_u32 v, s: integer, one
          int t = __builtin_bit_cast(int, x[i]) + it;
          x[i] = __builtin_bit_cast(float, t);
        }
        if (kKind == 9) {  // v_cmp_lt_f32 + v_cndmask_b32: the compare / select pair (two instructions per element)
          x[i] = (x[i] < x[(i + 3) & 15]) ? x[(i + 5) & 15] : x[i];
        }
        if (kKind == 10) x[i] = __builtin_floorf(x[i]) + 0.25f + a * 0.f;             // v_floor_f32 + v_add_f32 v, c
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i] + p[i].x + p[i].y;
  if (s == 12345.678f) out[0] = s;
}

template <int kKind>
void run(const char *name, int cus) {
  float *d;
  hipMalloc(&d, 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int waves_per_simd : {1, 2, 3, 4, 6, 8}) {
    const int blocks = cus * 4 * waves_per_simd;  // 64-thread blocks: one wave each, 4 SIMDs per CU
    hipLaunchKernelGGL(k<kKind>, dim3(blocks), dim3(64), 0, 0, d, 100, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<kKind>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)waves_per_simd * iters * 32;
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const double cycles = ms * 1e-3 * clk_khz * 1e3;
    printf("%-14s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction per SIMD (clock %d MHz)\n", name,
           waves_per_simd, ms, cycles / insts_per_simd, clk_khz / 1000);
  }
}

int main() {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  run<0>("v_fma_f32", cus);
  run<1>("v_pk_fma_f32", cus);
  run<2>("fma chain", cus);
  run<3>("v_mov_dpp", cus);
  run<4>("v_max3_f32", cus);
  run<5>("v_mul v,v", cus);
  run<6>("v_fma v,v,v", cus);
  run<7>("v_max3 v,v,v", cus);
  run<8>("v_add_u32 v,s", cus);
  run<9>("cmp+cndmask(2)", cus);
  run<10>("floor+add(2)", cus);
  return 0;
}
