// exec_mask.hip -- does a gfx950 SIMD skip the passes of a VALU instruction whose lanes are all masked off?
// Each wave runs a chain of independent VALU instructions under an EXEC mask chosen at launch; if the hardware skipped
// inactive 16- or 32-lane groups, a wave with one group active would retire its instructions 2-4x faster.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/exec_mask tools/microbench/exec_mask.hip && /tmp/exec_mask
#include <hip/hip_runtime.h>
#include <cstdio>

template <int kForm>
__global__ void __launch_bounds__(64) k(float *out, int iters, unsigned long long mask) {
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
  const bool on = (mask >> threadIdx.x) & 1ull;
  if (on) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (kForm == 0)
          asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %2, %2, %3, %4\n\tv_fma_f32 %3, %3, %4, %5\n\t"
                       "v_fma_f32 %4, %4, %5, %6\n\tv_fma_f32 %5, %5, %6, %7\n\tv_fma_f32 %6, %6, %7, %0\n\tv_fma_f32 %7, %7, %0, %1"
                       : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
        else
          asm volatile("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %1, %1, %2, %3\n\tv_max3_f32 %2, %2, %3, %4\n\tv_max3_f32 %3, %3, %4, %5\n\t"
                       "v_max3_f32 %4, %4, %5, %6\n\tv_max3_f32 %5, %5, %6, %7\n\tv_max3_f32 %6, %6, %7, %0\n\tv_max3_f32 %7, %7, %0, %1"
                       : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

int main() {
  const int waves = 256 * 4 * 4, iters = 20000;  // four waves per SIMD
  float *out;
  hipMalloc(&out, sizeof(float) * waves * 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  struct { const char *name; unsigned long long m; } masks[] = {
      {"all 64 lanes", ~0ull}, {"lanes 0-31", 0xffffffffull}, {"lanes 0-15", 0xffffull}, {"lanes 16-31", 0xffff0000ull},
      {"lanes 0-15 + 32-47", 0x0000ffff0000ffffull}, {"lane 0 only", 1ull}, {"every 4th lane", 0x1111111111111111ull}};
  for (int form = 0; form < 2; ++form)
    for (auto &mk : masks) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (form == 0) hipLaunchKernelGGL(k<0>, dim3(waves), dim3(64), 0, 0, out, iters, mk.m);
        else hipLaunchKernelGGL(k<1>, dim3(waves), dim3(64), 0, 0, out, iters, mk.m);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double insts_per_simd = 4.0 * iters * 32.0;  // four waves x iters x 32 instructions
      printf("%-12s %-22s %8.3f ms  %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", form ? "v_max3_f32" : "v_fma_f32",
             mk.name, best, best * 1e-3 * 2.4e9 / insts_per_simd);
    }
  return 0;
}
