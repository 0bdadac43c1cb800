// lds_atomic.hip -- what does a CONFLICT-FREE ds_add_f32 cost on gfx950?  Round 1 measured ~150 cycles per wave
// instruction for the reference's histogram atomics (many lanes on one address).  Here every lane adds into slots of its
// own ([slot][lane] layout: lane l always on bank l), the pattern of the descriptor's vertical pass, against the
// read-add-write it would replace.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_atomic tools/microbench/lds_atomic.hip && /tmp/lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>

template <int kMode>
__global__ void __launch_bounds__(64) k(float *out, const int *slots, int iters) {
  __shared__ float h[16 * 64];
  const int lane = threadIdx.x;
  for (int i = lane; i < 16 * 64; i += 64) h[i] = 0.0f;
  __syncthreads();
  int s = slots[lane];
  float v = 1.0f + lane * 0.001f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float *p = h + ((s + r * 5) & 7) * 64 + lane;
      if (kMode == 0) {  // read-modify-write of two slots (ds_read2st64 + 2 adds + ds_write2st64)
        const float a = p[0], b = p[64];
        p[0] = a + v;
        p[64] = b + v * 0.5f;
      } else {  // two LDS atomics without return
        __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float *)p, v, 0, 0, false);
        __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float *)(p + 64), v * 0.5f, 0, 0, false);
      }
    }
    s = (s * 5 + 1) & 7;
  }
  __syncthreads();
  float acc = 0.f;
  for (int i = 0; i < 16; ++i) acc += h[i * 64 + lane];
  out[blockIdx.x * 64 + lane] = acc;
}

int main() {
  const int waves = 256 * 4 * 4, iters = 2000;
  float *out;
  int *slots;
  hipMalloc(&out, sizeof(float) * waves * 64);
  hipMalloc(&slots, sizeof(int) * 64);
  int h_slots[64];
  for (int i = 0; i < 64; ++i) h_slots[i] = (i * 7 + 3) & 7;
  hipMemcpy(slots, h_slots, sizeof(h_slots), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(waves), dim3(64), 0, 0, out, slots, iters);
      else hipLaunchKernelGGL(k<1>, dim3(waves), dim3(64), 0, 0, out, slots, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    // per CU: 16 waves x iters x 8 visits
    const double visits_per_cu = 16.0 * iters * 8.0;
    printf("%-34s %8.3f ms  %.1f cycles per visit per CU (two slots; 16 waves per CU, 2.4 GHz)\n",
           mode ? "2 x ds_add_f32 (no return)" : "ds_read2st64 + 2 add + ds_write2st64", best, best * 1e-3 * 2.4e9 / visits_per_cu);
  }
  return 0;
}
