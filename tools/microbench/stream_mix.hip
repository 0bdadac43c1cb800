// Microbenchmark: what HBM rate does MI355X sustain for the blur+DoG traffic mix (1 float4 stream read,
// 7 float4 streams written), and how much do the store pattern (992-B unaligned row segments into 7 planes,
// as the strip kernel writes them) and its alignment matter?   hipcc --offload-arch=gfx950 -O3 -o stream_mix stream_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// (A) ideal: grid-stride, lane reads one float4 and writes 7 float4 (plane-major), fully aligned
__global__ void mix_ideal(const f4* __restrict__ in, f4* __restrict__ out, long n4, long plane4) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f4 v = in[i];
#pragma unroll
    for (int p = 0; p < 7; ++p) out[p * plane4 + i] = v * (float)(p + 1);
  }
}
// (B) write only, same as A without the read
__global__ void write_only(f4* __restrict__ out, long n4, long plane4) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f4 v = {1.f, 2.f, 3.f, (float)i};
#pragma unroll
    for (int p = 0; p < 7; ++p) out[p * plane4 + i] = v;
  }
}
// (C) strip pattern: wave owns `cols` columns (lanes write 16 B each, lanes [lo,hi) active), marches `rows` rows,
//     7 planes per row; strips start at strip*valid columns.  cols=256: valid=248 -> unaligned 992-B segments; valid=256 -> aligned
__global__ void strip_pattern(const float* __restrict__ in, float* __restrict__ out, int w, int h, int pitch, int rows_per_wave,
                              int valid, int do_read, int swz, int horiz, int nt) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (swz) {  // XCD-aware remap: blocks that share an XCD (same linear id mod 8) take consecutive strips
    const int nx = gridDim.x, ny = gridDim.y, nz = gridDim.z;
    const int total = nx * ny * nz;
    int id = (bz * ny + by) * nx + bx;
    if (total % 8 == 0) id = (id % 8) * (total / 8) + id / 8;
    bx = id % nx; by = (id / nx) % ny; bz = id / (nx * ny);
  }
  // horiz: the 4 waves of a block take 4 horizontally adjacent strips (same rows) instead of 4 row chunks
  const int strip = horiz ? bx * 4 + wv : bx;
  const int y0 = (horiz ? by : by * 4 + wv) * rows_per_wave;
  if (y0 >= h) return;
  const int y1 = min(y0 + rows_per_wave, h);
  const long plane = (long)h * pitch;
  const long img_off = (long)bz * plane;
  const int halo = (256 - valid) / 8;  // lanes each side
  const int c0 = strip * valid - halo * 4 + lane * 4;
  if (strip * valid >= w) return;
  const bool writer = lane >= halo && lane < 64 - halo && c0 < w && c0 >= 0;
  const int cl = min(max(c0, 0), w - 4);
  f4 acc = {0, 0, 0, 0};
  for (int y = y0; y < y1; ++y) {
    if (do_read) acc += *(const f4*)(in + img_off + (long)y * pitch + cl);
    if (writer) {
#pragma unroll
      for (int p = 0; p < 7; ++p) {
        f4* dst = (f4*)(out + img_off * 7 + p * plane + (long)y * pitch + c0);
        if (nt) __builtin_nontemporal_store(acc + (float)p, dst); else *dst = acc + (float)p;
      }
    }
  }
}
// (D) read only streaming
__global__ void read_only(const f4* __restrict__ in, float* __restrict__ sink, long n4) {
  f4 acc = {0, 0, 0, 0};
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) acc += in[i];
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

int main() {
  const int n = 64, w = 1920, h = 1080, pitch = 1920;
  const long plane = (long)h * pitch, n_px = plane * n;
  float *in, *out;
  CK(hipMalloc(&in, n_px * 4));
  CK(hipMalloc(&out, n_px * 4 * 7));
  CK(hipMemset(in, 0, n_px * 4));
  CK(hipMemset(out, 0, n_px * 4 * 7));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto timeit = [&](const char* name, double bytes, auto launch) {
    for (int i = 0; i < 2; ++i) launch();
    CK(hipDeviceSynchronize());
    float best = 1e9, tot = 0;
    const int reps = 8;
    for (int i = 0; i < reps; ++i) {
      CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best; tot += ms;
    }
    printf("%-52s avg %.3f ms  best %.3f ms  -> %.0f GB/s (best)\n", name, tot / reps, best, bytes / best / 1e6);
  };
  const long n4 = n_px / 4;
  const double rw = n_px * 32.0, wr = n_px * 28.0, rd = n_px * 4.0;
  for (int grid : {2048, 8192}) {
    char nm[96];
    snprintf(nm, 96, "A ideal 1R+7W float4 grid-stride, grid=%d", grid);
    timeit(nm, rw, [&] { hipLaunchKernelGGL(mix_ideal, dim3(grid), dim3(256), 0, 0, (const f4*)in, (f4*)out, n4, n4); });
    snprintf(nm, 96, "B write-only 7W, grid=%d", grid);
    timeit(nm, wr, [&] { hipLaunchKernelGGL(write_only, dim3(grid), dim3(256), 0, 0, (f4*)out, n4, n4); });
  }
  timeit("D read-only (7 planes = 3.7 GB)", wr, [&] { hipLaunchKernelGGL(read_only, dim3(4096), dim3(256), 0, 0, (const f4*)out, in, n4 * 7); });
  for (int rows : {32, 68}) {
    for (int valid : {224, 256}) {
      for (int nt : {0, 1}) for (int swz : {0, 1}) for (int horiz : {0, 1}) {
        char nm[128];
        snprintf(nm, 128, "C strips valid=%d rows/wave=%d nontemporal=%d xcd_swizzle=%d horiz=%d", valid, rows, nt, swz, horiz);
        const int strips = (w + valid - 1) / valid, chunks = (h + rows - 1) / rows;
        dim3 grid = horiz ? dim3((strips + 3) / 4, chunks, n) : dim3(strips, (chunks + 3) / 4, n);
        timeit(nm, rw, [&] { hipLaunchKernelGGL(strip_pattern, grid, dim3(256), 0, 0, in, out, w, h, pitch, rows, valid, 1, swz, horiz, nt); });
      }
    }
  }
  return 0;
}
