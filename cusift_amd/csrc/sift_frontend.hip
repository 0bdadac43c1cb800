// sift_frontend.hip -- the caller-side image front-end of the reference's programs, moved onto the device
// (SURVEY.md section 8f rank 2).  The callers decode an 8-bit image with OpenCV, convert it to float and optionally
// pre-blur it with cv::GaussianBlur(img, img, Size(3,3), 0.5) before upload (main.cpp:300-318, test/detector.cpp:19-27).
// Uploading the 8-bit pixels and converting here moves 4x fewer bytes over PCIe (2 MB instead of 8.3 MB per 1080p).
#include "sift_device.h"

namespace cusift {

// 8-bit -> float, exact (cv::Mat::convertTo(CV_32FC1)).  One lane converts 4 pixels: uchar4 in, float4 out.
__global__ void __launch_bounds__(256) u8_to_f32_kernel(float *__restrict__ dst, int dst_pitch, long dst_stride,
                                                       const unsigned char *__restrict__ src, int w, int h,
                                                       int src_pitch, long src_stride, int vec_ok) {
  const int y = blockIdx.y;
  const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (x4 >= w) return;
  const unsigned char *s = src + (long)blockIdx.z * src_stride + (long)y * src_pitch + x4;
  float *d = dst + (long)blockIdx.z * dst_stride + (long)y * dst_pitch + x4;
  if (vec_ok && x4 + 3 < w) {
    const uchar4 v = *reinterpret_cast<const uchar4 *>(s);
    *reinterpret_cast<float4 *>(d) = make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
  } else {
    for (int j = 0; j < 4 && x4 + j < w; ++j) d[j] = (float)s[j];
  }
}

// 3x3 separable Gaussian, the float path of cv::GaussianBlur(Size(3,3), sigma) as OpenCV's small symmetric filters
// evaluate it: k = {k1, k0, k1} = exp(-x^2/(2 sigma^2)) normalised; rows first: (S[x-1] + S[x+1])*k1 + S[x]*k0,
// then columns the same way; BORDER_REFLECT_101 (index -1 -> 1, n -> n-2).  In place is allowed (dst == src is NOT:
// the kernel reads a 3x3 neighbourhood; the host wrapper uses a scratch image).
__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  if (i < 0) return -i;
  if (i >= n) return 2 * n - 2 - i;
  return i;
}

__global__ void __launch_bounds__(256) gaussian3x3_kernel(float *__restrict__ dst, int dst_pitch, long dst_stride,
                                                         const float *__restrict__ src, int w, int h, int src_pitch,
                                                         long src_stride, float k0, float k1) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= w) return;
  src += (long)blockIdx.z * src_stride;
  dst += (long)blockIdx.z * dst_stride;
  const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
  float r[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const float *s = src + (long)reflect101(y - 1 + t, h) * src_pitch;
    r[t] = (s[xm] + s[xp]) * k1 + s[x] * k0;  // row filter (built with -ffp-contract=off: no fusion)
  }
  dst[(long)y * dst_pitch + x] = (r[0] + r[2]) * k1 + r[1] * k0;  // column filter
}

}  // namespace cusift
