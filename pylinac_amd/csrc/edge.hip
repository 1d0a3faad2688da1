// Sobel derivative with scipy semantics.
//
// Replaces: scipy.ndimage.sobel(float32 image, axis) at pylinac/core/image.py:1006-1007
// (BaseImage.gamma).  ndimage.sobel = correlate1d(input, [-1, 0, 1], axis) followed by
// correlate1d(., [1, 2, 1], other axis), mode='reflect', each pass accumulated in float64 and CAST
// INTO THE IMAGE DTYPE (scipy's anti-symmetric / symmetric correlate1d branches):
//     d   = T( x[0]*0 + (x[-1] - x[+1]) * (-1) )        along `axis`
//     out = T( d[0]*2 + (d[-1] + d[+1]) * 1 )           along the other axis
// One lane per output pixel; the six derivative inputs come straight from L2 (3x3 neighbourhood).
#include "pl_common.h"

namespace {

template <typename T>
__device__ __forceinline__ T deriv(const T* __restrict__ f, int h, int w, int r, int c, int axis) {
  double xm, x0, xp;
  if (axis == 0) {
    xm = (double)f[(size_t)pl_reflect(r - 1, h) * w + c];
    x0 = (double)f[(size_t)r * w + c];
    xp = (double)f[(size_t)pl_reflect(r + 1, h) * w + c];
  } else {
    xm = (double)f[(size_t)r * w + pl_reflect(c - 1, w)];
    x0 = (double)f[(size_t)r * w + c];
    xp = (double)f[(size_t)r * w + pl_reflect(c + 1, w)];
  }
  double acc = x0 * 0.0;
  acc = acc + (xm - xp) * -1.0;
  return pl_from_double<T>(acc);
}

template <typename T>
__global__ void __launch_bounds__(256)
sobel_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t total, int h, int w, int axis) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % w);
  const int64_t t = i / w;
  const int r = (int)(t % h);
  const T* f = in + (t / h) * (size_t)h * w;
  double dm, d0, dp;
  if (axis == 0) {  // derivative along rows, smoothing along columns
    dm = (double)deriv<T>(f, h, w, r, pl_reflect(c - 1, w), 0);
    d0 = (double)deriv<T>(f, h, w, r, c, 0);
    dp = (double)deriv<T>(f, h, w, r, pl_reflect(c + 1, w), 0);
  } else {
    dm = (double)deriv<T>(f, h, w, pl_reflect(r - 1, h), c, 1);
    d0 = (double)deriv<T>(f, h, w, r, c, 1);
    dp = (double)deriv<T>(f, h, w, pl_reflect(r + 1, h), c, 1);
  }
  double acc = d0 * 2.0;
  acc = acc + (dm + dp) * 1.0;
  out[i] = pl_from_double<T>(acc);
}

}  // namespace

extern "C" int pl_sobel(const void* in, void* out, int dtype, int64_t n, int h, int w, int axis,
                        void* stream) {
  PL_REQUIRE(in && out && in != out, "null or aliased pointers");
  PL_REQUIRE(n >= 0 && h > 0 && w > 0, "bad shape");
  PL_REQUIRE(axis == 0 || axis == 1, "axis must be 0 or 1");
  if (n == 0) return PL_OK;
  const int64_t total = n * (int64_t)h * w;
  PL_REQUIRE(pl_cdiv(total, 256) <= 0x7fffffffLL, "batch too large");
  PL_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL(sobel_kernel<T>, dim3((unsigned)pl_cdiv(total, 256)), dim3(256), 0,
                                       (hipStream_t)stream, (const T*)in, (T*)out, total, h, w, axis));
  return pl_check_launch("pl_sobel");
}
