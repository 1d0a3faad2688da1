// Circular / collapsed-circular profile sampling.
//
// Replaces: ndimage.map_coordinates(image, [y, x], order=0) as called by
//   CircleProfile._profile            pylinac/core/profile.py:2279-2283
//   CollapsedCircleProfile._profile   pylinac/core/profile.py:2473-2483 (sum over num_profiles radii,
//                                      then /= num_profiles)
// callers: Starshot (pylinac/starshot.py:770-782), CTP528 MTF (pylinac/ct.py:1561-1580), CatPhan
// origin search (pylinac/ct.py:2468-2495).
//
// Semantics (scipy 1.15.3, verified by probing): nearest neighbour index floor(c + 0.5); mode
// 'constant', cval 0: ANY coordinate outside [0, n-1] -- even fractionally -- yields 0.
// x = cos*r + cx and y = sin*r + cy are evaluated in float64 in exactly that order (no FMA) from
// host-computed cos/sin tables (numpy's libm values; device sin/cos may differ in the last ulp,
// which would move a sample across a rounding boundary).
// The radii are accumulated in their given order in float64, like `profile += ...`.
// One lane per sample; the gather goes through L2 (each ring touches every row it crosses once).
#include "pl_common.h"

namespace {

template <typename T>
__global__ void __launch_bounds__(256)
circle_profile_kernel(const T* __restrict__ img, int h, int w, const double* __restrict__ cosv,
                      const double* __restrict__ sinv, int nsamp, const double* __restrict__ radii, int nr,
                      const double* __restrict__ cx, const double* __restrict__ cy, double divisor,
                      double* __restrict__ out) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  const size_t frame = blockIdx.y;
  if (s >= nsamp) return;
  const T* f = img + frame * (size_t)h * w;
  const double c = cosv[s], sn = sinv[s];
  const double x0 = cx[frame], y0 = cy[frame];
  double acc = 0.0;
  for (int k = 0; k < nr; ++k) {
    const double r = radii[frame * nr + k];
    const double x = c * r + x0;
    const double y = sn * r + y0;
    double v = 0.0;
    if (x >= 0.0 && x <= (double)(w - 1) && y >= 0.0 && y <= (double)(h - 1)) {
      const int xi = (int)floor(x + 0.5), yi = (int)floor(y + 0.5);
      v = (double)f[(size_t)yi * w + xi];
    }
    acc = acc + v;
  }
  out[frame * (size_t)nsamp + s] = (divisor == 1.0) ? acc : acc / divisor;
}

}  // namespace

extern "C" int pl_circle_profile(const void* img, int dtype, int64_t n, int h, int w, const double* d_cos,
                                 const double* d_sin, int nsamp, const double* d_radii, int nr,
                                 const double* d_cx, const double* d_cy, double divisor, double* d_out,
                                 void* stream) {
  PL_REQUIRE(img && d_cos && d_sin && d_radii && d_cx && d_cy && d_out, "null pointer");
  PL_REQUIRE(n >= 0 && n <= 65535 && h > 0 && w > 0 && nsamp > 0 && nr > 0, "bad shape");
  PL_REQUIRE(divisor != 0.0, "zero divisor");
  if (n == 0) return PL_OK;
  dim3 grid((unsigned)pl_cdiv(nsamp, 256), (unsigned)n);
  PL_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL(circle_profile_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream,
                                       (const T*)img, h, w, d_cos, d_sin, nsamp, d_radii, nr, d_cx, d_cy,
                                       divisor, d_out));
  return pl_check_launch("pl_circle_profile");
}
