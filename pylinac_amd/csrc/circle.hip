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

// The same sampling on combine_surrounding_slices(slice +- k, "max") (pylinac/ct.py:3351-3386, called from
// CTP528CP504's circle profile, ct.py:1561-1580) WITHOUT building the combined slices: the maximum over the 2k + 1
// neighbouring slices is taken per tap (a ring of 20 radii touches ~2 % of a slice; the full combination moved every pixel
// of 7 slices).  np.max over exact integers commutes with the nearest-neighbour gather, so the samples are identical.
// Slice z of a volume reads z - k .. z + k of ITS volume the way the reference indexes its list (pl_combine_slices: a
// negative index wraps to the end of the volume, an index past the end -- IndexError in the reference -- reuses the last
// slice and the caller discards the profile).
// Lanes: EIGHT consecutive samples x EIGHT radii per wave (a lane takes radii rl, rl + 8, rl + 16 ...): the taps of one
// wave-wide gather then lie in a patch of about 4 x 3 pixels -- one to four cache lines -- where a wave of 64 consecutive
// samples of one radius spread over up to 32 rows (the kernel was bound by the address path: 47 cycles per gather
// instruction, SQ_WAIT_INST_ANY 85 %).  The taps are integers, so the sum over the radii is exact in ANY order: each lane
// adds its own radii, a three-step butterfly adds the eight lanes of a sample -- the same float64 as the reference's
// sequential `profile += ...`.  KPM >= 0: the 2 KPM + 1 slices are compile-time, so a tap's gathers are issued together
// (rounds 1-3: one runtime loop of 140 dependent gathers per sample).
constexpr int kCpRadLanes = 8, kCpSamples = 256 / kCpRadLanes;

template <typename T, int KPM>
__global__ void __launch_bounds__(256)
circle_profile_combined_kernel(const T* __restrict__ stack, int h, int w, const int64_t* __restrict__ slice_index,
                               int64_t per_volume, int k_pm, const double* __restrict__ cosv,
                               const double* __restrict__ sinv, int nsamp, const double* __restrict__ radii, int nr,
                               const double* __restrict__ cx, const double* __restrict__ cy, double divisor,
                               double* __restrict__ out, double* __restrict__ margin) {
  const int rl = threadIdx.x & (kCpRadLanes - 1);
  const int s_raw = blockIdx.x * kCpSamples + (threadIdx.x / kCpRadLanes);
  const bool live = s_raw < nsamp;
  const int s = live ? s_raw : nsamp - 1;                    // every lane stays for the butterfly
  const size_t frame = blockIdx.y;
  const int64_t g = slice_index ? slice_index[frame] : (int64_t)frame;
  const int64_t v0 = (g / per_volume) * per_volume, z = g - v0;
  const size_t per_frame = (size_t)h * w;
  const double c = cosv[s], sn = sinv[s];
  const double x0 = cx[frame], y0 = cy[frame];
  auto slice_of = [&](int d) {                               // the reference's list index z + d inside the slice's own volume
    int64_t q = z + d;
    if (q < 0) q += per_volume;
    if (q < 0) q = 0;
    if (q >= per_volume) q = per_volume - 1;
    return stack + (size_t)(v0 + q) * per_frame;
  };
  const double* rad = radii + frame * nr;
  // `margin` (optional, [frames], preset to +inf): how far the centre may move before ANY decision of this profile changes --
  // the smallest distance of a coordinate + 0.5 to an integer (the nearest-pixel choice) or of a coordinate to 0 / n - 1 (the
  // inside test).  ct.ctp528_batch samples about a centre fitted on the DEVICE and accepts the profile only if the exact
  // host fit (np.polyfit, bit for bit the reference's) lies closer than this margin: then every tap is the same pixel.
  double mrg = __longlong_as_double(0x7ff0000000000000LL);
  const bool want_margin = margin != nullptr;
  // a ring that stays a pixel clear of the frame on every side cannot meet a bounds decision: the inside test's margin is
  // then dropped from every tap (wave-uniform; the radii are sorted neither way, so the largest is looked up once)
  bool near_border = true;
  if (want_margin) {
    double rmax = 0.0;
    for (int k = 0; k < nr; ++k) rmax = fmax(rmax, fabs(rad[k]));
    near_border = !(x0 - rmax > 1.0 && x0 + rmax < (double)(w - 2) && y0 - rmax > 1.0 && y0 + rmax < (double)(h - 2));
  }
  auto tap = [&](double r, unsigned& off) {                  // -> inside?, offset of the nearest pixel
    const double x = c * r + x0;
    const double y = sn * r + y0;
    const bool in = x >= 0.0 && x <= (double)(w - 1) && y >= 0.0 && y <= (double)(h - 1);
    const double fx = floor(x + 0.5), fy = floor(y + 0.5);
    const int xi = (int)fx, yi = (int)fy;
    off = in ? (unsigned)yi * (unsigned)w + (unsigned)xi : 0u;
    if (want_margin) {
      // distance of c + 0.5 to the nearest integer = 0.5 - |frac - 0.5|; both coordinates, then the running minimum
      const double dx = ((x + 0.5) - fx) - 0.5, dy = ((y + 0.5) - fy) - 0.5;
      double m = 0.5 - fmax(fabs(dx), fabs(dy));
      if (near_border)                                       // |min(c, n - 1 - c)| = the distance to the nearer bound
        m = fmin(m, fmin(fabs(fmin(x, (double)(w - 1) - x)), fabs(fmin(y, (double)(h - 1) - y))));
      mrg = fmin(mrg, m);                                    // (a NaN coordinate never lowers it: the caller tests NaN centres)
    }
    return in;
  };
  double acc = 0.0;                                          // integers: exact whatever the order
  if constexpr (KPM >= 0) {
    constexpr int NS = 2 * KPM + 1;
    const T* base[NS];
#pragma unroll
    for (int d = 0; d < NS; ++d) base[d] = slice_of(d - KPM);
    for (int k = rl; k < nr; k += kCpRadLanes) {
      unsigned o0;
      const bool in0 = tap(rad[k], o0);
      T e0[NS];
#pragma unroll
      for (int d = 0; d < NS; ++d) e0[d] = base[d][o0];
      T m0 = e0[0];
#pragma unroll
      for (int d = 1; d < NS; ++d) m0 = e0[d] > m0 ? e0[d] : m0;
      acc = acc + (in0 ? (double)m0 : 0.0);
    }
  } else {
    for (int k = rl; k < nr; k += kCpRadLanes) {
      unsigned o0;
      const bool in0 = tap(rad[k], o0);
      T m = slice_of(-k_pm)[o0];
      for (int d = -k_pm + 1; d <= k_pm; ++d) {
        const T e = slice_of(d)[o0];
        m = e > m ? e : m;
      }
      acc = acc + (in0 ? (double)m : 0.0);
    }
  }
#pragma unroll
  for (int o = 1; o < kCpRadLanes; o <<= 1) acc = acc + __shfl_xor(acc, o, 64);
  if (live && rl == 0) out[frame * (size_t)nsamp + s] = (divisor == 1.0) ? acc : acc / divisor;
  if (want_margin) {
    mrg = pl_wave_reduce(mrg, [](double a, double b) { return a < b ? a : b; });
    // non-negative doubles order like their bit patterns
    if ((threadIdx.x & 63) == 0) atomicMin(reinterpret_cast<long long*>(margin + frame), __double_as_longlong(mrg));
  }
}

}  // namespace

extern "C" int pl_circle_profile_combined_ex(const void* stack, int dtype, int64_t n_stack, int h, int w,
                                             const int64_t* d_slice_index, int64_t m, int64_t slices_per_volume, int plusminus,
                                             const double* d_cos, const double* d_sin, int nsamp, const double* d_radii, int nr,
                                             const double* d_cx, const double* d_cy, double divisor, double* d_out,
                                             double* d_margin, void* stream) {
  PL_REQUIRE(stack && d_cos && d_sin && d_radii && d_cx && d_cy && d_out, "null pointer");
  PL_REQUIRE(m >= 0 && m <= 65535 && h > 0 && w > 0 && nsamp > 0 && nr > 0 && plusminus >= 0, "bad shape");
  PL_REQUIRE(slices_per_volume > 0 && n_stack > 0 && n_stack % slices_per_volume == 0, "the stack must hold whole volumes");
  PL_REQUIRE(d_slice_index || m == n_stack, "without a slice index every slice of the stack is sampled");
  PL_REQUIRE(divisor != 0.0, "zero divisor");
  PL_REQUIRE(dtype == PL_I16 || dtype == PL_U16 || dtype == PL_I32 || dtype == PL_U8, "integer slices");
  if (m == 0) return PL_OK;
  dim3 grid((unsigned)pl_cdiv(nsamp, kCpSamples), (unsigned)m);
  PL_REQUIRE((int64_t)h * w <= 0xffffffffLL, "frame too large");
#define CPC_LAUNCH(K)                                                                                                       \
  hipLaunchKernelGGL((circle_profile_combined_kernel<T, K>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)stack, h, w, \
                     d_slice_index, slices_per_volume, plusminus, d_cos, d_sin, nsamp, d_radii, nr, d_cx, d_cy, divisor, d_out, \
                     d_margin)
  PL_DISPATCH_DTYPE(dtype, T, {
    switch (plusminus) {
      case 0: CPC_LAUNCH(0); break;
      case 1: CPC_LAUNCH(1); break;
      case 2: CPC_LAUNCH(2); break;
      case 3: CPC_LAUNCH(3); break;
      default: CPC_LAUNCH(-1); break;
    }
  });
#undef CPC_LAUNCH
  return pl_check_launch("pl_circle_profile_combined");
}

extern "C" int pl_circle_profile_combined(const void* stack, int dtype, int64_t n_stack, int h, int w,
                                          const int64_t* d_slice_index, int64_t m, int64_t slices_per_volume, int plusminus,
                                          const double* d_cos, const double* d_sin, int nsamp, const double* d_radii, int nr,
                                          const double* d_cx, const double* d_cy, double divisor, double* d_out,
                                          void* stream) {
  return pl_circle_profile_combined_ex(stack, dtype, n_stack, h, w, d_slice_index, m, slices_per_volume, plusminus, d_cos, d_sin,
                                       nsamp, d_radii, nr, d_cx, d_cy, divisor, d_out, nullptr, stream);
}

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
