// CatPhan slice localisation, the edge-image half (pylinac/ct.py:391-392, 3327-3338) in ONE pass over the slices:
//   raw   = skimage.filters.scharr(slice.astype(float))            pylinac/ct.py:391, 3327
//   max(raw)                                                       the "no edges" test np.max(edges) < 0.1, ct.py:392
//   edges = skimage.filters.gaussian(raw, sigma)  (mode 'nearest') ct.py:3328 (ndimage.gaussian_filter underneath)
//   edges[disk].min(), .max()                                      the histogram range of threshold_otsu, ct.py:3334-3338
// Rounds 1-3 ran this as scharr -> gaussian axis 0 -> gaussian axis 1 -> minmax -> minmax_masked: five launches and
// 2 MiB of float64 per 512 x 512 slice written three times and read four times.  Here a workgroup owns a 24 x 64 output
// tile: the raw samples with a (radius + 1) halo go to LDS once (clamped coordinates: for the one pixel beyond the frame
// that scharr's 'reflect' border needs, reflection and clamping coincide), the Scharr magnitude is evaluated on the tile
// plus the Gaussian's halo -- at CLAMPED frame coordinates, which is exactly what mode 'nearest' feeds the filter --,
// then axis 0 and axis 1 of the Gaussian run from LDS with scipy's symmetric-kernel operation order (centre tap first,
// then (left + right) * weight from the outermost pair inwards: ni_filters.c NI_Correlate1D), and the three extrema
// leave as float64 atomics.  Every float64 operation and its order are those of scharr_kernel (ct.hip) and
// gauss_generic (gaussian.hip), so the plane is bit-identical to the separate entry points'.
#include "pl_common.h"

namespace {

constexpr int kEfThreads = 256;
// 24 x 64 tiles: 37 KB of LDS and (radius 4) 97 registers leave FOUR workgroups on a CU; the 32-row tile of the first version
// (48 KB, 129 registers) left three, and the kernel waits more than it computes
constexpr int kTH = 24, kTW = 64;

__device__ __forceinline__ void ef_atomic_min(double* addr, double v) {
  unsigned long long* a = reinterpret_cast<unsigned long long*>(addr);
  unsigned long long old = *a;
  while (v < __longlong_as_double((long long)old)) {
    const unsigned long long assumed = old;
    old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(v));
    if (old == assumed) break;
  }
}
__device__ __forceinline__ void ef_atomic_max(double* addr, double v) {
  unsigned long long* a = reinterpret_cast<unsigned long long*>(addr);
  unsigned long long old = *a;
  while (v > __longlong_as_double((long long)old)) {
    const unsigned long long assumed = old;
    old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(v));
    if (old == assumed) break;
  }
}

__global__ void ef_init_kernel(double* __restrict__ rawmax, double* __restrict__ mn, double* __restrict__ mx, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kEfThreads + threadIdx.x;
  if (i >= n) return;
  const double pinf = __longlong_as_double(0x7ff0000000000000LL), ninf = __longlong_as_double((long long)0xfff0000000000000ULL);
  rawmax[i] = ninf;
  mn[i] = pinf;
  mx[i] = ninf;
}

template <typename T, int RAD>
__global__ void __launch_bounds__(kEfThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
edge_field_kernel(const T* __restrict__ in, int h, int w, int tiles_r, int tiles_c, const double* __restrict__ wts,
                  const uint8_t* __restrict__ mask, double* __restrict__ out, double* __restrict__ rawmax,
                  double* __restrict__ mn, double* __restrict__ mx) {
  constexpr int SH = kTH + 2 * RAD, SW = kTW + 2 * RAD;       // Scharr tile
  constexpr int RH = SH + 2, RW = SW + 2;                     // raw tile
  constexpr int SP = SW + 1, VP = SW + 1;                     // odd pitches: column walks spread over the banks
  __shared__ T raw[RH * RW];
  __shared__ double S[SH * SP];
  __shared__ double V[kTH * VP];
  __shared__ double s_w[RAD + 1];
  __shared__ double s_red[3][kEfThreads / PL_WAVE];

  const int tid = threadIdx.x;
  unsigned b = blockIdx.x;
  const int tc = (int)(b % (unsigned)tiles_c);
  b /= (unsigned)tiles_c;
  const int tr = (int)(b % (unsigned)tiles_r);
  const int64_t f = b / (unsigned)tiles_r;
  const int r0 = tr * kTH, c0 = tc * kTW;
  const T* src = in + f * (int64_t)h * w;
  if (tid <= RAD) s_w[tid] = wts[tid];
  // ---- raw tile, clamped coordinates
  for (int i = tid; i < RH * RW; i += kEfThreads) {
    const int a = i / RW, bb = i - a * RW;
    int rr = r0 - RAD - 1 + a, cc = c0 - RAD - 1 + bb;
    rr = rr < 0 ? 0 : (rr >= h ? h - 1 : rr);
    cc = cc < 0 ? 0 : (cc >= w ? w - 1 : cc);
    raw[i] = src[(int64_t)rr * w + cc];
  }
  __syncthreads();
  // ---- Scharr magnitude at the clamped coordinates of every tile position (scharr_kernel's operations)
  double rmax = __longlong_as_double((long long)0xfff0000000000000ULL);
  for (int i = tid; i < SH * SW; i += kEfThreads) {
    const int si = i / SW, sj = i - si * SW;
    const int vr = r0 - RAD + si, vc = c0 - RAD + sj;
    const int rr = vr < 0 ? 0 : (vr >= h ? h - 1 : vr);
    const int cc = vc < 0 ? 0 : (vc >= w ? w - 1 : vc);
    const int a = rr - (r0 - RAD - 1), bb = cc - (c0 - RAD - 1);
    auto at = [&](int da, int db) { return (double)raw[(a + da) * RW + (bb + db)]; };
    const double ka = 0.1875, kb = 0.625;
    double s0 = 0.0;
    s0 = s0 + at(-1, -1) * -ka; s0 = s0 + at(-1, 0) * -kb; s0 = s0 + at(-1, 1) * -ka;
    s0 = s0 + at(1, -1) * ka;   s0 = s0 + at(1, 0) * kb;   s0 = s0 + at(1, 1) * ka;
    double s1 = 0.0;
    s1 = s1 + at(-1, -1) * -ka; s1 = s1 + at(-1, 1) * ka;
    s1 = s1 + at(0, -1) * -kb;  s1 = s1 + at(0, 1) * kb;
    s1 = s1 + at(1, -1) * -ka;  s1 = s1 + at(1, 1) * ka;
    double o = 0.0;
    o = o + s0 * s0;
    o = o + s1 * s1;
    const double e = sqrt(o) / 1.4142135623730951;
    S[si * SP + sj] = e;
    if (si >= RAD && si < RAD + kTH && sj >= RAD && sj < RAD + kTW && vr < h && vc < w) rmax = e > rmax ? e : rmax;
  }
  __syncthreads();
  // ---- Gaussian along axis 0: item = (8-row segment, column), a 8 + 2 RAD register window per item
  for (int i = tid; i < (kTH / 8) * SW; i += kEfThreads) {
    const int seg = i / SW, j = i - seg * SW;
    double win[8 + 2 * RAD];
#pragma unroll
    for (int k = 0; k < 8 + 2 * RAD; ++k) win[k] = S[(seg * 8 + k) * SP + j];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      double acc = win[q + RAD] * s_w[RAD];
#pragma unroll
      for (int k = RAD; k >= 1; --k) acc = acc + (win[q + RAD - k] + win[q + RAD + k]) * s_w[RAD - k];
      V[(seg * 8 + q) * VP + j] = acc;
    }
  }
  __syncthreads();
  // ---- Gaussian along axis 1 + store + masked extrema: lane = column, wave = 8 rows
  double lo = __longlong_as_double(0x7ff0000000000000LL), hi = __longlong_as_double((long long)0xfff0000000000000ULL);
  {
    const int j = tid & 63, wv = tid >> 6;
    const int cc = c0 + j;
    for (int q = 0; q < kTH / 4; ++q) {
      const int i = wv * (kTH / 4) + q;
      const int rr = r0 + i;
      const double* vrow = V + i * VP + j + RAD;
      double acc = vrow[0] * s_w[RAD];
#pragma unroll
      for (int k = RAD; k >= 1; --k) acc = acc + (vrow[-k] + vrow[k]) * s_w[RAD - k];
      if (rr < h && cc < w) {
        const int64_t p = (int64_t)rr * w + cc;
        out[f * (int64_t)h * w + p] = acc;
        if (!mask || mask[p]) { lo = acc < lo ? acc : lo; hi = acc > hi ? acc : hi; }
      }
    }
  }
  rmax = pl_wave_reduce(rmax, [](double a, double c) { return a > c ? a : c; });
  lo = pl_wave_reduce(lo, [](double a, double c) { return a < c ? a : c; });
  hi = pl_wave_reduce(hi, [](double a, double c) { return a > c ? a : c; });
  const int lane = tid & 63, wv = tid >> 6;
  if (lane == 0) { s_red[0][wv] = rmax; s_red[1][wv] = lo; s_red[2][wv] = hi; }
  __syncthreads();
  if (tid == 0) {
    for (int k = 1; k < kEfThreads / PL_WAVE; ++k) {
      rmax = s_red[0][k] > rmax ? s_red[0][k] : rmax;
      lo = s_red[1][k] < lo ? s_red[1][k] : lo;
      hi = s_red[2][k] > hi ? s_red[2][k] : hi;
    }
    ef_atomic_max(rawmax + f, rmax);
    ef_atomic_min(mn + f, lo);
    ef_atomic_max(mx + f, hi);
  }
}

template <typename T>
int ef_launch(const T* in, int64_t n, int h, int w, const double* wts, int radius, const uint8_t* mask, double* out,
              double* rawmax, double* mn, double* mx, hipStream_t st) {
  const int tiles_r = (int)pl_cdiv(h, kTH), tiles_c = (int)pl_cdiv(w, kTW);
  const int64_t blocks = n * tiles_r * tiles_c;
  if (blocks > 0x7fffffffLL) { pl_set_error("pl_scharr_gaussian: batch too large for one launch"); return PL_ERR_INVALID_ARG; }
  hipLaunchKernelGGL(ef_init_kernel, dim3((unsigned)pl_cdiv(n, kEfThreads)), dim3(kEfThreads), 0, st, rawmax, mn, mx, n);
#define EF_CASE(R)                                                                                                          \
  case R:                                                                                                                   \
    hipLaunchKernelGGL((edge_field_kernel<T, R>), dim3((unsigned)blocks), dim3(kEfThreads), 0, st, in, h, w, tiles_r, tiles_c, \
                       wts, mask, out, rawmax, mn, mx);                                                                     \
    break;
  switch (radius) {
    EF_CASE(1) EF_CASE(2) EF_CASE(3) EF_CASE(4) EF_CASE(5) EF_CASE(6) EF_CASE(7) EF_CASE(8)
    default: pl_set_error("pl_scharr_gaussian: radius 1..8"); return PL_ERR_UNSUPPORTED;
  }
#undef EF_CASE
  return pl_check_launch("pl_scharr_gaussian");
}

}  // namespace

extern "C" int pl_scharr_gaussian(const void* in, int dtype, int64_t n, int h, int w, const double* d_weights, int radius,
                                  const uint8_t* d_mask, double* d_out, double* d_rawmax, double* d_min, double* d_max,
                                  void* stream) {
  PL_REQUIRE(in && d_weights && d_out && d_rawmax && d_min && d_max, "null pointer");
  PL_REQUIRE(n >= 0 && h > 0 && w > 0, "bad shape");
  PL_REQUIRE(radius >= 1 && radius <= 8, "radius 1..8 (sigma <= 2 at truncate 4)");
  PL_REQUIRE(dtype == PL_I16 || dtype == PL_U16, "int16 / uint16 slices");
  if (n == 0) return PL_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PL_I16)
    return ef_launch<short>((const short*)in, n, h, w, d_weights, radius, d_mask, d_out, d_rawmax, d_min, d_max, st);
  return ef_launch<unsigned short>((const unsigned short*)in, n, h, w, d_weights, radius, d_mask, d_out, d_rawmax, d_min, d_max, st);
}
