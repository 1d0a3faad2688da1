// Device functions shared by the CatPhan edge-image kernels (edge_stream.hip) and the labelling kernel that consumes the
// float32 plane (slice_regions.hip): the exact Scharr magnitude from integer responses, and the exact value of one plane
// pixel recomputed by a whole wave.  pylinac/ct.py:391, 3327-3328 (skimage.filters.scharr + skimage.filters.gaussian).
#pragma once
// (included after pl_common.h by the .hip files that use it)

namespace {

__device__ __forceinline__ int es_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// skimage's Scharr magnitude sqrt(s0^2 + s1^2) / sqrt(2) from the integer responses S0 = 16 s0, S1 = 16 s1 (|S| <= 16 * 65535):
// o = (S0^2 + S1^2) / 256 is exact, so RN(sqrt(o)) = RN(sqrt(K)) / 16 with the integer K = S0^2 + S1^2 < 2^43, and the
// quotient by RN(sqrt(2)) becomes a quotient by 16 RN(sqrt(2)) (both scalings are powers of two).
//   * the square root is the compiler's own float64 expansion (v_rsq_f64 seed, Goldschmidt step, two fused residual
//     corrections: AMDGPULegalizerInfo::legalizeFSQRTF64) WITHOUT its range scaling (K is an integer: never below 2^-767)
//     and with the seed taken from max(K, 1), which runs K = 0 through the same chain to exactly 0 instead of a special case;
//   * the division by the constant is Markstein's three operations q0 = g r, rem = fma(-q0, c, g), q = fma(rem, r, q0) with
//     r = RN(1 / c).  Correct rounding for EVERY float64 g is proven by enumeration (tests/test_exact_sequences.py): the exact
//     value the last operation rounds lies within g / c * 4.001 * 2^-106 of g / c, only six mantissas g put g / c that close
//     to a rounding boundary, and all six round correctly.
__device__ __forceinline__ double es_edge_k(double x);
__device__ __forceinline__ double es_edge(int S0, int S1) {
  const double a = (double)S0, b = (double)S1;
  return es_edge_k(fma(a, a, b * b));                   // K, exact
}
// the same from K = S0^2 + S1^2 itself (an exact integer below 2^43)
__device__ __forceinline__ double es_edge_k(double x) {
  const long long xb = __double_as_longlong(x);
  const unsigned hi = max((unsigned)(xb >> 32), 0x3ff00000u);
  const double xs = __longlong_as_double((long long)(((unsigned long long)hi << 32) | (unsigned long long)(xb & 0xffffffffLL)));
  const double y = __builtin_amdgcn_rsq(xs);
  double g = x * y;
  double hh = y * 0.5;
  const double r0 = fma(-hh, g, 0.5);
  g = fma(g, r0, g);
  hh = fma(hh, r0, hh);
  const double d0 = fma(-g, g, x);
  g = fma(d0, hh, g);
  const double d1 = fma(-g, g, x);
  g = fma(d1, hh, g);                                    // RN(sqrt(K))
  constexpr double c16 = 0x1.6a09e667f3bcdp+4;           // 16 * 1.4142135623730951
  constexpr double r16 = 0x1.6a09e667f3bccp-5;           // RN(1 / 1.4142135623730951) / 16
  const double q0 = g * r16;
  const double rem = fma(-q0, c16, g);
  return fma(rem, r16, q0);                              // RN(RN(sqrt(K)) / 16 / RN(sqrt(2))) = np.sqrt(output) / np.sqrt(ndim)
}

// ---- exact value of ONE plane pixel, by a whole wave (the rare pixels a float32 plane cannot decide) ---------------------
// Every lane returns edges[r][c] as edge_stream_kernel computes it: the (2 rad + 1)^2 Scharr values at clamped coordinates
// (lanes share them out), axis 0 for the 2 rad + 1 columns, axis 1 on their results; `scratch` = (2 rad + 1)^2 + 2 rad + 1
// doubles of LDS private to the wave.  All 64 lanes must call it with the same arguments.
template <typename T>
__device__ double es_exact_wave(const T* __restrict__ src, int h, int w, int r, int c, const double* __restrict__ wts, int rad,
                                double* scratch) {
  const int lane = threadIdx.x & 63;
  const int win = 2 * rad + 1;
  auto px = [&](int rr, int cc) { return (int)src[(int64_t)es_clamp(rr, 0, h - 1) * w + es_clamp(cc, 0, w - 1)]; };
  for (int g = lane; g < win * win; g += PL_WAVE) {
    const int i = g / win, j = g - i * win;
    const int rr = es_clamp(r - rad + i, 0, h - 1), cc = es_clamp(c - rad + j, 0, w - 1);   // mode 'nearest'
    int S0 = 0, S1 = 0;
#pragma unroll
    for (int d = -1; d <= 1; ++d) {
      const int k = d == 0 ? 10 : 3;
      S0 += k * (px(rr + 1, cc + d) - px(rr - 1, cc + d));
      S1 += k * (px(rr + d, cc + 1) - px(rr + d, cc - 1));
    }
    scratch[g] = es_edge(S0, S1);
  }
  pl_wave_sync();
  if (lane < win) {
    double acc = scratch[rad * win + lane] * wts[rad];
    for (int k = rad; k >= 1; --k) acc = acc + (scratch[(rad - k) * win + lane] + scratch[(rad + k) * win + lane]) * wts[rad - k];
    scratch[win * win + lane] = acc;
  }
  pl_wave_sync();
  const double* v = scratch + win * win;
  double acc = v[rad] * wts[rad];
  for (int k = rad; k >= 1; --k) acc = acc + (v[rad - k] + v[rad + k]) * wts[rad - k];
  pl_wave_sync();                       // the scratch may be reused at once
  return acc;
}

constexpr int kEsMaxWin = 17;                            // radius 8
constexpr int kEsScratch = kEsMaxWin * kEsMaxWin + kEsMaxWin;

// the two float32 neighbours of a float32 value as float64: the float64 value it was rounded from lies between them
// (edge values are non-negative; 0 is exact: the smallest non-zero edge value is far above the float32 denormals)
__device__ __forceinline__ void es_f32_bracket(float v, double& vlo, double& vhi) {
  const unsigned b = __float_as_uint(v);
  vlo = b ? (double)__uint_as_float(b - 1u) : 0.0;
  vhi = b ? (double)__uint_as_float(b + 1u) : 0.0;
}
// the same with a radius of `n` bit patterns (n = 1: the plane holds RN32 of the exact value; the packed-float32 edge kernel's
// plane is within kEs32Bracket patterns of it, edge_stream32.hip).  A stored 0 is an exact 0 in both planes.
__device__ __forceinline__ void es_f32_bracket_n(float v, unsigned n, double& vlo, double& vhi) {
  const unsigned b = __float_as_uint(v);
  vlo = b > n ? (double)__uint_as_float(b - n) : 0.0;
  vhi = b ? (double)__uint_as_float(b + n) : 0.0;
}

}  // namespace
