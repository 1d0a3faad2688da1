// 1-D resampling of profiles.
//
// Replaces: scipy.interpolate.interp1d(x, y, kind="linear" | "cubic", bounds_error=False,
// fill_value="extrapolate") as called at pylinac/core/profile.py:1349-1358 (SingleProfile._interpolate:
// a detector profile of ~10^2-10^3 samples resampled to ~10x as many on linspace(x0-offset, xN+offset)).
//
// kind 0 (linear): scipy's _call_linear formula exactly -- hi = clip(searchsorted(x, xq, "left"), 1, L-1),
//   slope = (y_hi - y_lo) / (x_hi - x_lo), out = slope * (xq - x_lo) + y_lo; float64, no FMA: bit-identical.
// kind 1 (cubic): scipy builds make_interp_spline(x, y, k=3) = the not-a-knot interpolating cubic spline and
//   evaluates its B-spline form.  That spline is unique, so it is computed here in the classical form: second
//   derivatives M from the tridiagonal system with the two not-a-knot rows folded in (Thomas algorithm, one
//   lane per profile: L is small and the recurrence is sequential), then the piecewise cubic is evaluated
//   per query (end pieces extrapolate, as BSpline(extrapolate=True) does).  Agreement with scipy: ~1e-13
//   relative (different but equally stable arithmetic); tests state 1e-10.
#include "pl_common.h"

namespace {

constexpr int kThreads = 256;

// first index i in [0, n) with x[i] >= v (np.searchsorted side="left"), n if none
__device__ __forceinline__ int lower_bound(const double* __restrict__ x, int n, double v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (x[mid] < v) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

__global__ void interp_linear_kernel(const double* __restrict__ x, int64_t x_stride, const double* __restrict__ y,
                                     int L, const double* __restrict__ xq, int S, int64_t total,
                                     double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const int q = (int)(i % S);
  const int64_t p = i / S;
  const double* xs = x + p * x_stride;
  const double* ys = y + p * (int64_t)L;
  const double v = xq[q];
  int hi = lower_bound(xs, L, v);
  hi = hi < 1 ? 1 : (hi > L - 1 ? L - 1 : hi);
  const int lo = hi - 1;
  const double slope = (ys[hi] - ys[lo]) / (xs[hi] - xs[lo]);
  out[i] = slope * (v - xs[lo]) + ys[lo];
}

// second derivatives of the not-a-knot cubic spline; work: 2 L doubles per profile (c', d')
__global__ void spline_moments_kernel(const double* __restrict__ x, int64_t x_stride, const double* __restrict__ y,
                                      int L, int64_t n_profiles, double* __restrict__ M,
                                      double* __restrict__ work) {
  const int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (p >= n_profiles) return;
  const double* xs = x + p * x_stride;
  const double* ys = y + p * (int64_t)L;
  double* m = M + p * (int64_t)L;
  double* cp = work + p * 2 * (int64_t)L;
  double* dp = cp + L;
  const int n = L;
  auto h = [&](int i) { return xs[i + 1] - xs[i]; };
  auto rhs = [&](int i) { return 6.0 * ((ys[i + 1] - ys[i]) / h(i) - (ys[i] - ys[i - 1]) / h(i - 1)); };
  // unknowns M_1 .. M_{n-2}; row i: a_i M_{i-1} + b_i M_i + c_i M_{i+1} = r_i
  auto row = [&](int i, double& a, double& b, double& c, double& r) {
    a = h(i - 1);
    b = 2.0 * (h(i - 1) + h(i));
    c = h(i);
    r = rhs(i);
    if (i == 1) {  // M_0 = (1 + h0/h1) M_1 - (h0/h1) M_2
      const double t = h(0) / h(1);
      b += h(0) * (1.0 + t);
      c -= h(0) * t;
      a = 0.0;
    }
    if (i == n - 2) {  // M_{n-1} = (1 + h_{n-2}/h_{n-3}) M_{n-2} - (h_{n-2}/h_{n-3}) M_{n-3}
      const double t = h(n - 2) / h(n - 3);
      b += h(n - 2) * (1.0 + t);
      a -= h(n - 2) * t;
      c = 0.0;
    }
  };
  double a, b, c, r;
  row(1, a, b, c, r);
  cp[1] = c / b;
  dp[1] = r / b;
  for (int i = 2; i <= n - 2; ++i) {
    row(i, a, b, c, r);
    const double den = b - a * cp[i - 1];
    cp[i] = c / den;
    dp[i] = (r - a * dp[i - 1]) / den;
  }
  m[n - 2] = dp[n - 2];
  for (int i = n - 3; i >= 1; --i) m[i] = dp[i] - cp[i] * m[i + 1];
  {
    const double t0 = h(0) / h(1);
    m[0] = (1.0 + t0) * m[1] - t0 * m[2];
    const double t1 = h(n - 2) / h(n - 3);
    m[n - 1] = (1.0 + t1) * m[n - 2] - t1 * m[n - 3];
  }
}

__global__ void spline_eval_kernel(const double* __restrict__ x, int64_t x_stride, const double* __restrict__ y,
                                   const double* __restrict__ M, int L, const double* __restrict__ xq, int S,
                                   int64_t total, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const int q = (int)(i % S);
  const int64_t p = i / S;
  const double* xs = x + p * x_stride;
  const double* ys = y + p * (int64_t)L;
  const double* m = M + p * (int64_t)L;
  const double v = xq[q];
  int hi = lower_bound(xs, L, v);
  hi = hi < 1 ? 1 : (hi > L - 1 ? L - 1 : hi);
  const int lo = hi - 1;
  const double hh = xs[hi] - xs[lo];
  const double a = xs[hi] - v, b = v - xs[lo];
  out[i] = (m[lo] * a * a * a + m[hi] * b * b * b) / (6.0 * hh) + (ys[lo] / hh - m[lo] * hh / 6.0) * a +
           (ys[hi] / hh - m[hi] * hh / 6.0) * b;
}

// np.gradient(y) with unit spacing, edge_order 1: central differences inside, one-sided at the two ends
__global__ void gradient_kernel(const double* __restrict__ y, int L, int64_t total, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const int k = (int)(i % L);
  const double* p = y + (i - k);
  out[i] = k == 0 ? p[1] - p[0] : (k == L - 1 ? p[L - 1] - p[L - 2] : (p[k + 1] - p[k - 1]) / 2.0);
}

// ---- scipy.ndimage.zoom(values, zoom, order=3, mode="nearest", grid_mode) of 1-D profiles --------------------
// (ProfileBase.as_resampled, pylinac/core/profile.py:353-390; PhysicalProfileMixin.as_resampled :950-1011 with grid_mode).  scipy pads the input with 12 edge samples, runs the cubic
// B-spline prefilter (pole sqrt(3) - 2, mirror initialisation) over the padded array and evaluates the four-tap spline
// at i * (L - 1) / (S - 1) + 12 with clamped tap indices.  The prefilter is a sequential recursion: one lane per profile.
constexpr int kZoomPad = 12;

__global__ void zoom_prefilter_kernel(const double* __restrict__ y, int L, int64_t n_profiles, double* __restrict__ work) {
  const int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (p >= n_profiles) return;
  const double* v = y + p * (int64_t)L;
  const int n = L + 2 * kZoomPad;
  double* c = work + p * (int64_t)n;
  const double z = sqrt(3.0) - 2.0;
  const double gain = (1.0 - z) * (1.0 - 1.0 / z);
  for (int i = 0; i < n; ++i) {
    const int k = i - kZoomPad;
    c[i] = v[k < 0 ? 0 : (k > L - 1 ? L - 1 : k)] * gain;
  }
  double z_i = z;
  const double z_n_1 = pow(z, (double)(n - 1));
  double c0 = c[0] + z_n_1 * c[n - 1];
  for (int i = 1; i < n - 1; ++i) {
    c0 += z_i * (c[i] + z_n_1 * c[n - 1 - i]);
    z_i *= z;
  }
  c[0] = c0 / (1.0 - z_n_1 * z_n_1);
  for (int i = 1; i < n; ++i) c[i] += z * c[i - 1];
  c[n - 1] = (z * c[n - 2] + c[n - 1]) * z / (z * z - 1.0);
  for (int i = n - 2; i >= 0; --i) c[i] = z * (c[i + 1] - c[i]);
}

__global__ void zoom_eval_kernel(const double* __restrict__ work, int L, int S, int grid_mode, int64_t total,
                                 double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const int q = (int)(i % S);
  const int64_t p = i / S;
  const int n = L + 2 * kZoomPad;
  const double* c = work + p * (int64_t)n;
  // grid_mode: samples are cell centres (zoom = L / S, shift = zoom / 2 - 1 / 2); otherwise the end points coincide
  const double zoom = grid_mode ? (double)L / (double)S : (S > 1 ? (double)(L - 1) / (double)(S - 1) : 1.0);
  const double shift = grid_mode ? 0.5 * zoom - 0.5 : 0.0;
  const double cc = zoom * (double)q + shift + (double)kZoomPad;
  const double fl = floor(cc);
  const double yv = cc - fl, zv = 1.0 - yv;
  const double w1 = (yv * yv * (yv - 2.0) * 3.0 + 4.0) / 6.0;
  const double w2 = (zv * zv * (zv - 2.0) * 3.0 + 4.0) / 6.0;
  const double w0 = zv * zv * zv / 6.0;
  const double w3 = 1.0 - w0 - w1 - w2;
  const int start = (int)fl - 1;
  auto at = [&](int k) { return c[k < 0 ? 0 : (k > n - 1 ? n - 1 : k)]; };
  double t = 0.0;
  t += at(start) * w0;
  t += at(start + 1) * w1;
  t += at(start + 2) * w2;
  t += at(start + 3) * w3;
  out[i] = t;
}

}  // namespace

extern "C" int pl_zoom1d_cubic(const double* y, int64_t n_profiles, int length, int out_length, int grid_mode,
                               double* work, double* out, void* stream) {
  PL_REQUIRE(y && work && out, "null pointer");
  PL_REQUIRE(n_profiles >= 0 && length >= 2 && out_length >= 1, "bad shape");
  if (n_profiles == 0) return PL_OK;
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = n_profiles * (int64_t)out_length;
  PL_REQUIRE(pl_cdiv(total, kThreads) <= 0x7fffffffLL, "batch too large for one launch");
  hipLaunchKernelGGL(zoom_prefilter_kernel, dim3((unsigned)pl_cdiv(n_profiles, kThreads)), dim3(kThreads), 0, st, y,
                     length, n_profiles, work);
  hipLaunchKernelGGL(zoom_eval_kernel, dim3((unsigned)pl_cdiv(total, kThreads)), dim3(kThreads), 0, st, work, length,
                     out_length, grid_mode ? 1 : 0, total, out);
  return pl_check_launch("pl_zoom1d_cubic");
}

extern "C" int pl_gradient1d(const double* y, int64_t n_profiles, int length, double* out, void* stream) {
  PL_REQUIRE(y && out, "null pointer");
  PL_REQUIRE(n_profiles >= 0 && length >= 2, "np.gradient needs at least 2 samples");
  if (n_profiles == 0) return PL_OK;
  const int64_t total = n_profiles * (int64_t)length;
  PL_REQUIRE(pl_cdiv(total, kThreads) <= 0x7fffffffLL, "batch too large for one launch");
  hipLaunchKernelGGL(gradient_kernel, dim3((unsigned)pl_cdiv(total, kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                     y, length, total, out);
  return pl_check_launch("pl_gradient1d");
}

extern "C" int pl_interp1d(const double* x, int64_t x_stride, const double* y, int64_t n_profiles, int length,
                           const double* xq, int n_query, int kind, double* work, double* out, void* stream) {
  PL_REQUIRE(x && y && xq && out, "null pointer");
  PL_REQUIRE(n_profiles >= 0 && n_query >= 0, "bad shape");
  PL_REQUIRE(kind == 0 || kind == 1, "kind must be 0 (linear) or 1 (cubic)");
  PL_REQUIRE(x_stride == 0 || x_stride >= length, "x_stride must be 0 (shared abscissae) or >= length");
  if (kind == 0) PL_REQUIRE(length >= 2, "linear interpolation needs >= 2 samples");
  if (kind == 1) {
    // scipy: "The number of derivatives at boundaries does not match" below k+1 points
    PL_REQUIRE(length >= 4, "cubic interpolation needs >= 4 samples");
    PL_REQUIRE(work, "cubic interpolation needs a workspace of 3 * n_profiles * length doubles");
  }
  if (n_profiles == 0 || n_query == 0) return PL_OK;
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = n_profiles * (int64_t)n_query;
  PL_REQUIRE(pl_cdiv(total, kThreads) <= 0x7fffffffLL, "batch too large for one launch");
  if (kind == 0) {
    hipLaunchKernelGGL(interp_linear_kernel, dim3((unsigned)pl_cdiv(total, kThreads)), dim3(kThreads), 0, st, x,
                       x_stride, y, length, xq, n_query, total, out);
  } else {
    double* M = work;
    double* scratch = work + n_profiles * (int64_t)length;
    hipLaunchKernelGGL(spline_moments_kernel, dim3((unsigned)pl_cdiv(n_profiles, kThreads)), dim3(kThreads), 0, st,
                       x, x_stride, y, length, n_profiles, M, scratch);
    hipLaunchKernelGGL(spline_eval_kernel, dim3((unsigned)pl_cdiv(total, kThreads)), dim3(kThreads), 0, st, x,
                       x_stride, y, M, length, xq, n_query, total, out);
  }
  return pl_check_launch("pl_interp1d");
}
