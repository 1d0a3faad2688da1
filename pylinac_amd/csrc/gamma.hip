// 2-D gamma index (SURVEY.md section 8 "next" row f4, gamma part).
//
// Replaces: pylinac.core.gamma.gamma_2d (pylinac/core/gamma.py:229-330) -- a Python double loop over every
// reference pixel that gathers an evaluation disk of radius DTA + 1 (skimage.draw.disk) and takes
//   Gamma^2 = nanmin_k ( dist2[k] + (eval_n[r + dr_k, c + dc_k] - ref_n[r, c])^2 ),
// with both images divided by the dose-to-agreement first (global: pct/100 * reference.max(); local:
// pct/100 * reference, elementwise -- the evaluation image is divided by the REFERENCE pixel at its own
// position, as the reference writes it), the evaluation edge-padded by DTA, pixels with NaN or
// ref_n < threshold/100 set to fill_value, Gamma capped.  Same float64 operations in the same order, IEEE
// sqrt: bit-identical.  One lane per reference pixel, the disk offsets and their squared distances come
// from the host (a few dozen entries, computed with skimage's own formula).
#include "pl_common.h"

namespace {

constexpr int kThreads = 256;

// out = a / (scale * b)   (b == nullptr: out = a / scale)
__global__ void gamma_normalize_kernel(const double* __restrict__ a, const double* __restrict__ b, double scale,
                                       const double* __restrict__ frame_scale, int64_t per_frame, int64_t total,
                                       double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const double dose_ta = b ? scale * b[i] : scale * frame_scale[i / per_frame];
  out[i] = a[i] / dose_ta;
}

__global__ void gamma2d_kernel(const double* __restrict__ ref_n, const double* __restrict__ ev_n, int h, int w,
                               const int* __restrict__ dr, const int* __restrict__ dc,
                               const double* __restrict__ dist2, int k, double thr, double cap, double fill,
                               int64_t total, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % w);
  const int64_t t = i / w;
  const int r = (int)(t % h);
  const double* ev = ev_n + (t / h) * (int64_t)h * w;
  const double rp = ref_n[i];
  if (rp != rp || rp < thr) {            // math.isnan(ref_point) or ref_point < threshold_normalized
    out[i] = fill;
    return;
  }
  double best = __longlong_as_double(0x7ff8000000000000LL);   // nanmin of nothing but NaNs is NaN
  bool any = false;
  for (int q = 0; q < k; ++q) {
    int rr = r + dr[q], cc = c + dc[q];   // np.pad(mode="edge"): clamp
    rr = rr < 0 ? 0 : (rr >= h ? h - 1 : rr);
    cc = cc < 0 ? 0 : (cc >= w ? w - 1 : cc);
    const double d = ev[(int64_t)rr * w + cc] - rp;
    const double v = dist2[q] + d * d;
    if (v == v && (!any || v < best)) { best = v; any = true; }
  }
  const double cap2 = cap * cap;
  if (best >= cap2) { out[i] = cap; return; }   // NaN compares false: falls through to sqrt(NaN) = NaN
  out[i] = sqrt(best);
}

// gamma_1d (pylinac/core/gamma.py:333-455): one lane per reference point; the search samples are
// np.linspace(x - DTA, x + DTA, num) (start + k*step, last sample forced to the stop value), the evaluation
// profile is read through scipy's linear interp1d (slope form, extrapolating), Gamma = sqrt(dist^2 / DTA^2 +
// dose^2 / dose_ta^2) in Python's operation order, minimum taken sequentially like Python's min().  Agreement:
// sample positions and values bit-identical; gamma within 2 ulp (the reference squares with float ** 2 = libm
// pow, not always the correctly rounded product x*x).
__global__ void gamma1d_kernel(const double* __restrict__ ref, const double* __restrict__ ref_x, int n_ref,
                               const double* __restrict__ ev, const double* __restrict__ ev_x, int n_ev, double dta,
                               double dta2, int num, double threshold, double dose_ta_global, double dose_fraction,
                               int global_dose, double cap, double fill, double* __restrict__ gamma,
                               double* __restrict__ eval_vals, double* __restrict__ eval_xs, int* __restrict__ computed) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n_ref) return;
  const double x = ref_x[i], v = ref[i];
  if (v < threshold) {
    gamma[i] = fill;
    computed[i] = 0;
    return;
  }
  computed[i] = 1;
  const double start = x - dta, stop = x + dta;
  const double step = (stop - start) / (double)(num - 1);
  const double dose_ta = global_dose ? dose_ta_global : dose_fraction * v;
  double best = 0.0;
  for (int k = 0; k < num; ++k) {
    const double ex = (k == num - 1) ? stop : ((double)k * step + start);
    // scipy interp1d._call_linear with extrapolation
    int lo = 0, hi = n_ev;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (ev_x[mid] < ex) lo = mid + 1; else hi = mid;
    }
    int h1 = lo < 1 ? 1 : (lo > n_ev - 1 ? n_ev - 1 : lo);
    const int l1 = h1 - 1;
    const double slope = (ev[h1] - ev[l1]) / (ev_x[h1] - ev_x[l1]);
    const double evv = slope * (ex - ev_x[l1]) + ev[l1];
    eval_xs[(int64_t)i * num + k] = ex;
    eval_vals[(int64_t)i * num + k] = evv;
    const double dist = fabs(x - ex);
    const double dose = v - evv;
    const double cg = sqrt(dist * dist / dta2 + dose * dose / (dose_ta * dose_ta));
    if (k == 0 || cg < best) best = cg;
  }
  gamma[i] = cap < best ? cap : best;
}

// pylinac/core/gamma.py:105-227, gamma_geometric (Ju et al. 2008: distance from each reference point to the piecewise-linear
// evaluation curve in (x / DTA, dose / dose-to-agreement) space), one lane per reference point:
//   nr = ref * 100 / (max(ref) * dose_ta)      ne = ev * 100 / (max(ref) * dose_ta)      nrx = ref_x / dta      nex = ev_x / dta
//   vertices: from  max(argmin|nex - (nrx - dta)| - 1, 0)  to  min(argmin|nex - (nrx + dta)| + 1, m - 1)   (first arg-min; the
//   reference subtracts the UN-normalised dta from normalised positions, restated as is; the two searches swap for a
//   decreasing nex), and over consecutive vertex pairs (v1, v2) the distance of p = (nrx, nr) to the segment:
//     V = v1 - v2, P = p - v2, w = pinv([V.V]) * (V.P)   (1 x 1 pseudo-inverse: the reciprocal, 0 for a zero matrix),
//     w < 0 or 1 - w < 0 -> the nearer vertex;  else |p - (w v1 + (1 - w) v2)|
//   gamma = min(min over pairs, cap); points with nr < dose_threshold / dose_ta keep `fill`.
// Float64 in the reference's operation order; BLAS dot products / math.dist may differ in the last bit.
__global__ void gamma_geometric_kernel(const double* __restrict__ ref, const double* __restrict__ ref_x, int n_ref,
                                       const double* __restrict__ ev, const double* __restrict__ ev_x, int n_ev, double denom,
                                       double dta, double threshold, int decreasing, double cap, double fill,
                                       double* __restrict__ gamma) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n_ref) return;
  const double py = ref[i] * 100.0 / denom;
  const double px = ref_x[i] / dta;
  if (py < threshold) {
    gamma[i] = fill;
    return;
  }
  const double lo_t = px - dta, hi_t = px + dta;
  int a_lo = 0, a_hi = 0;
  double b_lo = 0.0, b_hi = 0.0;
  for (int j = 0; j < n_ev; ++j) {
    const double x = ev_x[j] / dta;
    const double dl = fabs(x - lo_t), dh = fabs(x - hi_t);
    if (j == 0 || dl < b_lo) { b_lo = dl; a_lo = j; }
    if (j == 0 || dh < b_hi) { b_hi = dh; a_hi = j; }
  }
  if (decreasing) { const int t = a_lo; a_lo = a_hi; a_hi = t; }
  const int left = a_lo - 1 > 0 ? a_lo - 1 : 0;
  const int right = a_hi + 1 < n_ev - 1 ? a_hi + 1 : n_ev - 1;
  double best = 0.0;
  bool any = false;
  for (int j = left; j < right; ++j) {
    const double v1x = ev_x[j] / dta, v1y = ev[j] * 100.0 / denom;
    const double v2x = ev_x[j + 1] / dta, v2y = ev[j + 1] * 100.0 / denom;
    const double Vx = v1x - v2x, Vy = v1y - v2y, Px = px - v2x, Py = py - v2y;
    const double vtv = Vx * Vx + Vy * Vy;
    const double inv = vtv != 0.0 ? 1.0 / vtv : 0.0;
    const double w = inv * (Vx * Px + Vy * Py);
    const double w2 = 1.0 - w;
    double d;
    if (w < 0.0 || w2 < 0.0) {
      const double d1 = sqrt((px - v1x) * (px - v1x) + (py - v1y) * (py - v1y));
      const double d2 = sqrt(Px * Px + Py * Py);
      d = d2 < d1 ? d2 : d1;
    } else {
      const double qx = px - (w * v1x + w2 * v2x), qy = py - (w * v1y + w2 * v2y);
      d = sqrt(qx * qx + qy * qy);
    }
    if (!any || d < best) best = d;
    any = true;
  }
  // (no pair: n_ev >= 2 makes right > left always)
  gamma[i] = cap < best ? cap : best;
}

// BaseImage.gamma (Bakai et al. eq. 6; pylinac/core/image.py:994-1016), in numpy's own mixed precision:
//   ref[ref < threshold * max(ref)] = nan                          (float64)
//   img_x, img_y = sobel(ref.astype(float32), 1 / 0)               (float32: pl_sobel on the array written here)
//   grad = hypot(img_x, img_y)                                     (float32: hypotf = sqrt in double, rounded)
//   denominator = sqrt((doseTA/100)**2 + (distTA_px**2) * grad**2) (float32: the Python scalars are weak, NEP 50)
//   gamma = abs(comp - ref) / denominator                          (float64 / float32 -> float64)
__global__ void bakai_mask_kernel(const double* __restrict__ ref, const double* __restrict__ frame_cut, int64_t per_frame,
                                  int64_t total, double* __restrict__ ref_masked, float* __restrict__ ref32) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  double v = ref[i];
  if (v < frame_cut[i / per_frame]) v = __longlong_as_double(0x7ff8000000000000LL);
  ref_masked[i] = v;
  ref32[i] = (float)v;
}

__global__ void bakai_gamma_kernel(const double* __restrict__ ref_masked, const double* __restrict__ comp,
                                   const float* __restrict__ gx, const float* __restrict__ gy, float dose2, float dist2,
                                   int64_t total, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const float grad = (float)sqrt((double)gx[i] * (double)gx[i] + (double)gy[i] * (double)gy[i]);   // hypotf
  const float g2 = grad * grad;
  const float den = sqrtf(dose2 + dist2 * g2);
  out[i] = fabs(comp[i] - ref_masked[i]) / (double)den;
}

}  // namespace

extern "C" int pl_bakai_mask(const double* d_ref, const double* d_frame_cut, int64_t n, int64_t per_frame,
                             double* d_ref_masked, float* d_ref32, void* stream) {
  PL_REQUIRE(d_ref && d_frame_cut && d_ref_masked && d_ref32, "null pointer");
  PL_REQUIRE(n >= 0 && per_frame > 0, "bad shape");
  const int64_t total = n * per_frame;
  if (total == 0) return PL_OK;
  PL_REQUIRE(pl_cdiv(total, kThreads) <= 0x7fffffffLL, "batch too large for one launch");
  hipLaunchKernelGGL(bakai_mask_kernel, dim3((unsigned)pl_cdiv(total, kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                     d_ref, d_frame_cut, per_frame, total, d_ref_masked, d_ref32);
  return pl_check_launch("pl_bakai_mask");
}

extern "C" int pl_bakai_gamma(const double* d_ref_masked, const double* d_comp, const float* d_grad_x,
                              const float* d_grad_y, float dose_term, float dist_term, int64_t total, double* d_out,
                              void* stream) {
  PL_REQUIRE(d_ref_masked && d_comp && d_grad_x && d_grad_y && d_out, "null pointer");
  PL_REQUIRE(total >= 0, "bad shape");
  if (total == 0) return PL_OK;
  PL_REQUIRE(pl_cdiv(total, kThreads) <= 0x7fffffffLL, "batch too large for one launch");
  hipLaunchKernelGGL(bakai_gamma_kernel, dim3((unsigned)pl_cdiv(total, kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                     d_ref_masked, d_comp, d_grad_x, d_grad_y, dose_term, dist_term, total, d_out);
  return pl_check_launch("pl_bakai_gamma");
}

extern "C" int pl_gamma1d(const double* d_ref, const double* d_ref_x, int n_ref, const double* d_eval,
                          const double* d_eval_x, int n_eval, double distance_to_agreement, double dta_squared,
                          int n_samples, double threshold, double dose_ta_global, double dose_fraction,
                          int global_dose, double gamma_cap, double fill_value, double* d_gamma, double* d_eval_vals,
                          double* d_eval_xs, int32_t* d_computed, void* stream) {
  PL_REQUIRE(d_ref && d_ref_x && d_eval && d_eval_x && d_gamma && d_eval_vals && d_eval_xs && d_computed, "null pointer");
  PL_REQUIRE(n_ref >= 0 && n_eval >= 2 && n_samples >= 2, "bad shape");
  if (n_ref == 0) return PL_OK;
  hipLaunchKernelGGL(gamma1d_kernel, dim3((unsigned)pl_cdiv(n_ref, kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                     d_ref, d_ref_x, n_ref, d_eval, d_eval_x, n_eval, distance_to_agreement, dta_squared, n_samples,
                     threshold, dose_ta_global, dose_fraction, global_dose, gamma_cap, fill_value, d_gamma,
                     d_eval_vals, d_eval_xs, d_computed);
  return pl_check_launch("pl_gamma1d");
}

extern "C" int pl_gamma_geometric(const double* d_ref, const double* d_ref_x, int n_ref, const double* d_eval,
                                  const double* d_eval_x, int n_eval, double dose_denominator, double distance_to_agreement,
                                  double threshold_normalized, int eval_x_decreasing, double gamma_cap, double fill_value,
                                  double* d_gamma, void* stream) {
  PL_REQUIRE(d_ref && d_ref_x && d_eval && d_eval_x && d_gamma, "null pointer");
  PL_REQUIRE(n_ref >= 0 && n_eval >= 2, "bad shape (the evaluation profile needs two samples)");
  PL_REQUIRE(distance_to_agreement > 0.0, "distance to agreement must be greater than 0");
  if (n_ref == 0) return PL_OK;
  hipLaunchKernelGGL(gamma_geometric_kernel, dim3((unsigned)pl_cdiv(n_ref, kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                     d_ref, d_ref_x, n_ref, d_eval, d_eval_x, n_eval, dose_denominator, distance_to_agreement,
                     threshold_normalized, eval_x_decreasing, gamma_cap, fill_value, d_gamma);
  return pl_check_launch("pl_gamma_geometric");
}

extern "C" int pl_gamma2d(const double* d_reference, const double* d_evaluation, int64_t n, int h, int w,
                          double dose_fraction, int global_dose, const double* d_ref_max, const int32_t* d_dr,
                          const int32_t* d_dc, const double* d_dist2, int n_offsets, double threshold_normalized,
                          double gamma_cap, double fill_value, double* d_work, double* d_out, void* stream) {
  PL_REQUIRE(d_reference && d_evaluation && d_dr && d_dc && d_dist2 && d_work && d_out, "null pointer");
  PL_REQUIRE(n >= 0 && h > 0 && w > 0 && n_offsets > 0, "bad shape");
  PL_REQUIRE(!global_dose || d_ref_max, "global dose needs the per-frame reference maxima");
  if (n == 0) return PL_OK;
  const int64_t per = (int64_t)h * w, total = n * per;
  PL_REQUIRE(pl_cdiv(total, kThreads) <= 0x7fffffffLL, "batch too large for one launch");
  hipStream_t st = (hipStream_t)stream;
  double* ref_n = d_work;
  double* ev_n = d_work + total;
  const unsigned blocks = (unsigned)pl_cdiv(total, kThreads);
  const double* local = global_dose ? nullptr : d_reference;
  hipLaunchKernelGGL(gamma_normalize_kernel, dim3(blocks), dim3(kThreads), 0, st, d_reference, local, dose_fraction,
                     d_ref_max, per, total, ref_n);
  hipLaunchKernelGGL(gamma_normalize_kernel, dim3(blocks), dim3(kThreads), 0, st, d_evaluation, local, dose_fraction,
                     d_ref_max, per, total, ev_n);
  hipLaunchKernelGGL(gamma2d_kernel, dim3(blocks), dim3(kThreads), 0, st, ref_n, ev_n, h, w, d_dr, d_dc, d_dist2,
                     n_offsets, threshold_normalized, gamma_cap, fill_value, total, d_out);
  return pl_check_launch("pl_gamma2d");
}
