// Canny edge detection (SURVEY.md section 8 "next" row f2, first half).
//
// Replaces: skimage.feature.canny (scikit-image 0.18.3, skimage/feature/_canny.py; third-party, not under
// /root/reference) as called by pylinac/planar_imaging.py:574-588
//   (feature.canny(image, sigma, low_threshold, high_threshold, use_quantiles=True), then measure.label).
// Pipeline (mask=None): smoothed = G(image) / (G(ones) + eps) with G = ndimage.gaussian_filter(mode='constant')
// [pl_gaussian2d_mode, mode 2]; isobel / jsobel = ndimage.sobel [pl_sobel]; the kernels of this file:
//   canny_magnitude  hypot(isobel, jsobel)
//   canny_nms        the four 45-degree sectors with linear interpolation between the two neighbours on either
//                    side of the gradient direction, exactly the comparisons of _canny.py (c2*w + c1*(1-w) <= m),
//                    interior pixels only (binary_erosion of an all-ones mask with border_value 0) and
//                    magnitude > 0
//   order statistics of the float64 magnitude image for the quantile thresholds (np.percentile's neighbours,
//                    found by bisection on an order-preserving 64-bit key; the lerp is numpy's, on the host)
//   canny_masks      low / high = local_maxima & (magnitude >= threshold)
//   hysteresis       pl_label (8-connected) on the low mask, canny_flag marks labels that contain a high pixel,
//                    canny_select keeps them.
#include "pl_common.h"

namespace {

constexpr int kThreads = 256;

__global__ void canny_normalise_kernel(const double* __restrict__ g_img, const double* __restrict__ g_ones,
                                       int64_t per_frame, int64_t total, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  out[i] = g_img[i] / (g_ones[i % per_frame] + 2.220446049250313e-16);   // bleed_over + np.finfo(float).eps
}

__global__ void canny_magnitude_kernel(const double* __restrict__ isob, const double* __restrict__ jsob, int64_t total,
                                       double* __restrict__ mag) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  mag[i] = hypot(isob[i], jsob[i]);
}

// masked_image = image where the mask is set, 0 elsewhere; mask_f = the mask as float64 (smooth_with_function_and_mask,
// skimage/feature/_canny.py: both go through the same Gaussian).  The mask is one plane for the batch or one per frame.
__global__ void canny_mask_prepare_kernel(const double* __restrict__ img, const unsigned char* __restrict__ mask, int64_t per_frame,
                                          int64_t mask_stride, int64_t total, double* __restrict__ masked, double* __restrict__ mask_f) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const bool on = mask[(i / per_frame) * mask_stride + i % per_frame] != 0;
  masked[i] = on ? img[i] : 0.0;
  mask_f[i] = on ? 1.0 : 0.0;
}

// mask: NULL (every pixel counts: the eroded mask is the frame's interior) or uint8, one plane per frame (stride per_frame) or
// one for the batch (stride 0): eroded_mask = binary_erosion(mask, 3 x 3 ones, border_value=0) -- the pixel and its eight
// neighbours are all inside the frame and all set
__global__ void canny_nms_kernel(const double* __restrict__ isob, const double* __restrict__ jsob,
                                 const double* __restrict__ mag, int h, int w, int64_t total,
                                 const unsigned char* __restrict__ mask, int64_t mask_stride,
                                 unsigned char* __restrict__ local_max) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % w);
  const int r = (int)((i / w) % h);
  unsigned char res = 0;
  const double m = mag[i];
  bool inside = r > 0 && c > 0 && r < h - 1 && c < w - 1;
  if (inside && mask) {
    const unsigned char* mk = mask + (i / ((int64_t)h * w)) * mask_stride + (int64_t)r * w + c;
#pragma unroll
    for (int dr = -1; dr <= 1; ++dr)
#pragma unroll
      for (int dc = -1; dc <= 1; ++dc) inside = inside && mk[(int64_t)dr * w + dc] != 0;
  }
  if (inside && m > 0.0) {
    const double is = isob[i], js = jsob[i];
    const double ai = fabs(is), aj = fabs(js);
    auto M = [&](int dr, int dc) { return mag[i + (int64_t)dr * w + dc]; };
    // the sectors are tested in the reference's order; a later sector that also contains the pixel (ties on the
    // sector boundaries) overwrites the earlier result, like the successive local_maxima[pts] = ... assignments
    const bool same = (is >= 0 && js >= 0) || (is <= 0 && js <= 0);
    const bool opp = (is <= 0 && js >= 0) || (is >= 0 && js <= 0);
    if (same && ai >= aj) {            // 0 - 45 degrees
      const double wgt = aj / ai;
      const bool cp = M(1, 1) * wgt + M(1, 0) * (1 - wgt) <= m;
      const bool cm = M(-1, -1) * wgt + M(-1, 0) * (1 - wgt) <= m;
      res = cp && cm;
    }
    if (same && ai <= aj) {            // 45 - 90 degrees
      const double wgt = ai / aj;
      const bool cp = M(1, 1) * wgt + M(0, 1) * (1 - wgt) <= m;
      const bool cm = M(-1, -1) * wgt + M(0, -1) * (1 - wgt) <= m;
      res = cp && cm;
    }
    if (opp && ai <= aj) {             // 90 - 135 degrees
      const double wgt = ai / aj;
      const bool cp = M(-1, 1) * wgt + M(0, 1) * (1.0 - wgt) <= m;
      const bool cm = M(1, -1) * wgt + M(0, -1) * (1.0 - wgt) <= m;
      res = cp && cm;
    }
    if (opp && ai >= aj) {             // 135 - 180 degrees
      const double wgt = aj / ai;
      const bool cp = M(-1, 1) * wgt + M(-1, 0) * (1 - wgt) <= m;
      const bool cm = M(1, -1) * wgt + M(1, 0) * (1 - wgt) <= m;
      res = cp && cm;
    }
  }
  local_max[i] = res;
}

__device__ __forceinline__ unsigned long long key_of(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double value_of(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

// k-th smallest values (0-based ranks) of each frame: bisection on the key, one 1024-lane workgroup per (frame, rank)
__global__ void __launch_bounds__(1024)
order_stats_f64_kernel(const double* __restrict__ vals, int64_t count, const int64_t* __restrict__ ranks, int n_ranks,
                       double* __restrict__ out) {
  __shared__ unsigned long long s_cnt;
  const int64_t frame = blockIdx.x / n_ranks;
  const int q = blockIdx.x % n_ranks;
  const double* v = vals + frame * count;
  const unsigned long long need = (unsigned long long)ranks[q] + 1ull;
  unsigned long long lo = 0, hi = ~0ull;
  while (lo < hi) {
    const unsigned long long mid = lo + ((hi - lo) >> 1);
    unsigned long long c = 0;
    for (int64_t i = threadIdx.x; i < count; i += 1024) c += key_of(v[i]) <= mid ? 1 : 0;
    c = pl_wave_reduce(c, [](unsigned long long a, unsigned long long b) { return a + b; });
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) atomicAdd(&s_cnt, c);
    __syncthreads();
    const unsigned long long tot = s_cnt;
    __syncthreads();
    if (tot >= need) hi = mid; else lo = mid + 1;
  }
  if (threadIdx.x == 0) out[blockIdx.x] = value_of(lo);
}

__global__ void canny_masks_kernel(const unsigned char* __restrict__ local_max, const double* __restrict__ mag,
                                   const double* __restrict__ thr /*[n][2] low, high*/, int64_t per_frame,
                                   int64_t total, unsigned char* __restrict__ low, unsigned char* __restrict__ high) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const int64_t f = i / per_frame;
  const bool lm = local_max[i] != 0;
  low[i] = lm && (mag[i] >= thr[2 * f]);
  high[i] = lm && (mag[i] >= thr[2 * f + 1]);
}

__global__ void canny_flag_kernel(const int32_t* __restrict__ labels, const unsigned char* __restrict__ high,
                                  int64_t per_frame, int64_t total, int32_t* __restrict__ good /*per pixel slot*/) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const int32_t l = labels[i];
  if (l > 0 && high[i]) good[(i / per_frame) * per_frame + (l - 1)] = 1;   // labels <= pixels: slot l-1 of the frame
}

__global__ void canny_select_kernel(const int32_t* __restrict__ labels, const int32_t* __restrict__ good,
                                    int64_t per_frame, int64_t total, unsigned char* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const int32_t l = labels[i];
  out[i] = (l > 0 && good[(i / per_frame) * per_frame + (l - 1)]) ? 1 : 0;
}

// skimage.transform.hough_line (0.18.3, _hough_transform.pyx; behaviour pinned by probing the installed build): every
// non-zero pixel (x = column, y = row) votes, for every angle j, into
// accum[round(cos(theta_j) * x + sin(theta_j) * y) + offset][j], offset = ceil(sqrt(h^2 + w^2)), C round() (half away
// from zero); the accumulator has 2 * offset rows (scikit-image >= 0.19 has one more).  cos / sin tables come from the
// host (numpy's values).
__global__ void hough_line_kernel(const unsigned char* __restrict__ img, int h, int w, const double* __restrict__ ct,
                                  const double* __restrict__ st, int n_theta, int offset, int64_t total,
                                  unsigned long long* __restrict__ accum) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i % n_theta);
  const int64_t p = i / n_theta;
  if (!img[p]) return;
  const double x = (double)(p % w), y = (double)(p / w);
  const long long idx = (long long)round(ct[j] * x + st[j] * y) + offset;
  if (idx >= 0 && idx < 2ll * offset) atomicAdd(&accum[idx * n_theta + j], 1ull);
}

}  // namespace

extern "C" int pl_hough_line(const unsigned char* d_image, int h, int w, const double* d_cos, const double* d_sin,
                             int n_theta, unsigned long long* d_accum, void* stream) {
  PL_REQUIRE(d_image && d_cos && d_sin && d_accum, "null pointer");
  PL_REQUIRE(h > 0 && w > 0 && n_theta > 0, "bad shape");
  const int offset = (int)ceil(sqrt((double)h * h + (double)w * w));
  const int64_t total = (int64_t)h * w * n_theta;
  PL_REQUIRE(pl_cdiv(total, kThreads) <= 0x7fffffffLL, "image x angles too large for one launch");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(d_accum, 0, (size_t)(2 * offset) * n_theta * sizeof(unsigned long long), st) != hipSuccess) {
    pl_set_error("pl_hough_line: memset failed");
    return PL_ERR_HIP;
  }
  hipLaunchKernelGGL(hough_line_kernel, dim3((unsigned)pl_cdiv(total, kThreads)), dim3(kThreads), 0, st, d_image, h, w,
                     d_cos, d_sin, n_theta, offset, total, d_accum);
  return pl_check_launch("pl_hough_line");
}

extern "C" int pl_canny_normalise(const double* d_g_img, const double* d_g_ones, int64_t n, int64_t per_frame,
                                  double* d_out, void* stream) {
  PL_REQUIRE(d_g_img && d_g_ones && d_out, "null pointer");
  PL_REQUIRE(n >= 0 && per_frame > 0, "bad shape");
  const int64_t total = n * per_frame;
  if (total == 0) return PL_OK;
  hipLaunchKernelGGL(canny_normalise_kernel, dim3((unsigned)pl_cdiv(total, kThreads)), dim3(kThreads), 0,
                     (hipStream_t)stream, d_g_img, d_g_ones, per_frame, total, d_out);
  return pl_check_launch("pl_canny_normalise");
}

extern "C" int pl_canny_nms_masked(const double* d_isobel, const double* d_jsobel, int64_t n, int h, int w, const unsigned char* d_mask,
                                   int mask_per_frame, double* d_magnitude, unsigned char* d_local_max, void* stream);

extern "C" int pl_canny_nms(const double* d_isobel, const double* d_jsobel, int64_t n, int h, int w, double* d_magnitude,
                            unsigned char* d_local_max, void* stream) {
  return pl_canny_nms_masked(d_isobel, d_jsobel, n, h, w, nullptr, 0, d_magnitude, d_local_max, stream);
}

extern "C" int pl_canny_mask_prepare(const double* d_img, const unsigned char* d_mask, int mask_per_frame, int64_t n, int64_t per_frame,
                                     double* d_masked, double* d_mask_f, void* stream) {
  PL_REQUIRE(d_img && d_mask && d_masked && d_mask_f, "null pointer");
  PL_REQUIRE(n >= 0 && per_frame > 0, "bad shape");
  const int64_t total = n * per_frame;
  if (total == 0) return PL_OK;
  PL_REQUIRE(pl_cdiv(total, kThreads) <= 0x7fffffffLL, "batch too large for one launch");
  hipLaunchKernelGGL(canny_mask_prepare_kernel, dim3((unsigned)pl_cdiv(total, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, d_img,
                     d_mask, per_frame, mask_per_frame ? per_frame : (int64_t)0, total, d_masked, d_mask_f);
  return pl_check_launch("pl_canny_mask_prepare");
}

extern "C" int pl_canny_nms_masked(const double* d_isobel, const double* d_jsobel, int64_t n, int h, int w, const unsigned char* d_mask,
                                   int mask_per_frame, double* d_magnitude, unsigned char* d_local_max, void* stream) {
  PL_REQUIRE(d_isobel && d_jsobel && d_magnitude && d_local_max, "null pointer");
  PL_REQUIRE(n >= 0 && h > 0 && w > 0, "bad shape");
  const int64_t total = n * (int64_t)h * w;
  if (total == 0) return PL_OK;
  PL_REQUIRE(pl_cdiv(total, kThreads) <= 0x7fffffffLL, "batch too large for one launch");
  const unsigned blocks = (unsigned)pl_cdiv(total, kThreads);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(canny_magnitude_kernel, dim3(blocks), dim3(kThreads), 0, st, d_isobel, d_jsobel, total, d_magnitude);
  hipLaunchKernelGGL(canny_nms_kernel, dim3(blocks), dim3(kThreads), 0, st, d_isobel, d_jsobel, d_magnitude, h, w, total, d_mask,
                     mask_per_frame ? (int64_t)h * w : (int64_t)0, d_local_max);
  return pl_check_launch("pl_canny_nms");
}

extern "C" int pl_order_stats_f64(const double* d_values, int64_t n, int64_t count, const int64_t* d_ranks, int n_ranks,
                                  double* d_out, void* stream) {
  PL_REQUIRE(d_values && d_ranks && d_out, "null pointer");
  PL_REQUIRE(n >= 0 && count > 0 && n_ranks > 0 && n * n_ranks <= 0x7fffffffLL, "bad shape");
  if (n == 0) return PL_OK;
  hipLaunchKernelGGL(order_stats_f64_kernel, dim3((unsigned)(n * n_ranks)), dim3(1024), 0, (hipStream_t)stream, d_values,
                     count, d_ranks, n_ranks, d_out);
  return pl_check_launch("pl_order_stats_f64");
}

extern "C" int pl_canny_hysteresis(const unsigned char* d_local_max, const double* d_magnitude, const double* d_thresholds,
                                   int64_t n, int h, int w, unsigned char* d_low, unsigned char* d_high,
                                   const int32_t* d_labels, int32_t* d_good, unsigned char* d_out, int phase,
                                   void* stream) {
  PL_REQUIRE(n >= 0 && h > 0 && w > 0, "bad shape");
  const int64_t per = (int64_t)h * w, total = n * per;
  if (total == 0) return PL_OK;
  const unsigned blocks = (unsigned)pl_cdiv(total, kThreads);
  hipStream_t st = (hipStream_t)stream;
  if (phase == 0) {        // thresholds -> low / high masks (the caller labels d_low, 8-connected, then calls phase 1)
    PL_REQUIRE(d_local_max && d_magnitude && d_thresholds && d_low && d_high, "null pointer");
    hipLaunchKernelGGL(canny_masks_kernel, dim3(blocks), dim3(kThreads), 0, st, d_local_max, d_magnitude, d_thresholds,
                       per, total, d_low, d_high);
  } else {
    PL_REQUIRE(d_high && d_labels && d_good && d_out, "null pointer");
    if (hipMemsetAsync(d_good, 0, (size_t)total * sizeof(int32_t), st) != hipSuccess) { pl_set_error("pl_canny_hysteresis: memset failed"); return PL_ERR_HIP; }
    hipLaunchKernelGGL(canny_flag_kernel, dim3(blocks), dim3(kThreads), 0, st, d_labels, d_high, per, total, d_good);
    hipLaunchKernelGGL(canny_select_kernel, dim3(blocks), dim3(kThreads), 0, st, d_labels, d_good, per, total, d_out);
  }
  return pl_check_launch("pl_canny_hysteresis");
}
