// Noise power spectrum, radial average and ESF-FFT MTF.
//
// Replaces (reference file:line):
//   pylinac/core/nps.py:35-79     noise_power_spectrum_2d  -> pl_nps2d
//   pylinac/core/nps.py:12-32     radial_average           -> pl_radial_average
//   pylinac/core/mtf.py:448-456   _compute_esf_mtf         -> pl_esf_mtf
//
// The reference calls numpy's pocketfft on small inputs (ROIs of 30-300 px, 1-D signals <= a few hundred
// samples zero-padded to >= 1024): the transform sizes are arbitrary (not powers of two) and the work is
// O(10^7) multiply-adds, so these are plain DFTs with exactly reduced twiddle indices ((k*m) mod n kept
// as an integer, sincospi of the reduced fraction) -- float64, agreement with pocketfft ~1e-13 relative
// (different summation order; tests state 1e-9 of the spectrum maximum).  No MFMA: nothing here is large
// enough to matter (SURVEY: "negligible; not a custom kernel priority").
#include "pl_common.h"

namespace {

constexpr int kThreads = 256;

// T[k] = exp(-2 pi i k / L), k = 0 .. L-1
__global__ void twiddle_kernel(double* __restrict__ tw, int L) {
  const int k = blockIdx.x * kThreads + threadIdx.x;
  if (k >= L) return;
  double s, c;
  sincospi(2.0 * (double)k / (double)L, &s, &c);
  tw[2 * k] = c;
  tw[2 * k + 1] = -s;
}

// per-ROI mean (np.mean over the cropped L x L block; float64 accumulation)
__global__ void roi_mean_kernel(const double* __restrict__ rois, int L, int64_t roi_stride, int row_stride,
                                double* __restrict__ means) {
  const double* a = rois + blockIdx.x * roi_stride;
  double acc = 0.0;
  for (int i = threadIdx.x; i < L * L; i += kThreads) acc += a[(size_t)(i / L) * row_stride + (i % L)];
  __shared__ double red[kThreads / PL_WAVE];
  acc = pl_wave_reduce(acc, [](double x, double y) { return x + y; });
  if ((threadIdx.x & (PL_WAVE - 1)) == 0) red[threadIdx.x / PL_WAVE] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < kThreads / PL_WAVE; ++i) t += red[i];
    means[blockIdx.x] = t / ((double)L * (double)L);
  }
}

// G[r][y][v] = sum_x (a[r][y][x] - mean_r) * T[(v x) mod L]
__global__ void row_dft_kernel(const double* __restrict__ rois, int L, int64_t roi_stride, int row_stride,
                               const double* __restrict__ means, const double* __restrict__ tw,
                               double* __restrict__ G, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const int v = (int)(i % L);
  const int64_t t = i / L;
  const int y = (int)(t % L);
  const int64_t r = t / L;
  const double* row = rois + r * roi_stride + (size_t)y * row_stride;
  const double mean = means[r];
  double re = 0.0, im = 0.0;
  int idx = 0;
  for (int x = 0; x < L; ++x) {
    const double a = row[x] - mean;
    re += a * tw[2 * idx];
    im += a * tw[2 * idx + 1];
    idx += v;
    if (idx >= L) idx -= L;
  }
  G[2 * i] = re;
  G[2 * i + 1] = im;
}

// out[fftshift(u), fftshift(v)] = scale * mean_r |sum_y G[r][y][v] T[(u y) mod L]|^2
__global__ void col_dft_power_kernel(const double* __restrict__ G, int L, int R, const double* __restrict__ tw,
                                     double scale, double* __restrict__ out) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= L * L) return;
  const int v = i % L, u = i / L;
  double acc = 0.0;
  for (int r = 0; r < R; ++r) {
    const double* g = G + (size_t)r * L * L * 2;
    double re = 0.0, im = 0.0;
    int idx = 0;
    for (int y = 0; y < L; ++y) {
      const double gr = g[2 * ((size_t)y * L + v)], gi = g[2 * ((size_t)y * L + v) + 1];
      const double tr = tw[2 * idx], ti = tw[2 * idx + 1];
      re += gr * tr - gi * ti;
      im += gr * ti + gi * tr;
      idx += u;
      if (idx >= L) idx -= L;
    }
    const double mag = hypot(re, im);  // np.abs(complex) then ** 2, as the reference writes it
    acc += mag * mag;
  }
  // np.fft.fftshift: shifted[j] = b[(j - L/2) mod L]  <=>  b[k] lands at (k + L/2) mod L
  const int us = (u + L / 2) % L, vs = (v + L / 2) % L;
  out[(size_t)us * L + vs] = scale * (acc / (double)R);
}

// floor(sqrt(d)) for an exact non-negative integer d
__device__ __forceinline__ int isqrt_floor(long long d) {
  long long b = (long long)sqrt((double)d);
  while (b * b > d) --b;
  while ((b + 1) * (b + 1) <= d) ++b;
  return (int)b;
}

// One lane per radius bin; it visits exactly the pixels of its ring, in raster order (the order in which
// np.bincount accumulates), so sums and counts are numpy's bit for bit.
__global__ void radial_average_kernel(const double* __restrict__ arr, int h, int w, int nbins,
                                      double* __restrict__ out) {
  const int b = blockIdx.x * kThreads + threadIdx.x;
  if (b >= nbins) return;
  const int cy = h / 2, cx = w / 2;  // np.floor(shape / 2)
  const long long lo = (long long)b * b, hi = (long long)(b + 1) * (b + 1);  // lo <= d < hi
  double sum = 0.0;
  long long cnt = 0;
  for (int y = 0; y < h; ++y) {
    const long long dy2 = (long long)(y - cy) * (y - cy);
    if (dy2 >= hi) continue;
    // |dx| in [a0, a1]: a0 = ceil(sqrt(max(lo - dy2, 0))), a1 = floor(sqrt(hi - 1 - dy2))
    const long long need = lo - dy2;
    int a0 = 0;
    if (need > 0) {
      a0 = isqrt_floor(need);
      if ((long long)a0 * a0 < need) ++a0;
    }
    const int a1 = isqrt_floor(hi - 1 - dy2);
    if (a0 > a1) continue;
    const double* row = arr + (size_t)y * w;
    // left interval: x = cx - a1 .. cx - a0 ; right interval: x = cx + max(a0, 1) .. cx + a1
    for (int x = max(cx - a1, 0); x <= min(cx - a0, w - 1); ++x) {
      sum += row[x];
      ++cnt;
    }
    for (int x = max(cx + (a0 > 0 ? a0 : 1), 0); x <= min(cx + a1, w - 1); ++x) {
      sum += row[x];
      ++cnt;
    }
  }
  out[b] = cnt ? sum / (double)cnt : 0.0;
}

// |fft(gradient(esf) * window, n)|[k] for k < n/2, one lane per (esf, k); X[0] magnitude kept for the
// normalisation.  np.gradient: central differences inside, one-sided first differences at the two ends.
__global__ void esf_dft_kernel(const double* __restrict__ esf, const int* __restrict__ lens,
                               const double* __restrict__ window, int E, int lmax, int n,
                               double* __restrict__ mag) {
  const int half = n / 2;
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= E * half) return;
  const int k = i % half, e = i / half;
  const int len = lens[e];
  const double* s = esf + (size_t)e * lmax;
  const double* wv = window + (size_t)e * lmax;
  double re = 0.0, im = 0.0;
  int idx = 0;
  for (int m = 0; m < len; ++m) {
    double g;
    if (m == 0) g = s[1] - s[0];
    else if (m == len - 1) g = s[len - 1] - s[len - 2];
    else g = (s[m + 1] - s[m - 1]) / 2.0;
    const double a = g * wv[m];
    double sn, cs;
    sincospi(2.0 * (double)idx / (double)n, &sn, &cs);
    re += a * cs;
    im -= a * sn;
    idx += k;
    if (idx >= n) idx -= n;
  }
  mag[i] = hypot(re, im);
}

// mtf_each[e][k] = mag[e][k] / mag[e][0]; mtf_mean[k] = mean over e (np.mean over axis 0: sequential sum)
__global__ void esf_normalise_kernel(const double* __restrict__ mag, int E, int half, double* __restrict__ each,
                                     double* __restrict__ mean) {
  const int k = blockIdx.x * kThreads + threadIdx.x;
  if (k >= half) return;
  double acc = 0.0;
  for (int e = 0; e < E; ++e) {
    const double v = mag[(size_t)e * half + k] / mag[(size_t)e * half];
    each[(size_t)e * half + k] = v;
    acc += v;
  }
  mean[k] = acc / (double)E;
}

}  // namespace

extern "C" int pl_nps2d(const double* rois, int64_t n_rois, int length, int64_t roi_stride, int row_stride,
                        double pixel_size, double* work, double* out, void* stream) {
  PL_REQUIRE(rois && work && out, "null pointer");
  PL_REQUIRE(n_rois > 0 && length > 0, "bad shape");
  PL_REQUIRE(row_stride >= length && roi_stride >= (int64_t)row_stride * (length - 1) + length, "bad strides");
  hipStream_t st = (hipStream_t)stream;
  const int L = length;
  double* tw = work;                              // 2 L
  double* means = tw + 2 * (size_t)L;             // n_rois
  double* G = means + n_rois;                     // 2 n_rois L L
  const int64_t total = n_rois * (int64_t)L * L;
  PL_REQUIRE(pl_cdiv(total, kThreads) <= 0x7fffffffLL, "batch too large for one launch");
  hipLaunchKernelGGL(twiddle_kernel, dim3((unsigned)pl_cdiv(L, kThreads)), dim3(kThreads), 0, st, tw, L);
  hipLaunchKernelGGL(roi_mean_kernel, dim3((unsigned)n_rois), dim3(kThreads), 0, st, rois, L, roi_stride,
                     row_stride, means);
  hipLaunchKernelGGL(row_dft_kernel, dim3((unsigned)pl_cdiv(total, kThreads)), dim3(kThreads), 0, st, rois, L,
                     roi_stride, row_stride, means, tw, G, total);
  const double scale = pixel_size * pixel_size / ((double)L * (double)L);
  hipLaunchKernelGGL(col_dft_power_kernel, dim3((unsigned)pl_cdiv((int64_t)L * L, kThreads)), dim3(kThreads), 0,
                     st, G, L, (int)n_rois, tw, scale, out);
  return pl_check_launch("pl_nps2d");
}

extern "C" int64_t pl_nps2d_work_doubles(int64_t n_rois, int length) {
  return 2 * (int64_t)length + n_rois + 2 * n_rois * (int64_t)length * length;
}

extern "C" int pl_radial_average(const double* arr, int h, int w, int nbins, double* out, void* stream) {
  PL_REQUIRE(arr && out, "null pointer");
  PL_REQUIRE(h > 0 && w > 0 && nbins > 0, "bad shape");
  hipLaunchKernelGGL(radial_average_kernel, dim3((unsigned)pl_cdiv(nbins, kThreads)), dim3(kThreads), 0,
                     (hipStream_t)stream, arr, h, w, nbins, out);
  return pl_check_launch("pl_radial_average");
}

extern "C" int pl_esf_mtf(const double* esf, const int* lens, const double* window, int n_esf, int lmax,
                          int num_samples, double* work, double* mtf_each, double* mtf_mean, void* stream) {
  PL_REQUIRE(esf && lens && window && work && mtf_each && mtf_mean, "null pointer");
  PL_REQUIRE(n_esf > 0 && lmax >= 2 && num_samples >= 2, "bad shape");
  hipStream_t st = (hipStream_t)stream;
  const int half = num_samples / 2;
  hipLaunchKernelGGL(esf_dft_kernel, dim3((unsigned)pl_cdiv((int64_t)n_esf * half, kThreads)), dim3(kThreads), 0,
                     st, esf, lens, window, n_esf, lmax, num_samples, work);
  hipLaunchKernelGGL(esf_normalise_kernel, dim3((unsigned)pl_cdiv(half, kThreads)), dim3(kThreads), 0, st, work,
                     n_esf, half, mtf_each, mtf_mean);
  return pl_check_launch("pl_esf_mtf");
}
