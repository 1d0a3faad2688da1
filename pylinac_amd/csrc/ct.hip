// CatPhan slice localisation primitives.
//
// Replaces, with the parameters pylinac passes (pylinac/ct.py:381-425 Slice.phantom_roi and
// :3315-3348 get_regions; scikit-image 0.18.3 + scipy semantics):
//   pl_scharr        skimage.filters.scharr(float image): two scipy.ndimage.convolve calls with the
//                    3x3 kernels edge (x) smooth/16 (mode='reflect'; non-zero taps accumulated from 0
//                    in raster order of the flipped kernel), squared, summed, sqrt, / sqrt(2)
//   pl_clip          np.clip(array, lo, hi)                                   (ct.py:400)
//   pl_hist_uniform  np.histogram(values, bins=256) as used by threshold_otsu on float data:
//                    numpy's exact edge-corrected bin assignment, optional pixel mask (draw.disk)
//   pl_compare       edges > thres  (strict)                                   (ct.py:3341)
//   pl_clear_border  skimage.segmentation.clear_border(bw, buffer_size): 8-connected components
//                    that touch the (buffer_size+1)-wide frame border band are removed
//   pl_region_stats  skimage.measure.regionprops raw sums per label: area, bbox, coordinate sums,
//                    intensity-weighted sums (centroid / weighted_centroid / area / bbox)
#include "pl_common.h"

namespace {

constexpr int kThreads = 256;

template <typename T>
__global__ void __launch_bounds__(kThreads)
scharr_kernel(const T* __restrict__ in, double* __restrict__ out, int64_t total, int h, int w) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= total) return;
  const int c = (int)(g % w);
  const int64_t t = g / w;
  const int r = (int)(t % h);
  const T* f = in + (t / h) * (size_t)h * w;
  const int rm = pl_reflect(r - 1, h), rp = pl_reflect(r + 1, h);
  const int cm = pl_reflect(c - 1, w), cp = pl_reflect(c + 1, w);
  auto at = [&](int rr, int cc) { return (double)f[(size_t)rr * w + cc]; };
  const double a = 0.1875, b = 0.625;  // 3/16, 10/16
  // edge along axis 0: flipped kernel rows (-1: -3 -10 -3 ; +1: 3 10 3) / 16, raster order
  double s0 = 0.0;
  s0 = s0 + at(rm, cm) * -a; s0 = s0 + at(rm, c) * -b; s0 = s0 + at(rm, cp) * -a;
  s0 = s0 + at(rp, cm) * a;  s0 = s0 + at(rp, c) * b;  s0 = s0 + at(rp, cp) * a;
  // edge along axis 1: flipped kernel (-3 0 3 ; -10 0 10 ; -3 0 3) / 16, raster order
  double s1 = 0.0;
  s1 = s1 + at(rm, cm) * -a; s1 = s1 + at(rm, cp) * a;
  s1 = s1 + at(r, cm) * -b;  s1 = s1 + at(r, cp) * b;
  s1 = s1 + at(rp, cm) * -a; s1 = s1 + at(rp, cp) * a;
  double o = 0.0;
  o = o + s0 * s0;
  o = o + s1 * s1;
  out[g] = sqrt(o) / 1.4142135623730951;  // np.sqrt(output) / np.sqrt(ndim)
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
clip_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t total, double lo, double hi) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= total) return;
  const T v = in[g];
  const T l = (T)lo, u = (T)hi;
  out[g] = v < l ? l : (v > u ? u : v);
}

// numpy's uniform-bin histogram: estimate the index, then correct it against the actual edges
__global__ void __launch_bounds__(kThreads)
hist_uniform_kernel(const double* __restrict__ in, const uint8_t* __restrict__ mask, int64_t per_frame,
                    const double* __restrict__ edges /* [n][nbins+1] */, int nbins,
                    uint32_t* __restrict__ counts /* [n][nbins], zeroed */) {
  extern __shared__ unsigned lbins[];
  const int64_t frame = blockIdx.y;
  const double* e = edges + frame * (nbins + 1);
  for (int i = threadIdx.x; i < nbins; i += kThreads) lbins[i] = 0;
  __syncthreads();
  const double first = e[0], last = e[nbins];
  const double denom = last - first;
  const double* src = in + frame * per_frame;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < per_frame; i += (int64_t)gridDim.x * kThreads) {
    if (mask && !mask[i]) continue;
    const double v = src[i];
    if (!(v >= first && v <= last)) continue;
    int idx = (int)(((v - first) / denom) * (double)nbins);
    if (idx == nbins) idx -= 1;
    if (idx < 0) idx = 0;
    if (v < e[idx]) idx -= 1;
    else if (v >= e[idx + 1] && idx != nbins - 1) idx += 1;
    atomicAdd(&lbins[idx], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nbins; i += kThreads)
    if (lbins[i]) atomicAdd(&counts[frame * nbins + i], lbins[i]);
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
compare_kernel(const T* __restrict__ in, int64_t total, int64_t per_frame, const double* __restrict__ thr,
               int thr_stride, int op, uint8_t* __restrict__ out) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= total) return;
  const double v = (double)in[g], t = thr[(g / per_frame) * thr_stride];
  const bool r = op == 0 ? (v >= t) : op == 1 ? (v > t) : op == 2 ? (v <= t) : (v < t);
  out[g] = r ? 1 : 0;
}

// ---- clear_border: flag the roots of components that own a pixel in the border band -------------
__global__ void band_flag_kernel(const int* __restrict__ Lall, int64_t total, int h, int w, int ext,
                                 uint8_t* __restrict__ flags) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= total) return;
  const int64_t per_frame = (int64_t)h * w;
  const int i = (int)(g % per_frame);
  const int r = i / w, c = i % w;
  if (r < ext || r >= h - ext || c < ext || c >= w - ext) {
    const int root = Lall[g];
    if (root >= 0) flags[(g / per_frame) * per_frame + root] = 1;
  }
}
__global__ void clear_apply_kernel(const uint8_t* __restrict__ mask, const int* __restrict__ Lall,
                                   const uint8_t* __restrict__ flags, int64_t total, int64_t per_frame,
                                   uint8_t* __restrict__ out) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= total) return;
  const int root = Lall[g];
  out[g] = (root >= 0 && !flags[(g / per_frame) * per_frame + root]) ? (mask[g] ? 1 : 0) : 0;
}

// ---- region sums -------------------------------------------------------------------------------
// stats layout per label (float64 x 10): area, rmin, cmin, rmax+1, cmax+1 (skimage bbox is half-open),
// sum r, sum c, sum w, sum w*r, sum w*c.   Integer quantities are accumulated exactly in a uint64
// side buffer and converted at the end; the weighted sums use float64 atomics (order-dependent in the
// last bits: the reference's moments are compared at 1e-9 relative, SURVEY asks 1e-5).
__global__ void region_init_kernel(unsigned long long* __restrict__ isum, double* __restrict__ wsum, int64_t rows) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= rows) return;
  isum[g * 7 + 0] = 0;                      // area
  isum[g * 7 + 1] = 0xffffffffffffffffull;  // rmin
  isum[g * 7 + 2] = 0xffffffffffffffffull;  // cmin
  isum[g * 7 + 3] = 0;                      // rmax
  isum[g * 7 + 4] = 0;                      // cmax
  isum[g * 7 + 5] = 0;                      // sum r
  isum[g * 7 + 6] = 0;                      // sum c
  wsum[g * 3 + 0] = 0.0; wsum[g * 3 + 1] = 0.0; wsum[g * 3 + 2] = 0.0;
}

__global__ void __launch_bounds__(kThreads)
region_accum_kernel(const int32_t* __restrict__ labels, const double* __restrict__ intensity, int64_t total,
                    int h, int w, int max_labels, unsigned long long* __restrict__ isum,
                    double* __restrict__ wsum, int32_t* __restrict__ overflow) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= total) return;
  const int lab = labels[g];
  if (lab <= 0) return;
  const int64_t per_frame = (int64_t)h * w;
  const int64_t frame = g / per_frame;
  if (lab > max_labels) { overflow[frame] = 1; return; }
  const int i = (int)(g % per_frame);
  const unsigned long long r = (unsigned long long)(i / w), c = (unsigned long long)(i % w);
  unsigned long long* s = isum + (frame * max_labels + (lab - 1)) * 7;
  atomicAdd(&s[0], 1ull);
  atomicMin(&s[1], r); atomicMin(&s[2], c);
  atomicMax(&s[3], r); atomicMax(&s[4], c);
  atomicAdd(&s[5], r); atomicAdd(&s[6], c);
  if (intensity) {
    const double v = intensity[g];
    double* ws = wsum + (frame * max_labels + (lab - 1)) * 3;
    atomicAdd(&ws[0], v);
    atomicAdd(&ws[1], v * (double)r);
    atomicAdd(&ws[2], v * (double)c);
  }
}

__global__ void region_finish_kernel(const unsigned long long* __restrict__ isum, const double* __restrict__ wsum,
                                     int64_t rows, double* __restrict__ out) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= rows) return;
  const unsigned long long* s = isum + g * 7;
  double* o = out + g * 10;
  const bool empty = s[0] == 0;
  o[0] = (double)s[0];
  o[1] = empty ? 0.0 : (double)s[1];
  o[2] = empty ? 0.0 : (double)s[2];
  o[3] = empty ? 0.0 : (double)(s[3] + 1);
  o[4] = empty ? 0.0 : (double)(s[4] + 1);
  o[5] = (double)s[5];
  o[6] = (double)s[6];
  o[7] = wsum[g * 3 + 0]; o[8] = wsum[g * 3 + 1]; o[9] = wsum[g * 3 + 2];
}

// ---- exact raw second moments per label (regionprops.orientation / eccentricity / centroid) ------
// m00, m10 = sum r, m01 = sum c, m20 = sum r*r, m02 = sum c*c, m11 = sum r*c in IMAGE coordinates, exact uint64.
// The central moments are translation invariant and are formed from these on the host in exact integer
// arithmetic (n*m20 - m10*m10, ...), so symmetric regions give mu20 == mu02 and mu11 == 0 exactly.
// A wave usually sits inside one region: the lanes that share the first pending label are summed with
// cross-lane butterflies and one lane issues the six atomics.
__global__ void __launch_bounds__(kThreads)
region_moments_kernel(const int32_t* __restrict__ labels, int64_t total, int h, int w, int max_labels,
                      unsigned long long* __restrict__ mom, int32_t* __restrict__ overflow) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const int64_t per_frame = (int64_t)h * w;
  int lab = 0;
  int64_t frame = 0;
  unsigned long long r = 0, c = 0;
  if (g < total) {
    lab = labels[g];
    frame = g / per_frame;
    const int i = (int)(g - frame * per_frame);
    r = (unsigned long long)(i / w);
    c = (unsigned long long)(i % w);
    if (lab > max_labels) { overflow[frame] = 1; lab = 0; }
    if (lab < 0) lab = 0;
  }
  // key = (frame, label): a wave can straddle two frames
  long long key = lab > 0 ? frame * (long long)max_labels + (lab - 1) : -1;
  unsigned long long pending = __ballot(key >= 0);
  const int lane = threadIdx.x & 63;
  while (pending) {
    const int leader = __builtin_ctzll(pending);
    const long long k = __shfl(key, leader, 64);
    const bool mine = key == k;
    const unsigned long long grp = __ballot(mine);
    unsigned long long v[6];
    v[0] = mine ? 1ull : 0ull;
    v[1] = mine ? r : 0ull;
    v[2] = mine ? c : 0ull;
    v[3] = mine ? r * r : 0ull;
    v[4] = mine ? c * c : 0ull;
    v[5] = mine ? r * c : 0ull;
#pragma unroll
    for (int q = 0; q < 6; ++q) v[q] = pl_wave_reduce(v[q], [](unsigned long long a, unsigned long long b) { return a + b; });
    if (lane == leader) {
      unsigned long long* s = mom + k * 6;
#pragma unroll
      for (int q = 0; q < 6; ++q) atomicAdd(&s[q], v[q]);
    }
    pending &= ~grp;
  }
}

}  // namespace

int pl_ccl_roots(const uint8_t* mask, int invert, int64_t n, int h, int w, int conn, int* L, hipStream_t st);

#define PL_CT_TOTAL()                                                                       \
  const int64_t per_frame = (int64_t)h * w, total = n * per_frame;                           \
  PL_REQUIRE(n >= 0 && h > 0 && w > 0, "bad shape");                                        \
  PL_REQUIRE(per_frame <= 0x7fffffffLL && pl_cdiv(total, kThreads) <= 0x7fffffffLL, "too large"); \
  if (n == 0) return PL_OK;                                                                 \
  hipStream_t st = (hipStream_t)stream;                                                     \
  const unsigned blocks = (unsigned)pl_cdiv(total, kThreads);

extern "C" int pl_scharr(const void* in, double* out, int dtype, int64_t n, int h, int w, void* stream) {
  PL_REQUIRE(in && out, "null pointer");
  PL_CT_TOTAL();
  PL_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL(scharr_kernel<T>, dim3(blocks), dim3(kThreads), 0, st, (const T*)in, out, total,
                                       h, w));
  return pl_check_launch("pl_scharr");
}

extern "C" int pl_clip(const void* in, void* out, int dtype, int64_t n, int64_t count, double lo, double hi,
                       void* stream) {
  PL_REQUIRE(in && out, "null pointer");
  PL_REQUIRE(n >= 0 && count > 0 && lo <= hi, "bad arguments");
  if (n == 0) return PL_OK;
  const int64_t total = n * count;
  PL_REQUIRE(pl_cdiv(total, kThreads) <= 0x7fffffffLL, "too large");
  PL_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL(clip_kernel<T>, dim3((unsigned)pl_cdiv(total, kThreads)), dim3(kThreads), 0,
                                       (hipStream_t)stream, (const T*)in, (T*)out, total, lo, hi));
  return pl_check_launch("pl_clip");
}

extern "C" int pl_hist_uniform(const double* in, const uint8_t* d_mask, int64_t n, int64_t count,
                               const double* d_edges, int nbins, uint32_t* d_counts, void* stream) {
  PL_REQUIRE(in && d_edges && d_counts, "null pointer");
  PL_REQUIRE(n >= 0 && n <= 65535 && count > 0 && nbins > 0 && nbins <= 8192, "bad arguments");
  if (n == 0) return PL_OK;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(d_counts, 0, (size_t)n * nbins * sizeof(uint32_t), st);
  if (e != hipSuccess) { pl_set_error("pl_hist_uniform: memset: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
  int bx = (int)pl_cdiv(count, (int64_t)kThreads * 16);
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(hist_uniform_kernel, dim3((unsigned)bx, (unsigned)n), dim3(kThreads), nbins * sizeof(unsigned), st,
                     in, d_mask, count, d_edges, nbins, d_counts);
  return pl_check_launch("pl_hist_uniform");
}

extern "C" int pl_compare(const void* in, int dtype, int64_t n, int64_t count, const double* d_thr,
                          int thr_stride, int op, uint8_t* d_out, void* stream) {
  PL_REQUIRE(in && d_thr && d_out, "null pointer");
  PL_REQUIRE(n >= 0 && count > 0 && op >= 0 && op <= 3 && (thr_stride == 0 || thr_stride == 1), "bad arguments");
  if (n == 0) return PL_OK;
  const int64_t total = n * count;
  PL_REQUIRE(pl_cdiv(total, kThreads) <= 0x7fffffffLL, "too large");
  PL_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL(compare_kernel<T>, dim3((unsigned)pl_cdiv(total, kThreads)), dim3(kThreads), 0,
                                       (hipStream_t)stream, (const T*)in, total, count, d_thr, thr_stride, op, d_out));
  return pl_check_launch("pl_compare");
}

extern "C" int pl_clear_border(const uint8_t* d_mask, uint8_t* d_out, int64_t n, int h, int w, int buffer_size,
                               int32_t* d_work, uint8_t* d_flags, void* stream) {
  PL_REQUIRE(d_mask && d_out && d_work && d_flags, "null pointer");
  PL_REQUIRE(buffer_size >= 0, "negative buffer");
  PL_CT_TOTAL();
  if (int rc = pl_ccl_roots(d_mask, 0, n, h, w, 8, d_work, st)) return rc;  // skimage label default: full connectivity
  hipError_t e = hipMemsetAsync(d_flags, 0, (size_t)total, st);
  if (e != hipSuccess) { pl_set_error("pl_clear_border: memset: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
  hipLaunchKernelGGL(band_flag_kernel, dim3(blocks), dim3(kThreads), 0, st, d_work, total, h, w, buffer_size + 1, d_flags);
  hipLaunchKernelGGL(clear_apply_kernel, dim3(blocks), dim3(kThreads), 0, st, d_mask, d_work, d_flags, total, per_frame,
                     d_out);
  return pl_check_launch("pl_clear_border");
}

extern "C" int pl_region_stats(const int32_t* d_labels, const double* d_intensity, int64_t n, int h, int w,
                               int max_labels, unsigned long long* d_isum, double* d_wsum, double* d_stats,
                               int32_t* d_overflow, void* stream) {
  PL_REQUIRE(d_labels && d_isum && d_wsum && d_stats && d_overflow, "null pointer");
  PL_REQUIRE(max_labels > 0, "max_labels must be positive");
  PL_CT_TOTAL();
  const int64_t rows = n * max_labels;
  hipError_t e = hipMemsetAsync(d_overflow, 0, (size_t)n * sizeof(int32_t), st);
  if (e != hipSuccess) { pl_set_error("pl_region_stats: memset: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
  hipLaunchKernelGGL(region_init_kernel, dim3((unsigned)pl_cdiv(rows, kThreads)), dim3(kThreads), 0, st, d_isum, d_wsum,
                     rows);
  hipLaunchKernelGGL(region_accum_kernel, dim3(blocks), dim3(kThreads), 0, st, d_labels, d_intensity, total, h, w,
                     max_labels, d_isum, d_wsum, d_overflow);
  hipLaunchKernelGGL(region_finish_kernel, dim3((unsigned)pl_cdiv(rows, kThreads)), dim3(kThreads), 0, st, d_isum,
                     d_wsum, rows, d_stats);
  return pl_check_launch("pl_region_stats");
}

extern "C" int pl_region_moments(const int32_t* d_labels, int64_t n, int h, int w, int max_labels,
                                 unsigned long long* d_mom, int32_t* d_overflow, void* stream) {
  PL_REQUIRE(d_labels && d_mom && d_overflow, "null pointer");
  PL_REQUIRE(max_labels > 0, "max_labels must be positive");
  PL_CT_TOTAL();
  hipError_t e = hipMemsetAsync(d_overflow, 0, (size_t)n * sizeof(int32_t), st);
  if (e == hipSuccess) e = hipMemsetAsync(d_mom, 0, (size_t)n * max_labels * 6 * sizeof(unsigned long long), st);
  if (e != hipSuccess) { pl_set_error("pl_region_moments: memset: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
  hipLaunchKernelGGL(region_moments_kernel, dim3(blocks), dim3(kThreads), 0, st, d_labels, total, h, w, max_labels,
                     d_mom, d_overflow);
  return pl_check_launch("pl_region_moments");
}
