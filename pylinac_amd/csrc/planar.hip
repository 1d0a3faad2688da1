// Hough-space peak search for the planar phantoms (SURVEY.md section 8 "next" row f2, second half).
//
// Replaces the dense part of skimage.transform.hough_line_peaks -> skimage.feature.peak._prominent_peaks
// (scikit-image 0.18.3) as called at pylinac/planar_imaging.py:3160-3166 on the accumulator of pl_hough_line
// (2 * ceil(diagonal) rows x 1001 angles for the reference's 40..50 degree band; min_distance = 70 mm in pixels, so
// the row window is a few hundred taps):
//   pl_max_filter1d      ndimage.maximum_filter1d(img, size = 2 * half + 1, axis, mode="constant", cval=0)
//   pl_peak_candidates   (img == img_max) & (img > threshold)   -- the pixels `img *= mask; img > threshold` keeps
// The candidate groups are then labelled with pl_label (8-connected); the greedy neighbourhood suppression walks a
// handful of groups in height order and stays on the host (planar.py).
// One lane per output element; the window loop re-reads neighbours through L1/L2 (axis 0: lanes = adjacent columns,
// coalesced rows; axis 1: adjacent lanes share all but one tap).
#include "pl_common.h"

namespace {

constexpr int kThreads = 256;

template <typename T>
__global__ void __launch_bounds__(kThreads)
max_filter1d_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t total, int h, int w, int axis, int half) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % w);
  const int r = (int)((i / w) % h);
  const int pos = axis == 0 ? r : c;
  const int len = axis == 0 ? h : w;
  const int64_t step = axis == 0 ? w : 1;
  const int lo = pos - half < 0 ? 0 : pos - half;
  const int hi = pos + half > len - 1 ? len - 1 : pos + half;
  const T* p = in + (i - (int64_t)pos * step);
  T m = p[(int64_t)lo * step];
  for (int k = lo + 1; k <= hi; ++k) {
    const T v = p[(int64_t)k * step];
    m = v > m ? v : m;
  }
  // the part of the window that hangs over the edge reads cval = 0
  if ((pos - half < 0 || pos + half > len - 1) && m < (T)0) m = (T)0;
  out[i] = m;
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
peak_candidates_kernel(const T* __restrict__ img, const T* __restrict__ img_max, int64_t total, double threshold,
                       unsigned char* __restrict__ mask) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const T v = img[i];
  mask[i] = (v == img_max[i] && (double)v > threshold) ? 1 : 0;
}

}  // namespace

extern "C" int pl_max_filter1d(const void* in, void* out, int dtype, int64_t n, int h, int w, int axis, int half,
                               void* stream) {
  PL_REQUIRE(in && out, "null pointer");
  PL_REQUIRE(in != out, "in-place filtering is not supported");
  PL_REQUIRE(n >= 0 && h > 0 && w > 0, "bad shape");
  PL_REQUIRE(axis == 0 || axis == 1, "axis must be 0 or 1");
  PL_REQUIRE(half >= 0, "half window must be >= 0");
  const int64_t total = n * (int64_t)h * w;
  if (total == 0) return PL_OK;
  PL_REQUIRE(pl_cdiv(total, kThreads) <= 0x7fffffffLL, "batch too large for one launch");
  PL_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL(max_filter1d_kernel<T>, dim3((unsigned)pl_cdiv(total, kThreads)), dim3(kThreads), 0,
                                       (hipStream_t)stream, (const T*)in, (T*)out, total, h, w, axis, half));
  return pl_check_launch("pl_max_filter1d");
}

extern "C" int pl_peak_candidates(const void* img, const void* img_max, int dtype, int64_t count, double threshold,
                                  unsigned char* mask, void* stream) {
  PL_REQUIRE(img && img_max && mask, "null pointer");
  PL_REQUIRE(count >= 0, "bad count");
  if (count == 0) return PL_OK;
  PL_REQUIRE(pl_cdiv(count, kThreads) <= 0x7fffffffLL, "batch too large for one launch");
  PL_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL(peak_candidates_kernel<T>, dim3((unsigned)pl_cdiv(count, kThreads)), dim3(kThreads),
                                       0, (hipStream_t)stream, (const T*)img, (const T*)img_max, count, threshold, mask));
  return pl_check_launch("pl_peak_candidates");
}
