// Disk / rectangle ROI statistics (SURVEY.md section 8 "next" row f3).
//
// Replaces: DiskROI.circle_mask + pixel_value / mean / std / min / max (pylinac/core/roi.py:104-140) and the
// axis-aligned RectangleROI.pixel_array statistics (:664-704), as used after phantom localisation for the
// CatPhan HU / uniformity / low-contrast ROIs (pylinac/ct.py:554-586).
//
//   disk pixels   skimage.draw.disk(center=(cy, cx), radius): bounding box ceil(center - r) .. floor(center + r),
//                 membership ((ri - r_org) / r)^2 + ((ci - c_org) / r)^2 < 1 with ri, ci the box-local indices and
//                 (r_org, c_org) = center - upper_left, float64 in skimage's operation order (draw.py ellipse /
//                 _ellipse_in_shape, rotation 0); pixels in raster order like np.nonzero
//   rectangle     array[r0:r1, c0:c1] (the caller applies the reference's rounding)
//   polygon       skimage.draw.polygon(r, c, shape) as RectangleROI.pixels_flat calls it for (rotated) rectangles
//                 (pylinac/core/roi.py:644-662): bounding box int(max(0, min)) .. min(size - 1, int(ceil(max))), every
//                 integer point kept unless scikit-image 0.18.3's point_in_polygon (the two-ray crossing test of Hao
//                 et al. 2018, 1e-12 vertex tolerance) says "outside": vertices and edge points belong to the ROI
//   statistics    np.mean / np.std (population, two-pass) / np.min / np.max / np.median (exact order statistics,
//                 mean of the two middle values for an even count)
//
// One workgroup per (frame, ROI): the ROI values are gathered into LDS as float64 (<= 16 384 pixels), reduced,
// and the median is found by bisection on an order-preserving 64-bit key.  Larger ROIs take the same passes straight
// from the frame (membership re-evaluated, pixels re-read per pass): slower per ROI, no size limit below 2^28 box pixels.
#include "pl_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxPix = 16384;

__device__ __forceinline__ unsigned long long key_of(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double value_of(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

// v: nv (row, col) pairs.  float64, no contraction: the quotient decides membership on edges.
__device__ __forceinline__ bool in_polygon(const double* __restrict__ v, int nv, double x, double y) {
  const double eps = 1e-12;
  int r_cross = 0, l_cross = 0;
  double x1 = v[2 * (nv - 1) + 1] - x, y1 = v[2 * (nv - 1)] - y;
  for (int i = 0; i < nv; ++i) {
    const double x0 = x1, y0 = y1;
    x1 = v[2 * i + 1] - x;
    y1 = v[2 * i] - y;
    if (-eps < x0 && x0 < eps && -eps < y0 && y0 < eps) return true;            // on a vertex
    if ((y0 > 0) != (y1 > 0) && (x1 * y0 - x0 * y1) / (y0 - y1) > 0) ++r_cross;
    if ((y0 < 0) != (y1 < 0) && (x1 * y0 - x0 * y1) / (y0 - y1) < 0) ++l_cross;
  }
  return ((r_cross & 1) != (l_cross & 1)) || (r_cross & 1);                     // on an edge, or inside
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
roi_stats_kernel(const T* __restrict__ frames, int h, int w, const double* __restrict__ rois, int rois_per_frame,
                 int64_t roi_frame_stride, int kind, int roi_doubles, double* __restrict__ out,
                 int32_t* __restrict__ status) {
  extern __shared__ double vals[];   // kMaxPix
  __shared__ int s_n;
  __shared__ double s_red[4][kThreads / PL_WAVE];
  __shared__ unsigned long long s_cnt;
  const int64_t item = blockIdx.x;
  const int64_t frame = item / rois_per_frame;
  const int k = (int)(item % rois_per_frame);
  const double* roi = rois + frame * roi_frame_stride + (int64_t)k * roi_doubles;
  const int nv = roi_doubles / 2;   // kind 2
  const T* f = frames + frame * (int64_t)h * w;
  double* o = out + item * 6;

  int r_lo, c_lo, nr, nc;
  double r_org = 0, c_org = 0, rad = 1;
  if (kind == 0) {            // disk: roi = (cx, cy, radius, -)
    const double cy = roi[1], cx = roi[0];
    rad = roi[2];
    r_lo = (int)ceil(cy - rad);
    c_lo = (int)ceil(cx - rad);
    nr = (int)floor(cy + rad) - r_lo + 1;
    nc = (int)floor(cx + rad) - c_lo + 1;
    r_org = cy - (double)r_lo;
    c_org = cx - (double)c_lo;
  } else if (kind == 1) {     // rectangle: roi = (r0, r1, c0, c1), half-open
    r_lo = (int)roi[0];
    c_lo = (int)roi[2];
    nr = (int)roi[1] - r_lo;
    nc = (int)roi[3] - c_lo;
  } else {                    // polygon: roi = nv x (row, col); the box is clipped to the frame like polygon(shape=...)
    double rmin = roi[0], rmax = roi[0], cmin = roi[1], cmax = roi[1];
    for (int i = 1; i < nv; ++i) {
      rmin = roi[2 * i] < rmin ? roi[2 * i] : rmin;
      rmax = roi[2 * i] > rmax ? roi[2 * i] : rmax;
      cmin = roi[2 * i + 1] < cmin ? roi[2 * i + 1] : cmin;
      cmax = roi[2 * i + 1] > cmax ? roi[2 * i + 1] : cmax;
    }
    // int(max(0, min)) .. min(size - 1, int(ceil(max))); the clamps keep the casts in range for far-away vertices
    r_lo = (int)(rmin > 0 ? (rmin < (double)h ? rmin : (double)h) : 0.0);
    c_lo = (int)(cmin > 0 ? (cmin < (double)w ? cmin : (double)w) : 0.0);
    const double r_top = ceil(rmax), c_top = ceil(cmax);
    const int r_hi = r_top < (double)(h - 1) ? (r_top < -1.0 ? -1 : (int)r_top) : h - 1;
    const int c_hi = c_top < (double)(w - 1) ? (c_top < -1.0 ? -1 : (int)c_top) : w - 1;
    nr = r_hi - r_lo + 1;
    nc = c_hi - c_lo + 1;
    if (nr <= 0 || nc <= 0) {   // the polygon misses the frame: np.mean of nothing
      if (threadIdx.x == 0) { status[item] = 3; for (int q = 0; q < 6; ++q) o[q] = __longlong_as_double(0x7ff8000000000000LL); }
      return;
    }
  }
  __shared__ int s_outside;
  if (threadIdx.x == 0) { s_n = 0; s_outside = 0; }
  __syncthreads();
  auto fail = [&](int code) {
    if (threadIdx.x == 0) { status[item] = code; for (int q = 0; q < 6; ++q) o[q] = __longlong_as_double(0x7ff8000000000000LL); }
  };
  const int64_t total64 = (int64_t)nr * nc;
  if (nr <= 0 || nc <= 0) { fail(kind == 1 ? 1 : 3); return; }
  if (total64 > (1LL << 28)) { fail(2); return; }     // a box of more than 2^28 pixels is not an ROI
  if (kind == 1 && !(r_lo >= 0 && c_lo >= 0 && r_lo + nr <= h && c_lo + nc <= w)) { fail(1); return; }
  const int total = (int)total64;
  // membership + value of box element e; a selected pixel that lies outside the frame is what makes the reference wrap
  // or raise (decided from the pixels actually selected, not from the box: a disk whose box row -1 selects nothing is fine)
  auto member = [&](int e, double& v) -> bool {
    const int ri = e / nc, ci = e % nc;
    bool in;
    if (kind == 0) {
      const double a = ((double)ri - r_org) / rad, b = ((double)ci - c_org) / rad;
      in = (a * a + b * b) < 1.0;
    } else if (kind == 1) {
      in = true;
    } else {
      in = in_polygon(roi, nv, (double)(c_lo + ci), (double)(r_lo + ri));
    }
    if (!in) return false;
    const int r = r_lo + ri, c = c_lo + ci;
    if (r < 0 || c < 0 || r >= h || c >= w) { s_outside = 1; return false; }
    v = (double)f[(int64_t)r * w + c];
    return true;
  };
  // ---- small ROIs: gather in raster order into LDS (ordered compaction, one box row chunk at a time); ROIs whose box or
  //      pixel count exceeds the LDS buffer are STREAMED instead: every pass re-evaluates the membership and re-reads the
  //      pixels (ACR large uniformity disks, large rectangles) --------------------------------------------------------------
  bool streaming = total > 4 * kMaxPix;
  if (!streaming) {
    for (int base = 0; base < total; base += kThreads) {
      const int e = base + threadIdx.x;
      double v = 0;
      const bool in = e < total && member(e, v);
      const unsigned long long bal = __ballot(in);
      const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
      __shared__ int s_wcnt[kThreads / PL_WAVE];
      if (lane == 0) s_wcnt[wv] = __popcll(bal);
      __syncthreads();
      int off = s_n;
      for (int q = 0; q < wv; ++q) off += s_wcnt[q];
      const int dst = off + __popcll(bal & ((1ull << lane) - 1ull));
      if (in && dst < kMaxPix) vals[dst] = v;
      __syncthreads();
      if (threadIdx.x == 0) { int t = 0; for (int q = 0; q < kThreads / PL_WAVE; ++q) t += s_wcnt[q]; s_n += t; }
      __syncthreads();
    }
    streaming = s_n > kMaxPix;
  }
  auto block_reduce = [&](double v, int slot, auto op) {
    v = pl_wave_reduce(v, op);
    if ((threadIdx.x & 63) == 0) s_red[slot][threadIdx.x >> 6] = v;
    __syncthreads();
    double t = s_red[slot][0];
    for (int q = 1; q < kThreads / PL_WAVE; ++q) t = op(t, s_red[slot][q]);
    __syncthreads();
    return t;
  };
  int n = s_n;
  if (streaming) {
    double c = 0;
    for (int e = threadIdx.x; e < total; e += kThreads) { double v; c += member(e, v) ? 1.0 : 0.0; }
    n = (int)block_reduce(c, 0, [](double a, double b) { return a + b; });
  }
  __syncthreads();
  if (s_outside) { fail(1); return; }
  if (n == 0) { fail(3); return; }
  auto for_each = [&](auto fn) {
    if (!streaming) {
      for (int i = threadIdx.x; i < n; i += kThreads) fn(vals[i]);
    } else {
      for (int e = threadIdx.x; e < total; e += kThreads) { double v; if (member(e, v)) fn(v); }
    }
  };
  // ---- sum / min / max, then the two-pass variance -------------------------------------------------------------
  const double inf = __longlong_as_double(0x7ff0000000000000LL);
  double sum = 0, mn = inf, mx = -inf;
  for_each([&](double v) {
    sum += v;
    mn = v < mn ? v : mn;
    mx = v > mx ? v : mx;
  });
  sum = block_reduce(sum, 0, [](double a, double b) { return a + b; });
  mn = block_reduce(mn, 1, [](double a, double b) { return a < b ? a : b; });
  mx = block_reduce(mx, 2, [](double a, double b) { return a > b ? a : b; });
  const double mean = sum / (double)n;
  double ss = 0;
  for_each([&](double v) {
    const double d = v - mean;
    ss += d * d;
  });
  ss = block_reduce(ss, 3, [](double a, double b) { return a + b; });
  // ---- median: smallest key with #{<= key} >= rank + 1, for the two middle ranks -----------------------------------
  auto kth = [&](int rank) {
    unsigned long long lo = 0, hi = ~0ull;
    while (lo < hi) {
      const unsigned long long mid = lo + ((hi - lo) >> 1);
      unsigned long long c = 0;
      for_each([&](double v) { c += key_of(v) <= mid ? 1 : 0; });
      c = pl_wave_reduce(c, [](unsigned long long a, unsigned long long b) { return a + b; });
      if (threadIdx.x == 0) s_cnt = 0;
      __syncthreads();
      if ((threadIdx.x & 63) == 0) atomicAdd(&s_cnt, c);
      __syncthreads();
      const unsigned long long tot = s_cnt;
      __syncthreads();
      if (tot >= (unsigned long long)rank + 1) hi = mid; else lo = mid + 1;
    }
    return value_of(lo);
  };
  const double m_hi = kth(n / 2);
  const double median = (n & 1) ? m_hi : (kth(n / 2 - 1) + m_hi) / 2.0;   // np.median: mean of the two middle values
  if (threadIdx.x == 0) {
    o[0] = (double)n;
    o[1] = mean;
    o[2] = sqrt(ss / (double)n);
    o[3] = mn;
    o[4] = mx;
    o[5] = median;
    status[item] = 0;
  }
}

}  // namespace

extern "C" int pl_roi_stats(const void* frames, int dtype, int64_t n, int h, int w, const double* d_rois,
                            int rois_per_frame, int64_t roi_frame_stride, int kind, double* d_out,
                            int32_t* d_status, void* stream) {
  PL_REQUIRE(frames && d_rois && d_out && d_status, "null pointer");
  PL_REQUIRE(n >= 0 && h > 0 && w > 0 && rois_per_frame > 0, "bad shape");
  PL_REQUIRE(kind == 0 || kind == 1, "kind must be 0 (disk) or 1 (rectangle)");
  PL_REQUIRE(roi_frame_stride == 0 || roi_frame_stride >= 4 * (int64_t)rois_per_frame, "bad ROI stride");
  if (n == 0) return PL_OK;
  const int64_t items = n * rois_per_frame;
  PL_REQUIRE(items <= 0x7fffffffLL, "batch too large");
  const size_t lds = (size_t)kMaxPix * sizeof(double);
  hipStream_t st = (hipStream_t)stream;
  PL_DISPATCH_DTYPE(dtype, T, {
    static std::atomic<bool> attr{false};
    if (!attr) {
      hipError_t e = hipFuncSetAttribute((const void*)roi_stats_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds);
      if (e != hipSuccess) { pl_set_error("pl_roi_stats: LDS attribute: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
      attr = true;
    }
    hipLaunchKernelGGL(roi_stats_kernel<T>, dim3((unsigned)items), dim3(kThreads), lds, st, (const T*)frames, h, w,
                       d_rois, rois_per_frame, roi_frame_stride, kind, 4, d_out, d_status);
  });
  return pl_check_launch("pl_roi_stats");
}

extern "C" int pl_polygon_roi_stats(const void* frames, int dtype, int64_t n, int h, int w, const double* d_vertices,
                                    int n_vertices, int rois_per_frame, int64_t roi_frame_stride, double* d_out,
                                    int32_t* d_status, void* stream) {
  PL_REQUIRE(frames && d_vertices && d_out && d_status, "null pointer");
  PL_REQUIRE(n >= 0 && h > 0 && w > 0 && rois_per_frame > 0, "bad shape");
  PL_REQUIRE(n_vertices >= 3 && n_vertices <= 64, "a polygon has 3..64 vertices");
  const int roi_doubles = 2 * n_vertices;
  PL_REQUIRE(roi_frame_stride == 0 || roi_frame_stride >= roi_doubles * (int64_t)rois_per_frame, "bad ROI stride");
  if (n == 0) return PL_OK;
  const int64_t items = n * rois_per_frame;
  PL_REQUIRE(items <= 0x7fffffffLL, "batch too large");
  const size_t lds = (size_t)kMaxPix * sizeof(double);
  hipStream_t st = (hipStream_t)stream;
  PL_DISPATCH_DTYPE(dtype, T, {
    static std::atomic<bool> attr{false};
    if (!attr) {
      hipError_t e = hipFuncSetAttribute((const void*)roi_stats_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds);
      if (e != hipSuccess) { pl_set_error("pl_polygon_roi_stats: LDS attribute: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
      attr = true;
    }
    hipLaunchKernelGGL(roi_stats_kernel<T>, dim3((unsigned)items), dim3(kThreads), lds, st, (const T*)frames, h, w,
                       d_vertices, rois_per_frame, roi_frame_stride, 2, roi_doubles, d_out, d_status);
  });
  return pl_check_launch("pl_polygon_roi_stats");
}
