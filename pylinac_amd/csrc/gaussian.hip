// Separable Gaussian with scipy-exact semantics.
//
// Replaces: scipy.ndimage.gaussian_filter as called at pylinac/core/array_utils.py:133
// (BaseImage.filter(kind="gaussian"), pylinac/core/image.py:695-712).
//
// Contract reproduced bit-for-bit (verified against scipy 1.15.3 by oracle/pylinac_oracle.py):
//   per axis (0 first, then 1):   acc  = x[0]*w[0]
//                                 acc += (x[-j] + x[+j]) * w[j]   j = radius .. 1 (outermost first)
//   in float64 without FMA contraction, borders 'reflect', result C-cast into the image dtype
//   (integer dtypes: truncation toward zero), and that dtype feeds the next axis.
//
// This is 61 float64 VALU operations per pixel per pass at sigma=5 (radius 20): the pass is bound
// by the FP64 issue rate (16 lanes/clk/SIMD), not by HBM -- see DESIGN.md.  The design therefore
// minimises everything that is NOT one of those 61 operations:
//   * vertical pass: one lane owns one column and NOUT consecutive rows; the NOUT+2*RAD inputs are
//     loaded once (coalesced 64-lane row segments, rows wave-uniform so the reflect index is
//     scalar) and converted once; the taps sit in SGPRs (uniform scalar loads).
//   * horizontal pass: a wave owns a 64*NOUT-pixel row segment; the segment (+halo) is converted
//     to float64 ONCE while being staged into LDS, with a 16-byte pad every 64 bytes so that the
//     per-lane ds_read_b128 windows (80-byte lane stride) are bank-conflict-free.
//   * block ids are remapped so that consecutive tiles of a frame share an XCD's L2 (halo rows).
#include "pl_common.h"

namespace {

constexpr int kThreads = 256;

// ------------------------------------------------------------------ vertical (axis 0), fast path
template <typename T, int RAD, int NOUT>
__global__ void __launch_bounds__(kThreads)
gauss_v_fast(const T* __restrict__ in, T* __restrict__ out, int h, int w, int col_tiles,
             int row_groups, const double* __restrict__ wts) {
  const unsigned nwg = gridDim.x;
  unsigned id = pl_xcd_remap(blockIdx.x, nwg);
  const int ct = id % col_tiles;
  id /= col_tiles;
  const int rg = id % row_groups;
  const size_t frame = id / row_groups;

  const int c = ct * kThreads + threadIdx.x;
  const int r0 = rg * NOUT;
  if (c >= w) return;
  const T* f = in + frame * (size_t)h * w;
  T* o = out + frame * (size_t)h * w;

  double x[NOUT + 2 * RAD];
  if (r0 - RAD >= 0 && r0 + NOUT + RAD <= h) {
    const T* p = f + (size_t)(r0 - RAD) * w + c;
#pragma unroll
    for (int k = 0; k < NOUT + 2 * RAD; ++k) x[k] = (double)p[(size_t)k * w];
  } else {
#pragma unroll
    for (int k = 0; k < NOUT + 2 * RAD; ++k) {
      int r = pl_reflect(r0 - RAD + k, h);  // wave-uniform
      x[k] = (double)f[(size_t)r * w + c];
    }
  }
#pragma unroll
  for (int i = 0; i < NOUT; ++i) {
    double acc = x[i + RAD] * wts[RAD];
#pragma unroll
    for (int j = RAD; j >= 1; --j) acc = acc + (x[i + RAD - j] + x[i + RAD + j]) * wts[RAD - j];
    if (r0 + i < h) o[(size_t)(r0 + i) * w + c] = pl_from_double<T>(acc);
  }
}

// ---------------------------------------------------------------- horizontal (axis 1), fast path
// LDS layout per wave: logical position p in [0, 64*NOUT + 2*RAD) <-> column c0 - RAD + p,
// stored at p + 2*(p >> 3) doubles (16-byte pad after every 64 bytes).
__device__ __forceinline__ int pad8(int p) { return p + ((p >> 3) << 1); }

template <typename T, int RAD, int NOUT>
__global__ void __launch_bounds__(kThreads)
gauss_h_fast(const T* __restrict__ in, T* __restrict__ out, int64_t rows_total, int w,
             int col_tiles, const double* __restrict__ wts) {
  static_assert(NOUT == 8, "pad8 assumes 8 outputs per lane");
  static_assert(2 * RAD <= PL_WAVE, "halo is loaded by one wave pass");
  constexpr int SEG = PL_WAVE * NOUT;
  constexpr int LOGICAL = SEG + 2 * RAD;
  constexpr int PADDED = LOGICAL + ((LOGICAL + 7) / 8) * 2;
  __shared__ __attribute__((aligned(16))) double lds[(kThreads / PL_WAVE) * PADDED];

  const int lane = threadIdx.x & (PL_WAVE - 1);
  const int wave = threadIdx.x / PL_WAVE;
  const unsigned nwg = gridDim.x;
  unsigned id = pl_xcd_remap(blockIdx.x, nwg);
  const int ct = id % col_tiles;
  const int64_t row = (int64_t)(id / col_tiles) * (kThreads / PL_WAVE) + wave;  // frame*h + r
  if (row >= rows_total) return;  // whole wave exits together; no block-level barrier is used

  const T* f = in + row * (size_t)w;
  T* o = out + row * (size_t)w;
  double* s = lds + wave * PADDED;
  const int c0 = ct * SEG;
  const int c = c0 + lane * NOUT;

  // stage: own NOUT pixels (vector load when possible) + halo, converted to f64 once
  if (c + NOUT <= w && (sizeof(T) * NOUT == 16) && ((reinterpret_cast<uintptr_t>(f + c) & 15) == 0)) {
    union { uint4 v; T e[NOUT]; } u;
    u.v = *reinterpret_cast<const uint4*>(f + c);
#pragma unroll
    for (int k = 0; k < NOUT; ++k) s[pad8(RAD + lane * NOUT + k)] = (double)u.e[k];
  } else {
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
      int cc = pl_reflect(c + k, w);
      s[pad8(RAD + lane * NOUT + k)] = (double)f[cc];
    }
  }
  if (lane < RAD) {
    int cc = pl_reflect(c0 - RAD + lane, w);
    s[pad8(lane)] = (double)f[cc];
  } else if (lane < 2 * RAD) {
    int p = SEG + lane;  // = RAD + SEG + (lane - RAD)
    int cc = pl_reflect(c0 - RAD + p, w);
    s[pad8(p)] = (double)f[cc];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

  if (c >= w) return;
  double x[NOUT + 2 * RAD];
  const double* win = s + 10 * lane;  // pad8(8*lane + k) = 10*lane + k + 2*(k>>3)
#pragma unroll
  for (int k = 0; k < NOUT + 2 * RAD; ++k) x[k] = win[k + ((k >> 3) << 1)];

  T res[NOUT];
#pragma unroll
  for (int i = 0; i < NOUT; ++i) {
    double acc = x[i + RAD] * wts[RAD];
#pragma unroll
    for (int j = RAD; j >= 1; --j) acc = acc + (x[i + RAD - j] + x[i + RAD + j]) * wts[RAD - j];
    res[i] = pl_from_double<T>(acc);
  }
  if (c + NOUT <= w && (sizeof(T) * NOUT == 16) && ((reinterpret_cast<uintptr_t>(o + c) & 15) == 0)) {
    union { uint4 v; T e[NOUT]; } u;
#pragma unroll
    for (int k = 0; k < NOUT; ++k) u.e[k] = res[k];
    *reinterpret_cast<uint4*>(o + c) = u.v;
  } else {
#pragma unroll
    for (int k = 0; k < NOUT; ++k)
      if (c + k < w) o[c + k] = res[k];
  }
}

// ------------------------------------------------------------------------- generic (any radius)
// One output per lane, taps streamed from global/L2 in scipy's order.  Correct for every radius,
// dtype and tiny frames (multiple reflections); used when no specialised instance exists.
template <typename T>
__global__ void __launch_bounds__(kThreads)
gauss_generic(const T* __restrict__ in, T* __restrict__ out, int64_t total, int h, int w, int axis,
              const double* __restrict__ wts, int rad) {
  int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % w);
  const int64_t t = i / w;
  const int r = (int)(t % h);
  const T* f = in + (t / h) * (size_t)h * w;
  double acc;
  if (axis == 0) {
    acc = (double)f[(size_t)r * w + c] * wts[rad];
    for (int j = rad; j >= 1; --j) {
      double a = (double)f[(size_t)pl_reflect(r - j, h) * w + c];
      double b = (double)f[(size_t)pl_reflect(r + j, h) * w + c];
      acc = acc + (a + b) * wts[rad - j];
    }
  } else {
    const T* frow = f + (size_t)r * w;
    acc = (double)frow[c] * wts[rad];
    for (int j = rad; j >= 1; --j) {
      double a = (double)frow[pl_reflect(c - j, w)];
      double b = (double)frow[pl_reflect(c + j, w)];
      acc = acc + (a + b) * wts[rad - j];
    }
  }
  out[i] = pl_from_double<T>(acc);
}

template <typename T, int RAD>
int launch_fast(const T* in, T* out, int64_t n, int h, int w, int axis, const double* wts,
                hipStream_t st) {
  constexpr int NOUT = 8;
  if (axis == 0) {
    int col_tiles = (int)pl_cdiv(w, kThreads);
    int row_groups = (int)pl_cdiv(h, NOUT);
    int64_t blocks = n * col_tiles * row_groups;
    if (blocks > 0x7fffffffLL) return -1;
    hipLaunchKernelGGL((gauss_v_fast<T, RAD, NOUT>), dim3((unsigned)blocks), dim3(kThreads), 0, st,
                       in, out, h, w, col_tiles, row_groups, wts);
  } else {
    int col_tiles = (int)pl_cdiv(w, PL_WAVE * NOUT);
    int64_t rows_total = n * h;
    int64_t blocks = pl_cdiv(rows_total, kThreads / PL_WAVE) * col_tiles;
    if (blocks > 0x7fffffffLL) return -1;
    hipLaunchKernelGGL((gauss_h_fast<T, RAD, NOUT>), dim3((unsigned)blocks), dim3(kThreads), 0, st,
                       in, out, rows_total, w, col_tiles, wts);
  }
  return 0;
}

template <typename T>
int gaussian1d_t(const T* in, T* out, int64_t n, int h, int w, int axis, const double* wts,
                 int radius, hipStream_t st) {
  int rc = -1;
  // specialised instances: radius = int(4*sigma+0.5) for sigma 1, 2, 3, 5
  const bool big_enough = (axis == 0 ? h : w) >= 1;
  if (big_enough && sizeof(T) == 2) {
    switch (radius) {
      case 4: rc = launch_fast<T, 4>(in, out, n, h, w, axis, wts, st); break;
      case 8: rc = launch_fast<T, 8>(in, out, n, h, w, axis, wts, st); break;
      case 12: rc = launch_fast<T, 12>(in, out, n, h, w, axis, wts, st); break;
      case 20: rc = launch_fast<T, 20>(in, out, n, h, w, axis, wts, st); break;
      default: break;
    }
  }
  if (rc != 0) {
    int64_t total = n * (int64_t)h * w;
    int64_t blocks = pl_cdiv(total, kThreads);
    if (blocks > 0x7fffffffLL) {
      pl_set_error("pl_gaussian1d: batch too large for one launch");
      return PL_ERR_INVALID_ARG;
    }
    hipLaunchKernelGGL(gauss_generic<T>, dim3((unsigned)blocks), dim3(kThreads), 0, st, in, out,
                       total, h, w, axis, wts, radius);
  }
  return pl_check_launch("pl_gaussian1d");
}

}  // namespace

extern "C" int pl_gaussian1d(const void* in, void* out, int dtype, int64_t n, int h, int w,
                             int axis, const double* d_weights, int radius, void* stream) {
  PL_REQUIRE(in && out && d_weights, "null pointer");
  PL_REQUIRE(in != out, "in-place operation is not supported");
  PL_REQUIRE(n >= 0 && h > 0 && w > 0, "bad shape");
  PL_REQUIRE(axis == 0 || axis == 1, "axis must be 0 or 1");
  PL_REQUIRE(radius >= 0, "negative radius");
  if (n == 0) return PL_OK;
  hipStream_t st = (hipStream_t)stream;
  PL_DISPATCH_DTYPE(dtype, T,
                    return gaussian1d_t<T>((const T*)in, (T*)out, n, h, w, axis, d_weights, radius, st));
  return PL_OK;
}

extern "C" int pl_gaussian2d(const void* in, void* out, void* tmp, int dtype, int64_t n, int h,
                             int w, const double* d_weights, int radius, void* stream) {
  PL_REQUIRE(tmp && tmp != in && tmp != out, "tmp must be a distinct buffer");
  int rc = pl_gaussian1d(in, tmp, dtype, n, h, w, 0, d_weights, radius, stream);
  if (rc != PL_OK) return rc;
  return pl_gaussian1d(tmp, out, dtype, n, h, w, 1, d_weights, radius, stream);
}
