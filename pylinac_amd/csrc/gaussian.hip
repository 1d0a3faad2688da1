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
// The exact sequence is 61 float64 VALU operations per pixel per pass at sigma=5 (radius 20): the
// pass is bound by the FP64 issue rate (16 lanes/clk/SIMD), not by HBM -- see DESIGN.md.  For integer
// images a 41-op float64 FMA chain decides the truncated result whenever it is provably the same
// (eval_window), the exact chain runs only on the ambiguous (constant/saturated) regions.  Everything
// that is NOT one of those float64 operations is minimised:
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

template <typename T> struct IsInt { static constexpr bool value = true; };
template <> struct IsInt<float> { static constexpr bool value = false; };
template <> struct IsInt<double> { static constexpr bool value = false; };

// scipy's exact float64 sequence for one output centred at x[c] (no FMA; -ffp-contract=off).
// OPAQUE = true launders every operand through an empty asm so that the compiler cannot share the
// pair sums with the FMA chain of eval_window (sharing them keeps 8x20 doubles alive -> spills).
template <int RAD, bool OPAQUE>
__device__ __forceinline__ double taps_exact(const double* x, int c, const double* __restrict__ wts) {
  auto ld = [](double v) -> double {
    if constexpr (OPAQUE) asm volatile("" : "+v"(v));
    return v;
  };
  double acc = ld(x[c]) * wts[RAD];
#pragma unroll
  for (int j = RAD; j >= 1; --j) acc = acc + (ld(x[c - j]) + ld(x[c + j])) * wts[RAD - j];
  return acc;
}

// NOUT outputs from a register window x[NOUT + 2*RAD].
// Integer dtypes: the result is trunc(S) where S is scipy's 61-op sequence.  A float64 FMA chain
// (41 ops: the pair sums are exact either way) gives S' with |S' - S| <= (2*RAD+2) * 2^-52 * max|partial|,
// so trunc(S') == trunc(S) unless S' lies within that bound of an integer -- which happens on
// constant / saturated regions (S ~= c * sum(w) = c +- 1e-12), practically never on noisy data.
// Only then is the exact sequence evaluated (for all NOUT outputs of the lane; rare, wave-coherent).
// Float dtypes have no truncation to hide behind: always the exact sequence.
template <typename T, int RAD, int NOUT>
__device__ __forceinline__ void eval_window(const double* x, const double* __restrict__ wts, T* res) {
  if constexpr (!IsInt<T>::value) {
#pragma unroll
    for (int i = 0; i < NOUT; ++i) res[i] = pl_from_double<T>(taps_exact<RAD, false>(x, i + RAD, wts));
  } else {
    double acc[NOUT];
    bool ambiguous = false;
    constexpr double kRel = (2 * RAD + 4) * 2.220446049250313e-16;  // 2x margin on the bound
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
      double a = x[i + RAD] * wts[RAD];
#pragma unroll
      for (int j = RAD; j >= 1; --j) a = __builtin_fma(x[i + RAD - j] + x[i + RAD + j], wts[RAD - j], a);
      acc[i] = a;
      // |fract(|a|) - 0.5| > 0.5 - eps  <=>  a within eps of an integer (4 f64 ops)
      const double mag = __builtin_fabs(a);
      const double eps = (sizeof(T) <= 2) ? kRel * 65536.0 : kRel * mag;
      const double off = __builtin_fabs(__builtin_amdgcn_fract(mag) - 0.5);
      // a == 0 exactly: every product is zero (or cancels to < 1e-9) -> trunc is 0 either way
      ambiguous |= (a != 0.0) && (off > 0.5 - eps);
    }
    if (ambiguous) {
#pragma unroll
      for (int i = 0; i < NOUT; ++i) acc[i] = taps_exact<RAD, true>(x, i + RAD, wts);
    }
#pragma unroll
    for (int i = 0; i < NOUT; ++i) res[i] = pl_from_double<T>(acc[i]);
  }
}

// ------------------------------------------------------------------ vertical (axis 0), fast path
template <typename T, int RAD, int NOUT>
__global__ void __launch_bounds__(kThreads, 4)
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

  // row pointers are wave-uniform (SALU address arithmetic); the lane's 32-bit byte offset is
  // constant -> global_load saddr + voffset form, no per-load VALU address arithmetic
  const unsigned coff = (unsigned)c * (unsigned)sizeof(T);
  double x[NOUT + 2 * RAD];
  if (r0 - RAD >= 0 && r0 + NOUT + RAD <= h) {
    const T* row = f + (size_t)(r0 - RAD) * w;
#pragma unroll
    for (int k = 0; k < NOUT + 2 * RAD; ++k) {
      x[k] = (double)*reinterpret_cast<const T*>(reinterpret_cast<const char*>(row) + coff);
      row += w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < NOUT + 2 * RAD; ++k) {
      const T* row = f + (size_t)pl_reflect(r0 - RAD + k, h) * w;  // wave-uniform
      x[k] = (double)*reinterpret_cast<const T*>(reinterpret_cast<const char*>(row) + coff);
    }
  }
  T res[NOUT];
  eval_window<T, RAD, NOUT>(x, wts, res);
  T* orow = o + (size_t)r0 * w;
#pragma unroll
  for (int i = 0; i < NOUT; ++i) {
    if (r0 + i < h) *reinterpret_cast<T*>(reinterpret_cast<char*>(orow) + coff) = res[i];
    orow += w;
  }
}

// ---------------------------------------------------------------- horizontal (axis 1), fast path
// LDS layout per wave: logical position p in [0, 64*NOUT + 2*RAD) <-> column c0 - RAD + p,
// stored at p + 2*(p >> 3) doubles (16-byte pad after every 64 bytes).
__device__ __forceinline__ constexpr int pad8(int p) { return p + ((p >> 3) << 1); }

template <typename T, int RAD, int NOUT>
__global__ void __launch_bounds__(kThreads, 4)
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
    for (int k = 0; k < NOUT; ++k) s[10 * lane + pad8(RAD + k)] = (double)u.e[k];
  } else {
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
      int cc = pl_reflect(c + k, w);
      s[10 * lane + pad8(RAD + k)] = (double)f[cc];
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
  eval_window<T, RAD, NOUT>(x, wts, res);
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
              const double* __restrict__ wts, int rad, int mode) {
  // mode 0: scipy 'reflect'; 1: 'nearest' (skimage.filters.gaussian's default); 2: 'constant' with cval 0
  // (skimage.feature.canny smooths with mode='constant'): an index outside the line reads 0
  auto border = [mode](int i, int n) {
    if (mode == 0) return pl_reflect(i, n);
    if (mode == 1) return i < 0 ? 0 : (i >= n ? n - 1 : i);
    return (i < 0 || i >= n) ? -1 : i;
  };
  int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % w);
  const int64_t t = i / w;
  const int r = (int)(t % h);
  const T* f = in + (t / h) * (size_t)h * w;
  double acc;
  if (axis == 0 && r - rad >= 0 && r + rad < h) {           // interior: no border arithmetic per tap
    const T* p = f + (size_t)r * w + c;
    acc = (double)p[0] * wts[rad];
    for (int j = rad; j >= 1; --j) acc = acc + ((double)p[-(ptrdiff_t)j * w] + (double)p[(ptrdiff_t)j * w]) * wts[rad - j];
  } else if (axis == 1 && c - rad >= 0 && c + rad < w) {
    const T* p = f + (size_t)r * w + c;
    acc = (double)p[0] * wts[rad];
    for (int j = rad; j >= 1; --j) acc = acc + ((double)p[-j] + (double)p[j]) * wts[rad - j];
  } else if (axis == 0) {
    acc = (double)f[(size_t)r * w + c] * wts[rad];
    for (int j = rad; j >= 1; --j) {
      const int ia = border(r - j, h), ib = border(r + j, h);
      double a = ia < 0 ? 0.0 : (double)f[(size_t)ia * w + c];
      double b = ib < 0 ? 0.0 : (double)f[(size_t)ib * w + c];
      acc = acc + (a + b) * wts[rad - j];
    }
  } else {
    const T* frow = f + (size_t)r * w;
    acc = (double)frow[c] * wts[rad];
    for (int j = rad; j >= 1; --j) {
      const int ia = border(c - j, w), ib = border(c + j, w);
      double a = ia < 0 ? 0.0 : (double)frow[ia];
      double b = ib < 0 ? 0.0 : (double)frow[ib];
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

}  // namespace

// gaussian_rw.hip: register-window packed-float32 decision kernels for 16-bit images.  covers() = 1 when the launcher
// takes this call; launch(): 0 = launched, -1 = not covered.  The launcher wants the taps in HOST memory (they travel as
// kernel arguments).
int pl_gauss_rw_covers(const void* in, const void* out, int h, int w, int axis, int radius);
int pl_gauss_rw_launch(const void* in, void* out, int is_signed, int64_t n, int h, int w, int axis,
                       const double* h_wts, int radius, hipStream_t st);

// gaussian_mm.hip: exact-integer matrix-core kernel for 16-bit frames, radius <= 24, even width, at least 64 x 64:
// both axes in one kernel (the axis-0 result stays in LDS); same protocol
int pl_gauss_mm2d_covers(const void* in, const void* out, int h, int w, int radius);
int pl_gauss_mm2d_launch(const void* in, void* out, int is_signed, int64_t n, int h, int w, const double* h_wts, int radius,
                         hipStream_t st);

namespace {
// 16-bit frames, reflect mode, radius 4 / 8 / 12 / 20, even width (axis 0) or width % 16 == 0 (axis 1): the register-
// window kernels of gaussian_rw.hip (256 x 1024^2, sigma 5 on MI355X: 0.46 / 0.42 ms per pass in their first version
// against 0.61 / 0.61 for the float64 kernels below and 0.56 / 0.47 for round 1's LDS-tile packed kernels, which are
// gone).  Everything else -- other dtypes, radii, border modes, ragged widths -- runs the float64 kernels.
template <typename T>
int gaussian1d_t(const T* in, T* out, int64_t n, int h, int w, int axis, const double* wts, const double* h_wts,
                 int radius, hipStream_t st, int mode = 0) {
  int rc = -1;
  // the register-window kernels take the taps as kernel arguments: they need the HOST copy.  Without one (h_wts == NULL)
  // nothing is fetched back -- no hidden stream synchronisation, legal under stream capture -- and the float64 kernels
  // below, which read the device copy, produce the same frames
  if (sizeof(T) == 2 && mode == 0 && h_wts && pl_gauss_rw_covers(in, out, h, w, axis, radius)) {
    rc = pl_gauss_rw_launch(in, out, (T)-1 < (T)0, n, h, w, axis, h_wts, radius, st);
    if (rc == 0) return pl_check_launch("pl_gaussian1d");
  }
  // specialised instances: radius = int(4*sigma+0.5) for sigma 1, 2, 3, 5
  const bool big_enough = (axis == 0 ? h : w) >= 1;
  if (big_enough && sizeof(T) == 2 && mode == 0) {
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
                       total, h, w, axis, wts, radius, mode);
  }
  return pl_check_launch("pl_gaussian1d");
}

}  // namespace

extern "C" int pl_gaussian1d(const void* in, void* out, int dtype, int64_t n, int h, int w, int axis,
                             const double* d_weights, const double* h_weights, int radius, void* stream) {
  PL_REQUIRE(in && out && d_weights, "null pointer");
  PL_REQUIRE(in != out, "in-place operation is not supported");
  PL_REQUIRE(n >= 0 && h > 0 && w > 0, "bad shape");
  PL_REQUIRE(axis == 0 || axis == 1, "axis must be 0 or 1");
  PL_REQUIRE(radius >= 0, "negative radius");
  if (n == 0) return PL_OK;
  hipStream_t st = (hipStream_t)stream;
  PL_DISPATCH_DTYPE(dtype, T,
                    return gaussian1d_t<T>((const T*)in, (T*)out, n, h, w, axis, d_weights, h_weights, radius, st));
  return PL_OK;
}

extern "C" int pl_gaussian2d_mode(const void* in, void* out, void* tmp, int dtype, int64_t n, int h, int w,
                                  const double* d_weights, int radius, int mode, void* stream) {
  PL_REQUIRE(in && out && tmp && d_weights, "null pointer");
  PL_REQUIRE(tmp != in && tmp != out && in != out, "buffers must be distinct");
  PL_REQUIRE(n >= 0 && h > 0 && w > 0 && radius >= 0, "bad shape");
  PL_REQUIRE(mode >= 0 && mode <= 2, "mode must be 0 (reflect), 1 (nearest) or 2 (constant, cval 0)");
  if (n == 0) return PL_OK;
  hipStream_t st = (hipStream_t)stream;
  PL_DISPATCH_DTYPE(dtype, T, {
    int rc = gaussian1d_t<T>((const T*)in, (T*)tmp, n, h, w, 0, d_weights, nullptr, radius, st, mode);
    if (rc != PL_OK) return rc;
    return gaussian1d_t<T>((const T*)tmp, (T*)out, n, h, w, 1, d_weights, nullptr, radius, st, mode);
  });
  return PL_OK;
}

extern "C" int pl_gaussian2d(const void* in, void* out, void* tmp, int dtype, int64_t n, int h, int w,
                             const double* d_weights, const double* h_weights, int radius, void* stream) {
  PL_REQUIRE(tmp && tmp != in && tmp != out, "tmp must be a distinct buffer");
  // both axes in one launch on the matrix cores: needs the host copy of the taps (kernel arguments); without it the two
  // passes below run (float64 kernels on the device copy) -- never a device-to-host fetch inside a launch path
  if ((dtype == PL_U16 || dtype == PL_I16) && in && out && d_weights && h_weights && in != out && n > 0 && h > 0 && w > 0 &&
      pl_gauss_mm2d_covers(in, out, h, w, radius)) {
    hipStream_t st = (hipStream_t)stream;
    if (pl_gauss_mm2d_launch(in, out, dtype == PL_I16, n, h, w, h_weights, radius, st) == 0) return pl_check_launch("pl_gaussian2d");
  }
  int rc = pl_gaussian1d(in, tmp, dtype, n, h, w, 0, d_weights, h_weights, radius, stream);
  if (rc != PL_OK) return rc;
  return pl_gaussian1d(tmp, out, dtype, n, h, w, 1, d_weights, h_weights, radius, stream);
}
