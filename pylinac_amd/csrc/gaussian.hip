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
#include <stdlib.h>

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

// ------------------------------------------------- horizontal pass fused with the 3x3 median filter
// BaseImage.filter(s, "gaussian") followed by BaseImage.filter(3, "median") (pylinac/core/image.py:
// 695-712 twice; the PF noise filter is the size-3 median, pylinac/picketfence.py:226).
// One 768-lane workgroup owns TR output rows of one frame over the full width:
//   phase 1  its 12 waves compute the horizontal Gaussian of the TR+2 rows the median needs
//            (same per-wave LDS-staged window evaluation as gauss_h_fast) and keep the truncated
//            results in LDS (u16/i16 rows, never written to HBM);
//   phase 2  lane = column slides down the LDS rows: sorted horizontal triples, median of nine =
//            med3(max3(lows), med3(mids), min3(highs)); one coalesced global store per row.
// HBM traffic: reads the axis-0 result once (+2 halo rows per band from L2), writes the median
// once -- the unfused pair costs an extra frame write + read.  The (TR+2)/TR recompute is the price.
constexpr int kFusedThreads = 768;  // 12 waves = 3 per SIMD: <=168 VGPRs, no spills in the task loop

template <typename T, int RAD>
__global__ void __launch_bounds__(kFusedThreads)
gauss_h_median3_kernel(const T* __restrict__ in, T* __restrict__ out, int h, int w, int tr, int bands,
                       int hpitch, const double* __restrict__ wts) {
  constexpr int NOUT = 8;
  constexpr int SEG = PL_WAVE * NOUT;
  constexpr int LOGICAL = SEG + 2 * RAD;
  constexpr int PADDED = LOGICAL + ((LOGICAL + 7) / 8) * 2;
  constexpr int WAVES = kFusedThreads / PL_WAVE;
  extern __shared__ __attribute__((aligned(16))) unsigned char fused_smem[];
  double* stage = reinterpret_cast<double*>(fused_smem);                 // [WAVES][PADDED]
  T* hrows = reinterpret_cast<T*>(stage + WAVES * PADDED);               // [tr+2][hpitch]

  const unsigned id = pl_xcd_remap(blockIdx.x, gridDim.x);
  const int band = id % bands;
  const size_t frame = id / bands;
  const int r0 = band * tr;
  const int rows_here = (r0 + tr <= h) ? tr : (h - r0);
  const T* f = in + frame * (size_t)h * w;
  T* o = out + frame * (size_t)h * w;
  const int lane = threadIdx.x & (PL_WAVE - 1);
  const int wave = threadIdx.x / PL_WAVE;
  const int segs = (w + SEG - 1) / SEG;
  const int tasks = (rows_here + 2) * segs;
  double* s = stage + wave * PADDED;

  for (int task = wave; task < tasks; task += WAVES) {
    const int k = task / segs;               // LDS row 0..rows_here+1  <->  frame row r0-1+k
    const int c0 = (task % segs) * SEG;
    const T* frow = f + (size_t)pl_reflect(r0 - 1 + k, h) * w;
    const int c = c0 + lane * NOUT;
    if (c + NOUT <= w && ((reinterpret_cast<uintptr_t>(frow + c) & 15) == 0)) {
      union { uint4 v; T e[NOUT]; } u;
      u.v = *reinterpret_cast<const uint4*>(frow + c);
#pragma unroll
      for (int q = 0; q < NOUT; ++q) s[10 * lane + pad8(RAD + q)] = (double)u.e[q];
    } else {
#pragma unroll
      for (int q = 0; q < NOUT; ++q) s[10 * lane + pad8(RAD + q)] = (double)frow[pl_reflect(c + q, w)];
    }
    if (lane < RAD) {
      s[pad8(lane)] = (double)frow[pl_reflect(c0 - RAD + lane, w)];
    } else if (lane < 2 * RAD) {
      const int p = SEG + lane;
      s[pad8(p)] = (double)frow[pl_reflect(c0 - RAD + p, w)];
    }
    // order this wave's LDS writes before its LDS reads (in-order per wave; compiler barrier)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (c < w) {
      double x[NOUT + 2 * RAD];
      const double* win = s + 10 * lane;
#pragma unroll
      for (int q = 0; q < NOUT + 2 * RAD; ++q) x[q] = win[q + ((q >> 3) << 1)];
      union { uint4 v; T e[NOUT]; } r;
      eval_window<T, RAD, NOUT>(x, wts, r.e);
      *reinterpret_cast<uint4*>(hrows + (size_t)k * hpitch + c) = r.v;  // hpitch % 8 == 0: aligned
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();   // the staging window is reused by this wave's next task
  }
  __syncthreads();

  // phase 2: 3x3 median over the LDS rows.  Work item = (column pair, third of the band): the pair's
  // four neighbours come from three 32-bit LDS reads, the sorted horizontal triples of the last
  // three rows rotate through registers (loop unrolled by 3 so the slots are static).
  {
    const int npairs = (w + 1) >> 1;
    const int rs = (rows_here + 2) / 3;                   // rows per item
    const int items = npairs * 3;
    const unsigned* hw = reinterpret_cast<const unsigned*>(hrows);
    const int wpitch = hpitch >> 1;
    for (int it = threadIdx.x; it < items; it += kFusedThreads) {
      const int pi = it % npairs;
      const int i0 = (it / npairs) * rs;
      const int i1 = (i0 + rs < rows_here) ? i0 + rs : rows_here;
      if (i0 >= i1) continue;
      const int ca = 2 * pi;
      const bool interior = (pi >= 1) && (ca + 2 < w);
      const int c_m1 = pl_reflect(ca - 1, w), c_p1 = pl_reflect(ca + 1, w), c_p2 = pl_reflect(ca + 2, w);
      int lo0[3], mi0[3], hi0[3], lo1[3], mi1[3], hi1[3];
      auto load_row = [&](int k, int slot) {
        int a, b, c, d;
        if (interior) {
          const unsigned* p = hw + (size_t)k * wpitch + pi;
          const unsigned L = p[-1], C = p[0], R = p[1];
          a = (int)(T)(L >> 16); b = (int)(T)(C & 0xffffu); c = (int)(T)(C >> 16); d = (int)(T)(R & 0xffffu);
        } else {
          const T* p = hrows + (size_t)k * hpitch;
          a = (int)p[c_m1]; b = (int)p[ca]; c = (int)p[c_p1]; d = (int)p[c_p2];
        }
        lo0[slot] = min(min(a, b), c); hi0[slot] = max(max(a, b), c); mi0[slot] = pl_smed3(a, b, c);
        lo1[slot] = min(min(b, c), d); hi1[slot] = max(max(b, c), d); mi1[slot] = pl_smed3(b, c, d);
      };
      auto emit = [&](int i) {
        const int m0 = pl_smed3(max(max(lo0[0], lo0[1]), lo0[2]), pl_smed3(mi0[0], mi0[1], mi0[2]),
                                min(min(hi0[0], hi0[1]), hi0[2]));
        const int m1 = pl_smed3(max(max(lo1[0], lo1[1]), lo1[2]), pl_smed3(mi1[0], mi1[1], mi1[2]),
                                min(min(hi1[0], hi1[1]), hi1[2]));
        T* op = o + (size_t)(r0 + i) * w + ca;
        if (ca + 1 < w && ((reinterpret_cast<uintptr_t>(op) & 3) == 0)) {
          *reinterpret_cast<unsigned*>(op) = ((unsigned)m0 & 0xffffu) | ((unsigned)m1 << 16);
        } else {
          op[0] = (T)m0;
          if (ca + 1 < w) op[1] = (T)m1;
        }
      };
      // LDS row k holds frame row r0-1+k: output row i needs LDS rows i, i+1, i+2
      load_row(i0, 0);
      load_row(i0 + 1, 1);
      int i = i0;
      for (; i + 3 <= i1; i += 3) {
        load_row(i + 2, 2); emit(i);
        load_row(i + 3, 0); emit(i + 1);
        load_row(i + 4, 1); emit(i + 2);
      }
      if (i < i1) { load_row(i + 2, 2); emit(i); ++i; }
      if (i < i1) { load_row(i + 2, 0); emit(i); }
    }
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
  if (axis == 0) {
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

// gaussian_pk.hip: packed-float32 decision kernels for 16-bit images (0 = launched, -1 = not covered)
int pl_gauss_pk_launch(const void* in, void* out, int is_signed, int64_t n, int h, int w, int axis,
                       const double* wts, int radius, hipStream_t st);

namespace {
// Packed-float32 decision kernels (gaussian_pk.hip, bit-identical to the float64 kernels below): the default
// for 16-bit frames on both axes (256 x 1024^2, sigma 5: axis 1 0.48 vs 0.62 ms, axis 0 0.57 vs 0.61 ms; shapes
// they do not cover -- odd widths on axis 0, other radii -- fall through to the float64 kernels).
// PL_GAUSS_PK=0 pins the float64 kernels (A/B measurements, parity tests of both paths).
bool use_pk_path(int /*axis*/) {
  static const bool on = [] {
    const char* e = getenv("PL_GAUSS_PK");
    return !(e && e[0] == '0');
  }();
  return on;
}

template <typename T>
int gaussian1d_t(const T* in, T* out, int64_t n, int h, int w, int axis, const double* wts,
                 int radius, hipStream_t st, int mode = 0) {
  int rc = -1;
  if (sizeof(T) == 2 && mode == 0 && use_pk_path(axis)) {
    rc = pl_gauss_pk_launch(in, out, (T)-1 < (T)0, n, h, w, axis, wts, radius, st);
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

namespace {
template <typename T, int RAD>
int launch_fused(const T* in, T* out, int64_t n, int h, int w, const double* wts, hipStream_t st) {
  constexpr int PADDED = (512 + 2 * RAD) + ((512 + 2 * RAD + 7) / 8) * 2;
  const int hpitch = (w + 7) & ~7;
  const size_t stage_bytes = (size_t)(kFusedThreads / PL_WAVE) * PADDED * sizeof(double);
  const size_t lds_budget = 158 * 1024 - stage_bytes;
  int tr = (int)(lds_budget / ((size_t)hpitch * sizeof(T))) - 2;
  if (tr > 46) tr = 46;
  if (tr > h) tr = h;
  if (tr < 4) return -1;  // very wide frames: unfused path
  // prefer a band height whose (tr+2) x segments task count divides evenly over the waves
  {
    const int segs = (int)pl_cdiv(w, 512), waves = kFusedThreads / PL_WAVE;
    for (int t = tr; t >= tr - 8 && t >= 4; --t)
      if (((t + 2) * segs) % waves == 0) { tr = t; break; }
  }
  const size_t lds = stage_bytes + (size_t)(tr + 2) * hpitch * sizeof(T);
  const int bands = (int)pl_cdiv(h, tr);
  if (n * bands > 0x7fffffffLL) return -1;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)gauss_h_median3_kernel<T, RAD>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      (void)hipGetLastError();
      return -1;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL((gauss_h_median3_kernel<T, RAD>), dim3((unsigned)(n * bands)), dim3(kFusedThreads), lds,
                     st, in, out, h, w, tr, bands, hpitch, wts);
  return 0;
}
}  // namespace

// Horizontal Gaussian pass fused with a 3x3 median:  out = median3(gauss_axis1(in)).
// Falls back to the two separate kernels (through tmp) for unsupported dtype/radius/width.
extern "C" int pl_gauss_h_median3(const void* in, void* out, void* tmp, int dtype, int64_t n, int h,
                                  int w, const double* d_weights, int radius, void* stream);
extern "C" int pl_median2d(const void* in, void* out, int dtype, int64_t n, int h, int w, int size,
                           void* stream);

extern "C" int pl_gauss_h_median3(const void* in, void* out, void* tmp, int dtype, int64_t n, int h,
                                  int w, const double* d_weights, int radius, void* stream) {
  PL_REQUIRE(in && out && tmp && d_weights, "null pointer");
  PL_REQUIRE(in != out && tmp != in && tmp != out, "buffers must be distinct");
  PL_REQUIRE(n >= 0 && h > 0 && w > 0 && radius >= 0, "bad shape");
  if (n == 0) return PL_OK;
  hipStream_t st = (hipStream_t)stream;
  int rc = -1;
  if (h > 1 && (dtype == PL_U16 || dtype == PL_I16)) {
#define PL_FUSED_CASE(R)                                                                                   \
  case R:                                                                                                  \
    rc = (dtype == PL_U16)                                                                                 \
             ? launch_fused<unsigned short, R>((const unsigned short*)in, (unsigned short*)out, n, h, w,  \
                                               d_weights, st)                                              \
             : launch_fused<short, R>((const short*)in, (short*)out, n, h, w, d_weights, st);             \
    break;
    switch (radius) {
      PL_FUSED_CASE(4)
      PL_FUSED_CASE(8)
      PL_FUSED_CASE(12)
      PL_FUSED_CASE(20)
      default: break;
    }
#undef PL_FUSED_CASE
  }
  if (rc == 0) return pl_check_launch("pl_gauss_h_median3");
  rc = pl_gaussian1d(in, tmp, dtype, n, h, w, 1, d_weights, radius, stream);
  if (rc != PL_OK) return rc;
  return pl_median2d(tmp, out, dtype, n, h, w, 3, stream);
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
    int rc = gaussian1d_t<T>((const T*)in, (T*)tmp, n, h, w, 0, d_weights, radius, st, mode);
    if (rc != PL_OK) return rc;
    return gaussian1d_t<T>((const T*)tmp, (T*)out, n, h, w, 1, d_weights, radius, st, mode);
  });
  return PL_OK;
}

extern "C" int pl_gaussian2d(const void* in, void* out, void* tmp, int dtype, int64_t n, int h,
                             int w, const double* d_weights, int radius, void* stream) {
  PL_REQUIRE(tmp && tmp != in && tmp != out, "tmp must be a distinct buffer");
  int rc = pl_gaussian1d(in, tmp, dtype, n, h, w, 0, d_weights, radius, stream);
  if (rc != PL_OK) return rc;
  return pl_gaussian1d(tmp, out, dtype, n, h, w, 1, d_weights, radius, stream);
}
