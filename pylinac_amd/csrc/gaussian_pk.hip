// Packed-FP32 decision path for the scipy-exact Gaussian on 16-bit images.
//
// Replaces: scipy.ndimage.gaussian_filter on uint16 / int16 frames as called at
// pylinac/core/array_utils.py:133 (BaseImage.filter(kind="gaussian"), pylinac/core/image.py:695-712).
//
// The contract (gaussian.hip header) is trunc(S), S = scipy's float64 tap sequence.  The float64 chain is
// bound by the FP64 issue rate (one wave instruction per 4 clocks); v_pk_fma_f32 issues at the same rate
// but carries two pixels.  So the result is DECIDED in packed float32 and only the undecidable pixels see
// float64:
//
//   S = m*W + t,  t = sum_j w_j * (x[-j] + x[+j] - 2m) + w_0 * (x[0] - m)       (real arithmetic)
//
//   m   a lower bound of every sample in the window (per-column minimum of the staged tile in the axis-0 pass,
//       minimum of the 8-sample blocks the lane's own window covers in the axis-1 pass), so every term of t is >= 0 and the partial sums of the
//       float32 chain grow monotonically: |t_hat - t| <= (RAD + 2) * 2^-24 * t   (RAD+1 roundings of the
//       chain + the float32 rounding of the taps; the pair sums and x - m are exact, < 2^24);
//   W   = w_0 + 2*sum w_j in float64; |m * (W - 1)| <= 65535 * |W - 1| is added to the margin (unnormalised
//       taps make the margin > 0.5: every pixel then takes the exact path -- slow, still right);
//   =>  trunc(S) = m + floor(t_hat) whenever frac(t_hat) is farther than
//       delta = 1.02 * (RAD + 2) * 2^-24 * t_hat + 65535 * |W - 1| + 1e-6 from 0 and 1.
//
// Typical EPID content (1 % noise on a 40 000 plateau) leaves ~0.7 % of the pixels undecided.  Their codes go
// on a workgroup list in LDS; after one barrier the listed pixels are recomputed with scipy's exact float64
// sequence FROM THE STAGED LDS TILE (x = x' + m is exact; the reflected halo is already there), 64 per wave
// pass, and overwrite the first-pass stores.  A workgroup whose list overflows (constant or saturated tiles:
// S sits within 1e-11 of an integer everywhere) recomputes its whole tile that way.  (Measured dead ends: the
// same fix-up fed by global gathers doubled the kernel time -- load latency with three waves parked at the
// barrier -- and a deferred second launch spent 5-14 ms on 41 scattered cache lines per pixel.)
// Bit-identical to gauss_v_fast / gauss_h_fast by construction; tests compare against scipy on noisy,
// constant, saturated and ragged frames.
//
// STATUS: default for 16-bit frames on both axes (256 x 1024^2, sigma 5, MI355X: axis 1 0.48 ms vs 0.62 ms for the
// float64 kernel; axis 0 0.57 vs 0.61 ms -- its three-barrier structure sits at 53 % VALU utilisation; taking the
// axis-0 offset from block minima like axis 1, which removes the first barrier, measured 0.59 ms: no gain).
// PL_GAUSS_PK=0 pins the float64 kernels.  History (profiles/r01c_*): with a wave-wide row
// minimum the axis-1 kernel left ~6 % of the in-field pixels undecided (every 552-sample span that contains a
// field edge) and ran 0.54 ms; the lane-local minimum below brought that to 0.51 ms (0.48 without the wave-wide offset) once the register spills it
// first caused (20 scratch accesses per wave doubled the run time) were removed.  Axis-0 variants that were
// measured and dropped: a persistent strip-walking version of the LDS-tile kernel with register prefetch
// (0.84 ms); a barrier-free sliding REGISTER window (one wave walks down 64 column pairs, window shift and
// re-offset fused into one packed add per sample, undecided outputs re-read from a per-wave LDS dump of the
// window): 46.8 VALU/px but 0.75-0.92 ms -- 96 window registers + prefetch leave two waves per SIMD, the
// compiler still spills, and the steps that cross a field edge (every lane undecided) serialise.
//
// u16 -> f32 without cvt instructions: (x16 | 0x4B000000) is the float 2^23 + x16; one v_pk_add_f32 with
// -(2^23 + m) yields x - m for two pixels (exact).  int16 is XOR-biased into the unsigned domain first.
#include <stdlib.h>

#include "pl_common.h"

namespace {

constexpr int kPkThreads = 256;
constexpr int kListCap = 512;

typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }

// Undecided pixels, from LDS-staged m-subtracted float samples: v(k) = sample at window position k
// (0 .. 2*RAD, centre RAD), m = the subtracted minimum, wsum = float64 sum of all taps.
// Tier 2: a float64 FMA chain over v plus m*wsum differs from scipy's sequence S by less than 4e-9 for
// 16-bit data ((4*RAD + 8) * 2^-53 * 65536 < 7e-10), so it decides trunc(S) unless it lands that close to
// an integer.  Tier 3 (constant / saturated neighbourhoods): scipy's exact sequence -- x = v + m is an
// exact small integer in float64, so every pair sum, product and running sum rounds as scipy's does.
template <int RAD, typename F>
__device__ __forceinline__ double exact_from_lds(F v, double m, double wsum, const double* __restrict__ wts) {
  // samples are re-read per tier instead of being held: 41 doubles would cost 82 registers on top of the caller's
  double a = (double)v(RAD) * wts[RAD];
#pragma unroll
  for (int j = RAD; j >= 1; --j) a = __builtin_fma((double)v(RAD - j) + (double)v(RAD + j), wts[RAD - j], a);
  a = __builtin_fma(m, wsum, a);
  const double off = __builtin_fabs(__builtin_amdgcn_fract(__builtin_fabs(a)) - 0.5);
  if (off > 0.5 - 4e-9) {
    a = ((double)v(RAD) + m) * wts[RAD];
#pragma unroll
    for (int j = RAD; j >= 1; --j)
      a = a + (((double)v(RAD - j) + m) + ((double)v(RAD + j) + m)) * wts[RAD - j];
  }
  return a;
}

// taps as float32 in SGPRs + the margin constants (wave-uniform)
template <int RAD>
struct PkTaps {
  float w[RAD + 1];  // w[j] = tap at offset j (j = 0 centre)
  float c0, c1;
};

// s_wf: the taps converted to float32 ONCE per workgroup (wave 0, before the staging barrier); every wave then
// reads them from LDS -- 21 v_cvt_f32_f64 per wave in the hot segment were ~10 % of its issue time
template <int RAD>
__device__ __forceinline__ void load_taps(const float* __restrict__ s_wf, PkTaps<RAD>& t, float c0) {
#pragma unroll
  for (int j = 0; j <= RAD; ++j) t.w[j] = s_wf[j];
  t.c0 = c0;
  t.c1 = (RAD + 2) * 5.9604645e-08f * 1.02f;
}
template <int RAD>
__device__ __forceinline__ void publish_taps(const double* __restrict__ wts, float* __restrict__ s_wf, int lane) {
  if (lane <= RAD) s_wf[lane] = (float)wts[RAD - lane];
}

// margin constant c0 = 65535*|W-1| + 1e-6 (float64 sum of the taps; its own error <= 1e-14 is added)
template <int RAD>
__device__ __forceinline__ float margin_c0(const double* __restrict__ wts, double& wsum) {
  double s = 0.0;
  bool nonneg = wts[RAD] >= 0.0;
#pragma unroll
  for (int j = RAD; j >= 1; --j) {
    s += wts[RAD - j];
    nonneg = nonneg && (wts[RAD - j] >= 0.0);
  }
  s = wts[RAD] + 2.0 * s;
  wsum = s;
  double c = 65535.0 * (__builtin_fabs(s - 1.0) + 1e-14) + 1e-6;
  // a negative tap breaks the monotone-partial-sum bound: margin > 0.5 sends every pixel to the exact path
  return (c > 4.0 || !nonneg) ? 4.0f : (float)c;
}

// One pair of outputs from a window of pair-packed, m-subtracted samples x[0 .. 2*RAD] centred at x[RAD].
// Returns the two truncated results packed as 2 x 16 bit (biased domain) and sets the fail bits.
template <int RAD>
__device__ __forceinline__ unsigned decide_pair(const f2* x, const PkTaps<RAD>& t, f2 mf, bool m0_zero,
                                                bool m1_zero, bool is_signed, unsigned& fail) {
  f2 acc = (x[0] + x[2 * RAD]) * splat(t.w[RAD]);
#pragma unroll
  for (int j = RAD - 1; j >= 1; --j)
    acc = __builtin_elementwise_fma(x[RAD - j] + x[RAD + j], splat(t.w[j]), acc);
  acc = __builtin_elementwise_fma(x[RAD], splat(t.w[0]), acc);

  const f2 fl = __builtin_elementwise_floor(acc);
  const f2 fr = acc - fl;
  const f2 lim = __builtin_elementwise_fma(acc, splat(-t.c1), splat(0.5f - t.c0));  // 0.5 - delta
  const f2 d = fr - splat(0.5f);
  // all-zero window over m == 0: S is exactly 0
  const bool ok0 = (__builtin_fabsf(d.x) < lim.x) || (m0_zero && acc.x == 0.0f);
  const bool ok1 = (__builtin_fabsf(d.y) < lim.y) || (m1_zero && acc.y == 0.0f);
  fail = (ok0 ? 0u : 1u) | (ok1 ? 0u : 2u);
  const f2 rf = fl + mf;  // integers < 2^17: exact
  unsigned r0 = (unsigned)rf.x, r1 = (unsigned)rf.y;
  if (is_signed) {  // biased value = floor(S) + 32768; C truncation rounds negative S toward zero
    r0 += (r0 < 32768u) ? 1u : 0u;
    r1 += (r1 < 32768u) ? 1u : 0u;
  }
  return (r0 & 0xffffu) | (r1 << 16);
}

// raw dword (two 16-bit samples) -> {2^23 + lo16, 2^23 + hi16} as floats
__device__ __forceinline__ f2 magic_pair(unsigned raw) {
  const unsigned lo = (raw & 0xffffu) | 0x4B000000u;
  const unsigned hi = __builtin_amdgcn_perm(0x4B000000u, raw, 0x070c0302u);
  return f2{__uint_as_float(lo), __uint_as_float(hi)};
}
__device__ __forceinline__ float magic_lo(unsigned raw) { return __uint_as_float((raw & 0xffffu) | 0x4B000000u); }
__device__ __forceinline__ float magic_hi(unsigned raw) {
  return __uint_as_float(__builtin_amdgcn_perm(0x4B000000u, raw, 0x070c0302u));
}

__device__ __forceinline__ us2 as_us2(unsigned v) {
  union { unsigned u; us2 s; } c;
  c.u = v;
  return c.s;
}

// workgroup list of undecided pixels
struct FixList {
  unsigned cnt;
  unsigned item[kListCap];
};
__device__ __forceinline__ void push_fails(FixList& fl, unsigned failmask, unsigned tid) {
  while (failmask) {  // lanes without undecided outputs skip the loop
    const int b = __builtin_ctz(failmask);
    failmask &= failmask - 1;
    const unsigned i = atomicAdd(&fl.cnt, 1u);
    if (i < (unsigned)kListCap) fl.item[i] = (tid << 5) | (unsigned)b;
  }
}

// ------------------------------------------------------------------------ axis 0 (vertical) pass
// Workgroup tile = 128 columns x kVRows rows (+ 2*RAD halo rows) staged ONCE through LDS as m-subtracted
// float pairs; lane = column pair, wave w owns kVRows/4 rows in groups of 8 outputs.
// m = per-column minimum over the whole staged tile (a valid lower bound for every window in it).
// OCC = workgroups per CU the register allocation is bounded for: 3 (134 VGPRs, the configuration measured in round 1)
// or 4 (128 VGPRs: one 8-byte value spilled and reloaded once per 8-row group; a fourth wave per SIMD to hide the LDS /
// barrier latency that keeps the VALU only ~half busy).  4 is selected with PL_GAUSS_V_OCC=4 until it has been timed.
template <typename T, int RAD, int kVRows, int NW, int OCC = 3>
__global__ void __launch_bounds__(NW * PL_WAVE, (NW == 8 ? 2 : (kVRows == 64 ? 2 : OCC)))
gauss_v_pk(const T* __restrict__ in, T* __restrict__ out, int h, int w, int col_tiles, int row_tiles,
           const double* __restrict__ wts) {
  constexpr int NOUT = 8, WIN = NOUT + 2 * RAD;
  constexpr int TROWS = kVRows + 2 * RAD;          // staged rows
  constexpr int WAVES = NW;
  constexpr int SHARE = kVRows / WAVES;             // output rows per wave
  constexpr int PER = (TROWS + WAVES - 1) / WAVES;  // staged rows per wave
  constexpr bool kSigned = (T)-1 < (T)0;
  constexpr unsigned kBias = kSigned ? 0x80008000u : 0u;
  __shared__ f2 tile[TROWS * PL_WAVE];              // [row][column pair]
  __shared__ unsigned s_min[(WAVES + 1) * PL_WAVE];  // per-wave partial minima, then the final ones
  __shared__ FixList fix;
  __shared__ float s_c0;
  __shared__ float s_wf[RAD + 1];
  __shared__ double s_wsum;

  unsigned id = pl_xcd_remap(blockIdx.x, gridDim.x);
  const int ct = id % col_tiles;
  id /= col_tiles;
  const int rt = id % row_tiles;
  const size_t frame = id / row_tiles;
  const int tid = threadIdx.x;
  const int lane = tid & (PL_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / PL_WAVE);  // scalar: row arithmetic stays on the SALU
  const int c = ct * (2 * PL_WAVE) + 2 * lane;
  const int r0 = rt * kVRows;
  const bool active = c < w;  // w is even: the pair is inside or outside together
  const T* f = in + frame * (size_t)h * w;
  T* o = out + frame * (size_t)h * w;

  if (tid == 0) fix.cnt = 0;
  if (wave == 0) {
    double ws;
    const float c0 = margin_c0<RAD>(wts, ws);
    publish_taps<RAD>(wts, s_wf, lane);
    if (tid == 0) {
      s_c0 = c0;
      s_wsum = ws;
    }
  }

  // ---- load: wave `wave` takes staged rows wave, wave+4, ... (row index wave-uniform -> scalar reflect)
  unsigned raw[PER];
  const unsigned coff = (unsigned)c * (unsigned)sizeof(T);
  if (r0 - RAD >= 0 && r0 - RAD + TROWS <= h) {   // interior tile: no reflection, one pointer bump per row
    const T* row = f + (size_t)(r0 - RAD + wave) * w;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      raw[k] = 0xffffffffu;   // neutral for the minimum (biased domain)
      if (wave + k * WAVES < TROWS && active)
        raw[k] = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(row) + coff) ^ kBias;
      row += (size_t)WAVES * w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int tr = wave + k * WAVES;
      raw[k] = 0xffffffffu;
      if (tr < TROWS && active) {
        const T* row = f + (size_t)pl_reflect(r0 - RAD + tr, h) * w;
        raw[k] = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(row) + coff) ^ kBias;
      }
    }
  }
  us2 mn = as_us2(raw[0]);
#pragma unroll
  for (int k = 1; k < PER; ++k) mn = __builtin_elementwise_min(mn, as_us2(raw[k]));
  {
    union { us2 s; unsigned u; } cv;
    cv.s = mn;
    s_min[wave * PL_WAVE + lane] = cv.u;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < WAVES; ++q) mn = __builtin_elementwise_min(mn, as_us2(s_min[q * PL_WAVE + lane]));
  if (wave == 0) {
    union { us2 s; unsigned u; } cv;
    cv.s = mn;
    s_min[WAVES * PL_WAVE + lane] = cv.u;
  }
  const f2 mf = f2{(float)mn.x, (float)mn.y};
  const f2 nb = -(splat(8388608.0f) + mf);
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int tr = wave + k * WAVES;
    if (tr < TROWS) tile[tr * PL_WAVE + lane] = magic_pair(raw[k]) + nb;
  }
  __syncthreads();

  PkTaps<RAD> taps;
  load_taps<RAD>(s_wf, taps, s_c0);
  const bool m0z = kSigned ? (mn.x == 32768) : (mn.x == 0);
  const bool m1z = kSigned ? (mn.y == 32768) : (mn.y == 0);

  unsigned failmask = 0;
#pragma unroll 1
  for (int g = 0; g < SHARE / NOUT; ++g) {
    const int lr = wave * SHARE + g * NOUT;  // first output row of the group, tile-local
    f2 x[WIN];
#pragma unroll
    for (int k = 0; k < WIN; ++k) x[k] = tile[(lr + k) * PL_WAVE + lane];
    unsigned res[NOUT];
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
      unsigned fb;
      res[i] = decide_pair<RAD>(x + i, taps, mf, m0z, m1z, kSigned, fb) ^ kBias;
      failmask |= fb << (g * 16 + 2 * i);
    }
    if (active) {
      T* orow = o + (size_t)(r0 + lr) * w;
#pragma unroll
      for (int i = 0; i < NOUT; ++i) {
        if (r0 + lr + i < h) *reinterpret_cast<unsigned*>(reinterpret_cast<char*>(orow) + coff) = res[i];
        orow += w;
      }
    }
  }
  {  // rows beyond the frame / inactive column pairs are never undecided
    const int nv = h - (r0 + wave * SHARE);  // valid rows of this wave's share
    const unsigned rows = !active || nv <= 0 ? 0u : (nv >= 16 ? 0xffffffffu : ((1u << (2 * nv)) - 1u));
    failmask &= rows;
  }
  push_fails(fix, failmask, (unsigned)tid);
  __syncthreads();  // list complete; first-pass stores ordered before the corrections below

  const unsigned cnt = fix.cnt;
  if (cnt == 0) return;
  // code b = g*16 + 2*i + half of thread t  ->  tile row (t/64)*SHARE + g*8 + i, column pair t%64
  auto fix_one = [&](unsigned t, unsigned b) {
    const int l = (int)(t & (PL_WAVE - 1));
    const int lrow = (int)(t / PL_WAVE) * SHARE + (int)(b >> 4) * NOUT + (int)((b & 15u) >> 1);
    const int half = (int)(b & 1u);
    const int cc = ct * (2 * PL_WAVE) + 2 * l + half;
    const int rr = r0 + lrow;
    if (rr >= h || cc >= w) return;
    const us2 mm = as_us2(s_min[WAVES * PL_WAVE + l]);
    const double m = (double)(half ? mm.y : mm.x) - (kSigned ? 32768.0 : 0.0);
    const float* col = reinterpret_cast<const float*>(tile + lrow * PL_WAVE + l) + half;
    const double acc = exact_from_lds<RAD>([&](int k) { return col[k * 2 * PL_WAVE]; }, m, s_wsum, wts);
    o[(size_t)rr * w + cc] = pl_from_double<T>(acc);
  };
  if (cnt <= (unsigned)kListCap) {
    for (unsigned e = tid; e < cnt; e += NW * PL_WAVE) {
      const unsigned code = fix.item[e];
      fix_one(code >> 5, code & 31u);
    }
  } else {
    for (unsigned b = 0; b < (unsigned)(2 * SHARE); ++b) fix_one((unsigned)tid, b);
  }
}

// ---------------------------------------------------------------------- axis 1 (horizontal) pass
// A wave owns a ROW PAIR x 512 columns; the two rows travel in the two halves of every packed
// operation.  LDS per wave: position p <-> column c0 - RAD + p holds {row a, row b} (8 bytes) at
// p + 2*(p >> 3) (16-byte pad after every 64 bytes: the 80-byte lane stride of the ds_read_b128
// windows is bank-conflict-free).  Rows are independent, so a pair may straddle two frames.
__device__ __forceinline__ constexpr int pad8(int p) { return p + ((p >> 3) << 1); }

template <typename T, int RAD>
__global__ void __launch_bounds__(kPkThreads, 4)
gauss_h_pk(const T* __restrict__ in, T* __restrict__ out, int64_t rows_total, int w, int col_tiles,
           const double* __restrict__ wts) {
  constexpr int NOUT = 8, WIN = NOUT + 2 * RAD;
  constexpr int SEG = PL_WAVE * NOUT;
  constexpr int LOGICAL = SEG + 2 * RAD;
  constexpr int PADDED = LOGICAL + ((LOGICAL + 7) / 8) * 2;
  constexpr int WAVES = kPkThreads / PL_WAVE;
  constexpr bool kSigned = (T)-1 < (T)0;
  constexpr unsigned kBias = kSigned ? 0x80008000u : 0u;
  constexpr unsigned kBias1 = kSigned ? 0x8000u : 0u;
  static_assert(2 * RAD <= PL_WAVE, "halo is loaded by one wave pass");
  __shared__ __attribute__((aligned(16))) f2 lds[WAVES * PADDED];
  __shared__ __attribute__((aligned(16))) f2 s_blockmin[WAVES][LOGICAL / 8 + 1];  // minima of 8-position blocks
  __shared__ FixList fix;
  __shared__ float s_c0;
  __shared__ float s_wf[RAD + 1];
  __shared__ double s_wsum;

  const int tid = threadIdx.x;
  const int lane = tid & (PL_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / PL_WAVE);
  const unsigned lid = pl_xcd_remap(blockIdx.x, gridDim.x);
  const int ct = lid % col_tiles;
  const int64_t pair0 = (int64_t)(lid / col_tiles) * WAVES;  // first row pair of the workgroup
  const int64_t ra = (pair0 + wave) * 2;                     // rows ra, ra+1 (frame*h + r)
  const bool have_a = ra < rows_total;
  const bool have_b = ra + 1 < rows_total;
  const T* fa = in + (have_a ? ra : 0) * (size_t)w;
  const T* fb = in + (have_b ? ra + 1 : (have_a ? ra : 0)) * (size_t)w;
  f2* s = lds + wave * PADDED;
  const int c0 = ct * SEG;
  const int c = c0 + lane * NOUT;

  if (tid == 0) fix.cnt = 0;
  if (wave == 0) {
    double ws;
    const float cc0 = margin_c0<RAD>(wts, ws);
    publish_taps<RAD>(wts, s_wf, lane);
    if (tid == 0) {
      s_c0 = cc0;
      s_wsum = ws;
    }
  }

  // ---- load: own 8 pixels of both rows (+ halo), biased to unsigned
  unsigned qa[NOUT / 2], qb[NOUT / 2];
  const bool vec = (c + NOUT <= w) && ((reinterpret_cast<uintptr_t>(fa + c) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(fb + c) & 15) == 0);
  if (vec) {
    const uint4 va = *reinterpret_cast<const uint4*>(fa + c);
    const uint4 vb = *reinterpret_cast<const uint4*>(fb + c);
    qa[0] = va.x ^ kBias; qa[1] = va.y ^ kBias; qa[2] = va.z ^ kBias; qa[3] = va.w ^ kBias;
    qb[0] = vb.x ^ kBias; qb[1] = vb.y ^ kBias; qb[2] = vb.z ^ kBias; qb[3] = vb.w ^ kBias;
  } else {
#pragma unroll
    for (int k = 0; k < NOUT / 2; ++k) {
      const int c1 = pl_reflect(c + 2 * k, w), c2 = pl_reflect(c + 2 * k + 1, w);
      qa[k] = (((unsigned)(unsigned short)fa[c1]) | ((unsigned)(unsigned short)fa[c2] << 16)) ^ kBias;
      qb[k] = (((unsigned)(unsigned short)fb[c1]) | ((unsigned)(unsigned short)fb[c2] << 16)) ^ kBias;
    }
  }
  unsigned ha = 0, hb = 0;  // halo sample (one per lane < 2*RAD), low 16 bits
  int hp = -1;
  if (lane < 2 * RAD) {
    hp = lane < RAD ? lane : SEG + lane;
    const int cc = pl_reflect(c0 - RAD + hp, w);
    ha = ((unsigned)(unsigned short)fa[cc]) ^ kBias1;
    hb = ((unsigned)(unsigned short)fb[cc]) ^ kBias1;
  }
  // The staged samples carry NO wave-wide offset (biased 16-bit values are exact in float32 as they are); the
  // lane-local minimum below is the only offset.  (An earlier version subtracted the wave-wide row minimum first:
  // two 6-step wave reductions per wave for nothing once the local minimum exists.)
  const f2 mf = splat(0.0f);
  const f2 nb = splat(-8388608.0f);

  // ---- stage {row a, row b} pairs, m-subtracted, as float2
#pragma unroll
  for (int k = 0; k < NOUT / 2; ++k) {
    s[10 * lane + pad8(RAD + 2 * k)] = f2{magic_lo(qa[k]), magic_lo(qb[k])} + nb;
    s[10 * lane + pad8(RAD + 2 * k + 1)] = f2{magic_hi(qa[k]), magic_hi(qb[k])} + nb;
  }
  if (hp >= 0) s[pad8(hp)] = f2{magic_lo(ha), magic_lo(hb)} + nb;
  __syncthreads();  // staging visible (per wave), fix.cnt / s_c0 / taps visible (workgroup)

  PkTaps<RAD> taps;
  load_taps<RAD>(s_wf, taps, s_c0);

  unsigned failmask = 0;
  unsigned res[NOUT];  // res[i] = {row a px i, row b px i}
  {
    const f2* win = s + 10 * lane;
    // ---- lane-local minimum.  The staged samples are relative to the wave-wide row minimum; wherever the
    // 552-sample span holds a field edge that offset leaves t ~ 40 000 on the plateau and 10 % of those pixels
    // undecided.  Lane l's window is exactly the 8-position blocks l .. l+NBLK-1: every lane reduces its
    // own first block (the first NBLK-1 lanes also the blocks past lane 63), the block minima are exchanged through
    // LDS, and the window is re-offset by their minimum d >= 0 (exact: integers < 2^17) while it is loaded
    // (d first, window second: holding the whole window across the exchange spilled and doubled the run time).
    static_assert(WIN % 8 == 0, "a window must be a whole number of 8-position blocks");
    constexpr int NBLK = WIN / 8;            // blocks a window covers (RAD 20: 6)
    f2 d;
    {
      f2 bmin = win[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) bmin = f2{__builtin_fminf(bmin.x, win[k].x), __builtin_fminf(bmin.y, win[k].y)};
      f2* bm = s_blockmin[wave];
      bm[lane] = bmin;
      if (lane < NBLK - 1) {                 // blocks 64 .. 64+NBLK-2 start past the last lane's own block
        const f2* extra = s + pad8(8 * (PL_WAVE + lane));
        f2 e = extra[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) e = f2{__builtin_fminf(e.x, extra[k].x), __builtin_fminf(e.y, extra[k].y)};
        bm[PL_WAVE + lane] = e;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      d = bmin;
#pragma unroll
      for (int q = 1; q < NBLK; ++q) {
        const f2 v = bm[lane + q];
        d = f2{__builtin_fminf(d.x, v.x), __builtin_fminf(d.y, v.y)};
      }
    }
    f2 x[WIN];
#pragma unroll
    for (int k = 0; k < WIN; ++k) x[k] = win[k + ((k >> 3) << 1)] - d;
    const f2 mfl = mf + d;                                       // = the local minimum itself (biased domain)
    const float zero_b = kSigned ? 32768.0f : 0.0f;               // actual value 0 in the biased domain
    const bool z0 = mfl.x == zero_b, z1 = mfl.y == zero_b;
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
      unsigned fbits;
      res[i] = decide_pair<RAD>(x + i, taps, mfl, z0, z1, kSigned, fbits) ^ kBias;
      failmask |= fbits << (2 * i);
    }
  }
  {  // outputs that do not exist (beyond the width, unpaired last row) are never undecided
    const int nv = (w - c) < 0 ? 0 : ((w - c) > NOUT ? NOUT : (w - c));
    const unsigned cols = nv >= NOUT ? 0xffffu : ((1u << (2 * nv)) - 1u);
    failmask &= cols & ((have_a ? 0x5555u : 0u) | (have_b ? 0xaaaau : 0u));
  }
  T* oa = out + (have_a ? ra : 0) * (size_t)w;
  T* ob = out + (have_b ? ra + 1 : 0) * (size_t)w;
  if (c < w) {
    // transpose the {a,b} pairs into one 8-pixel vector per row
    unsigned pa[NOUT / 2], pb[NOUT / 2];
#pragma unroll
    for (int k = 0; k < NOUT / 2; ++k) {
      pa[k] = (res[2 * k] & 0xffffu) | (res[2 * k + 1] << 16);
      pb[k] = (res[2 * k] >> 16) | (res[2 * k + 1] & 0xffff0000u);
    }
    const bool vst = (c + NOUT <= w) && ((reinterpret_cast<uintptr_t>(oa + c) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(ob + c) & 15) == 0);
    if (vst) {
      if (have_a) *reinterpret_cast<uint4*>(oa + c) = uint4{pa[0], pa[1], pa[2], pa[3]};
      if (have_b) *reinterpret_cast<uint4*>(ob + c) = uint4{pb[0], pb[1], pb[2], pb[3]};
    } else {
#pragma unroll
      for (int k = 0; k < NOUT; ++k) {
        if (c + k < w) {
          if (have_a) oa[c + k] = (T)(res[k] & 0xffffu);
          if (have_b) ob[c + k] = (T)(res[k] >> 16);
        }
      }
    }
  }
  push_fails(fix, failmask, (unsigned)tid);
  __syncthreads();  // list complete; first-pass stores ordered before the corrections below

  const unsigned cnt = fix.cnt;
  if (cnt == 0) return;
  // code b = 2*i + row select of thread t  ->  wave t/64, window start 8*(t%64) + i
  auto fix_one = [&](unsigned t, unsigned b) {
    const int wv = (int)(t / PL_WAVE), l = (int)(t & (PL_WAVE - 1));
    const int i = (int)(b >> 1), sel = (int)(b & 1u);
    const int64_t row = (pair0 + wv) * 2 + sel;
    const int cc = c0 + l * NOUT + i;
    if (row >= rows_total || cc >= w) return;
    const double m = kSigned ? -32768.0 : 0.0;                  // the staged samples are the biased values themselves
    const float* base = reinterpret_cast<const float*>(lds + wv * PADDED) + sel;
    // window position p0 + k lives at 10*l + pad8(i + k): one lane-dependent shift per 8 positions
    const float* win = base + 20 * l;
    const double acc = exact_from_lds<RAD>(
        [&](int k) {
          const int q = i + k;  // 0 .. 2*RAD + 7
          return win[2 * (q + ((q >> 3) << 1))];
        },
        m, s_wsum, wts);
    out[row * (size_t)w + cc] = pl_from_double<T>(acc);
  };
  if (cnt <= (unsigned)kListCap) {
    for (unsigned e = tid; e < cnt; e += kPkThreads) {
      const unsigned code = fix.item[e];
      fix_one(code >> 5, code & 31u);
    }
  } else {
    for (unsigned b = 0; b < (unsigned)(2 * NOUT); ++b) fix_one((unsigned)tid, b);
  }
}

template <typename T, int RAD, int kVRows>
int launch_pk_t(const T* in, T* out, int64_t n, int h, int w, int axis, const double* wts, hipStream_t st) {
  constexpr int WAVES = kPkThreads / PL_WAVE;
  if (axis == 0) {
    if ((w & 1) || (reinterpret_cast<uintptr_t>(in) & 3) || (reinterpret_cast<uintptr_t>(out) & 3)) return -1;
    const int col_tiles = (int)pl_cdiv(w, 2 * PL_WAVE);
    const int row_tiles = (int)pl_cdiv(h, kVRows);
    const int64_t blocks = n * col_tiles * row_tiles;
    if (blocks > 0x7fffffffLL) return -1;
    // measured alternatives: 64-row tiles with 4 waves 0.69 ms, 64-row tiles with 8 waves 0.78 ms (32 rows: 0.58)
    static const bool occ4 = [] {
      const char* e = getenv("PL_GAUSS_V_OCC");
      return e && e[0] == '4';
    }();
    if (occ4)
      hipLaunchKernelGGL((gauss_v_pk<T, RAD, kVRows, 4, 4>), dim3((unsigned)blocks), dim3(kPkThreads), 0, st, in, out,
                         h, w, col_tiles, row_tiles, wts);
    else
      hipLaunchKernelGGL((gauss_v_pk<T, RAD, kVRows, 4>), dim3((unsigned)blocks), dim3(kPkThreads), 0, st, in, out,
                         h, w, col_tiles, row_tiles, wts);
  } else {
    const int col_tiles = (int)pl_cdiv(w, PL_WAVE * 8);
    const int64_t rows_total = n * h;
    const int64_t blocks = pl_cdiv(pl_cdiv(rows_total, 2), WAVES) * col_tiles;
    if (blocks > 0x7fffffffLL) return -1;
    hipLaunchKernelGGL((gauss_h_pk<T, RAD>), dim3((unsigned)blocks), dim3(kPkThreads), 0, st, in, out,
                       rows_total, w, col_tiles, wts);
  }
  return 0;
}

}  // namespace

// 0 = launched; -1 = shape / alignment / radius not covered (caller uses the float64 kernels)
int pl_gauss_pk_launch(const void* in, void* out, int is_signed, int64_t n, int h, int w, int axis,
                       const double* wts, int radius, hipStream_t st) {
#define PL_PK_CASE(R)                                                                                        \
  case R:                                                                                                    \
    return is_signed ? launch_pk_t<short, R, 32>((const short*)in, (short*)out, n, h, w, axis, wts, st)     \
                     : launch_pk_t<unsigned short, R, 32>((const unsigned short*)in, (unsigned short*)out,   \
                                                          n, h, w, axis, wts, st);
  switch (radius) {
    PL_PK_CASE(4)
    PL_PK_CASE(8)
    PL_PK_CASE(12)
    PL_PK_CASE(20)
    default: return -1;
  }
#undef PL_PK_CASE
}
