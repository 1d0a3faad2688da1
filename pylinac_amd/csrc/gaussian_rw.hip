// Register-window packed-FP32 decision kernels for the scipy-exact Gaussian on 16-bit images.
//
// Replaces: scipy.ndimage.gaussian_filter on uint16 / int16 frames as called at
// pylinac/core/array_utils.py:133 (BaseImage.filter(kind="gaussian"), pylinac/core/image.py:695-712).
//
// Contract (gaussian.hip header): trunc(S), S = scipy's float64 tap sequence.  Decision arithmetic as in round 1:
//
//   S = m*W + t,  t = sum_j w_j * (x[-j] + x[+j] - 2m) + w_0 * (x[0] - m)       (real arithmetic)
//
//   m   a lower bound of every sample of the lane's OWN window (lane-local minimum: no exchange), so every term of t is
//       >= 0, the partial sums of the float32 chain grow monotonically and |t_hat - t| <= (RAD + 2) * 2^-24 * t;
//   W   = w_0 + 2*sum w_j in float64; |m * (W - 1)| <= 65535 * |W - 1| is added to the margin;
//   =>  trunc(S) = m + floor(t_hat) whenever frac(t_hat) is farther than
//       delta = 1.02 * (RAD + 2) * 2^-24 * t_hat + 65535 * |W - 1| + 1e-6 from 0 and 1.
//   m == 0 (true value zero): S = t >= 0 exactly, so t_hat <= 0.5 decides trunc(S) = 0 (all-zero windows included).
//
// What changed against round 1's LDS-tile kernels (gaussian_pk.hip, 0.60 / 0.49 ms per pass, VALU 53 % busy behind
// three workgroup barriers): every WAVE now owns its window outright -- it loads the rows / columns it needs straight
// into registers, takes its own minimum, converts once and produces NOUT output rows (axis 0: lane = column pair) or
// NOUT output pixels of a row pair (axis 1: lane = NOUT consecutive columns of two rows, halo exchanged through a
// wave-private LDS row copy).  There is NO barrier in front of the arithmetic; waves of a workgroup only meet once, at
// the end, to share the list of undecided pixels (axis 0) -- axis 1 has no workgroup barrier at all.
//
// Undecided pixels (~0.4 % on EPID content): their codes go on a list in LDS and are recomputed from the RAW samples
// the waves copied to LDS on their way in (axis 0: the workgroup's 4*NOUT + 2*RAD rows x 128 columns; axis 1: the
// wave's own two rows), one pixel per lane: float64 FMA chain first, scipy's exact sequence when that lands within
// 4e-9 of an integer.  A list that overflows (constant / saturated regions: S sits 1e-11 from an integer everywhere)
// makes every lane recompute its own outputs that way.
//
// u16 -> f32 without cvt instructions: (x16 | 0x4B000000) is the float 2^23 + x16; one v_pk_add_f32 with
// -(2^23 + m) yields x - m for two pixels (exact).  int16 is XOR-biased into the unsigned domain first.
#include <type_traits>

#include "pl_common.h"

namespace {

constexpr int kRwThreads = 256;
constexpr int kRwWaves = kRwThreads / PL_WAVE;
constexpr int kRwListCap = 512;

typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 splat2(float v) { return f2{v, v}; }
__device__ __forceinline__ us2 to_us2(unsigned v) {
  union { unsigned u; us2 s; } c;
  c.u = v;
  return c.s;
}
__device__ __forceinline__ unsigned from_us2(us2 v) {
  union { unsigned u; us2 s; } c;
  c.s = v;
  return c.u;
}
__device__ __forceinline__ float magic_lo16(unsigned raw) {
  return __uint_as_float(__builtin_amdgcn_perm(0x4B000000u, raw, 0x070c0100u));
}
__device__ __forceinline__ float magic_hi16(unsigned raw) {
  return __uint_as_float(__builtin_amdgcn_perm(0x4B000000u, raw, 0x070c0302u));
}

// Everything a pass needs to know about the taps, computed ONCE on the host (rw_make_params) and handed over by value
// as kernel arguments: scalar loads put it in SGPRs, no wave spends VALU slots converting taps.
template <int RAD>
struct RwParams {
  float w[RAD + 1];   // float32 tap at offset j (j = 0 centre)
  float c1;           // relative margin (RAD + 2) * 2^-24 * 1.02
  float c2;           // 0.5 - c0, c0 = 65535*|W-1| + 1e-6, W = float64 sum of all taps
  double wd[RAD + 1]; // the float64 taps (offset j) for the exact path
};

template <int RAD>
RwParams<RAD> rw_make_params(const double* h_wts /* 2*RAD+1 taps, centre at RAD */) {
  RwParams<RAD> p;
  double s = 0.0;
  bool nonneg = h_wts[RAD] >= 0.0;
  for (int j = RAD; j >= 1; --j) {
    s += h_wts[RAD - j];
    nonneg = nonneg && (h_wts[RAD - j] >= 0.0);
  }
  s = h_wts[RAD] + 2.0 * s;
  const double c = 65535.0 * (__builtin_fabs(s - 1.0) + 1e-14) + 1e-6;
  // a negative tap breaks the monotone-partial-sum bound: margin > 0.5 sends every pixel to the exact path
  const float c0 = (c > 4.0 || !nonneg) ? 4.0f : (float)c;
  p.c2 = 0.5f - c0;
  p.c1 = (RAD + 2) * 5.9604645e-08f * 1.02f;
  for (int j = 0; j <= RAD; ++j) {
    p.w[j] = (float)h_wts[RAD - j];
    p.wd[j] = h_wts[RAD - j];
  }
  return p;
}

// One pair of outputs from a window of pair-packed, m-subtracted samples x[0 .. 2*RAD] centred at x[RAD].
// zfloor = {0.5 where the column's minimum is the value zero, else 0}; mpk = the two minima packed 2 x 16 bit (biased).
// Returns the two truncated results packed 2 x 16 bit (biased domain); shifts the two fail bits into `fail`
// (first the .x element, then .y: bit order documented at the callers).
template <int RAD>
__device__ __forceinline__ unsigned rw_decide(const f2* x, const RwParams<RAD>& t, f2 zfloor, unsigned mpk,
                                              bool is_signed, unsigned& fail) {
  f2 acc = (x[0] + x[2 * RAD]) * splat2(t.w[RAD]);
#pragma unroll
  for (int j = RAD - 1; j >= 1; --j)
    acc = __builtin_elementwise_fma(x[RAD - j] + x[RAD + j], splat2(t.w[j]), acc);
  acc = __builtin_elementwise_fma(x[RAD], splat2(t.w[0]), acc);
  acc = f2{__builtin_fmaxf(acc.x, zfloor.x), __builtin_fmaxf(acc.y, zfloor.y)};

  const f2 fr = f2{__builtin_amdgcn_fractf(acc.x), __builtin_amdgcn_fractf(acc.y)};
  const f2 lim = __builtin_elementwise_fma(acc, splat2(-t.c1), splat2(t.c2));  // 0.5 - delta
  const f2 d = fr - splat2(0.5f);
  // undecided <=> |d| > lim <=> sign bit of lim - |d|; one funnel shift appends it to the lane's fail word (no
  // compare -> mask -> select chain and none of its SGPR-hazard wait states)
  fail = __builtin_amdgcn_alignbit(fail, __float_as_uint(lim.x - __builtin_fabsf(d.x)), 31);
  fail = __builtin_amdgcn_alignbit(fail, __float_as_uint(lim.y - __builtin_fabsf(d.y)), 31);
  const unsigned r0 = (unsigned)acc.x, r1 = (unsigned)acc.y;  // floor: acc >= 0
  unsigned r = __builtin_amdgcn_perm(r1, r0, 0x05040100u) ;    // {r0.lo16, r1.lo16}
  us2 rr = to_us2(r) + to_us2(mpk);                            // integers < 2^16 per half: no carry between halves
  if (is_signed)  // biased value = floor(S) + 32768; C truncation rounds negative S toward zero
    rr = rr + us2{(unsigned short)(rr.x < 32768u ? 1 : 0), (unsigned short)(rr.y < 32768u ? 1 : 0)};
  return from_us2(rr);
}

// scipy's value for one pixel from raw samples v(k), k = 0 .. 2*RAD (actual values as doubles, centre RAD):
// float64 FMA chain first (differs from scipy's sequence by < 7e-10 for 16-bit data), scipy's exact sequence when
// that lands within 4e-9 of an integer (constant / saturated neighbourhoods).
template <int RAD, typename F>
__device__ __forceinline__ double rw_exact(F v, const RwParams<RAD>& P) {
  double a = v(RAD) * P.wd[0];
#pragma unroll
  for (int j = RAD; j >= 1; --j) a = __builtin_fma(v(RAD - j) + v(RAD + j), P.wd[j], a);
  const double off = __builtin_fabs(__builtin_amdgcn_fract(__builtin_fabs(a)) - 0.5);
  if (off > 0.5 - 4e-9) {
    a = v(RAD) * P.wd[0];
#pragma unroll
    for (int j = RAD; j >= 1; --j) a = a + (v(RAD - j) + v(RAD + j)) * P.wd[j];
  }
  return a;
}

struct RwFixList {
  unsigned cnt;
  unsigned item[kRwListCap];
};
__device__ __forceinline__ void rw_push_fails(RwFixList& fl, unsigned failmask, unsigned tid) {
  while (failmask) {  // lanes without undecided outputs skip the loop
    const int b = __builtin_ctz(failmask);
    failmask &= failmask - 1;
    const unsigned i = atomicAdd(&fl.cnt, 1u);
    if (i < (unsigned)kRwListCap) fl.item[i] = (tid << 5) | (unsigned)b;
  }
}

// ------------------------------------------------------------------------ axis 0 (vertical) pass
// Workgroup = 4 waves stacked vertically over one 128-column strip: wave v produces output rows
// r0 + v*NOUT .. + NOUT-1 from its own window of NOUT + 2*RAD rows; lane = column pair (one dword per row).
template <typename T, int RAD, int NOUT>
__global__ void __launch_bounds__(kRwThreads, (NOUT + 2 * RAD) * 2 + 24 <= 128 ? 4 : 3)
gauss_v_rw(const T* __restrict__ in, T* __restrict__ out, int h, int w, int col_tiles, int row_tiles,
           const RwParams<RAD> taps) {
  constexpr int WIN = NOUT + 2 * RAD;
  constexpr int TOUT = NOUT * kRwWaves;      // output rows per workgroup
  constexpr int TROWS = TOUT + 2 * RAD;      // rows the workgroup touches
  constexpr int PER = TROWS / kRwWaves;      // raw rows each wave copies to LDS
  static_assert(RAD % 2 == 0 && TROWS % kRwWaves == 0, "row shares must be whole");
  static_assert(2 * NOUT <= 32, "fail bits of a lane fit one dword");
  constexpr bool kSigned = (T)-1 < (T)0;
  constexpr unsigned kBias = kSigned ? 0x80008000u : 0u;
  __shared__ unsigned s_raw[TROWS * PL_WAVE];  // [tile row][column pair], biased raw dwords
  __shared__ RwFixList fix;

  unsigned id = pl_xcd_remap(blockIdx.x, gridDim.x);
  const int ct = id % col_tiles;
  id /= col_tiles;
  const int rt = id % row_tiles;
  const size_t frame = id / row_tiles;
  const int tid = threadIdx.x;
  const int lane = tid & (PL_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / PL_WAVE);
  const int c = ct * (2 * PL_WAVE) + 2 * lane;
  const int r0 = rt * TOUT;                  // first output row of the workgroup
  const int wr0 = r0 + wave * NOUT;          // first output row of the wave
  const bool active = c < w;                 // w is even: the pair is inside or outside together
  const T* f = in + frame * (size_t)h * w;
  T* o = out + frame * (size_t)h * w;
  // inactive lanes (columns beyond the width) read column 0 instead of being masked: every load below is then one
  // unconditional buffer access (frame base in the resource, row offset in an SGPR, lane offset in one VGPR -- no
  // per-load address arithmetic on the VALU); their results are never stored and never listed
  const unsigned coff = active ? (unsigned)c * (unsigned)sizeof(T) : 0u;
  const __amdgpu_buffer_rsrc_t rin = pl_make_rsrc(f);
  const __amdgpu_buffer_rsrc_t rout = pl_make_rsrc(o);
  const unsigned pitch = (unsigned)w * (unsigned)sizeof(T);

  if (tid == 0) fix.cnt = 0;

  // ---- window rows wr0 - RAD + k, k < WIN, straight into registers (row index wave-uniform -> scalar reflect)
  unsigned raw[WIN];
  if (wr0 - RAD >= 0 && wr0 - RAD + WIN <= h) {  // interior: one scalar add per row
    unsigned soff = (unsigned)(wr0 - RAD) * pitch;
#pragma unroll
    for (int k = 0; k < WIN; ++k) {
      raw[k] = pl_buffer_load_u32(rin, coff, soff) ^ kBias;
      soff += pitch;
    }
  } else {
#pragma unroll
    for (int k = 0; k < WIN; ++k)
      raw[k] = pl_buffer_load_u32(rin, coff, (unsigned)pl_reflect(wr0 - RAD + k, h) * pitch) ^ kBias;
  }
  // ---- raw copy for the fix-ups: wave v owns tile rows [v*PER, (v+1)*PER) = its window rows k0 .. k0+PER-1
  {
    unsigned* dst = s_raw + wave * PER * PL_WAVE + lane;
    auto copy = [&](auto kc) {
      constexpr int k0 = decltype(kc)::value;
#pragma unroll
      for (int j = 0; j < PER; ++j) dst[j * PL_WAVE] = raw[k0 + j];
    };
    if (wave == 0) copy(std::integral_constant<int, 0>{});
    else if (wave == 1) copy(std::integral_constant<int, RAD / 2>{});
    else if (wave == 2) copy(std::integral_constant<int, RAD>{});
    else copy(std::integral_constant<int, 3 * RAD / 2>{});
  }
  us2 mn = to_us2(raw[0]);
#pragma unroll
  for (int k = 1; k < WIN; ++k) mn = __builtin_elementwise_min(mn, to_us2(raw[k]));
  const unsigned mpk = from_us2(mn);
  const f2 nb = -(splat2(8388608.0f) + f2{(float)mn.x, (float)mn.y});
  const unsigned short zero_b = kSigned ? 32768 : 0;  // the value 0 in the biased domain
  const f2 zfloor = f2{mn.x == zero_b ? 0.5f : 0.0f, mn.y == zero_b ? 0.5f : 0.0f};

  f2 x[WIN];
#pragma unroll
  for (int k = 0; k < WIN; ++k) x[k] = f2{magic_lo16(raw[k]), magic_hi16(raw[k])} + nb;

  unsigned failmask = 0;  // after the loop: bit 2*(NOUT-1-i) + 1 = column c of output row i, + 0 = column c+1
  {
    unsigned res[NOUT];
#pragma unroll
    for (int i = 0; i < NOUT; ++i) res[i] = rw_decide<RAD>(x + i, taps, zfloor, mpk, kSigned, failmask) ^ kBias;
    if (active) {
      unsigned soff = (unsigned)wr0 * pitch;
#pragma unroll
      for (int i = 0; i < NOUT; ++i) {
        if (wr0 + i < h) pl_buffer_store_u32(res[i], rout, coff, soff);
        soff += pitch;
      }
    }
  }
  {  // rows beyond the frame / inactive column pairs are never undecided
    const int nv = h - wr0;  // valid rows of this wave
    constexpr unsigned kAll = (2 * NOUT < 32) ? ((1u << ((2 * NOUT) & 31)) - 1u) : 0xffffffffu;
    unsigned keep = kAll;
    if (!active || nv <= 0) keep = 0u;
    else if (nv < NOUT) keep &= ~((1u << (2 * (NOUT - nv))) - 1u);
    failmask &= keep;
  }
  rw_push_fails(fix, failmask, (unsigned)tid);
  __syncthreads();  // list + raw tile complete; first-pass stores ordered before the corrections below

  const unsigned cnt = fix.cnt;
  if (cnt == 0) return;
  // bit b of thread t  ->  output row (t/64)*NOUT + (NOUT-1 - b/2), column pair t%64, half = 1 - (b&1)
  auto fix_one = [&](unsigned t, unsigned b) {
    const int l = (int)(t & (PL_WAVE - 1));
    const int lrow = (int)(t / PL_WAVE) * NOUT + (NOUT - 1 - (int)(b >> 1));
    const int half = 1 - (int)(b & 1u);
    const int cc = ct * (2 * PL_WAVE) + 2 * l + half;
    const int rr = r0 + lrow;
    if (rr >= h || cc >= w) return;
    const unsigned short* col = reinterpret_cast<const unsigned short*>(s_raw + lrow * PL_WAVE + l) + half;
    const double acc = rw_exact<RAD>(
        [&](int k) { return (double)((int)col[k * 2 * PL_WAVE] - (kSigned ? 32768 : 0)); }, taps);
    o[(size_t)rr * w + cc] = pl_from_double<T>(acc);
  };
  if (cnt <= (unsigned)kRwListCap) {
    for (unsigned e = tid; e < cnt; e += kRwThreads) {
      const unsigned code = fix.item[e];
      fix_one(code >> 5, code & 31u);
    }
  } else {
    for (unsigned b = 0; b < (unsigned)(2 * NOUT); ++b) fix_one((unsigned)tid, b);
  }
}

// ---------------------------------------------------------------------- axis 1 (horizontal) pass
// A wave owns a ROW PAIR x SEG = 64*NOUT columns; lane = NOUT consecutive columns; the two rows travel in the two halves
// of every packed operation.  The wave copies its two raw row segments (+ RAD halo columns each side, reflected at the
// frame edge) into a private LDS strip -- that is both the halo exchange (every lane then reads its NOUT + 2*RAD window
// with 16-byte reads at a 2*NOUT-byte lane stride) and the sample store for the fix-ups.  No workgroup barrier.
template <typename T, int RAD, int NOUT>
__global__ void __launch_bounds__(kRwThreads, (NOUT + 2 * RAD) * 2 + 24 <= 128 ? 4 : 3)
gauss_h_rw(const T* __restrict__ in, T* __restrict__ out, int64_t rows_total, int w, int col_tiles,
           const RwParams<RAD> taps) {
  constexpr int WIN = NOUT + 2 * RAD;
  constexpr int SEG = PL_WAVE * NOUT;
  constexpr int STRIP = SEG + 2 * RAD;              // samples per staged row
  constexpr int STRIP_PAD = (STRIP + 7) & ~7;       // 16-byte multiple
  static_assert(NOUT % 8 == 0 && WIN % 8 == 0 && RAD % 4 == 0, "16-byte window reads, 8-byte aligned own-pixel writes");
  static_assert(2 * RAD <= PL_WAVE, "halo is loaded by one wave pass");
  static_assert(2 * NOUT <= 32, "fail bits of a lane fit one dword");
  constexpr bool kSigned = (T)-1 < (T)0;
  constexpr unsigned kBias = kSigned ? 0x80008000u : 0u;
  constexpr unsigned short kBias1 = kSigned ? 0x8000u : 0u;
  constexpr int kWaveCap = 64;                      // undecided pixels a wave lists before it recomputes everything
  __shared__ __attribute__((aligned(16))) unsigned short s_row[kRwWaves][2][STRIP_PAD];
  __shared__ unsigned s_cnt[kRwWaves];
  __shared__ unsigned s_item[kRwWaves][kWaveCap];

  const int tid = threadIdx.x;
  const int lane = tid & (PL_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / PL_WAVE);
  const unsigned lid = pl_xcd_remap(blockIdx.x, gridDim.x);
  const int ct = lid % col_tiles;
  const int64_t ra = ((int64_t)(lid / col_tiles) * kRwWaves + wave) * 2;  // rows ra, ra+1 (frame*h + r)
  if (ra >= rows_total) return;                       // wave-uniform; no workgroup barrier below
  const bool have_b = ra + 1 < rows_total;
  const T* fa = in + ra * (size_t)w;
  const T* fb = in + (have_b ? ra + 1 : ra) * (size_t)w;
  const int c0 = ct * SEG;
  const int c = c0 + lane * NOUT;
  const bool active = c < w;                          // w % NOUT == 0: a lane is wholly inside or outside
  unsigned short* sa = s_row[wave][0];
  unsigned short* sb = s_row[wave][1];

  if (lane == 0) s_cnt[wave] = 0;
  // ---- own NOUT pixels of both rows -> LDS position RAD + lane*NOUT; halo sample (lanes < 2*RAD) -> its position
  if (active) {
#pragma unroll
    for (int q = 0; q < NOUT / 8; ++q) {
      const uint4 va = *reinterpret_cast<const uint4*>(fa + c + 8 * q);
      const uint4 vb = *reinterpret_cast<const uint4*>(fb + c + 8 * q);
      uint2* da = reinterpret_cast<uint2*>(sa + RAD + lane * NOUT + 8 * q);   // 8-byte aligned: RAD % 4 == 0
      uint2* db = reinterpret_cast<uint2*>(sb + RAD + lane * NOUT + 8 * q);
      da[0] = uint2{va.x ^ kBias, va.y ^ kBias}; da[1] = uint2{va.z ^ kBias, va.w ^ kBias};
      db[0] = uint2{vb.x ^ kBias, vb.y ^ kBias}; db[1] = uint2{vb.z ^ kBias, vb.w ^ kBias};
    }
  } else if (c < w + RAD) {                           // partial last tile: the right halo lies in this lane's own positions
#pragma unroll 1
    for (int k = 0; k < NOUT; ++k) {
      const int cc = pl_reflect(c + k, w);
      sa[RAD + lane * NOUT + k] = (unsigned short)((unsigned short)fa[cc] ^ kBias1);
      sb[RAD + lane * NOUT + k] = (unsigned short)((unsigned short)fb[cc] ^ kBias1);
    }
  }
  if (lane < 2 * RAD) {
    const int hp = lane < RAD ? lane : SEG + lane;    // strip position; column c0 - RAD + hp
    const int cc = pl_reflect(c0 - RAD + hp, w);
    sa[hp] = (unsigned short)((unsigned short)fa[cc] ^ kBias1);
    sb[hp] = (unsigned short)((unsigned short)fb[cc] ^ kBias1);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

  // ---- window: strip positions lane*NOUT .. + WIN-1 of both rows (16-byte reads)
  unsigned qa[WIN / 2], qb[WIN / 2];
  {
    const uint4* pa = reinterpret_cast<const uint4*>(sa + lane * NOUT);
    const uint4* pb = reinterpret_cast<const uint4*>(sb + lane * NOUT);
#pragma unroll
    for (int q = 0; q < WIN / 8; ++q) {
      const uint4 va = pa[q], vb = pb[q];
      qa[4 * q] = va.x; qa[4 * q + 1] = va.y; qa[4 * q + 2] = va.z; qa[4 * q + 3] = va.w;
      qb[4 * q] = vb.x; qb[4 * q + 1] = vb.y; qb[4 * q + 2] = vb.z; qb[4 * q + 3] = vb.w;
    }
  }
  us2 ma = to_us2(qa[0]), mb = to_us2(qb[0]);
#pragma unroll
  for (int k = 1; k < WIN / 2; ++k) {
    ma = __builtin_elementwise_min(ma, to_us2(qa[k]));
    mb = __builtin_elementwise_min(mb, to_us2(qb[k]));
  }
  const unsigned short m_a = ma.x < ma.y ? ma.x : ma.y, m_b = mb.x < mb.y ? mb.x : mb.y;
  const unsigned mpk = (unsigned)m_a | ((unsigned)m_b << 16);
  const f2 nb = -(splat2(8388608.0f) + f2{(float)m_a, (float)m_b});
  const unsigned short zero_b = kSigned ? 32768 : 0;
  const f2 zfloor = f2{m_a == zero_b ? 0.5f : 0.0f, m_b == zero_b ? 0.5f : 0.0f};

  f2 x[WIN];  // x[k] = {row a, row b} at strip position lane*NOUT + k
#pragma unroll
  for (int k = 0; k < WIN / 2; ++k) {
    x[2 * k] = f2{magic_lo16(qa[k]), magic_lo16(qb[k])} + nb;
    x[2 * k + 1] = f2{magic_hi16(qa[k]), magic_hi16(qb[k])} + nb;
  }

  unsigned failmask = 0;  // after the loop: bit 2*(NOUT-1-i) + 1 = row a pixel i, + 0 = row b pixel i
  unsigned res[NOUT];     // res[i] = {row a px i, row b px i}
#pragma unroll
  for (int i = 0; i < NOUT; ++i) res[i] = rw_decide<RAD>(x + i, taps, zfloor, mpk, kSigned, failmask) ^ kBias;
  failmask &= (2 * NOUT < 32) ? ((1u << ((2 * NOUT) & 31)) - 1u) : 0xffffffffu;
  if (!active) failmask = 0;
  if (!have_b) failmask &= 0xaaaaaaaau;

  T* oa = out + ra * (size_t)w;
  T* ob = out + (have_b ? ra + 1 : ra) * (size_t)w;
  if (active) {
    // transpose the {a,b} pairs into one vector per row
#pragma unroll
    for (int q = 0; q < NOUT / 8; ++q) {
      unsigned pa[4], pb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned e = res[8 * q + 2 * k], o2 = res[8 * q + 2 * k + 1];
        pa[k] = __builtin_amdgcn_perm(o2, e, 0x05040100u);   // {e.lo, o2.lo}
        pb[k] = __builtin_amdgcn_perm(o2, e, 0x07060302u);   // {e.hi, o2.hi}
      }
      *reinterpret_cast<uint4*>(oa + c + 8 * q) = uint4{pa[0], pa[1], pa[2], pa[3]};
      if (have_b) *reinterpret_cast<uint4*>(ob + c + 8 * q) = uint4{pb[0], pb[1], pb[2], pb[3]};
    }
  }

  // ---- undecided pixels: wave-private list, recomputed from the wave's own LDS strip
  {
    unsigned fm = failmask;
    while (fm) {
      const int b = __builtin_ctz(fm);
      fm &= fm - 1;
      const unsigned i = atomicAdd(&s_cnt[wave], 1u);
      if (i < (unsigned)kWaveCap) s_item[wave][i] = ((unsigned)lane << 5) | (unsigned)b;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const unsigned cnt = s_cnt[wave];
  if (cnt == 0) return;
  // bit b of lane l -> pixel i = NOUT-1 - b/2 of row (b&1 ? a : b)
  auto fix_one = [&](unsigned l, unsigned b) {
    const int i = NOUT - 1 - (int)(b >> 1);
    const bool row_a = (b & 1u) != 0;
    const int cc = c0 + (int)l * NOUT + i;
    if (cc >= w || (!row_a && !have_b)) return;
    const unsigned short* s = (row_a ? sa : sb) + (int)l * NOUT + i;   // window of pixel i starts at strip position l*NOUT + i
    const double acc = rw_exact<RAD>([&](int k) { return (double)((int)s[k] - (kSigned ? 32768 : 0)); }, taps);
    (row_a ? oa : ob)[cc] = pl_from_double<T>(acc);
  };
  if (cnt <= (unsigned)kWaveCap) {
    for (unsigned e = lane; e < cnt; e += PL_WAVE) {
      const unsigned code = s_item[wave][e];
      fix_one(code >> 5, code & 31u);
    }
  } else {
    for (unsigned b = 0; b < (unsigned)(2 * NOUT); ++b) fix_one((unsigned)lane, b);
  }
}

template <typename T, int RAD, int NOUT>
int launch_rw_t(const T* in, T* out, int64_t n, int h, int w, int axis, const double* h_wts, hipStream_t st) {
  const RwParams<RAD> P = rw_make_params<RAD>(h_wts);
  if (axis == 0) {
    if ((w & 1) || (reinterpret_cast<uintptr_t>(in) & 3) || (reinterpret_cast<uintptr_t>(out) & 3)) return -1;
    const int col_tiles = (int)pl_cdiv(w, 2 * PL_WAVE);
    const int row_tiles = (int)pl_cdiv(h, NOUT * kRwWaves);
    const int64_t blocks = n * col_tiles * row_tiles;
    if (blocks > 0x7fffffffLL || (int64_t)h * w * (int64_t)sizeof(T) > 0xffffffffLL) return -1;
    hipLaunchKernelGGL((gauss_v_rw<T, RAD, NOUT>), dim3((unsigned)blocks), dim3(kRwThreads), 0, st, in, out, h, w,
                       col_tiles, row_tiles, P);
  } else {
    if ((w % NOUT) || (reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return -1;
    const int col_tiles = (int)pl_cdiv(w, PL_WAVE * NOUT);
    const int64_t rows_total = n * h;
    const int64_t blocks = pl_cdiv(pl_cdiv(rows_total, 2), kRwWaves) * col_tiles;
    if (blocks > 0x7fffffffLL) return -1;
    hipLaunchKernelGGL((gauss_h_rw<T, RAD, NOUT>), dim3((unsigned)blocks), dim3(kRwThreads), 0, st, in, out,
                       rows_total, w, col_tiles, P);
  }
  return 0;
}

}  // namespace

// 1 when pl_gauss_rw_launch covers this call (the caller then needs the taps in HOST memory)
int pl_gauss_rw_covers(const void* in, const void* out, int h, int w, int axis, int radius) {
  if (!(radius == 4 || radius == 8 || radius == 12 || radius == 20)) return 0;
  if (axis == 0)
    return !((w & 1) || (reinterpret_cast<uintptr_t>(in) & 3) || (reinterpret_cast<uintptr_t>(out) & 3) ||
             (int64_t)h * w * 2 > 0xffffffffLL);
  return !((w % 16) || (reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(out) & 15));
}

// 0 = launched; -1 = shape / alignment / radius not covered (caller uses the float64 kernels).  h_wts: HOST memory.
int pl_gauss_rw_launch(const void* in, void* out, int is_signed, int64_t n, int h, int w, int axis,
                       const double* wts, int radius, hipStream_t st) {
#define PL_RW_CASE(R, N)                                                                                     \
  if (radius == R)                                                                                           \
    return is_signed ? launch_rw_t<short, R, N>((const short*)in, (short*)out, n, h, w, axis, wts, st)      \
                     : launch_rw_t<unsigned short, R, N>((const unsigned short*)in, (unsigned short*)out,    \
                                                         n, h, w, axis, wts, st);
  PL_RW_CASE(4, 16)
  PL_RW_CASE(8, 16)
  PL_RW_CASE(12, 16)
  PL_RW_CASE(20, 16)
#undef PL_RW_CASE
  return -1;
}
