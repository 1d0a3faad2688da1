// CatPhan slice localisation, the edge image in PACKED FLOAT32 (round 6) -- the same plane as edge_stream.hip's
//   edges = skimage.filters.gaussian(skimage.filters.scharr(slice.astype(float)), sigma)        pylinac/ct.py:391, 3327-3328
// at less than half the vector instructions, for consumers that DECIDE from it and recompute exactly what they cannot decide.
//
// edge_stream_kernel spends 69 vector instructions per 64-pixel row, 40 of them float64 operations in scipy's order (no FMA),
// and float64 issues at the rate of unpacked float32 on this chip: only PACKED float32 (v_pk_*_f32: two pixels per lane and
// instruction) is cheaper.  Here a lane owns TWO adjacent columns of a 128-column strip:
//   * Scharr in float32 on integers below 2^24 (the responses 16 s0, 16 s1 of 16-bit pixels are at most 16 * 65535): exact;
//   * K = S0^2 + S1^2, sqrt (v_sqrt_f32, 1 ulp), the constant 1 / (16 sqrt 2), then both 9-tap sums as packed FMAs -- NOT
//     scipy's order and not its precision: the stored value v lies within kEs32Bracket float32 bit patterns of the exact
//     float64 value e (error budget below; tests measure the distance on every plane they make);
//   * a stored 0 is an exact 0 (every term of the sums is >= 0 and none underflows: the smallest non-zero term is ~1e-9).
// What must be exact stays exact:
//   * the consumers (pl_edge_otsu_ex, pl_edge_regions_ex with bracket = kEs32Bracket) decide bins / thresholds in the bit
//     domain from [bits(v) - B, bits(v) + B] and recompute a pixel from the 16-bit slice (es_exact_wave, scipy's float64
//     sequence) when the bracket straddles a decision;
//   * the exact EXTREMA over the selection (np.histogram's range, pylinac/ct.py:3334-3338): every lane keeps its smallest /
//     largest selected value with its position and the runner-up's value ("ghost"); a wave hands over the lanes within 2 B
//     patterns of its own extremum; es32_refine_kernel recomputes exactly the candidates within 2 B patterns of the SLICE's
//     float32 extremum -- the exact extremum is among them (e(p*) <= e(q) and both lie within B patterns of their stored
//     values, so bits(v(p*)) <= bits(v(q)) + 2 B) -- and reports a slice as unresolved (status 1: the caller repeats it on the
//     exact path) when a ghost or a full list says a candidate may be missing;
//   * max(raw) only feeds the "no edges" test np.max(edges) < 0.1 (ct.py:392).  The largest float32 raw value c = sqrt(K) / (16
//     sqrt 2) (5u) gives K back exactly as round((c 16 sqrt 2)^2) while K < 2^19 (raw < 32): there the reported maximum is the
//     exact float64 value; above, it is the float32 one (within 1e-6, far from any use).
// Error budget (u = 2^-24, all terms positive, so relative errors add without cancellation): K two roundings 2u, sqrt u + 2u,
// the constant and its product 2u -> raw 5u; axis 0: tap rounding u, pair sum u, five accumulations 5u -> 12u; axis 1 the
// same -> 19u <= 19 bit patterns (adjacent float32 patterns are at least 2^-24 apart, relatively); kEs32Bracket = 32.
#include "pl_common.h"
#include "edge_exact.h"

#ifndef PL_E32_AHEAD
#define PL_E32_AHEAD 4      // rows a wave has in flight ahead of the one it works on
#endif
#ifndef PL_E32_WANT_FACTOR
#define PL_E32_WANT_FACTOR 4
#endif

namespace {

constexpr int kE32Threads = 256;
constexpr int kE32Waves = kE32Threads / PL_WAVE;
constexpr unsigned kEs32Bracket = 32;
constexpr int kE32Cap = 128;                              // candidate entries per slice and side

typedef float f2 __attribute__((ext_vector_type(2)));

struct E32Cand { unsigned bits, ghost; int row, col; };

__device__ __forceinline__ float e32_from_prev(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float e32_from_next(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ f2 e32_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 e32_splat(float v) { return f2{v, v}; }

__global__ void e32_init_kernel(unsigned* __restrict__ kmax, unsigned* __restrict__ mn, unsigned* __restrict__ mx,
                                unsigned* __restrict__ cnt, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kE32Threads + threadIdx.x;
  if (i >= n) return;
  kmax[i] = 0u;
  mn[i] = 0x7f800000u;                                     // +inf: nothing selected yet
  mx[i] = 0u;                                              // (with mn = +inf this reads as "no maximum either")
  cnt[2 * i] = 0u;
  cnt[2 * i + 1] = 0u;
}

template <typename T, int RAD>
__global__ void __launch_bounds__(kE32Threads)
edge_stream32_kernel(const T* __restrict__ in, int h, int w, int strips, int segs, int seg_rows, int64_t items,
                     const double* __restrict__ wts, const int* __restrict__ spans, float* __restrict__ out,
                     unsigned* __restrict__ kmax_bits, unsigned* __restrict__ mn_bits, unsigned* __restrict__ mx_bits,
                     E32Cand* __restrict__ cands, unsigned* __restrict__ cand_cnt) {
  constexpr int WIN = 2 * RAD + 1, HL = (RAD + 2) / 2, OUTL = PL_WAVE - 2 * HL, OUTW = 2 * OUTL, PAD = RAD + (RAD & 1);
  constexpr bool kSignedT = (T)-1 < (T)0;
  __shared__ __attribute__((aligned(8))) float vbuf[kE32Waves][2 * PL_WAVE + 2 * PAD];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t item = (int64_t)blockIdx.x * kE32Waves + wv;
  if (item >= items) return;                               // a whole wave leaves; the kernel has no workgroup barrier
  const int s = (int)(item % strips);
  const int g = (int)((item / strips) % segs);
  const int64_t f = item / ((int64_t)strips * segs);

  float tw[RAD + 1];
#pragma unroll
  for (int k = 0; k <= RAD; ++k) tw[k] = (float)wts[k];    // RN32 of scipy's taps (tw[RAD] = centre)

  const int c_base = s * OUTW - 2 * HL;                    // even
  const int vc = c_base + 2 * lane;                        // the lane's columns: vc, vc + 1
  const int pc = es_clamp(vc, 0, w - 2);                   // the pair it loads (w is even: a pair is inside or outside the frame)
  const int r0 = g * seg_rows, r1 = min(h, r0 + seg_rows);
  const int vstart = r0 - RAD, vend = r1 + RAD;
  const T* src = in + f * (int64_t)h * w;
  const bool out_lane = lane >= HL && lane < PL_WAVE - HL && vc < w;
  const bool fix_left = c_base < 0, fix_right = c_base + 2 * PL_WAVE - 1 > w - 1;
  const int lane_first = -c_base / 2;                      // holds column 0 in .x   (when fix_left)
  const int lane_last = (w - 2 - c_base) / 2;              // holds column w - 1 in .y (when fix_right)
  const bool left_of = vc < 0, right_of = vc >= w;

  auto unpack = [&](unsigned d) -> f2 {
    float lo, hi;
    if (kSignedT) { lo = (float)(int)(short)(d & 0xffffu); hi = (float)((int)d >> 16); }
    else { lo = (float)(d & 0xffffu); hi = (float)(d >> 16); }
    if (fix_left | fix_right) {                            // (wave-uniform) pairs beyond the frame repeat its edge column
      hi = left_of ? lo : hi;
      lo = right_of ? hi : lo;
    }
    return f2{lo, hi};
  };
  const f2 three = e32_splat(3.0f), ten = e32_splat(10.0f);
  const float cinv = 0x1.6a09e6p-5f;                       // RN32(1 / (16 sqrt 2))
  auto edge_row = [&](f2 ra, f2 rb, f2 rc) -> f2 {
    const f2 dv = rc - ra;                                 // vertical difference: 16 x the 'edge' taps (exact integers)
    const f2 sv = e32_fma(rb, ten, three * (ra + rc));     // vertical smoothing
    // the horizontal neighbours of the pair (c, c + 1): c - 1 is the previous lane's .y, c + 2 the next lane's .x; the own
    // halves enter crosswise, which is an operand swizzle of the packed instruction, not a move
    const f2 pd = f2{e32_from_prev(dv.y), e32_from_next(dv.x)};      // {dv[c - 1], dv[c + 2]}
    const f2 ps = f2{e32_from_prev(sv.y), e32_from_next(sv.x)};      // {sv[c - 1], sv[c + 2]}
    const f2 S0 = e32_fma(dv, ten, three * (pd + dv.yx));            // 3 (left + right) + 10 centre
    const f2 S1 = sv.yx - ps;                                        // {sv[c+1] - sv[c-1], -(sv[c+2] - sv[c])}: only its square is used
    const f2 K = e32_fma(S0, S0, S1 * S1);
    f2 e = f2{__builtin_amdgcn_sqrtf(K.x), __builtin_amdgcn_sqrtf(K.y)} * e32_splat(cinv);
    if (fix_left) {                                        // mode 'nearest' of the Gaussian: the edge value of column 0 / w - 1
      const float t = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e.x), lane_first));
      e = lane < lane_first ? e32_splat(t) : e;
    }
    if (fix_right) {
      const float t = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e.y), lane_last));
      e = lane > lane_last ? e32_splat(t) : e;
    }
    return e;
  };

  const int fr = max(vstart, 0);
  const int hm1 = h - 1;
  const unsigned last_row = (unsigned)hm1 * (unsigned)w;
  auto ldo = [&](unsigned row_off) { return *reinterpret_cast<const unsigned*>(src + row_off + (unsigned)pc); };
  f2 ra = unpack(ldo((unsigned)max(fr - 1, 0) * (unsigned)w)), rb = unpack(ldo((unsigned)fr * (unsigned)w));
  unsigned noff = min((unsigned)(fr + 1) * (unsigned)w, last_row);
  f2 rc = unpack(ldo(noff));
  // PL_E32_AHEAD rows in flight per wave: with ONE (rounds up to r06z) a CU's 32 waves had 8 KB on their way, and the
  // kernel's reads ran at what that buys against a microsecond and a half of latency (4.4 GB/s per CU: its 1.15 ms)
  unsigned pend[PL_E32_AHEAD];
#pragma unroll
  for (int j = 0; j < PL_E32_AHEAD; ++j) {
    noff = min(noff + (unsigned)w, last_row);
    pend[j] = ldo(noff);
  }
  f2 e0 = edge_row(ra, rb, rc);
  f2 E[WIN];
#pragma unroll
  for (int i = 0; i < WIN; ++i) E[i] = e0;
  const bool has_spans = spans != nullptr;
  float* orow = out + (f * h + r0) * (int64_t)w;
  const PL_CONSTANT_AS int* srow = pl_constant_ptr(has_spans ? spans + 2 * r0 : nullptr);
  float* vb = vbuf[wv];

  // Everything below is kept as float32 BIT PATTERNS (the values are >= 0: patterns order like values, unsigned compares and
  // v_min_u32 / v_max_u32 need no NaN canonicalisation).
  unsigned rmx_x = 0u, rmx_y = 0u;                         // max of the raw Scharr magnitude over the segment's rows
  // the lane's extrema over the selection: value, position, and the runner-up's value (the "ghost").  Minimum side: an
  // unselected pixel reads +inf; anything below thr_lo takes the slow path.  Maximum side: unselected reads 0 (a selection
  // of zeros only needs no candidate: a stored 0 is an exact 0).
  constexpr unsigned kInf = 0x7f800000u, kB2 = 2u * kEs32Bracket;
  unsigned m1 = kInf, g1 = kInf, thr_lo = kInf;
  unsigned M1 = 0u, G1 = 0u, thr_hi = 0u;
  int mr = 0, mcol = 0, Mr = 0, Mcol = 0;
  auto note_min = [&](unsigned v, int row, int col) {
    if (v < thr_lo) {
      if (v < m1) { g1 = m1; m1 = v; mr = row; mcol = col; }
      else g1 = v < g1 ? v : g1;
      thr_lo = m1 ? m1 + kB2 + 1u : 0u;                    // an exact 0 needs no bracket: nothing is below it
    }
  };
  auto note_max = [&](unsigned v, int row, int col) {
    if (v > thr_hi) {
      if (v > M1) { G1 = M1; M1 = v; Mr = row; Mcol = col; }
      else G1 = v > G1 ? v : G1;
      thr_hi = M1 > kB2 + 1u ? M1 - kB2 - 1u : 0u;
    }
  };

  for (int base = fr + 1; base < vend; base += WIN) {
#pragma unroll
    for (int i = 0; i < WIN; ++i) {
      const int vr = base + i;
      ra = rb; rb = rc; rc = unpack(pend[0]);
#pragma unroll
      for (int j = 0; j + 1 < PL_E32_AHEAD; ++j) pend[j] = pend[j + 1];
      noff = min(noff + (unsigned)w, last_row);
      pend[PL_E32_AHEAD - 1] = ldo(noff);                  // in flight for PL_E32_AHEAD steps
      const f2 en = edge_row(ra, rb, rc);
      E[i] = en;
      if (__builtin_expect(vr > hm1, 0)) {                 // rows below the frame repeat the last edge row (wave-uniform)
        E[i] = E[(i + WIN - 1) % WIN];
        asm volatile("" : "+v"(E[i]));
      }
      const int ro = vr - RAD;
      if (ro >= r0 && ro < r1) {
        const f2 ctr = E[(i + WIN - RAD) % WIN];
        rmx_x = max(rmx_x, __float_as_uint(ctr.x));        // (halo lanes hold genuine columns or copies of them)
        rmx_y = max(rmx_y, __float_as_uint(ctr.y));
        f2 a0 = ctr * e32_splat(tw[RAD]);
#pragma unroll
        for (int k = RAD; k >= 1; --k)
          a0 = e32_fma(E[(i + 2 * WIN - RAD - k) % WIN] + E[(i + WIN - RAD + k) % WIN], e32_splat(tw[RAD - k]), a0);
        *reinterpret_cast<f2*>(vb + 2 * lane + PAD) = a0;
        pl_wave_sync();
        f2 a1 = a0 * e32_splat(tw[RAD]);
#pragma unroll
        for (int k = RAD; k >= 1; --k) {
          const f2 l = f2{vb[2 * lane + PAD - k], vb[2 * lane + PAD - k + 1]};
          const f2 r = f2{vb[2 * lane + PAD + k], vb[2 * lane + PAD + k + 1]};
          a1 = e32_fma(l + r, e32_splat(tw[RAD - k]), a1);
        }
        pl_wave_sync();                                    // (orders the next step's write behind these reads)
        bool sx = out_lane, sy = out_lane;
        if (has_spans) {
          const int c0 = srow[0], c1 = srow[1];
          sx = sx & (vc >= c0) & (vc < c1);
          sy = sy & (vc + 1 >= c0) & (vc + 1 < c1);
          srow += 2;
        }
        // fast test: is either selected value beyond the lane's thresholds?  (rare once the extrema have settled)
        const unsigned bx = __float_as_uint(a1.x), by = __float_as_uint(a1.y);
        const unsigned lo_x = sx ? bx : kInf, lo_y = sy ? by : kInf;
        const unsigned hi_x = sx ? bx : 0u, hi_y = sy ? by : 0u;
        if (__ballot(min(lo_x, lo_y) < thr_lo) != 0ull) {
          note_min(lo_x, ro, vc);
          note_min(lo_y, ro, vc + 1);
        }
        if (__ballot(max(hi_x, hi_y) > thr_hi) != 0ull) {
          note_max(hi_x, ro, vc);
          note_max(hi_y, ro, vc + 1);
        }
        if (out_lane) *reinterpret_cast<f2*>(orow + vc) = a1;
        orow += w;
      }
    }
  }
  // ---- hand over: the raw maximum, the wave's extrema, the candidates within 2 B patterns of them
  {
    // (lane 0's .x and lane 63's .y saw a zero neighbour across the strip's end: not a pixel's value)
    unsigned kb = lane == 0 ? rmx_y : (lane == PL_WAVE - 1 ? rmx_x : max(rmx_x, rmx_y));
    kb = pl_wave_reduce(kb, [](unsigned a, unsigned b) { return a > b ? a : b; });
    if (lane == 0 && kb) atomicMax(kmax_bits + f, kb);
  }
  const unsigned wmin = pl_wave_reduce(m1, [](unsigned a, unsigned b) { return a < b ? a : b; });
  if (wmin != kInf) {
    if (lane == 0) atomicMin(mn_bits + f, wmin);
    if (wmin != 0u && m1 <= wmin + kB2) {
      const unsigned slot = atomicAdd(cand_cnt + 2 * f, 1u);
      if (slot < (unsigned)kE32Cap) cands[(f * 2) * kE32Cap + slot] = E32Cand{m1, g1, mr, mcol};
    }
  }
  const unsigned wmax = pl_wave_reduce(M1, [](unsigned a, unsigned b) { return a > b ? a : b; });
  if (wmax != 0u) {
    if (lane == 0) atomicMax(mx_bits + f, wmax);
    if (M1 != 0u && M1 + kB2 >= wmax) {
      const unsigned slot = atomicAdd(cand_cnt + 2 * f + 1, 1u);
      if (slot < (unsigned)kE32Cap) cands[(f * 2 + 1) * kE32Cap + slot] = E32Cand{M1, G1, Mr, Mcol};
    }
  }
}

// One wave per slice: the exact float64 extrema from the candidates, the raw maximum, and the slice's status.
template <typename T>
__global__ void __launch_bounds__(PL_WAVE)
es32_refine_kernel(const T* __restrict__ raw, int h, int w, const double* __restrict__ wts, int rad,
                   const unsigned* __restrict__ kmax_bits, const unsigned* __restrict__ mn_bits, const unsigned* __restrict__ mx_bits,
                   const E32Cand* __restrict__ cands, const unsigned* __restrict__ cand_cnt, double* __restrict__ rawmax,
                   double* __restrict__ dmin, double* __restrict__ dmax, int32_t* __restrict__ status) {
  __shared__ double scratch[kEsScratch];
  const int64_t f = blockIdx.x;
  const int lane = threadIdx.x;
  const T* src = raw + f * (int64_t)h * w;
  const double pinf = __longlong_as_double(0x7ff0000000000000LL), ninf = __longlong_as_double((long long)0xfff0000000000000ULL);
  const unsigned mnb = mn_bits[f], mxb = mx_bits[f];
  int bad = 0;
  double lo = pinf, hi = ninf;
  if (mnb != 0x7f800000u) {                                // something was selected
    const unsigned B2 = 2u * kEs32Bracket;
    if (mnb == 0u) lo = 0.0;                               // a stored 0 is an exact 0
    else {
      const unsigned cnt = cand_cnt[2 * f];
      if (cnt > (unsigned)kE32Cap) bad = 1;
      const unsigned m = cnt < (unsigned)kE32Cap ? cnt : (unsigned)kE32Cap;
      for (unsigned i = 0; i < m; ++i) {                   // (wave-uniform)
        const E32Cand c = cands[(f * 2) * kE32Cap + i];
        if (c.bits > mnb + B2) continue;
        if (c.ghost <= mnb + B2) bad = 1;                  // the lane held a second candidate it could not keep
        const double v = es_exact_wave(src, h, w, c.row, c.col, wts, rad, scratch);
        lo = v < lo ? v : lo;
      }
    }
    if (mxb == 0u) hi = 0.0;
    else {
      const unsigned cnt = cand_cnt[2 * f + 1];
      if (cnt > (unsigned)kE32Cap) bad = 1;
      const unsigned m = cnt < (unsigned)kE32Cap ? cnt : (unsigned)kE32Cap;
      for (unsigned i = 0; i < m; ++i) {
        const E32Cand c = cands[(f * 2 + 1) * kE32Cap + i];
        if (c.bits + B2 < mxb) continue;
        if (c.ghost != 0u && c.ghost + B2 >= mxb) bad = 1;
        const double v = es_exact_wave(src, h, w, c.row, c.col, wts, rad, scratch);
        hi = v > hi ? v : hi;
      }
    }
    if (lo == pinf || hi == ninf) bad = 1;                 // (cannot happen: the extremum's own lane is a candidate)
  }
  if (lane == 0) {
    dmin[f] = lo;
    dmax[f] = hi;
    // the float32 maximum c = sqrt(K) / (16 sqrt 2) (1 +- 5u) gives K back exactly while K < 2^19
    const double c = (double)__uint_as_float(kmax_bits[f]);
    const double kk = rint((c * 0x1.6a09e667f3bcdp+4) * (c * 0x1.6a09e667f3bcdp+4));
    rawmax[f] = kk < 524288.0 ? es_edge_k(kk) : c;
    status[f] = bad;
  }
}

template <typename T>
int e32_launch(const T* in, int64_t n, int h, int w, const double* wts, int radius, const int* spans, float* out,
               unsigned char* work, double* rawmax, double* dmin, double* dmax, int32_t* status, hipStream_t st) {
  const int hl = (radius + 2) / 2, outw = 2 * (PL_WAVE - 2 * hl);
  const int strips = (int)pl_cdiv(w, outw);
  const int64_t want = (int64_t)PL_E32_WANT_FACTOR * pl_cu_count() * 32;   // waves the launch should at least have
  int segs = (int)pl_cdiv(want, n * strips);
  const int max_segs = (int)pl_cdiv(h, 32);
  if (segs > max_segs) segs = max_segs;
  if (segs < 1) segs = 1;
  const int seg_rows = (int)pl_cdiv(h, segs);
  segs = (int)pl_cdiv(h, seg_rows);
  const int64_t items = n * strips * segs;
  const int64_t blocks = pl_cdiv(items, kE32Waves);
  if (blocks > 0x7fffffffLL) { pl_set_error("pl_edge_plane32: batch too large for one launch"); return PL_ERR_INVALID_ARG; }
  unsigned* kmax = reinterpret_cast<unsigned*>(work);
  unsigned* mn = kmax + n;
  unsigned* mx = mn + n;
  unsigned* cnt = mx + n;
  E32Cand* cands = reinterpret_cast<E32Cand*>(work + (((size_t)5 * n * 4 + 15) & ~(size_t)15));
  hipLaunchKernelGGL(e32_init_kernel, dim3((unsigned)pl_cdiv(n, kE32Threads)), dim3(kE32Threads), 0, st, kmax, mn, mx, cnt, n);
#define E32_CASE(R)                                                                                                          \
  case R:                                                                                                                    \
    hipLaunchKernelGGL((edge_stream32_kernel<T, R>), dim3((unsigned)blocks), dim3(kE32Threads), 0, st, in, h, w, strips,     \
                       segs, seg_rows, items, wts, spans, out, kmax, mn, mx, cands, cnt);                                    \
    break;
  switch (radius) {
    E32_CASE(1) E32_CASE(2) E32_CASE(3) E32_CASE(4) E32_CASE(5) E32_CASE(6) E32_CASE(7) E32_CASE(8)
    default: pl_set_error("pl_edge_plane32: radius 1..8"); return PL_ERR_UNSUPPORTED;
  }
#undef E32_CASE
  hipLaunchKernelGGL((es32_refine_kernel<T>), dim3((unsigned)n), dim3(PL_WAVE), 0, st, in, h, w, wts, radius, kmax, mn, mx, cands,
                     cnt, rawmax, dmin, dmax, status);
  return pl_check_launch("pl_edge_plane32");
}

}  // namespace

extern "C" int pl_edge_plane32_bracket(void) { return (int)kEs32Bracket; }

extern "C" int64_t pl_edge_plane32_work_bytes(int64_t n) {
  if (n < 0) return 0;
  return (int64_t)((5 * n * 4 + 15) & ~(int64_t)15) + (int64_t)n * 2 * kE32Cap * (int64_t)sizeof(E32Cand);
}

extern "C" int pl_edge_plane32(const void* in, int dtype, int64_t n, int h, int w, const double* d_weights, int radius,
                               const int32_t* d_row_spans, float* d_out, unsigned char* d_work, double* d_rawmax, double* d_min,
                               double* d_max, int32_t* d_status, void* stream) {
  PL_REQUIRE(in && d_weights && d_out && d_work && d_rawmax && d_min && d_max && d_status, "null pointer");
  PL_REQUIRE(n >= 0 && n <= 0x3fffffLL && h > 0 && w > 0 && (int64_t)h * w < 0x7fffffffLL, "bad shape");
  PL_REQUIRE((w & 1) == 0 && w >= 2, "even width (a lane loads two columns as one dword); odd widths: pl_edge_plane");
  PL_REQUIRE(radius >= 1 && radius <= 8, "radius 1..8 (sigma <= 2 at truncate 4)");
  PL_REQUIRE(dtype == PL_I16 || dtype == PL_U16, "int16 / uint16 slices");
  PL_REQUIRE(((uintptr_t)in & 3) == 0 && ((uintptr_t)d_out & 7) == 0 && ((uintptr_t)d_work & 15) == 0, "alignment: slices 4, plane 8, work 16 bytes");
  if (n == 0) return PL_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PL_I16)
    return e32_launch<short>((const short*)in, n, h, w, d_weights, radius, d_row_spans, d_out, d_work, d_rawmax, d_min, d_max, d_status, st);
  return e32_launch<unsigned short>((const unsigned short*)in, n, h, w, d_weights, radius, d_row_spans, d_out, d_work, d_rawmax,
                                    d_min, d_max, d_status, st);
}
