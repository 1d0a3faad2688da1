// CatPhan slice localisation, the edge-image half as a STREAMING pass (pylinac/ct.py:391-392, 3327-3340):
//   raw   = skimage.filters.scharr(slice.astype(float))            pylinac/ct.py:391, 3327
//   max(raw)                                                       the "no edges" test np.max(edges) < 0.1, ct.py:392
//   edges = skimage.filters.gaussian(raw, sigma)  (mode 'nearest') ct.py:3328 (ndimage.gaussian_filter underneath)
//   edges[disk].min(), .max()                                      the histogram range of threshold_otsu, ct.py:3334-3338
//   thres = threshold_otsu(edges[disk]) * 0.8                      ct.py:3338-3340 (256 bins, np.histogram's edges)
//
// Round 3 (csrc/edge_field.hip, deleted) ran 24 x 64 tiles with three workgroup barriers each, evaluated the Scharr magnitude
// 1.5 times per pixel through twelve float64 multiply-adds, and wrote a float64 plane (2 MiB per 512 x 512 slice, four times
// the slice) that the histogram and the labelling kernel read again: 1.05 + 0.29 ms per 320 slices, 0.8 % of the HBM roofline.
//
// pl_edge_plane -- MARCHING STRIPS, one wave each, no workgroup barrier: a lane owns one column of a 64-column strip and walks
// down a segment of rows.
//   * Scharr in INTEGERS: every tap of scharr's two 3 x 3 kernels is a multiple of 1/16 and the pixels are 16-bit integers, so
//     16 s0 and 16 s1 are exact int32 values and separable (vertical difference / smoothing per lane from a three-row register
//     window, the horizontal half from the two neighbour lanes by DPP wave shifts); o = s0^2 + s1^2 is an exact float64
//     whatever the order of the reference's twelve accumulations (every partial sum is a multiple of 1/16 below 2^53), so
//     sqrt(o) / sqrt(2) -- the same two IEEE operations as the reference -- gives the same bits;
//   * axis 0 of the Gaussian from a (2 radius + 1)-row REGISTER window of the lane's column (the loop is unrolled by the
//     window length so that every slot is a fixed register), axis 1 through a wave-private LDS row; both in scipy's
//     symmetric-kernel order (centre tap, then (left + right) * weight from the outermost pair inwards, NI_Correlate1D);
//   * mode 'nearest' = the edge value at CLAMPED coordinates: rows beyond the frame repeat the first / last edge row, lanes
//     beyond the frame copy the lane that holds column 0 / w - 1;
//   * the plane leaves as float32 (1 MiB per slice; RN of the float64 value) or float64; the three extrema stay exact float64.
// pl_edge_otsu -- the 256-bin histogram of the selected pixels and skimage's threshold from it in one launch.  On a float32
// plane a pixel's bin is decided from the two float32 neighbours of its value (the exact value lies between them); the
// few pixels per slice whose interval straddles a bin edge are recomputed exactly from the 16-bit slice (es_exact_wave: the
// whole wave evaluates the (2 radius + 1)^2 Scharr values, the operation order of the streaming kernel).  The workgroup that
// finishes a slice last runs the class statistics (otsu_counts_kernel's order of operations).
#include "pl_common.h"
#include "edge_exact.h"

namespace {

constexpr int kEsThreads = 256;
constexpr int kEsWaves = kEsThreads / PL_WAVE;

// DPP wave shifts whose first / last lane receives 0 (bound_ctrl): no copy of the source into the destination first, as the
// "keeps its own value" form of pl_common.h needs; the two outermost lanes of a strip are halo that nothing reads
__device__ __forceinline__ int es_from_prev(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ int es_from_next(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }

// non-negative float64 values (and -inf / +inf as "empty") order like their bit patterns read as SIGNED 64-bit integers
__device__ __forceinline__ void es_atomic_max(double* addr, double v) {
  atomicMax(reinterpret_cast<long long*>(addr), __double_as_longlong(v));
}
__device__ __forceinline__ void es_atomic_min(double* addr, double v) {
  atomicMin(reinterpret_cast<long long*>(addr), __double_as_longlong(v));
}

// v_min_f64 / v_max_f64 return the other operand when one is a NaN: a lane is taken out of a running minimum / maximum by
// replacing the HIGH dword of its value with a NaN pattern -- one v_cndmask instead of a compare and two selects per
// extremum (round 6: the extrema and selects were a quarter of the kernel's vector instructions outside the float64 sums)
__device__ __forceinline__ double es_nan_unless(double v, bool keep) {
  const long long b = __double_as_longlong(v);
  const unsigned hi = keep ? (unsigned)((unsigned long long)b >> 32) : 0x7ff80000u;
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | ((unsigned long long)b & 0xffffffffull)));
}
__device__ __forceinline__ double es_min_num(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double es_max_num(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__global__ void es_init_kernel(double* __restrict__ rawmax, double* __restrict__ mn, double* __restrict__ mx, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kEsThreads + threadIdx.x;
  if (i >= n) return;
  const double pinf = __longlong_as_double(0x7ff0000000000000LL), ninf = __longlong_as_double((long long)0xfff0000000000000ULL);
  rawmax[i] = ninf;
  mn[i] = pinf;
  mx[i] = ninf;
}

template <typename T, int RAD, typename OutT>
__global__ void __launch_bounds__(kEsThreads)
edge_stream_kernel(const T* __restrict__ in, int h, int w, int strips, int segs, int seg_rows, int64_t items,
                   const double* __restrict__ wts, const int* __restrict__ spans, const uint8_t* __restrict__ mask,
                   OutT* __restrict__ out, double* __restrict__ rawmax, double* __restrict__ mn, double* __restrict__ mx) {
  constexpr int WIN = 2 * RAD + 1, HALO = RAD + 1, OUTW = PL_WAVE - 2 * HALO;
  __shared__ double vbuf[kEsWaves][PL_WAVE + 2 * RAD];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t item = (int64_t)blockIdx.x * kEsWaves + wv;
  if (item >= items) return;                             // a whole wave leaves; the kernel has no workgroup barrier
  const int s = (int)(item % strips);
  const int g = (int)((item / strips) % segs);
  const int64_t f = item / ((int64_t)strips * segs);

  double tw[RAD + 1];
#pragma unroll
  for (int k = 0; k <= RAD; ++k) tw[k] = wts[k];

  const int c_base = s * OUTW - HALO;
  const int vc = c_base + lane;
  const int cc = es_clamp(vc, 0, w - 1);
  const int r0 = g * seg_rows, r1 = min(h, r0 + seg_rows);
  const int vstart = r0 - RAD, vend = r1 + RAD;          // the edge rows (virtual: clamped into the frame) the segment needs
  const T* src = in + f * (int64_t)h * w;
  const bool out_lane = lane >= HALO && lane < PL_WAVE - HALO && vc < w;
  const bool fix_left = c_base < 0, fix_right = c_base + PL_WAVE - 1 > w - 1;
  const int lane_first = -c_base, lane_last = w - 1 - c_base;     // the lanes that hold column 0 / w - 1 (when fix_*)

  const double pinf = __longlong_as_double(0x7ff0000000000000LL), ninf = __longlong_as_double((long long)0xfff0000000000000ULL);
  double lo = pinf, hi = ninf, rmax = ninf;
  auto edge_row = [&](int ra, int rb, int rc) {
    const int dv = rc - ra;                                // vertical difference: 16 * the 'edge' taps
    const int sv = 3 * (ra + rc) + 10 * rb;                // vertical smoothing
    const int dl = es_from_prev(dv), dr = es_from_next(dv);
    const int sl = es_from_prev(sv), sr = es_from_next(sv);
    double e = es_edge(3 * (dl + dr) + 10 * dv, sr - sl);
    if (fix_left) { const double t = __shfl(e, lane_first, PL_WAVE); e = lane < lane_first ? t : e; }
    if (fix_right) { const double t = __shfl(e, lane_last, PL_WAVE); e = lane > lane_last ? t : e; }
    return e;
  };
  // prologue: the first edge row that lies inside the frame fills the whole window -- edge rows above the frame (first
  // segment) are copies of it (mode 'nearest'), elsewhere the other slots are overwritten before the first output reads them
  const int fr = max(vstart, 0);
  const int hm1 = h - 1;
  // raw rows are addressed by a 32-bit element offset that walks down with the march and stops at the last row (the clamp
  // scharr's border handling amounts to): two scalar instructions per step instead of a 64-bit multiply
  const unsigned last_row = (unsigned)hm1 * (unsigned)w;
  auto ldo = [&](unsigned row_off) { return (int)src[row_off + (unsigned)cc]; };
  int ra = ldo((unsigned)max(fr - 1, 0) * (unsigned)w), rb = ldo((unsigned)fr * (unsigned)w);
  unsigned noff = min((unsigned)(fr + 1) * (unsigned)w, last_row);       // the row below edge row fr
  int rc = ldo(noff);
  noff = min(noff + (unsigned)w, last_row);
  int pending = ldo(noff);                                               // the row below edge row fr + 1
  double e = edge_row(ra, rb, rc);
  double E[WIN];
#pragma unroll
  for (int i = 0; i < WIN; ++i) E[i] = e;
  const bool has_out = out != nullptr, has_mask = mask != nullptr, has_spans = spans != nullptr;
  OutT* orow = has_out ? out + (f * h + r0) * (int64_t)w : nullptr;   // wave-uniform: the lane's column is the store's vector offset
  const uint8_t* mrow = has_mask ? mask + (int64_t)r0 * w : nullptr;
  const PL_CONSTANT_AS int* srow = pl_constant_ptr(has_spans ? spans + 2 * r0 : nullptr);    // scalar loads
  double* vb = vbuf[wv];

  // slot i of the window holds edge row base + i.  Every step computes its edge row (the last iteration may run up to
  // 2 * radius rows past the segment: clamped loads, results unused); only the output half of a step is conditional, so the
  // raw-row rotation and the window slots are pure register renaming in the unrolled body
  for (int base = fr + 1; base < vend; base += WIN) {
#pragma unroll
    for (int i = 0; i < WIN; ++i) {
      const int vr = base + i;
      ra = rb; rb = rc; rc = pending;
      noff = min(noff + (unsigned)w, last_row);
      pending = ldo(noff);                                 // in flight for a whole step
      const double en = edge_row(ra, rb, rc);
      // rows below the frame (last segment only) repeat the last edge row: a register copy on the rare side of a branch
      E[i] = en;
      if (__builtin_expect(vr > hm1, 0)) {                 // (wave-uniform: a branch around two moves, not a select per row)
        E[i] = E[(i + WIN - 1) % WIN];
        asm volatile("" : "+v"(E[i]));
      }
      const int ro = vr - RAD;
      if (ro >= r0 && ro < r1) {
        const double ctr = E[(i + WIN - RAD) % WIN];
        double a0 = ctr * tw[RAD];
#pragma unroll
        for (int k = RAD; k >= 1; --k)
          a0 = a0 + (E[(i + 2 * WIN - RAD - k) % WIN] + E[(i + WIN - RAD + k) % WIN]) * tw[RAD - k];
        vb[lane + RAD] = a0;
        pl_wave_sync();
        double a1 = a0 * tw[RAD];
#pragma unroll
        for (int k = RAD; k >= 1; --k) a1 = a1 + (vb[lane + RAD - k] + vb[lane + RAD + k]) * tw[RAD - k];
        pl_wave_sync();                                    // (orders the next step's write behind these reads)
        bool sel = out_lane;
        if (has_spans) {
          sel = sel & (vc >= srow[0]) & (vc < srow[1]);
          srow += 2;
        } else if (has_mask) {
          sel = sel & (mrow[cc] != 0);                     // every lane reads inside the frame (clamped column)
          mrow += w;
        }
        rmax = es_max_num(rmax, es_nan_unless(ctr, out_lane));
        const double a1s = es_nan_unless(a1, sel);
        lo = es_min_num(lo, a1s);
        hi = es_max_num(hi, a1s);
        if (has_out) {
          if (out_lane) orow[vc] = (OutT)a1;
          orow += w;
        }
      }
    }
  }
  rmax = pl_wave_reduce(rmax, [](double a, double c) { return a > c ? a : c; });
  lo = pl_wave_reduce(lo, [](double a, double c) { return a < c ? a : c; });
  hi = pl_wave_reduce(hi, [](double a, double c) { return a > c ? a : c; });
  if (lane == 0) {
    if (rmax != ninf) es_atomic_max(rawmax + f, rmax);
    if (lo != pinf) es_atomic_min(mn + f, lo);
    if (hi != ninf) es_atomic_max(mx + f, hi);
  }
}

template <typename T, typename OutT>
int es_launch(const T* in, int64_t n, int h, int w, const double* wts, int radius, const int* spans, const uint8_t* mask,
              OutT* out, double* rawmax, double* mn, double* mx, hipStream_t st) {
  const int outw = PL_WAVE - 2 * (radius + 1);
  const int strips = (int)pl_cdiv(w, outw);
  // segments: enough waves to fill the chip several times over (8 waves per SIMD resident), at least 32 rows each so that
  // the 2 * radius extra edge rows of a segment stay a small share
  const int64_t want = 4LL * pl_cu_count() * 32;
  int segs = (int)pl_cdiv(want, n * strips);
  const int max_segs = (int)pl_cdiv(h, 32);
  if (segs > max_segs) segs = max_segs;
  if (segs < 1) segs = 1;
  const int seg_rows = (int)pl_cdiv(h, segs);
  segs = (int)pl_cdiv(h, seg_rows);
  const int64_t items = n * strips * segs;
  const int64_t blocks = pl_cdiv(items, kEsWaves);
  if (blocks > 0x7fffffffLL) { pl_set_error("pl_edge_plane: batch too large for one launch"); return PL_ERR_INVALID_ARG; }
  hipLaunchKernelGGL(es_init_kernel, dim3((unsigned)pl_cdiv(n, kEsThreads)), dim3(kEsThreads), 0, st, rawmax, mn, mx, n);
#define ES_CASE(R)                                                                                                         \
  case R:                                                                                                                  \
    hipLaunchKernelGGL((edge_stream_kernel<T, R, OutT>), dim3((unsigned)blocks), dim3(kEsThreads), 0, st, in, h, w, strips, \
                       segs, seg_rows, items, wts, spans, mask, out, rawmax, mn, mx);                                      \
    break;
  switch (radius) {
    ES_CASE(1) ES_CASE(2) ES_CASE(3) ES_CASE(4) ES_CASE(5) ES_CASE(6) ES_CASE(7) ES_CASE(8)
    default: pl_set_error("pl_edge_plane: radius 1..8"); return PL_ERR_UNSUPPORTED;
  }
#undef ES_CASE
  return pl_check_launch("pl_edge_plane");
}

// np.histogram(values, 256) + skimage.filters.threshold_otsu's class statistics (pylinac/ct.py:3334-3340), one launch:
// grid (parts, n); each workgroup bins a band of rows of its slice into an LDS histogram, adds it to the slice's counts in
// global memory, and the workgroup that arrives last runs the threshold (otsu_counts_kernel's operations).
template <typename PlaneT, typename T>
__global__ void __launch_bounds__(kEsThreads)
edge_otsu_kernel(const PlaneT* __restrict__ plane, const T* __restrict__ raw, int h, int w, const double* __restrict__ wts, int rad,
                 const int* __restrict__ spans, const uint8_t* __restrict__ mask, const double* __restrict__ lo_all,
                 const double* __restrict__ hi_all, double scale, uint32_t* __restrict__ work /* [n][258], zeroed */,
                 double* __restrict__ thr, double* __restrict__ otsu_raw, unsigned bracket) {
  constexpr int NB = 256;
  constexpr int kCopies = 8, kStride = NB + 1;
  __shared__ double s_edge[NB + 1];
  // two lives of one block (20 KB with the edges: eight workgroups per CU -- the loop below waits on memory, waves hide it):
  // the histogram copies and the exact recomputation's scratch while the band is binned, the class statistics afterwards
  constexpr int kPhase1 = kEsWaves * kEsScratch + (kCopies * kStride + 1) / 2, kPhase2 = 7 * NB;
  __shared__ double s_block[kPhase1 > kPhase2 ? kPhase1 : kPhase2];
  double(*s_scratch)[kEsScratch] = reinterpret_cast<double(*)[kEsScratch]>(s_block);
  unsigned* s_hist = reinterpret_cast<unsigned*>(s_block + kEsWaves * kEsScratch);
  double *s_c = s_block, *s_p = s_c + NB, *s_w1 = s_p + NB, *s_s1 = s_w1 + NB, *s_w2 = s_s1 + NB, *s_m2 = s_w2 + NB,
         *s_var = s_m2 + NB;
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t f = blockIdx.y;
  const int parts = gridDim.x, part = blockIdx.x;
  const double first = lo_all[f], last = hi_all[f];
  uint32_t* counts = work + f * (NB + 2);     // 256 bins, the arrival ticket, the number of exact evaluations
  // np.linspace(first, last, 257) (linspace_edges_kernel)
  {
    const double delta = last - first;
    const double step = delta / (double)NB;
    s_edge[tid] = step == 0.0 ? ((double)tid / (double)NB) * delta + first : (double)tid * step + first;
    if (tid == 0) s_edge[NB] = last;
    for (int k = tid; k < kCopies * kStride; k += kEsThreads) s_hist[k] = 0;
  }
  __syncthreads();
  const bool usable = first < last;                       // an empty or constant selection has no histogram to take
  // Float32 planes, the common case in the BIT domain (round 6; r06e: the kernel issued 38 vector instructions per pixel, most of them
  // the float64 bracket: two conversions, subtract / multiply / convert, two 8-byte LDS reads, three compares).  The exact
  // value of a pixel lies between the float32 neighbours of its stored value v (non-negative: bit patterns order like values):
  //     tab[k].x = bits(the smallest float32 >= edge k) + B        prev^B(v) >= edge k      <=>  bits(v) >= tab[k].x
  //     tab[k].y = how many bit patterns from there on satisfy     next^B(v) <  edge k + 1  (<= `last` for the last bin)
  // (B = `bracket`: 1 for a plane that stores RN32 of the exact value, kEs32Bracket for the packed-float32 kernel's plane)
  // so "certainly in bin k" is ONE unsigned compare (bits(v) - x < y) against one 8-byte entry, with the bin estimated in
  // float32.  Whatever that does not place -- a pixel next to an edge, a stored 0, an estimate one bin off -- takes the
  // float64 logic below unchanged, including the exact recomputation from the slice.  A/B on one box (profiles/r06g_*):
  // 521 -> 435 us per 2000 slices.  (A first form that also asked the table about the neighbouring bins was SLOWER, 674 us.)
  __shared__ uint2 s_tab[NB];
  if constexpr (sizeof(PlaneT) == 4) {
    if (usable) {
      const double e0 = s_edge[tid], e1 = s_edge[tid + 1];
      unsigned lo_b = 1u, hi_b = 0u;
      if (e0 < 3.0e38 && e1 < 3.0e38) {
        if (e0 > 0.0) {
          float c = (float)e0;                              // RN; step up when it rounded down
          if ((double)c < e0) c = __uint_as_float(__float_as_uint(c) + 1u);
          lo_b = __float_as_uint(c) + bracket;              // bits(v) - bracket >= bits(c)
        }
        if (e1 > 0.0) {
          float c = (float)e1;                              // the largest float32 < e1 (<= e1 for the closed last bin)
          const bool over = tid == NB - 1 ? ((double)c > e1) : ((double)c >= e1);
          unsigned cb = __float_as_uint(c);
          if (over) cb = cb ? cb - 1u : 0u;
          hi_b = cb + 1u > bracket ? cb + 1u - bracket : 0u;   // bits(v) + bracket <= cb  <=>  bits(v) < cb + 1 - bracket
        }
      }
      s_tab[tid] = uint2{lo_b, hi_b > lo_b ? hi_b - lo_b : 0u};
    }
    __syncthreads();
  }
  if (usable) {
    const double inv = (double)NB / (last - first);
    const float first32 = (float)first, inv32 = (float)inv;
    // the lanes of a wave spread their counts over kCopies copies of the table (smooth planes put most of a wave into one or
    // two bins: a same-address LDS atomic serialises); copy c sits kStride words further, i.e. in the next bank
    unsigned* hist = s_hist + (lane & (kCopies - 1)) * kStride;
    const int rows_per = (h + parts - 1) / parts;
    const int rb = part * rows_per, re = min(h, rb + rows_per);
    const PlaneT* pl = plane + f * (int64_t)h * w;
    const T* src = raw ? raw + f * (int64_t)h * w : nullptr;
    // A chunk = up to 512 pixels of one row (8 loads per lane); the NEXT chunk's loads are issued before this one is binned:
    // the kernel is a stream of 4-byte loads whose rate is bytes in flight / memory latency (measured: with 4 loads per
    // wave in flight 2.2 TB/s, and neither the LDS atomics nor the edge reads nor the waves per SIMD moved it)
    constexpr int U = 8;
    struct Chunk { int r, cb, c1; };
    const PL_CONSTANT_AS int* cspans = pl_constant_ptr(spans);
    auto row_span = [&](int r, int& c0, int& c1) {
      c0 = 0; c1 = w;
      if (spans) { c0 = cspans[2 * r]; c1 = cspans[2 * r + 1]; }
    };
    auto first_chunk = [&](int r) {                        // the first non-empty chunk at or after row r (r >= re: none)
      Chunk ck{r, 0, 0};
      while (ck.r < re) {
        int c0;
        row_span(ck.r, c0, ck.c1);
        ck.cb = c0;
        if (c0 < ck.c1) break;
        ck.r += kEsWaves;
      }
      return ck;
    };
    auto next_chunk = [&](Chunk ck) {
      ck.cb += U * PL_WAVE;
      if (ck.cb < ck.c1) return ck;
      return first_chunk(ck.r + kEsWaves);
    };
    auto load_chunk = [&](const Chunk& ck, PlaneT (&pv)[U], unsigned& vmask) {
      vmask = 0;
      if (ck.r >= re) return;                              // wave-uniform
      const PlaneT* prow = pl + (int64_t)ck.r * w;
      const uint8_t* mrow = mask ? mask + (int64_t)ck.r * w : nullptr;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = ck.cb + u * PL_WAVE + lane;
        bool v = c < ck.c1;
        const int cl = v ? c : ck.c1 - 1;                  // every lane loads inside the row
        pv[u] = prow[cl];
        if (mrow) v = v & (mrow[cl] != 0);
        vmask |= v ? (1u << u) : 0u;
      }
    };
    auto bin_pixel = [&](int r, int c0u, PlaneT pvu, bool vld) {
          if constexpr (sizeof(PlaneT) == 4) {
            const unsigned vb = __float_as_uint(pvu);
            int k = (int)((pvu - first32) * inv32);
            k = k < 0 ? 0 : (k > NB - 1 ? NB - 1 : k);
            const uint2 e = s_tab[k];
            const bool placed = vld & ((vb - e.x) < e.y);
            if (placed) atomicAdd(&hist[k], 1u);
            if (__ballot(vld & !placed) == 0ull) return;
            vld = vld & !placed;                            // the wave's other pixels: the float64 logic
          }
          double vlo, vhi;
          if constexpr (sizeof(PlaneT) == 4) es_f32_bracket_n(pvu, bracket, vlo, vhi);
          else { vlo = (double)pvu; vhi = vlo; }
          // the bin of vlo as estimated, and whether the whole bracket provably lies in it
          int idx = (int)((vlo - first) * inv);
          idx = idx < 0 ? 0 : (idx > NB - 1 ? NB - 1 : idx);
          const double e0 = s_edge[idx], e1 = s_edge[idx + 1];
          const bool fast = vld & (vlo >= e0) & ((vhi < e1) | ((idx == NB - 1) & (vhi <= last)));
          int bin = fast ? idx : -1;
          if (__ballot(vld & !fast)) {                // rare: an estimate off by one, a bracket across an edge
            bool exact_needed = false;
            if (vld && !fast) {
              if (vlo >= first && vhi <= last) {
                while (idx > 0 && vlo < s_edge[idx]) --idx;
                while (idx < NB - 1 && vlo >= s_edge[idx + 1]) ++idx;
                if (idx == NB - 1 || vhi < s_edge[idx + 1]) bin = idx;
                else exact_needed = true;
              } else if constexpr (sizeof(PlaneT) == 4) {
                exact_needed = vhi >= first && vlo <= last;  // the bracket straddles an end of the range
              }                                              // (a float64 value outside the range is dropped, like np.histogram)
            }
            if constexpr (sizeof(PlaneT) == 4) {
              unsigned long long todo = __ballot(exact_needed);
              while (todo) {                                 // wave-uniform
                const int l = __builtin_ctzll(todo);
                todo &= todo - 1;
                const double v = es_exact_wave(src, h, w, r, c0u + l, wts, rad, s_scratch[wv]);
                if (lane == l) atomicAdd(&counts[NB + 1], 1u);
                if (lane == l && v >= first && v <= last) {
                  int k = (int)((v - first) * inv);
                  k = k < 0 ? 0 : (k > NB - 1 ? NB - 1 : k);
                  while (k > 0 && v < s_edge[k]) --k;
                  while (k < NB - 1 && v >= s_edge[k + 1]) ++k;
                  bin = k;
                }
              }
            }
          }
          if (bin >= 0) atomicAdd(&hist[bin], 1u);
    };
    auto bin_chunk = [&](const Chunk& ck, const PlaneT (&pv)[U], unsigned vmask) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (ck.cb + u * PL_WAVE >= ck.c1) break;           // wave-uniform
        bin_pixel(ck.r, ck.cb + u * PL_WAVE, pv[u], (vmask >> u) & 1u);
      }
    };
    PlaneT pa[U], pb[U];
    unsigned ma, mb;
    Chunk ca = first_chunk(rb + wv);
    load_chunk(ca, pa, ma);
    while (ca.r < re) {
      const Chunk cb2 = next_chunk(ca);
      load_chunk(cb2, pb, mb);
      bin_chunk(ca, pa, ma);
      if (cb2.r >= re) break;
      ca = next_chunk(cb2);
      load_chunk(ca, pa, ma);
      bin_chunk(cb2, pb, mb);
    }
  }
  __syncthreads();
  unsigned own = 0;                                        // this workgroup's count of bin `tid`
  if (usable)
    for (int k = 0; k < kCopies; ++k) own += s_hist[k * kStride + tid];
  if (parts == 1) {
    // the workgroup IS its slice: no merge through global atomics, no arrival ticket, and above all no device-scope fences
    // (each one writes the L2 back; with a fence pair in all 2 000 workgroups they were a quarter of the kernel)
    counts[tid] = own;
    __syncthreads();                                       // the block is about to change lives
  } else {
    // Several workgroups per slice (small batches).  Every access to `counts` is a device-scope atomic, performed at the
    // device's coherence point, so ordering is all that is needed: the merging atomics RETURN (the wave waits for them
    // because the barrier below consumes the value), then one lane takes the arrival ticket, and the last workgroup reads
    // the sums back with atomics too -- no __threadfence (an L2 write-back each on this chip)
    unsigned old = 0;
    if (own) old = atomicAdd(&counts[tid], own);
    const int never = __syncthreads_or(old == 0xffffffffu);   // (a count cannot reach 2^32 - 1: h * w < 2^31)
    if (tid == 0) s_last = atomicAdd(&counts[NB], never ? 0u : 1u) == (unsigned)(parts - 1);
    __syncthreads();
    if (!s_last) return;
    own = atomicAdd(&counts[tid], 0u);
  }
  // ---- skimage 0.18.3 threshold_otsu on the finished counts (otsu_counts_kernel in ct.hip states the reference's order of
  // operations).  Round 3's form -- one lane walking two 256-step chains through LDS, then a 254-step arg-max -- cost 36 us
  // per workgroup, a quarter of this kernel when every workgroup is its slice's last.  Here:
  //   * weight1 / weight2 (cumulative COUNTS) are integer prefix sums: exact in any order, so they are parallel scans;
  //   * the two cumulative sums of counts * centres must keep numpy's left-to-right float64 order: wave 0 runs the forward
  //     chain, wave 1 the reversed one, each from registers (v_readlane feeds the next term, no LDS round trip per step);
  //   * np.argmax (first maximum; a NaN wins and the first NaN wins) is a ballot over the 255 variances.
  const int i = tid;
  const unsigned cu = own;
  const double ci = (double)cu;
  const double centre = (s_edge[i] + s_edge[i + 1]) / 2.0;
  unsigned* s_cu = reinterpret_cast<unsigned*>(s_var);     // the variances come later
  s_cu[i] = cu;
  s_p[i] = ci * centre;
  __syncthreads();
  if (wv < 2) {
    const bool fwd = wv == 0;
    // bin held by this lane in block b: forward b * 64 + lane, backward 255 - (b * 64 + lane)
    double term[4], mine[4];
    unsigned cnt[4];
#pragma unroll
    for (int bq = 0; bq < 4; ++bq) {
      const int k = fwd ? bq * 64 + lane : NB - 1 - (bq * 64 + lane);
      term[bq] = s_p[k];
      cnt[bq] = s_cu[k];
    }
    double run = 0.0;
    unsigned carry = 0;
#pragma unroll
    for (int bq = 0; bq < 4; ++bq) {
      for (int l = 0; l < PL_WAVE; ++l) {
        run = run + pl_readlane_f64(term[bq], l);          // every lane keeps the same `run`
        if (lane == l) mine[bq] = run;
      }
      unsigned c = cnt[bq];                                // inclusive scan of the counts inside the block
#pragma unroll
      for (int o = 1; o < PL_WAVE; o <<= 1) {
        const unsigned t = __shfl_up(c, o, PL_WAVE);
        if (lane >= o) c += t;
      }
      c += carry;
      carry = __shfl(c, PL_WAVE - 1, PL_WAVE);
      const int k = fwd ? bq * 64 + lane : NB - 1 - (bq * 64 + lane);
      if (fwd) { s_w1[k] = (double)c; s_s1[k] = mine[bq]; }
      else { s_w2[k] = (double)c; s_m2[k] = mine[bq]; }
    }
  }
  __syncthreads();
  double var = 0.0;
  if (i < NB - 1) {
    const double mean1 = s_s1[i] / s_w1[i];
    const double mean2 = s_m2[i + 1] / s_w2[i + 1];
    const double d = mean1 - mean2;
    var = (s_w1[i] * s_w2[i + 1]) * (d * d);
  }
  __syncthreads();                                         // s_cu (aliased by s_var) is no longer read
  const bool live = i < NB - 1;
  const bool isnan_ = live && var != var;
  // per wave: first NaN lane, the maximum, then the first lane holding the workgroup's maximum
  const unsigned long long nanb = __ballot(isnan_);
  double wmax = live && !isnan_ ? var : __longlong_as_double((long long)0xfff0000000000000ULL);
  wmax = pl_wave_reduce(wmax, [](double a, double c) { return a > c ? a : c; });
  if (lane == 0) { s_var[wv] = wmax; reinterpret_cast<int*>(s_var + 8)[wv] = nanb ? wv * 64 + (int)__builtin_ctzll(nanb) : NB; }
  __syncthreads();
  double gmax = s_var[0];
  int first_nan = reinterpret_cast<int*>(s_var + 8)[0];
  for (int k = 1; k < kEsWaves; ++k) {
    gmax = s_var[k] > gmax ? s_var[k] : gmax;
    const int fn = reinterpret_cast<int*>(s_var + 8)[k];
    first_nan = fn < first_nan ? fn : first_nan;
  }
  const unsigned long long eqb = __ballot(live && !isnan_ && var == gmax);
  __syncthreads();
  if (lane == 0) reinterpret_cast<int*>(s_var + 8)[wv] = eqb ? wv * 64 + (int)__builtin_ctzll(eqb) : NB;
  __syncthreads();
  if (i == 0) {
    double otsu;
    if (!(first < last)) {
      otsu = first == last ? first : __longlong_as_double(0x7ff8000000000000LL);
    } else {
      int best_i = first_nan;
      if (best_i >= NB) {
        for (int k = 0; k < kEsWaves; ++k) {
          const int q = reinterpret_cast<int*>(s_var + 8)[k];
          best_i = q < best_i ? q : best_i;
        }
      }
      if (best_i >= NB) best_i = 0;                        // (cannot happen: some variance equals the maximum)
      otsu = (s_edge[best_i] + s_edge[best_i + 1]) / 2.0;
    }
    if (otsu_raw) otsu_raw[f] = otsu;
    thr[f] = otsu * scale;
  }
}

}  // namespace

extern "C" int pl_edge_plane(const void* in, int dtype, int64_t n, int h, int w, const double* d_weights, int radius,
                             const int32_t* d_row_spans, const uint8_t* d_mask, void* d_out, int out_dtype,
                             double* d_rawmax, double* d_min, double* d_max, void* stream) {
  PL_REQUIRE(in && d_weights && d_rawmax && d_min && d_max, "null pointer");
  PL_REQUIRE(n >= 0 && h > 0 && w > 0 && (int64_t)h * w < 0x7fffffffLL, "bad shape");
  PL_REQUIRE(radius >= 1 && radius <= 8, "radius 1..8 (sigma <= 2 at truncate 4)");
  PL_REQUIRE(dtype == PL_I16 || dtype == PL_U16, "int16 / uint16 slices");
  PL_REQUIRE(out_dtype == PL_F32 || out_dtype == PL_F64, "float32 or float64 plane");
  PL_REQUIRE(!(d_row_spans && d_mask), "row spans or a byte mask, not both");
  if (n == 0) return PL_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PL_I16) {
    if (out_dtype == PL_F32)
      return es_launch<short, float>((const short*)in, n, h, w, d_weights, radius, d_row_spans, d_mask, (float*)d_out, d_rawmax, d_min, d_max, st);
    return es_launch<short, double>((const short*)in, n, h, w, d_weights, radius, d_row_spans, d_mask, (double*)d_out, d_rawmax, d_min, d_max, st);
  }
  if (out_dtype == PL_F32)
    return es_launch<unsigned short, float>((const unsigned short*)in, n, h, w, d_weights, radius, d_row_spans, d_mask, (float*)d_out, d_rawmax, d_min, d_max, st);
  return es_launch<unsigned short, double>((const unsigned short*)in, n, h, w, d_weights, radius, d_row_spans, d_mask, (double*)d_out, d_rawmax, d_min, d_max, st);
}

/* round 3's entry point, kept: the float64 plane with a byte mask */
extern "C" int pl_scharr_gaussian(const void* in, int dtype, int64_t n, int h, int w, const double* d_weights, int radius,
                                  const uint8_t* d_mask, double* d_out, double* d_rawmax, double* d_min, double* d_max,
                                  void* stream) {
  PL_REQUIRE(d_out, "null pointer");
  return pl_edge_plane(in, dtype, n, h, w, d_weights, radius, nullptr, d_mask, d_out, PL_F64, d_rawmax, d_min, d_max, stream);
}

/* np.histogram(plane[selection], 256) over [d_min, d_max] + skimage's threshold_otsu on it, per frame, one launch */
extern "C" int pl_edge_otsu_ex(const void* d_plane, int plane_dtype, const void* in_raw, int dtype, int64_t n, int h, int w,
                               const double* d_weights, int radius, const int32_t* d_row_spans, const uint8_t* d_mask,
                               const double* d_min, const double* d_max, double scale, uint32_t* d_work, double* d_thr,
                               double* d_raw_otsu, int bracket, void* stream) {
  PL_REQUIRE(bracket >= 1 && bracket <= 4096, "bracket: 1 .. 4096 float32 bit patterns");
  PL_REQUIRE(d_plane && d_min && d_max && d_work && d_thr, "null pointer");
  PL_REQUIRE(n >= 0 && n <= 65535 && h > 0 && w > 0, "bad shape");
  PL_REQUIRE(plane_dtype == PL_F32 || plane_dtype == PL_F64, "float32 or float64 plane");
  PL_REQUIRE(plane_dtype == PL_F64 || (in_raw && d_weights && radius >= 1 && radius <= 8 && (dtype == PL_I16 || dtype == PL_U16)),
             "a float32 plane needs the int16 / uint16 slices and the taps it was made from");
  PL_REQUIRE(!(d_row_spans && d_mask), "row spans or a byte mask, not both");
  if (n == 0) return PL_OK;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(d_work, 0, (size_t)n * 258 * sizeof(uint32_t), st);
  if (e != hipSuccess) { pl_set_error("pl_edge_otsu: memset: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
  // bands: enough workgroups to fill the chip at small batches, at least 32 rows each
  int parts = (int)pl_cdiv(2LL * pl_cu_count(), n);
  const int max_parts = (int)pl_cdiv(h, 32);
  if (parts > max_parts) parts = max_parts;
  if (parts < 1) parts = 1;
  const dim3 grid((unsigned)parts, (unsigned)n);
#define EO_LAUNCH(P, T)                                                                                                     \
  hipLaunchKernelGGL((edge_otsu_kernel<P, T>), grid, dim3(kEsThreads), 0, st, (const P*)d_plane, (const T*)in_raw, h, w,    \
                     d_weights, radius, d_row_spans, d_mask, d_min, d_max, scale, d_work, d_thr, d_raw_otsu, (unsigned)bracket)
  if (plane_dtype == PL_F32) {
    if (dtype == PL_I16) EO_LAUNCH(float, short);
    else EO_LAUNCH(float, unsigned short);
  } else {
    EO_LAUNCH(double, short);
  }
#undef EO_LAUNCH
  return pl_check_launch("pl_edge_otsu");
}

extern "C" int pl_edge_otsu(const void* d_plane, int plane_dtype, const void* in_raw, int dtype, int64_t n, int h, int w,
                            const double* d_weights, int radius, const int32_t* d_row_spans, const uint8_t* d_mask,
                            const double* d_min, const double* d_max, double scale, uint32_t* d_work, double* d_thr,
                            double* d_raw_otsu, void* stream) {
  return pl_edge_otsu_ex(d_plane, plane_dtype, in_raw, dtype, n, h, w, d_weights, radius, d_row_spans, d_mask, d_min, d_max, scale,
                         d_work, d_thr, d_raw_otsu, 1, stream);
}
