// Circular / collapsed-circular profile sampling.
//
// Replaces: ndimage.map_coordinates(image, [y, x], order=0) as called by
//   CircleProfile._profile            pylinac/core/profile.py:2279-2283
//   CollapsedCircleProfile._profile   pylinac/core/profile.py:2473-2483 (sum over num_profiles radii,
//                                      then /= num_profiles)
// callers: Starshot (pylinac/starshot.py:770-782), CTP528 MTF (pylinac/ct.py:1561-1580), CatPhan
// origin search (pylinac/ct.py:2468-2495).
//
// Semantics (scipy 1.15.3, verified by probing): nearest neighbour index floor(c + 0.5); mode
// 'constant', cval 0: ANY coordinate outside [0, n-1] -- even fractionally -- yields 0.
// x = cos*r + cx and y = sin*r + cy are evaluated in float64 in exactly that order (no FMA) from
// host-computed cos/sin tables (numpy's libm values; device sin/cos may differ in the last ulp,
// which would move a sample across a rounding boundary).
// The radii are accumulated in their given order in float64, like `profile += ...`.
// One lane per sample; the gather goes through L2 (each ring touches every row it crosses once).
#include "pl_common.h"

namespace {

template <typename T>
__global__ void __launch_bounds__(256)
circle_profile_kernel(const T* __restrict__ img, int h, int w, const double* __restrict__ cosv,
                      const double* __restrict__ sinv, int nsamp, const double* __restrict__ radii, int nr,
                      const double* __restrict__ cx, const double* __restrict__ cy, double divisor,
                      double* __restrict__ out) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  const size_t frame = blockIdx.y;
  if (s >= nsamp) return;
  const T* f = img + frame * (size_t)h * w;
  const double c = cosv[s], sn = sinv[s];
  const double x0 = cx[frame], y0 = cy[frame];
  double acc = 0.0;
  for (int k = 0; k < nr; ++k) {
    const double r = radii[frame * nr + k];
    const double x = c * r + x0;
    const double y = sn * r + y0;
    double v = 0.0;
    if (x >= 0.0 && x <= (double)(w - 1) && y >= 0.0 && y <= (double)(h - 1)) {
      const int xi = (int)floor(x + 0.5), yi = (int)floor(y + 0.5);
      v = (double)f[(size_t)yi * w + xi];
    }
    acc = acc + v;
  }
  out[frame * (size_t)nsamp + s] = (divisor == 1.0) ? acc : acc / divisor;
}

// The same sampling on combine_surrounding_slices(slice +- k, "max") (pylinac/ct.py:3351-3386, called from
// CTP528CP504's circle profile, ct.py:1561-1580) WITHOUT building the combined slices: the maximum over the 2k + 1
// neighbouring slices is taken per tap (a ring of 20 radii touches ~2 % of a slice; the full combination moved every pixel
// of 7 slices).  np.max over exact integers commutes with the nearest-neighbour gather, so the samples are identical.
// Slice z of a volume reads z - k .. z + k of ITS volume the way the reference indexes its list (pl_combine_slices: a
// negative index wraps to the end of the volume, an index past the end -- IndexError in the reference -- reuses the last
// slice and the caller discards the profile).
// Lanes: EIGHT consecutive samples x EIGHT radii per wave (a lane takes radii rl, rl + 8, rl + 16 ...): the taps of one
// wave-wide gather then lie in a patch of about 4 x 3 pixels -- one to four cache lines -- where a wave of 64 consecutive
// samples of one radius spread over up to 32 rows (the kernel was bound by the address path: 47 cycles per gather
// instruction, SQ_WAIT_INST_ANY 85 %).  The taps are integers, so the sum over the radii is exact in ANY order: each lane
// adds its own radii, a three-step butterfly adds the eight lanes of a sample -- the same float64 as the reference's
// sequential `profile += ...`.  KPM >= 0: the 2 KPM + 1 slices are compile-time, so a tap's gathers are issued together
// (rounds 1-3: one runtime loop of 140 dependent gathers per sample).
constexpr int kCpRadLanes = 8, kCpSamples = 256 / kCpRadLanes;

template <typename T, int KPM>
__global__ void __launch_bounds__(256)
circle_profile_combined_kernel(const T* __restrict__ stack, int h, int w, const int64_t* __restrict__ slice_index,
                               int64_t per_volume, int k_pm, const double* __restrict__ cosv,
                               const double* __restrict__ sinv, int nsamp, const double* __restrict__ radii, int nr,
                               const double* __restrict__ cx, const double* __restrict__ cy, double divisor,
                               double* __restrict__ out, double* __restrict__ margin) {
  const int rl = threadIdx.x & (kCpRadLanes - 1);
  const int s_raw = blockIdx.x * kCpSamples + (threadIdx.x / kCpRadLanes);
  const bool live = s_raw < nsamp;
  const int s = live ? s_raw : nsamp - 1;                    // every lane stays for the butterfly
  const size_t frame = blockIdx.y;
  const int64_t g = slice_index ? slice_index[frame] : (int64_t)frame;
  const int64_t v0 = (g / per_volume) * per_volume, z = g - v0;
  const size_t per_frame = (size_t)h * w;
  const double c = cosv[s], sn = sinv[s];
  const double x0 = cx[frame], y0 = cy[frame];
  auto slice_of = [&](int d) {                               // the reference's list index z + d inside the slice's own volume
    int64_t q = z + d;
    if (q < 0) q += per_volume;
    if (q < 0) q = 0;
    if (q >= per_volume) q = per_volume - 1;
    return stack + (size_t)(v0 + q) * per_frame;
  };
  const double* rad = radii + frame * nr;
  // `margin` (optional, [frames], preset to +inf): how far the centre may move before ANY decision of this profile changes --
  // the smallest distance of a coordinate + 0.5 to an integer (the nearest-pixel choice) or of a coordinate to 0 / n - 1 (the
  // inside test).  ct.ctp528_batch samples about a centre fitted on the DEVICE and accepts the profile only if the exact
  // host fit (np.polyfit, bit for bit the reference's) lies closer than this margin: then every tap is the same pixel.
  double mrg = __longlong_as_double(0x7ff0000000000000LL);
  const bool want_margin = margin != nullptr;
  // a ring that stays a pixel clear of the frame on every side cannot meet a bounds decision: the inside test's margin is
  // then dropped from every tap (wave-uniform; the radii are sorted neither way, so the largest is looked up once)
  bool near_border = true;
  if (want_margin) {
    double rmax = 0.0;
    for (int k = 0; k < nr; ++k) rmax = fmax(rmax, fabs(rad[k]));
    near_border = !(x0 - rmax > 1.0 && x0 + rmax < (double)(w - 2) && y0 - rmax > 1.0 && y0 + rmax < (double)(h - 2));
  }
  auto tap = [&](double r, unsigned& off) {                  // -> inside?, offset of the nearest pixel
    const double x = c * r + x0;
    const double y = sn * r + y0;
    const bool in = x >= 0.0 && x <= (double)(w - 1) && y >= 0.0 && y <= (double)(h - 1);
    const double fx = floor(x + 0.5), fy = floor(y + 0.5);
    const int xi = (int)fx, yi = (int)fy;
    off = in ? (unsigned)yi * (unsigned)w + (unsigned)xi : 0u;
    if (want_margin) {
      // distance of c + 0.5 to the nearest integer = 0.5 - |frac - 0.5|; both coordinates, then the running minimum
      const double dx = ((x + 0.5) - fx) - 0.5, dy = ((y + 0.5) - fy) - 0.5;
      double m = 0.5 - fmax(fabs(dx), fabs(dy));
      if (near_border)                                       // |min(c, n - 1 - c)| = the distance to the nearer bound
        m = fmin(m, fmin(fabs(fmin(x, (double)(w - 1) - x)), fabs(fmin(y, (double)(h - 1) - y))));
      mrg = fmin(mrg, m);                                    // (a NaN coordinate never lowers it: the caller tests NaN centres)
    }
    return in;
  };
  double acc = 0.0;                                          // integers: exact whatever the order
  if constexpr (KPM >= 0) {
    constexpr int NS = 2 * KPM + 1;
    const T* base[NS];
#pragma unroll
    for (int d = 0; d < NS; ++d) base[d] = slice_of(d - KPM);
    for (int k = rl; k < nr; k += kCpRadLanes) {
      unsigned o0;
      const bool in0 = tap(rad[k], o0);
      T e0[NS];
#pragma unroll
      for (int d = 0; d < NS; ++d) e0[d] = base[d][o0];
      T m0 = e0[0];
#pragma unroll
      for (int d = 1; d < NS; ++d) m0 = e0[d] > m0 ? e0[d] : m0;
      acc = acc + (in0 ? (double)m0 : 0.0);
    }
  } else {
    for (int k = rl; k < nr; k += kCpRadLanes) {
      unsigned o0;
      const bool in0 = tap(rad[k], o0);
      T m = slice_of(-k_pm)[o0];
      for (int d = -k_pm + 1; d <= k_pm; ++d) {
        const T e = slice_of(d)[o0];
        m = e > m ? e : m;
      }
      acc = acc + (in0 ? (double)m : 0.0);
    }
  }
#pragma unroll
  for (int o = 1; o < kCpRadLanes; o <<= 1) acc = acc + __shfl_xor(acc, o, 64);
  if (live && rl == 0) out[frame * (size_t)nsamp + s] = (divisor == 1.0) ? acc : acc / divisor;
  if (want_margin) {
    mrg = pl_wave_reduce(mrg, [](double a, double b) { return a < b ? a : b; });
    // non-negative doubles order like their bit patterns
    if ((threadIdx.x & 63) == 0) atomicMin(reinterpret_cast<long long*>(margin + frame), __double_as_longlong(mrg));
  }
}

// ---- the ring staged in LDS ----------------------------------------------------------------------------------------------------
// CTP528's ring is thin: 20 radii within +-4 % of 94 px at 2x sampling are 24 560 taps x 7 slices = 172 000 gathers per
// profile, but they land on ~6 000 different pixels.  The kernel above is bound by the texture-address path (one wave-wide
// gather instruction costs it ~47 cycles whatever it fetches: r05 counters).  Here a workgroup owns one profile: it first
// copies the ANNULUS of the ring's bounding box -- per box row the one or two chords between r_lo - 2.5 and r_hi + 2.5 px,
// the maximum over the 2 KPM + 1 slices formed on the way -- into LDS with row-contiguous loads, then takes every tap from
// LDS.  A quarter of the loads, none of them scattered; the arithmetic of a tap (float64, the reference's order) is
// unchanged.  The annulus is stored COMPACTLY (a row table {offset, left start, right start, lengths} + the chords' pixels
// one after the other: 13 KB + 3 KB for CTP528 where the whole box would take 82 KB and leave one workgroup per CU: r06n),
// so four workgroups share a CU and one's staging hides under the others' taps.
// r_lo / r_hi are the CALLER'S promise about |radii|.  A tap reads LDS only if its pixel lies inside a staged chord -- an
// integer test against the row table -- and is fetched from the slices themselves otherwise (a radius outside the promise,
// an annulus that does not fit the LDS handed in): a wrong promise costs time, never a sample.
#ifndef PL_RING_VARIANT
#define PL_RING_VARIANT 0   // stopwatch builds: 1 = staging only (no taps), 2 = row table + taps (no pixel staged: garbage samples)
#endif
constexpr int kRingThreads = 512, kRingRadLanes = 4, kRingSamples = kRingThreads / kRingRadLanes, kRingRadRegs = 8;

template <typename T, int KPM>
__global__ void __launch_bounds__(kRingThreads, 8)   // eight waves per SIMD = four workgroups per CU: 64 registers
circle_ring_kernel(const T* __restrict__ stack, int h, int w, const int64_t* __restrict__ slice_index, int64_t per_volume,
                   const double* __restrict__ cosv, const double* __restrict__ sinv, int nsamp,
                   const double* __restrict__ radii, int nr, const double* __restrict__ cx, const double* __restrict__ cy,
                   double divisor, double r_lo, double r_hi, int rows_cap, int pix_cap, double* __restrict__ out,
                   double* __restrict__ margin) {
  // dynamic LDS: int4 row table [rows_cap] = {offset, left start, right start, left length | right length << 16}, then the pixels
  extern __shared__ __attribute__((aligned(16))) unsigned char ring_lds[];
  int4* const rows = reinterpret_cast<int4*>(ring_lds);
  T* const pix = reinterpret_cast<T*>(ring_lds + (size_t)rows_cap * sizeof(int4));
  __shared__ int s_total, s_wsum[kRingThreads / 64];
  constexpr int NS = 2 * KPM + 1;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // Workgroups go to the eight XCDs in turn, each with an L2 of its own.  Profile z reads slices z - k .. z + k, so NEIGHBOURING
  // profiles share 2 k of their 2 k + 1 slices: XCD x takes the x-th EIGHTH of the profiles (consecutive ones at the same time),
  // not every eighth one -- the chords are 24-byte pieces of 64-byte sectors, and with the neighbours on other XCDs every
  // piece came from HBM seven times (r06o: staging 134 of the kernel's 189 us, at HBM's sector rate).
  size_t frame;
  {
    const unsigned nb = gridDim.x, per = nb / 8u, rem = nb % 8u, xcd = blockIdx.x % 8u, k = blockIdx.x / 8u;
    frame = (size_t)xcd * per + (xcd < rem ? xcd : rem) + k;
  }
  const int64_t g = slice_index ? slice_index[frame] : (int64_t)frame;
  const int64_t v0 = (g / per_volume) * per_volume, z = g - v0;
  const size_t per_frame = (size_t)h * w;
  const T* base[NS];
#pragma unroll
  for (int d = 0; d < NS; ++d) {                             // the reference's list index z + d inside the slice's own volume
    int64_t q = z + (d - KPM);
    if (q < 0) q += per_volume;
    if (q < 0) q = 0;
    if (q >= per_volume) q = per_volume - 1;
    base[d] = stack + (size_t)(v0 + q) * per_frame;
  }
  const double x0 = cx[frame], y0 = cy[frame];
  // the box: every pixel a tap of radius <= r_hi can select, clipped to the frame (a NaN centre leaves it empty)
  const double R = r_hi + 2.0;
  int bx0 = 0, bx1 = -1, by0 = 0, by1 = -1;
  if (x0 - R > -2.0e9 && x0 + R < 2.0e9 && y0 - R > -2.0e9 && y0 + R < 2.0e9) {
    bx0 = (int)floor(x0 - R); bx1 = (int)ceil(x0 + R);
    by0 = (int)floor(y0 - R); by1 = (int)ceil(y0 + R);
    bx0 = bx0 < 0 ? 0 : bx0; by0 = by0 < 0 ? 0 : by0;
    bx1 = bx1 > w - 1 ? w - 1 : bx1; by1 = by1 > h - 1 ? h - 1 : by1;
  }
  const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
  bool staged = bw > 0 && bh > 0 && bh <= rows_cap && bw < 32768;
  if (staged) {
    // ---- the row table: thread t plans rows t, t + 512, ..: the outer chord [l0, r1], cut in two by the inner disc where that
    // leaves pixels out (float32 on box coordinates: the band is 1.8 px wider on either side than any tap's pixel can lie)
    const float fx0 = (float)(x0 - (double)bx0), fy0 = (float)(y0 - (double)by0);
    const float ro = (float)r_hi + 2.5f, ri = fmaxf((float)r_lo - 2.5f, 0.0f);
    const float ro2 = ro * ro, ri2 = ri * ri;
    int carry = 0;                                           // pixels of the rows before this pass (workgroup-uniform)
    for (int py0 = 0; py0 < bh; py0 += kRingThreads) {
      const int py = py0 + (int)threadIdx.x;
      int l0 = 0, r0 = 0, nl = 0, nrt = 0;
      if (py < bh) {
        const float dy = (float)py - fy0, dy2 = dy * dy;
        if (ro2 >= dy2) {
          const float a = sqrtf(ro2 - dy2);
          l0 = (int)floorf(fx0 - a);
          int r1 = (int)ceilf(fx0 + a);
          l0 = l0 < 0 ? 0 : l0;
          r1 = r1 > bw - 1 ? bw - 1 : r1;
          int l1 = r1;
          r0 = r1 + 1;
          if (ri2 > dy2) {
            const float b = sqrtf(ri2 - dy2);
            const int e = (int)ceilf(fx0 - b), sgt = (int)floorf(fx0 + b);   // pixels strictly between them lie inside the disc
            if (sgt - e > 1) { l1 = e; r0 = sgt; }
          }
          l1 = l1 > r1 ? r1 : l1;
          r0 = r0 < l0 ? l0 : r0;
          r0 = r0 <= l1 ? l1 + 1 : r0;                       // (segments never overlap)
          nl = l1 >= l0 ? l1 - l0 + 1 : 0;
          nrt = r1 >= r0 ? r1 - r0 + 1 : 0;
        }
      }
      // exclusive prefix of nl + nrt over the 512 rows of this pass
      const int cnt = nl + nrt;
      int inc = cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
      if (lane == 63) s_wsum[wv] = inc;
      __syncthreads();
      int before = carry, all = 0;
      for (int k = 0; k < kRingThreads / 64; ++k) { const int v = s_wsum[k]; before += k < wv ? v : 0; all += v; }
      if (py < bh) rows[py] = make_int4(before + inc - cnt, l0, r0, nl | (nrt << 16));
      carry += all;
      __syncthreads();
    }
    if (threadIdx.x == 0) s_total = carry;
    __syncthreads();
    staged = s_total <= pix_cap;
    if (staged && PL_RING_VARIANT != 2) {
      // ---- the pixels: a wave takes FOUR rows per trip and issues their 4 x NS loads before the first is used (one row per trip
      // made the wave pay a memory round trip per row: r06m).  Lanes 0-31 walk the left chord, lanes 32-63 the right one (a
      // row with one chord: its second 32 pixels).  Loads are unconditional on clamped addresses; only the LDS store is guarded.
      const int half = lane >> 5, hl = lane & 31;
      constexpr int RU = 4;
      for (int py0 = wv * RU; py0 < bh; py0 += (kRingThreads / 64) * RU) {    // wave-uniform
        int st[RU], n_mine[RU], dst[RU], step[RU], longest[RU];
#pragma unroll
        for (int q = 0; q < RU; ++q) {
          const int4 t = rows[py0 + q < bh ? py0 + q : bh - 1];
          const int nl = py0 + q < bh ? (t.w & 0xffff) : 0, nrt = py0 + q < bh ? (t.w >> 16) : 0;
          const bool two = nrt > 0;
          // this lane's run: `n_mine` pixels from box column `st`, stored from LDS element `dst`
          st[q] = half == 0 ? t.y : (two ? t.z : t.y + 32);
          n_mine[q] = half == 0 ? nl : (two ? nrt : nl - 32);
          dst[q] = half == 0 ? t.x : (two ? t.x + nl : t.x + 32);
          step[q] = two ? 32 : 64;
          longest[q] = two ? (nl > nrt ? nl : nrt) : nl;
        }
        T e0[RU][NS];
#pragma unroll
        for (int q = 0; q < RU; ++q) {
          const int pyc = py0 + q < bh ? py0 + q : bh - 1;
          int pxc = st[q] + hl;
          pxc = pxc < 0 ? 0 : (pxc > bw - 1 ? bw - 1 : pxc);
          const unsigned off = (unsigned)(by0 + pyc) * (unsigned)w + (unsigned)(bx0 + pxc);
#pragma unroll
          for (int d = 0; d < NS; ++d) e0[q][d] = base[d][off];
        }
#pragma unroll
        for (int q = 0; q < RU; ++q) {
          T m0 = e0[q][0];
#pragma unroll
          for (int d = 1; d < NS; ++d) m0 = e0[q][d] > m0 ? e0[q][d] : m0;
          if (hl < n_mine[q]) pix[dst[q] + hl] = m0;
        }
#pragma unroll
        for (int q = 0; q < RU; ++q) {                                         // chords longer than one trip (few rows)
          for (int j = step[q]; j < longest[q]; j += step[q]) {
            if (hl + j < n_mine[q]) {
              const unsigned off = (unsigned)(by0 + py0 + q) * (unsigned)w + (unsigned)(bx0 + st[q] + hl + j);
              T m0 = base[0][off];
#pragma unroll
              for (int d = 1; d < NS; ++d) { const T e = base[d][off]; m0 = e > m0 ? e : m0; }
              pix[dst[q] + hl + j] = m0;
            }
          }
        }
      }
    }
  }
  __syncthreads();

  const double* rad = radii + frame * nr;
  double mrg = __longlong_as_double(0x7ff0000000000000LL);
  const bool want_margin = margin != nullptr;
  // a ring that stays a pixel clear of the frame on every side cannot meet a bounds decision (|cos|, |sin| <= 1 and rounding
  // moves a coordinate by 1e-13): the inside test and its margin are dropped from every tap (workgroup-uniform)
  bool near_border = true;
  {
    double rmax = 0.0;
    bool finite = true;                                      // (fmax drops a NaN radius; its taps must stay "outside")
    for (int k = 0; k < nr; ++k) { rmax = fmax(rmax, fabs(rad[k])); finite = finite && rad[k] == rad[k]; }
    near_border = !(finite && x0 - rmax > 1.0 && x0 + rmax < (double)(w - 2) && y0 - rmax > 1.0 && y0 + rmax < (double)(h - 2));
  }
  // a lane's radii (rl, rl + 4, ..) do not change from sample to sample: the first 32 of a profile live in registers
  const int rl = threadIdx.x & (kRingRadLanes - 1);
  double myr[kRingRadRegs];
#pragma unroll
  for (int m = 0; m < kRingRadRegs; ++m) myr[m] = rl + kRingRadLanes * m < nr ? rad[rl + kRingRadLanes * m] : 0.0;
  const int n_passes = PL_RING_VARIANT == 1 ? 0 : (nsamp + kRingSamples - 1) / kRingSamples;
  int s = (int)(threadIdx.x / kRingRadLanes);
  s = s < nsamp ? s : nsamp - 1;
  double c = cosv[s], sn = sinv[s];
  for (int pass = 0; pass < n_passes; ++pass) {              // workgroup-uniform
    const int s_raw = pass * kRingSamples + (int)(threadIdx.x / kRingRadLanes);
    const bool live = s_raw < nsamp;
    s = live ? s_raw : nsamp - 1;                            // every lane stays for the butterfly
    // the next pass's angle is asked for before this pass's taps
    int s_next = s_raw + kRingSamples;
    s_next = s_next < nsamp ? s_next : nsamp - 1;
    const double c_next = cosv[s_next], sn_next = sinv[s_next];
    double acc = 0.0;                                        // integers: exact whatever the order
    auto tap = [&](double r) {
      const double x = c * r + x0;
      const double y = sn * r + y0;
      const bool in = !near_border || (x >= 0.0 && x <= (double)(w - 1) && y >= 0.0 && y <= (double)(h - 1));
      const double fx = floor(x + 0.5), fy = floor(y + 0.5);
      const int xi = in ? (int)fx : 0, yi = in ? (int)fy : 0;
      if (want_margin && live) {
        const double dx = ((x + 0.5) - fx) - 0.5, dy = ((y + 0.5) - fy) - 0.5;
        double m = 0.5 - fmax(fabs(dx), fabs(dy));
        if (near_border)
          m = fmin(m, fmin(fabs(fmin(x, (double)(w - 1) - x)), fabs(fmin(y, (double)(h - 1) - y))));
        mrg = fmin(mrg, m);
      }
      T m0 = 0;
      if (in) {
        const int qx = xi - bx0;
        const unsigned qy = (unsigned)(yi - by0);
        int idx = -1;
        if (staged && qy < (unsigned)bh) {
          const int4 t = rows[qy];
          const unsigned dl = (unsigned)(qx - t.y), dr = (unsigned)(qx - t.z);
          const unsigned nl = (unsigned)(t.w & 0xffff), nrt = (unsigned)(t.w >> 16);
          idx = dl < nl ? t.x + (int)dl : (dr < nrt ? t.x + (int)nl + (int)dr : -1);
        }
        if (idx >= 0) {
          m0 = pix[idx];
        } else {                                             // outside the promise: the slices themselves
          const unsigned off = (unsigned)yi * (unsigned)w + (unsigned)xi;
          m0 = base[0][off];
#pragma unroll
          for (int d = 1; d < NS; ++d) { const T e = base[d][off]; m0 = e > m0 ? e : m0; }
        }
      }
      acc = acc + (in ? (double)m0 : 0.0);
    };
#pragma unroll
    for (int m = 0; m < kRingRadRegs; ++m)
      if (rl + kRingRadLanes * m < nr) tap(myr[m]);
    for (int k = rl + kRingRadLanes * kRingRadRegs; k < nr; k += kRingRadLanes) tap(rad[k]);
#pragma unroll
    for (int o = 1; o < kRingRadLanes; o <<= 1) acc = acc + __shfl_xor(acc, o, 64);
    if (live && rl == 0) out[frame * (size_t)nsamp + s] = (divisor == 1.0) ? acc : acc / divisor;
    c = c_next;
    sn = sn_next;
  }
  if (want_margin) {
    mrg = pl_wave_reduce(mrg, [](double a, double b) { return a < b ? a : b; });
    if (lane == 0) atomicMin(reinterpret_cast<long long*>(margin + frame), __double_as_longlong(mrg));
  }
}

}  // namespace

extern "C" int pl_circle_profile_combined_ex(const void* stack, int dtype, int64_t n_stack, int h, int w,
                                             const int64_t* d_slice_index, int64_t m, int64_t slices_per_volume, int plusminus,
                                             const double* d_cos, const double* d_sin, int nsamp, const double* d_radii, int nr,
                                             const double* d_cx, const double* d_cy, double divisor, double* d_out,
                                             double* d_margin, void* stream) {
  PL_REQUIRE(stack && d_cos && d_sin && d_radii && d_cx && d_cy && d_out, "null pointer");
  PL_REQUIRE(m >= 0 && m <= 65535 && h > 0 && w > 0 && nsamp > 0 && nr > 0 && plusminus >= 0, "bad shape");
  PL_REQUIRE(slices_per_volume > 0 && n_stack > 0 && n_stack % slices_per_volume == 0, "the stack must hold whole volumes");
  PL_REQUIRE(d_slice_index || m == n_stack, "without a slice index every slice of the stack is sampled");
  PL_REQUIRE(divisor != 0.0, "zero divisor");
  PL_REQUIRE(dtype == PL_I16 || dtype == PL_U16 || dtype == PL_I32 || dtype == PL_U8, "integer slices");
  if (m == 0) return PL_OK;
  dim3 grid((unsigned)pl_cdiv(nsamp, kCpSamples), (unsigned)m);
  PL_REQUIRE((int64_t)h * w <= 0xffffffffLL, "frame too large");
#define CPC_LAUNCH(K)                                                                                                       \
  hipLaunchKernelGGL((circle_profile_combined_kernel<T, K>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)stack, h, w, \
                     d_slice_index, slices_per_volume, plusminus, d_cos, d_sin, nsamp, d_radii, nr, d_cx, d_cy, divisor, d_out, \
                     d_margin)
  PL_DISPATCH_DTYPE(dtype, T, {
    switch (plusminus) {
      case 0: CPC_LAUNCH(0); break;
      case 1: CPC_LAUNCH(1); break;
      case 2: CPC_LAUNCH(2); break;
      case 3: CPC_LAUNCH(3); break;
      default: CPC_LAUNCH(-1); break;
    }
  });
#undef CPC_LAUNCH
  return pl_check_launch("pl_circle_profile_combined");
}

// pl_circle_profile_combined_ex for a THIN ring: the caller promises r_lo <= |radius| <= r_hi for every radius of every
// profile (a promise that turns out wrong costs time, not correctness: see circle_ring_kernel).  Rings whose annulus (row
// table + pixels) does not fit 64 KB of LDS, and 2 k + 1 > 7 slices, go to pl_circle_profile_combined_ex.
extern "C" int pl_circle_profile_ring(const void* stack, int dtype, int64_t n_stack, int h, int w,
                                      const int64_t* d_slice_index, int64_t m, int64_t slices_per_volume, int plusminus,
                                      const double* d_cos, const double* d_sin, int nsamp, const double* d_radii, int nr,
                                      const double* d_cx, const double* d_cy, double divisor, double r_lo, double r_hi,
                                      double* d_out, double* d_margin, void* stream) {
  PL_REQUIRE(r_lo >= 0.0 && r_hi >= r_lo, "0 <= r_lo <= r_hi");
  PL_REQUIRE(dtype == PL_I16 || dtype == PL_U16 || dtype == PL_I32 || dtype == PL_U8, "integer slices");
  const size_t esz = dtype == PL_I32 ? 4 : (dtype == PL_U8 ? 1 : 2);
  // rows of the bounding box, and the staged band's area + two pixels of rounding per chord end
  const double rows_d = 2.0 * ceil(r_hi + 2.0) + 3.0;
  const double ro = r_hi + 4.0, ri = r_lo > 4.0 ? r_lo - 4.0 : 0.0;
  const double pix_d = 3.14159265358979 * (ro * ro - ri * ri) + 8.0 * rows_d + 64.0;
  const double bytes = rows_d * 16.0 + pix_d * (double)esz;
  if (!(bytes <= 64.0 * 1024.0) || plusminus > 3)
    return pl_circle_profile_combined_ex(stack, dtype, n_stack, h, w, d_slice_index, m, slices_per_volume, plusminus, d_cos,
                                         d_sin, nsamp, d_radii, nr, d_cx, d_cy, divisor, d_out, d_margin, stream);
  PL_REQUIRE(stack && d_cos && d_sin && d_radii && d_cx && d_cy && d_out, "null pointer");
  PL_REQUIRE(m >= 0 && m <= 0x7fffffffLL && h > 0 && w > 0 && nsamp > 0 && nr > 0 && plusminus >= 0, "bad shape");
  PL_REQUIRE(slices_per_volume > 0 && n_stack > 0 && n_stack % slices_per_volume == 0, "the stack must hold whole volumes");
  PL_REQUIRE(d_slice_index || m == n_stack, "without a slice index every slice of the stack is sampled");
  PL_REQUIRE(divisor != 0.0, "zero divisor");
  PL_REQUIRE((int64_t)h * w <= 0xffffffffLL, "frame too large");
  if (m == 0) return PL_OK;
  const int rows_cap = (int)rows_d, pix_cap = (int)pix_d;
  const size_t lds = ((size_t)rows_cap * 16 + (size_t)pix_cap * esz + 15) & ~(size_t)15;
#define RING_LAUNCH(K)                                                                                                        \
  hipLaunchKernelGGL((circle_ring_kernel<T, K>), dim3((unsigned)m), dim3(kRingThreads), lds, (hipStream_t)stream,             \
                     (const T*)stack, h, w, d_slice_index, slices_per_volume, d_cos, d_sin, nsamp, d_radii, nr, d_cx, d_cy,   \
                     divisor, r_lo, r_hi, rows_cap, pix_cap, d_out, d_margin)
  PL_DISPATCH_DTYPE(dtype, T, {
    switch (plusminus) {
      case 0: RING_LAUNCH(0); break;
      case 1: RING_LAUNCH(1); break;
      case 2: RING_LAUNCH(2); break;
      default: RING_LAUNCH(3); break;
    }
  });
#undef RING_LAUNCH
  return pl_check_launch("pl_circle_profile_ring");
}

extern "C" int pl_circle_profile_combined(const void* stack, int dtype, int64_t n_stack, int h, int w,
                                          const int64_t* d_slice_index, int64_t m, int64_t slices_per_volume, int plusminus,
                                          const double* d_cos, const double* d_sin, int nsamp, const double* d_radii, int nr,
                                          const double* d_cx, const double* d_cy, double divisor, double* d_out,
                                          void* stream) {
  return pl_circle_profile_combined_ex(stack, dtype, n_stack, h, w, d_slice_index, m, slices_per_volume, plusminus, d_cos, d_sin,
                                       nsamp, d_radii, nr, d_cx, d_cy, divisor, d_out, nullptr, stream);
}

extern "C" int pl_circle_profile(const void* img, int dtype, int64_t n, int h, int w, const double* d_cos,
                                 const double* d_sin, int nsamp, const double* d_radii, int nr,
                                 const double* d_cx, const double* d_cy, double divisor, double* d_out,
                                 void* stream) {
  PL_REQUIRE(img && d_cos && d_sin && d_radii && d_cx && d_cy && d_out, "null pointer");
  PL_REQUIRE(n >= 0 && n <= 65535 && h > 0 && w > 0 && nsamp > 0 && nr > 0, "bad shape");
  PL_REQUIRE(divisor != 0.0, "zero divisor");
  if (n == 0) return PL_OK;
  dim3 grid((unsigned)pl_cdiv(nsamp, 256), (unsigned)n);
  PL_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL(circle_profile_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream,
                                       (const T*)img, h, w, d_cos, d_sin, nsamp, d_radii, nr, d_cx, d_cy,
                                       divisor, d_out));
  return pl_check_launch("pl_circle_profile");
}
