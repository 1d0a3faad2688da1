// Batched 1-D peak finding with scipy.signal.find_peaks semantics, plus pylinac's own pre/post
// steps.
//
// Replaces: pylinac.core.profile.find_peaks (pylinac/core/profile.py:2545-2623) and
// _parse_peak_args (:2626-2649), i.e. the call
//     scipy.signal.find_peaks(trimmed, rel_height=1-fwxm_height, width=min_width,
//                             height=threshold, distance=peak_separation, prominence=required)
// followed by "keep the max_number largest by peak_props[peak_sort], re-sorted left to right".
//
// One 256-lane workgroup per profile.  Stages (the order of scipy's filters is preserved):
//   A  min/max of the FULL profile (ratio threshold: height = min + thr*(max-min))
//   B  local maxima (strict rise, plateau -> midpoint (l+r)/2, strict fall; ends never peaks)
//      + height filter, compacted IN ORDER with ballot/popcount scans
//   C  distance filter: priority = height, highest first, processed sequentially by one lane
//      exactly like _select_by_peak_distance (ties: stable order -- scipy's own tie order comes
//      from np.argsort's default introsort and is implementation-defined)
//   D  prominences + bases: one lane per peak walks outwards in LDS
//   E  prominence filter   F  widths at rel_height (+ width filter)
//   G  top-max_number by key (stable, reversed)   H  ordered output compaction
// Float64 throughout, operations in scipy's order, so the float results are bit-identical.
#include <math.h>

#include "pl_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kStageMax = 4096;   // profiles up to this length are staged in LDS
constexpr int kMaxCand = 4096;    // candidate peaks kept per profile
constexpr int kShortMax = 128;    // search regions up to this length: one wave per profile

struct Scan { int wave[kThreads / PL_WAVE]; };

// NT lanes work on one profile: 256 (a whole workgroup; the barrier is the workgroup's) or 64 (one wave of a
// four-profile workgroup: the wave is in lock step, the "barrier" only orders its LDS traffic for the compiler)
template <int NT>
__device__ __forceinline__ void group_sync() {
  if constexpr (NT == PL_WAVE) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  } else {
    __syncthreads();
  }
}

template <int NT>
__device__ __forceinline__ int block_flag_scan(int flag, int* total, Scan* s, int tid) {
  const unsigned long long b = __ballot(flag);
  const int lane = tid & 63, wv = tid >> 6;
  const int pre = __popcll(b & ((1ull << lane) - 1ull));
  if constexpr (NT == PL_WAVE) {
    *total = __popcll(b);
    return pre;
  } else {
    if (lane == 0) s->wave[wv] = __popcll(b);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NT / PL_WAVE; ++k) {
      if (k < wv) base += s->wave[k];
      tot += s->wave[k];
    }
    __syncthreads();
    *total = tot;
    return base + pre;
  }
}

struct Widths { double width, height, lip, rip; };

// scipy's walks are sequential by definition (while loops over neighbouring samples).  A lane-private walk pays
// one LDS round trip per step, and a beam profile has a handful of peaks whose walks cross the whole profile
// (measured: 190 us per launch, one lane busy).  So every walk is done by a whole WAVE for one peak at a time:
// the 64 lanes test 64 consecutive steps at once, a ballot finds the first step at which scipy's loop
// condition fails, and only the steps before it take part in the result.  Same result, 1/64 of the trips.
// All arguments are wave-uniform; every lane returns the same value.

// scipy:  i = peak; while (i_min < i && height < x[i]) --i;
__device__ __forceinline__ int walk_left_while_above(const double* xs, int start, int stop, double h) {
  const int lane = threadIdx.x & (PL_WAVE - 1);
  for (int t0 = 0;; t0 += PL_WAVE) {
    const int j = start - (t0 + lane);
    const bool fail = !(stop < j) || !(h < xs[j]);   // j > stop >= 0 whenever xs[j] is read
    const unsigned long long b = __ballot(fail);
    if (b) return start - (t0 + __builtin_ctzll(b));
  }
}
// scipy:  i = peak; while (i < i_max && height < x[i]) ++i;
__device__ __forceinline__ int walk_right_while_above(const double* xs, int start, int stop, double h) {
  const int lane = threadIdx.x & (PL_WAVE - 1);
  for (int t0 = 0;; t0 += PL_WAVE) {
    const int j = start + (t0 + lane);
    const bool fail = !(j < stop) || !(h < xs[j]);   // j < stop <= m - 1 whenever xs[j] is read
    const unsigned long long b = __ballot(fail);
    if (b) return start + (t0 + __builtin_ctzll(b));
  }
}

__device__ __forceinline__ Widths peak_width(const double* xs, int pk, int lb, int rb, double prom,
                                             double rel_height) {
  Widths r;
  const double h = xs[pk] - prom * rel_height;
  r.height = h;
  int i = walk_left_while_above(xs, pk, lb, h);
  double lip = (double)i;
  if (xs[i] < h) lip += (h - xs[i]) / (xs[i + 1] - xs[i]);
  i = walk_right_while_above(xs, pk, rb, h);
  double rip = (double)i;
  if (xs[i] < h) rip -= (h - xs[i]) / (xs[i - 1] - xs[i]);
  r.lip = lip;
  r.rip = rip;
  r.width = rip - lip;
  return r;
}

// scipy _peak_prominences, one side:  i = base = peak; min = x[peak];
//   while (in range && x[i] <= x[peak]) { if (x[i] < min) { min = x[i]; base = i; } i += dir; }
// Step t visits peak + DIR*t.  Each lane keeps the minimum over its own steps (strict <, so its earliest step
// wins a tie); the final reduction takes the smallest value and, among equal values, the earliest step --
// the sample the sequential loop would have kept.
template <int DIR>
__device__ __forceinline__ void prominence_side(const double* xs, int pk, int m, double& out_min, int& out_base) {
  const int lane = threadIdx.x & (PL_WAVE - 1);
  const double xp = xs[pk];
  double mn = xp;
  int step = 0;
  for (int t0 = 0;; t0 += PL_WAVE) {
    const int t = t0 + lane;
    const int j = pk + DIR * t;
    const bool inside = DIR < 0 ? (j >= 0) : (j <= m - 1);
    const double v = inside ? xs[j] : xp;
    const bool fail = !inside || !(v <= xp);
    const unsigned long long b = __ballot(fail);
    const int first = b ? __builtin_ctzll(b) : PL_WAVE;
    if (lane < first && v < mn) { mn = v; step = t; }
    if (b) break;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(mn, o, 64);
    const int os = __shfl_xor(step, o, 64);
    if (ov < mn || (ov == mn && os < step)) { mn = ov; step = os; }
  }
  out_min = mn;
  out_base = pk + DIR * step;
}

struct PeakLds {                                   // the LDS of one profile's search
  unsigned char* smem;                             // dynamic block: candidate tables (+ the staged profile)
  Scan* scan;
  double* s_red;                                   // 2 x (kThreads / PL_WAVE)
  int* s_cnt;
};

// One profile, NT lanes (tid = 0 .. NT-1): `xfull` has `len` samples, the search region is [rlo, rhi) clipped to it; results go
// to the profile's own output rows (o_count / o_status one element, o_idx / o_lb / o_rb `cap`, o_p 6 x cap).  Every lane of the
// group calls it; all of them return together.
// (`prm` BY VALUE: with a reference to the kernel's by-value parameter struct the gfx950 build ranked peak_sort="widths"
// wrongly -- tests/test_gpu_parity.py::test_find_peaks_vs_oracle_random caught it on the device, the CPU emulator did not.)
template <bool STAGE, int NT>
__device__ __forceinline__ void find_peaks_profile(const double* __restrict__ xfull, int len, int rlo, int rhi, const pl_peak_params prm,
                                                   int cap, int maxc, const PeakLds L, int tid, int32_t* __restrict__ o_count,
                                                   int32_t* __restrict__ o_idx, int32_t* __restrict__ o_lb, int32_t* __restrict__ o_rb,
                                                   double* __restrict__ o_p, int32_t* __restrict__ o_status, const double sign = 1.0) {
  // `sign` = -1: the search runs on the NEGATED profile (find_valleys, pylinac/core/profile.py: peaks of -values); staged
  // profiles only (the negation happens while the region is copied to LDS; x * 1.0 and x * -1.0 are exact)
  unsigned char* smem = L.smem;
  Scan& scan = *L.scan;
  double* s_red = L.s_red;
  int& s_cnt = *L.s_cnt;
  int lo = rlo < 0 ? 0 : rlo;
  int hi = rhi > len ? len : rhi;
  if (hi < lo) hi = lo;
  const int m = hi - lo;

  // LDS carve-up: [prom f64][width f64][idx][lb][rb][keep] x maxc, then optional staged profile
  double* s_prom = reinterpret_cast<double*>(smem);
  double* s_width = s_prom + maxc;
  int* s_idx = reinterpret_cast<int*>(s_width + maxc);
  int* s_lb = s_idx + maxc;
  int* s_rb = s_lb + maxc;
  int* s_keep = s_rb + maxc;
  double* s_x = reinterpret_cast<double*>(s_keep + maxc + (maxc & 1));

  if (len <= 0) {   // empty profile (e.g. a window that was rejected upstream)
    if (tid == 0) { *o_count = 0; *o_status = 0; }
    return;
  }
  // ---- A: height threshold -------------------------------------------------------------------
  double height = prm.threshold;
  if (prm.threshold_is_ratio) {
    double mn = sign * xfull[0], mx = mn;
    for (int i = tid; i < len; i += NT) {
      double v = sign * xfull[i];
      mn = v < mn ? v : mn;
      mx = v > mx ? v : mx;
    }
    mn = pl_wave_reduce(mn, [](double a, double b) { return a < b ? a : b; });
    mx = pl_wave_reduce(mx, [](double a, double b) { return a > b ? a : b; });
    if ((tid & 63) == 0) { s_red[tid >> 6] = mn; s_red[4 + (tid >> 6)] = mx; }
    group_sync<NT>();
    for (int k = 0; k < NT / PL_WAVE; ++k) {
      mn = s_red[k] < mn ? s_red[k] : mn;
      mx = s_red[4 + k] > mx ? s_red[4 + k] : mx;
    }
    height = mn + prm.threshold * (mx - mn);  // pylinac/core/profile.py:2633-2635
  }

  const double* xs;
  if constexpr (STAGE) {
    for (int i = tid; i < m; i += NT) s_x[i] = sign * xfull[lo + i];
    xs = s_x;
  } else {
    xs = xfull + lo;
  }
  if (tid == 0) s_cnt = 0;
  group_sync<NT>();

  // ---- B: local maxima + height filter, ordered compaction -----------------------------------
  int overflow = 0;
  for (int base = 0; base < m; base += NT) {
    const int i = base + tid;
    int flag = 0, mid = 0;
    if (i >= 1 && i < m - 1 && xs[i - 1] < xs[i]) {
      int a = i + 1;
      while (a < m - 1 && xs[a] == xs[i]) ++a;
      if (xs[a] < xs[i]) {
        mid = (i + a - 1) / 2;
        flag = (xs[mid] >= height) ? 1 : 0;
      }
    }
    int tot;
    const int off = block_flag_scan<NT>(flag, &tot, &scan, tid);
    const int cur = s_cnt;
    if (flag) {
      if (cur + off < maxc) s_idx[cur + off] = mid; else overflow = 1;
    }
    group_sync<NT>();
    if (tid == 0) s_cnt = cur + tot;
    group_sync<NT>();
  }
  int P = s_cnt;
  if (P > maxc) { P = maxc; overflow = 1; }
  overflow = (NT == PL_WAVE ? (__ballot(overflow) != 0ull ? 1 : 0) : __syncthreads_or(overflow));

  // ---- C: distance filter --------------------------------------------------------------------
  if (prm.distance > 1 && P > 1) {
    int* s_order = s_lb;  // scratch: bases are not computed yet
    for (int j = tid; j < P; j += NT) {
      const double hj = xs[s_idx[j]];
      int r = 0;
      for (int k = 0; k < P; ++k) {
        const double hk = xs[s_idx[k]];
        r += (hk < hj || (hk == hj && k < j)) ? 1 : 0;
      }
      s_order[r] = j;
      s_keep[j] = 1;
    }
    group_sync<NT>();
    if (tid == 0) {
      const int d = prm.distance;
      for (int i = P - 1; i >= 0; --i) {
        const int j = s_order[i];
        if (!s_keep[j]) continue;
        int k = j - 1;
        while (k >= 0 && s_idx[j] - s_idx[k] < d) { s_keep[k] = 0; --k; }
        k = j + 1;
        while (k < P && s_idx[k] - s_idx[j] < d) { s_keep[k] = 0; ++k; }
      }
    }
    group_sync<NT>();
    int* s_tmp = s_rb;
    if (tid == 0) s_cnt = 0;
    group_sync<NT>();
    for (int base = 0; base < P; base += NT) {
      const int j = base + tid;
      const int flag = (j < P) ? s_keep[j] : 0;
      int tot;
      const int off = block_flag_scan<NT>(flag, &tot, &scan, tid);
      const int cur = s_cnt;
      if (flag) s_tmp[cur + off] = s_idx[j];
      group_sync<NT>();
      if (tid == 0) s_cnt = cur + tot;
      group_sync<NT>();
    }
    P = s_cnt;
    for (int j = tid; j < P; j += NT) s_idx[j] = s_tmp[j];
    group_sync<NT>();
  }

  // ---- D/E/F: prominences, bases, widths, filters (one wave per peak, see the walk helpers) -------
  for (int p = tid / PL_WAVE; p < P; p += NT / PL_WAVE) {
    const int pk = s_idx[p];
    const double xp = xs[pk];
    double left_min, right_min;
    int lb, rb;
    prominence_side<-1>(xs, pk, m, left_min, lb);
    prominence_side<+1>(xs, pk, m, right_min, rb);
    const double prom = xp - (left_min > right_min ? left_min : right_min);
    int keep = (!prm.has_prominence || prom >= prm.prominence_min) ? 1 : 0;
    const Widths wd = peak_width(xs, pk, lb, rb, prom, prm.rel_height);
    keep = keep && (wd.width >= prm.width_min);
    if ((tid & (PL_WAVE - 1)) == 0) {
      s_prom[p] = prom;
      s_width[p] = wd.width;
      s_lb[p] = lb;
      s_rb[p] = rb;
      s_keep[p] = keep;
    }
  }
  group_sync<NT>();

  // ---- G: keep the max_number largest by key (np.argsort(kind=stable)[::-1][:max_number]) ------
  if (prm.max_number > 0) {
    // s_keep is read-only during the ranking; a peak to drop is tagged by complementing its
    // (non-negative) right base, then untagged after the barrier.
    for (int p = tid; p < P; p += NT) {
      if (!s_keep[p]) continue;
      const double kp = prm.sort_key == PL_SORT_PROMINENCES ? s_prom[p]
                        : prm.sort_key == PL_SORT_PEAK_HEIGHTS ? xs[s_idx[p]] : s_width[p];
      int ahead = 0;
      for (int k = 0; k < P; ++k) {
        if (!s_keep[k]) continue;
        const double kk = prm.sort_key == PL_SORT_PROMINENCES ? s_prom[k]
                          : prm.sort_key == PL_SORT_PEAK_HEIGHTS ? xs[s_idx[k]] : s_width[k];
        ahead += (kk > kp || (kk == kp && k > p)) ? 1 : 0;
      }
      if (ahead >= prm.max_number) s_rb[p] = ~s_rb[p];
    }
    group_sync<NT>();
    for (int p = tid; p < P; p += NT)
      if (s_rb[p] < 0) { s_rb[p] = ~s_rb[p]; s_keep[p] = 0; }
    group_sync<NT>();
  }

  // ---- H: ordered output: destinations by an ordered scan, then one wave per kept peak ---------------
  if (tid == 0) s_cnt = 0;
  group_sync<NT>();
  for (int base = 0; base < P; base += NT) {
    const int p = base + tid;
    const int flag = (p < P) ? s_keep[p] : 0;
    int tot;
    const int off = block_flag_scan<NT>(flag, &tot, &scan, tid);
    const int cur = s_cnt;
    if (p < P) s_keep[p] = flag ? (cur + off + 1) : 0;   // destination + 1
    group_sync<NT>();
    if (tid == 0) s_cnt = cur + tot;
    group_sync<NT>();
  }
  for (int p = tid / PL_WAVE; p < P; p += NT / PL_WAVE) {
    const int dst = s_keep[p] - 1;
    if (dst < 0 || dst >= cap) continue;                 // wave-uniform
    const int pk = s_idx[p];
    const Widths wd = peak_width(xs, pk, s_lb[p], s_rb[p], s_prom[p], prm.rel_height);
    if ((tid & (PL_WAVE - 1)) == 0) {
      o_idx[dst] = pk + lo;  // only the indices are shifted (pylinac/core/profile.py:2613)
      o_lb[dst] = s_lb[p];
      o_rb[dst] = s_rb[p];
      o_p[0 * cap + dst] = xs[pk];
      o_p[1 * cap + dst] = s_prom[p];
      o_p[2 * cap + dst] = wd.width;
      o_p[3 * cap + dst] = wd.height;
      o_p[4 * cap + dst] = wd.lip;
      o_p[5 * cap + dst] = wd.rip;
    }
  }
  if (tid == 0) {
    const int total = s_cnt;
    *o_count = total < cap ? total : cap;
    *o_status = overflow ? 2 : (total > cap ? 1 : 0);
  }
}

// STAGE = true: the (trimmed) profile lives in LDS and every walk is a ds_read; keeping the two
// cases in separate instantiations lets the compiler know the address space (a runtime select
// between an LDS and a global pointer degrades every access to a slow FLAT load).
template <bool STAGE, int NT>
__global__ void __launch_bounds__(kThreads)
find_peaks_kernel(const double* __restrict__ x, int64_t nprof, int slot_bytes, int len_all, const int32_t* __restrict__ lens,
                  const int32_t* __restrict__ regions, int64_t stride,
                  pl_peak_params prm, int cap, int maxc, int32_t* __restrict__ d_count, int32_t* __restrict__ d_idx,
                  int32_t* __restrict__ d_lb, int32_t* __restrict__ d_rb, double* __restrict__ d_props,
                  int32_t* __restrict__ d_status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  __shared__ Scan scan_a[kThreads / NT];
  __shared__ double s_red_a[kThreads / NT][2 * (kThreads / PL_WAVE)];
  __shared__ int s_cnt_a[kThreads / NT];
  const int slot = threadIdx.x / NT, tid = threadIdx.x % NT;
  const PeakLds L{smem_all + (size_t)slot * slot_bytes, &scan_a[slot], s_red_a[slot], &s_cnt_a[slot]};
  const int64_t prof = (int64_t)blockIdx.x * (kThreads / NT) + slot;
  if (prof >= nprof) return;                     // NT == 64: a whole wave leaves; NT == 256: never taken
  const int len = lens ? lens[prof] : len_all;   // ragged batches: per-profile length
  // search region: the batch's, or this profile's own [lo, hi) (python slice semantics resolved by the caller)
  const int rlo = regions ? regions[2 * prof] : prm.region_lo, rhi = regions ? regions[2 * prof + 1] : prm.region_hi;
  find_peaks_profile<STAGE, NT>(x + prof * stride, len, rlo, rhi, prm, cap, maxc, L, tid, d_count + prof, d_idx + prof * cap,
                                d_lb + prof * cap, d_rb + prof * cap, d_props + prof * 6 * (int64_t)cap, d_status + prof);
}

// FWXMProfile.field_edge_idx / center_idx / field_width_px from the single most prominent peak
// (pylinac/core/profile.py:602-611, 322-327, 339-344): record = {n_peaks, peak_idx, height,
// prominence, left, right, |r-l|/2+l, max(r,l)-min(r,l)}; NaN when the profile has no peak.
__device__ __forceinline__ void fwxm_record_one(int c, const int32_t* __restrict__ idx, const double* __restrict__ p, int cap,
                                                double* __restrict__ o) {
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  o[0] = (double)c;
  if (c <= 0) {
    for (int k = 1; k < 8; ++k) o[k] = nan;
    return;
  }
  const double l = p[4 * cap], r = p[5 * cap];
  o[1] = (double)idx[0];
  o[2] = p[0];
  o[3] = p[1 * cap];
  o[4] = l;
  o[5] = r;
  o[6] = fabs(r - l) / 2 + l;
  o[7] = (r > l ? r : l) - (r < l ? r : l);
}

__global__ void fwxm_record_kernel(const int32_t* __restrict__ count, const int32_t* __restrict__ idx,
                                   const double* __restrict__ props, int cap, int64_t n,
                                   double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fwxm_record_one(count[i], idx + i * cap, props + i * 6 * (int64_t)cap, cap, out + i * 8);
}

// The tail of the EPID pipeline in ONE launch, one workgroup per frame: per-band column sums of the thresholded frame ->
// np.mean(frame, 0) (pylinac/picketfence.py:747-750: float64 sum of integers / count) -> find_peaks on that profile ->
// the FWXM record.  Three launches (pl_colsum_to_mean, pl_find_peaks, pl_fwxm_record: 4 + 25 + 4 us of kernels and the gaps
// between them) and the column-sum memset + 64-bit atomics in front of them become one.
template <bool STAGE>
__global__ void __launch_bounds__(kThreads)
colparts_profile_fwxm_kernel(const uint32_t* __restrict__ parts, int bands, int w, int h, pl_peak_params prm, int cap, int maxc,
                             double* __restrict__ profile, int32_t* __restrict__ d_count, int32_t* __restrict__ d_idx,
                             int32_t* __restrict__ d_lb, int32_t* __restrict__ d_rb, double* __restrict__ d_props,
                             int32_t* __restrict__ d_status, double* __restrict__ fwxm) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  __shared__ Scan scan;
  __shared__ double s_red[2 * (kThreads / PL_WAVE)];
  __shared__ int s_cnt;
  const int64_t frame = blockIdx.x;
  const int tid = threadIdx.x;
  const uint32_t* pf = parts + frame * (int64_t)bands * w;
  double* prof = profile + frame * (int64_t)w;
  for (int i = tid; i < w; i += kThreads) {
    unsigned long long cs = 0;
    for (int b = 0; b < bands; ++b) cs += pf[(size_t)b * w + i];
    prof[i] = (double)cs / (double)h;               // np.mean of integers: float64 sum / count
  }
  __syncthreads();                                  // the profile row (global memory) is this workgroup's own
  const PeakLds L{smem_all, &scan, s_red, &s_cnt};
  find_peaks_profile<STAGE, kThreads>(prof, w, prm.region_lo, prm.region_hi, prm, cap, maxc, L, tid, d_count + frame,
                                      d_idx + frame * cap, d_lb + frame * cap, d_rb + frame * cap,
                                      d_props + frame * 6 * (int64_t)cap, d_status + frame);
  __syncthreads();
  if (tid == 0) fwxm_record_one(d_count[frame], d_idx + frame * cap, d_props + frame * 6 * (int64_t)cap, cap, fwxm + frame * 8);
}

// CTP528CP504.mtf's searches (pylinac/ct.py:1511-1544) for every (profile, line-pair region) pair in ONE launch, a wave per
// pair: the `max_number` most prominent peaks inside the region, and -- when exactly that many were found -- the valleys
// (peaks of the negated profile) between the outermost two of them.  Rounds 1-3 issued sixteen pl_find_peaks_regions
// launches plus the torch glue between them per batch.
constexpr int kPvMaxRegions = 16, kPvCap = 8;
struct PvRegions {
  pl_peak_params pk[kPvMaxRegions], vl[kPvMaxRegions];
  int n;
};

__global__ void __launch_bounds__(kThreads)
peak_valley_kernel(const double* __restrict__ x, int64_t nprof, int64_t stride, int len, PvRegions R, int slot_bytes, int maxc,
                   int cap_p, int cap_v, int32_t* __restrict__ d_pk_count, double* __restrict__ d_pk_height,
                   int32_t* __restrict__ d_vl_count, double* __restrict__ d_vl_value, double* __restrict__ d_means) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  constexpr int NW = kThreads / PL_WAVE;
  __shared__ Scan scan_a[NW];
  __shared__ double s_red_a[NW][2 * NW];
  __shared__ int s_cnt_a[NW];
  __shared__ int32_t o_cnt[NW], o_st[NW], o_idx[NW][kPvCap], o_lb[NW][kPvCap], o_rb[NW][kPvCap];
  __shared__ double o_p[NW][6 * kPvCap];
  const int slot = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / PL_WAVE)), tid = threadIdx.x % PL_WAVE;
  const int64_t unit = (int64_t)blockIdx.x * NW + slot;
  if (unit >= nprof * R.n) return;                     // a whole wave leaves
  const int64_t prof = unit / R.n;
  const int k = (int)(unit - prof * R.n);
  const PeakLds L{smem_all + (size_t)slot * slot_bytes, &scan_a[slot], s_red_a[slot], &s_cnt_a[slot]};
  const double* xp = x + prof * stride;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  const pl_peak_params pk = R.pk[k];
  find_peaks_profile<true, PL_WAVE>(xp, len, pk.region_lo, pk.region_hi, pk, cap_p, maxc, L, tid, &o_cnt[slot], o_idx[slot],
                                    o_lb[slot], o_rb[slot], o_p[slot], &o_st[slot]);
  group_sync<PL_WAVE>();
  const int cnt = o_cnt[slot];
  const int lo = cnt > 0 ? o_idx[slot][0] : 0, hi = cnt > 0 ? o_idx[slot][cnt - 1] : 0;   // peaks leave in index order
  if (tid == 0) d_pk_count[unit] = cnt;
  const double hv = tid < cnt && tid < cap_p ? o_p[slot][tid] : nan;      // the peak heights stay in registers for the mean
  if (tid < cap_p) d_pk_height[unit * cap_p + tid] = hv;
  group_sync<PL_WAVE>();
  int vcnt = 0;
  if (cnt == pk.max_number && cnt > 0) {               // wave-uniform
    const pl_peak_params vl = R.vl[k];
    find_peaks_profile<true, PL_WAVE>(xp, len, lo, hi, vl, cap_v, maxc, L, tid, &o_cnt[slot], o_idx[slot], o_lb[slot], o_rb[slot],
                                      o_p[slot], &o_st[slot], -1.0);
    group_sync<PL_WAVE>();
    vcnt = o_cnt[slot];
  }
  if (tid == 0) d_vl_count[unit] = vcnt;
  if (tid < cap_v) d_vl_value[unit * cap_v + tid] = tid < vcnt ? xp[o_idx[slot][tid]] : nan;   // values[valley_idxs]
  if (d_means) {                                          // wave-uniform
    // max_values.mean() / min_values.mean() (pylinac/ct.py:1530, 1536): np.mean of fewer than eight values is the plain
    // left-to-right float64 sum divided by the count; an empty selection gives NaN.  The peak mean is only meaningful when
    // the region held exactly max_number peaks -- the caller's `break` test -- and is NaN otherwise.
    double pm = nan, vm = nan;
    if (cnt == pk.max_number && cnt > 0) {
      double sum = __shfl(hv, 0, PL_WAVE);
      for (int j = 1; j < cnt; ++j) sum = sum + __shfl(hv, j, PL_WAVE);
      pm = sum / (double)cnt;
      if (vcnt > 0) {
        double vs = xp[o_idx[slot][0]];
        for (int j = 1; j < vcnt; ++j) vs = vs + xp[o_idx[slot][j]];
        vm = vs / (double)vcnt;
      }
    }
    if (tid == 0) {
      d_means[unit * 2] = pm;
      d_means[unit * 2 + 1] = vm;
    }
  }
}

}  // namespace

extern "C" int pl_fwxm_record(const int32_t* d_count, const int32_t* d_idx, const double* d_props,
                              int cap, int64_t n, double* d_out, void* stream) {
  PL_REQUIRE(d_count && d_idx && d_props && d_out, "null pointer");
  PL_REQUIRE(n >= 0 && cap > 0, "bad shape");
  if (n == 0) return PL_OK;
  hipLaunchKernelGGL(fwxm_record_kernel, dim3((unsigned)pl_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     d_count, d_idx, d_props, cap, n, d_out);
  return pl_check_launch("pl_fwxm_record");
}

extern "C" int pl_find_peaks_var(const double* d_x, int64_t n, int len, const int32_t* d_lens, int64_t stride,
                                 const pl_peak_params* params, int cap, int32_t* d_count, int32_t* d_idx,
                                 int32_t* d_left_base, int32_t* d_right_base, double* d_props,
                                 int32_t* d_status, void* stream);
extern "C" int pl_find_peaks_regions(const double* d_x, int64_t n, int len, const int32_t* d_lens, int64_t stride,
                                     const pl_peak_params* params, const int32_t* d_regions, int cap, int32_t* d_count,
                                     int32_t* d_idx, int32_t* d_left_base, int32_t* d_right_base, double* d_props,
                                     int32_t* d_status, void* stream);

extern "C" int pl_find_peaks(const double* d_x, int64_t n, int len, int64_t stride,
                             const pl_peak_params* params, int cap, int32_t* d_count, int32_t* d_idx,
                             int32_t* d_left_base, int32_t* d_right_base, double* d_props,
                             int32_t* d_status, void* stream) {
  return pl_find_peaks_var(d_x, n, len, nullptr, stride, params, cap, d_count, d_idx, d_left_base, d_right_base,
                           d_props, d_status, stream);
}

// ragged form: profile i has d_lens[i] <= len samples (row stride `stride`); the search region of
// `params` is clipped to each profile's own length
extern "C" int pl_find_peaks_var(const double* d_x, int64_t n, int len, const int32_t* d_lens, int64_t stride,
                                 const pl_peak_params* params, int cap, int32_t* d_count, int32_t* d_idx,
                                 int32_t* d_left_base, int32_t* d_right_base, double* d_props,
                                 int32_t* d_status, void* stream) {
  return pl_find_peaks_regions(d_x, n, len, d_lens, stride, params, nullptr, cap, d_count, d_idx, d_left_base,
                               d_right_base, d_props, d_status, stream);
}

// per-profile search regions: d_regions int32[n][2] = [lo, hi) of profile i (NULL: the region of `params`).  The
// reference's per-image loops search each profile where ITS peaks were (CTP528CP504.mtf: valleys between the outermost
// peaks of a line-pair region, pylinac/ct.py:1526-1533).
extern "C" int pl_find_peaks_regions(const double* d_x, int64_t n, int len, const int32_t* d_lens, int64_t stride,
                                     const pl_peak_params* params, const int32_t* d_regions, int cap, int32_t* d_count,
                                     int32_t* d_idx, int32_t* d_left_base, int32_t* d_right_base, double* d_props,
                                     int32_t* d_status, void* stream) {
  PL_REQUIRE(d_x && params && d_count && d_idx && d_left_base && d_right_base && d_props && d_status,
             "null pointer");
  PL_REQUIRE(n >= 0 && n <= 0x7fffffffLL && len > 0 && stride >= len && cap > 0, "bad shape");
  PL_REQUIRE(params->distance >= 1, "distance must be >= 1");
  if (n == 0) return PL_OK;
  int lo = params->region_lo < 0 ? 0 : params->region_lo;
  int hi = params->region_hi > len ? len : params->region_hi;
  int m = hi > lo ? hi - lo : 0;
  if (d_regions) m = len;                      // per-profile regions: size the tables for the longest possible one
  int maxc = m / 2 + 1;
  if (maxc > kMaxCand) maxc = kMaxCand;
  const int stage_x = (m <= kStageMax) ? 1 : 0;
  size_t lds = (size_t)maxc * (8 + 8 + 4 * 4) + 8 + (stage_x ? (size_t)m * 8 : 0);
  lds = (lds + 15) & ~(size_t)15;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)find_peaks_kernel<true, kThreads>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)find_peaks_kernel<false, kThreads>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              150 * 1024);
    if (e != hipSuccess) { pl_set_error("pl_find_peaks: LDS attribute: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
    attr_set = true;
  }
  if (stage_x && m <= kShortMax)
    // short profiles (picket-fence windows, 38 samples each, half a million per batch): one WAVE per profile, four
    // profiles per workgroup, no workgroup barrier anywhere
    hipLaunchKernelGGL((find_peaks_kernel<true, PL_WAVE>), dim3((unsigned)pl_cdiv(n, kThreads / PL_WAVE)), dim3(kThreads),
                       lds * (kThreads / PL_WAVE), (hipStream_t)stream, d_x, n, (int)lds, len, d_lens, d_regions, stride,
                       *params, cap, maxc, d_count, d_idx, d_left_base, d_right_base, d_props, d_status);
  else if (stage_x)
    hipLaunchKernelGGL((find_peaks_kernel<true, kThreads>), dim3((unsigned)n), dim3(kThreads), lds, (hipStream_t)stream, d_x,
                       n, (int)lds, len, d_lens, d_regions, stride, *params, cap, maxc, d_count, d_idx, d_left_base,
                       d_right_base, d_props, d_status);
  else
    hipLaunchKernelGGL((find_peaks_kernel<false, kThreads>), dim3((unsigned)n), dim3(kThreads), lds, (hipStream_t)stream, d_x,
                       n, (int)lds, len, d_lens, d_regions, stride, *params, cap, maxc, d_count, d_idx, d_left_base,
                       d_right_base, d_props, d_status);
  return pl_check_launch("pl_find_peaks");
}

// per-band column sums (pl_median3_threshold_colparts_u16) -> mean profile, its peaks, the FWXM record: see the kernel
extern "C" int pl_colparts_profile_fwxm(const uint32_t* d_parts, int64_t n, int bands, int w, int h, const pl_peak_params* params,
                                        int cap, double* d_profile, int32_t* d_count, int32_t* d_idx, int32_t* d_left_base,
                                        int32_t* d_right_base, double* d_props, int32_t* d_status, double* d_fwxm, void* stream) {
  PL_REQUIRE(d_parts && params && d_profile && d_count && d_idx && d_left_base && d_right_base && d_props && d_status && d_fwxm,
             "null pointer");
  PL_REQUIRE(n >= 0 && n <= 0x7fffffffLL && bands > 0 && w > 0 && h > 0 && cap > 0, "bad shape");
  PL_REQUIRE(params->distance >= 1, "distance must be >= 1");
  if (n == 0) return PL_OK;
  const int lo = params->region_lo < 0 ? 0 : params->region_lo;
  const int hi = params->region_hi > w ? w : params->region_hi;
  const int m = hi > lo ? hi - lo : 0;
  int maxc = m / 2 + 1;
  if (maxc > kMaxCand) maxc = kMaxCand;
  const bool stage_x = m <= kStageMax;
  size_t lds = (size_t)maxc * (8 + 8 + 4 * 4) + 8 + (stage_x ? (size_t)m * 8 : 0);
  lds = (lds + 15) & ~(size_t)15;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)colparts_profile_fwxm_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       150 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)colparts_profile_fwxm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) { pl_set_error("pl_colparts_profile_fwxm: LDS attribute: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
    attr_set = true;
  }
  if (stage_x)
    hipLaunchKernelGGL(colparts_profile_fwxm_kernel<true>, dim3((unsigned)n), dim3(kThreads), lds, (hipStream_t)stream, d_parts, bands,
                       w, h, *params, cap, maxc, d_profile, d_count, d_idx, d_left_base, d_right_base, d_props, d_status, d_fwxm);
  else
    hipLaunchKernelGGL(colparts_profile_fwxm_kernel<false>, dim3((unsigned)n), dim3(kThreads), lds, (hipStream_t)stream, d_parts, bands,
                       w, h, *params, cap, maxc, d_profile, d_count, d_idx, d_left_base, d_right_base, d_props, d_status, d_fwxm);
  return pl_check_launch("pl_colparts_profile_fwxm");
}

/* CTP528CP504.mtf's peak and valley searches for every (profile, region) pair in one launch: see peak_valley_kernel */
extern "C" int pl_peak_valley_regions(const double* d_x, int64_t n, int len, int64_t stride, const pl_peak_params* peak_params,
                                      const pl_peak_params* valley_params, int nregions, int cap_p, int cap_v,
                                      int32_t* d_pk_count, double* d_pk_height, int32_t* d_vl_count, double* d_vl_value,
                                      double* d_means, void* stream) {
  PL_REQUIRE(d_x && peak_params && valley_params && d_pk_count && d_pk_height && d_vl_count && d_vl_value, "null pointer");
  PL_REQUIRE(n >= 0 && len > 0 && stride >= len, "bad shape");
  PL_REQUIRE(nregions >= 1 && nregions <= kPvMaxRegions && cap_p >= 1 && cap_p <= kPvCap && cap_v >= 1 && cap_v <= kPvCap,
             "1..16 regions, capacities 1..8");
  PL_REQUIRE(n * nregions <= 0x7fffffffLL, "batch too large");
  if (n == 0) return PL_OK;
  PvRegions R;
  R.n = nregions;
  int m = 0;
  for (int k = 0; k < nregions; ++k) {
    R.pk[k] = peak_params[k];
    R.vl[k] = valley_params[k];
    PL_REQUIRE(R.pk[k].distance >= 1 && R.vl[k].distance >= 1, "distance must be >= 1");
    PL_REQUIRE(R.pk[k].max_number >= 1 && R.pk[k].max_number <= cap_p, "a peak count of 1..cap_p per region");
    const int lo = R.pk[k].region_lo < 0 ? 0 : R.pk[k].region_lo, hi = R.pk[k].region_hi > len ? len : R.pk[k].region_hi;
    if (hi - lo > m) m = hi - lo;
  }
  PL_REQUIRE(m <= 1024, "search regions of at most 1024 samples (a wave per region)");
  const int maxc = m / 2 + 1;
  size_t lds = (size_t)maxc * (8 + 8 + 4 * 4) + 8 + (size_t)m * 8;
  lds = (lds + 15) & ~(size_t)15;
  const int64_t units = n * nregions;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)peak_valley_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) { pl_set_error("pl_peak_valley_regions: LDS attribute: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
    attr_set = true;
  }
  hipLaunchKernelGGL(peak_valley_kernel, dim3((unsigned)pl_cdiv(units, kThreads / PL_WAVE)), dim3(kThreads),
                     lds * (kThreads / PL_WAVE), (hipStream_t)stream, d_x, n, stride, len, R, (int)lds, maxc, cap_p, cap_v,
                     d_pk_count, d_pk_height, d_vl_count, d_vl_value, d_means);
  return pl_check_launch("pl_peak_valley_regions");
}
