// pylinac.core.profile.find_peaks as DEVICE functions (see peaks.hip for the algorithm and its reference lines): the search
// of one profile by 64 or 256 lanes, usable from any kernel that has the profile in memory -- the stand-alone launches of
// peaks.hip, the CTP528 peak/valley kernel, the EPID tail, and the picket-fence window kernel, which searches the window
// profile it has just built instead of writing it out for a second launch.
// (included after pl_common.h by the .hip files that use it)
#pragma once
#include <math.h>

namespace {

constexpr int kPkThreads = 256;   // lanes of a whole-workgroup search
constexpr int kStageMax = 4096;   // profiles up to this length are staged in LDS
constexpr int kMaxCand = 4096;    // candidate peaks kept per profile
constexpr int kShortMax = 128;    // search regions up to this length: one wave per profile

struct Scan { int wave[kPkThreads / PL_WAVE]; };

// NT lanes work on one profile: 256 (a whole workgroup; the barrier is the workgroup's) or 64 (one wave of a
// four-profile workgroup: the wave is in lock step, the "barrier" only orders its LDS traffic for the compiler)
template <int NT>
__device__ __forceinline__ void group_sync() {
  if constexpr (NT == PL_WAVE) {
    pl_wave_sync();
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
  int step = 0, rounds = 0;
  for (int t0 = 0;; t0 += PL_WAVE) {
    const int t = t0 + lane;
    const int j = pk + DIR * t;
    const bool inside = DIR < 0 ? (j >= 0) : (j <= m - 1);
    const double v = inside ? xs[j] : xp;
    const bool fail = !inside || !(v <= xp);
    const unsigned long long b = __ballot(fail);
    const int first = b ? __builtin_ctzll(b) : PL_WAVE;
    if (lane < first && v < mn) { mn = v; step = t; }
    ++rounds;
    if (b) break;
  }
  // the smallest value and, among equal values, the earliest step
  const double best = pl_wave_reduce_idem(mn, [](double a, double b) { return a < b ? a : b; });
  if (rounds == 1) {
    // one round (the walk ended within 64 samples -- every picket-fence window): a lane's step is its own index, so the
    // earliest step is the lowest lane that holds the minimum -- a ballot instead of a second reduction; a minimum equal to the
    // peak itself was never "updated": step 0
    const unsigned long long at = __ballot(mn == best);
    step = best == xp ? 0 : __builtin_ctzll(at);
  } else {
    step = pl_wave_reduce_idem(mn == best ? step : 0x7fffffff, [](int a, int b) { return a < b ? a : b; });
  }
  mn = best;
  out_min = mn;
  out_base = pk + DIR * step;
}

struct PeakLds {                                   // the LDS of one profile's search
  unsigned char* smem;                             // dynamic block: candidate tables (+ the staged profile)
  Scan* scan;
  double* s_red;                                   // 2 x (kPkThreads / PL_WAVE)
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
    mn = pl_wave_reduce_idem(mn, [](double a, double b) { return a < b ? a : b; });
    mx = pl_wave_reduce_idem(mx, [](double a, double b) { return a > b ? a : b; });
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

  // ---- the FWXM search (FWXMProfile.field_edge_idx: max_number = 1 by prominence, no prominence / width / distance filter):
  // the one peak that survives stage G is the most prominent candidate -- among equals the LAST (np.argsort(kind="stable")
  // [::-1][:1]) -- so the candidate tables, the O(P^2) ranking and the two ordered scans of stages G / H are skipped: every
  // wave keeps the best of its candidates in registers and the wave that holds the overall best measures its width.
  // (Round 5 tried to go further for one-wave profiles of <= 64 samples -- lane i holds sample i; if exactly one candidate holds
  // the global maximum M its bases are the side minima and any other candidate's prominence is bounded by h' - min(profile),
  // so five wave reductions replace the walks -- bit-identical on every exit, and NOT faster: a picket-fence window has two
  // or three candidates, whose walks cost the same five reductions.  profiles/r05b_pf_window_variants.txt; not kept.)
  if (prm.max_number == 1 && prm.sort_key == PL_SORT_PROMINENCES && !prm.has_prominence && !(prm.width_min > 0.0)) {
    double best_prom = -1.0;                               // prominences are >= 0
    int best_p = -1, best_lb = 0, best_rb = 0;
    for (int p = tid / PL_WAVE; p < P; p += NT / PL_WAVE) {
      const int pk = s_idx[p];
      double left_min, right_min;
      int lb, rb;
      prominence_side<-1>(xs, pk, m, left_min, lb);
      prominence_side<+1>(xs, pk, m, right_min, rb);
      const double prom = xs[pk] - (left_min > right_min ? left_min : right_min);
      if (prom >= best_prom) { best_prom = prom; best_p = p; best_lb = lb; best_rb = rb; }   // p ascends: the last of equals
    }
    // the waves' bests meet in s_red (stage A is done with it): [prominence x 4][candidate index x 4]; the wave that owns the
    // winner still has its bases in registers and reports
    int win_p = best_p;
    if constexpr (NT > PL_WAVE) {
      const int wv = tid >> 6;
      group_sync<NT>();
      if ((tid & 63) == 0) { s_red[wv] = best_prom; s_red[4 + wv] = (double)best_p; }
      group_sync<NT>();
      double win_prom = -1.0;
      win_p = -1;
      for (int k = 0; k < NT / PL_WAVE; ++k) {
        const double pr = s_red[k];
        const int pp = (int)s_red[4 + k];
        if (pp >= 0 && (win_p < 0 || pr > win_prom || (pr == win_prom && pp > win_p))) { win_prom = pr; win_p = pp; }
      }
    }
    if (best_p >= 0 && best_p == win_p) {                  // wave-uniform: the winner's wave
      const int pk = s_idx[best_p];
      const Widths wd = peak_width(xs, pk, best_lb, best_rb, best_prom, prm.rel_height);
      if ((tid & (PL_WAVE - 1)) == 0) {
        o_idx[0] = pk + lo;
        o_lb[0] = best_lb;
        o_rb[0] = best_rb;
        o_p[0 * cap] = xs[pk];
        o_p[1 * cap] = best_prom;
        o_p[2 * cap] = wd.width;
        o_p[3 * cap] = wd.height;
        o_p[4 * cap] = wd.lip;
        o_p[5 * cap] = wd.rip;
      }
    }
    if (tid == 0) {
      *o_count = win_p >= 0 ? 1 : 0;
      *o_status = overflow ? 2 : 0;
    }
    return;
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
    // the width of every candidate is only needed for a width filter or a ranking by width; the peaks that are reported get
    // theirs in stage H anyway (FWXM searches -- max_number = 1 by prominence, no width limit -- skip this walk)
    double width = 0.0;
    if (prm.width_min > 0.0 || (prm.max_number > 0 && prm.sort_key == PL_SORT_WIDTHS)) {
      width = peak_width(xs, pk, lb, rb, prom, prm.rel_height).width;
      keep = keep && (width >= prm.width_min);
    }
    if ((tid & (PL_WAVE - 1)) == 0) {
      s_prom[p] = prom;
      s_width[p] = width;
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

}  // namespace
