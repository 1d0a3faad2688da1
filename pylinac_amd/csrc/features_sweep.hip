// BB / disk finder: the WHOLE threshold sweep of pylinac's find_features for one window in one workgroup
// (BASELINE config #4's per-image BB search).
//
// Replaces pylinac/metrics/utils.py:120-181 (after `sample = stretch(sample, 0, 1)`):
//   while cutoff <= imax and len(total_features) < max_number:
//       labels  = measure.label(sample > cutoff, connectivity=1);  segmentation.clear_border(labels)
//       regions = measure.regionprops(labels, intensity_image=sample)  ->  the five predicates of
//       pylinac/metrics/features.py (is_right_size_bb, is_round, is_right_circumference, is_symmetric, is_solid)
//       new points = Point(weighted_centroid[1], weighted_centroid[0]), de-duplicated by min_separation
//       cutoff += step
// as called through SizedDiskLocator (pylinac/metrics/image.py:564-612) by WLBaseImage.find_bb_centroids
// (pylinac/winston_lutz.py:788-806).
//
// Round 1 ran every level as nine launches over the batch (compare, five labelling kernels on a global label plane,
// three region-table kernels, the candidate kernel) with a host poll every eight levels: 145 us per 134 x 134 window.
// Here a workgroup keeps its window in LDS for the whole sweep:
//   * level map  L(p) = #{k : sample(p) > cutoff_k}  (exact float64 comparisons, once); mask_k(p) = L(p) > k because the
//     cutoffs increase;
//   * per level: row runs of the mask (one lane per row), run ids in raster order, 4-connected merge of vertically
//     overlapping runs with a union-find over RUN ids in LDS (links point to the smaller id, so a component's root is its
//     first run in raster order = scikit-image's label order), per-root area / bbox, candidate roots by the cheap
//     necessary conditions, then the same per-candidate crop analysis as features.hip (8-connected hole fill,
//     perimeter codes, exact convex area, weighted centroid from the float64 samples);
//   * the loop ends at the first level that completes max_number features, like the reference.
// Results are identical to features.hip's level-by-level path (tests run both on the same windows).
#include "pl_common.h"

// Phase stopwatch (-DPL_SWEEP_TIMING, development builds only): per-workgroup s_memtime totals of the level loop's phases.
#ifndef PL_SWEEP_TIMING
#define PL_SWEEP_TIMING 0
#endif
#if PL_SWEEP_TIMING
__device__ unsigned long long pl_sweep_dbg[8];
#define SW_STAMP(k) do { if (tid == 0) { const long long t_ = clock64(); tacc[k] += t_ - tlast; tlast = t_; } } while (0)
#else
#define SW_STAMP(k) do { } while (0)
#endif

namespace {

constexpr int kSwThreads = 256;
constexpr int kSwMaxSide = 160;          // window side limit (coordinates are packed in 8 bits)
constexpr int kSwMaxRuns = 4096;         // row runs of one level: the large table (second pass)
constexpr int kSwMidRuns = 1536;         // ... the second pass's: 67 KB of LDS per workgroup, two workgroups per CU
constexpr int kSwMaxCrop = 64;           // candidate bbox side limit for the crop analysis
constexpr int kSwThirdLds = 52224;       // LDS budget (static + dynamic) of the FIRST pass: three workgroups per CU (160 KB)
constexpr int kSwMaxHullPts = 8 * kSwMaxCrop;
constexpr int kSwMaxOut = 8;
constexpr int kSwMaxLevels = 64;

struct SweepParams {
  double dpmm, radius_mm, tol_mm, min_sep_px;
  int max_number;
  int nlevels;
  double cut[kSwMaxLevels];
};

__device__ __forceinline__ long long sw_cross2(int ax, int ay, int bx, int by, int cx, int cy) {
  return (long long)(bx - ax) * (cy - ay) - (long long)(by - ay) * (cx - ax);
}

__device__ __forceinline__ unsigned sw_find(const unsigned* parent, unsigned i) {
  unsigned p = parent[i];
  while (p != i) {
    i = p;
    p = parent[i];
  }
  return i;
}

__device__ __forceinline__ void sw_unite(unsigned* parent, unsigned a, unsigned b) {
  bool done;
  do {
    a = sw_find(parent, a);
    b = sw_find(parent, b);
    if (a < b) {
      const unsigned old = atomicMin(&parent[b], a);
      done = (old == b);
      b = old;
    } else if (b < a) {
      const unsigned old = atomicMin(&parent[a], b);
      done = (old == a);
      a = old;
    } else {
      done = true;
    }
  } while (!done);
}

// Where the window's samples come from: `sample` (float64 windows, already stretched), or -- f16 != nullptr -- straight from the
// uint16 frames: the sample of SizedDiskRegion.calculate / find_features (pylinac/metrics/image.py:564-612,
// pylinac/metrics/utils.py:112-118) is a chain of MONOTONE float64 maps of the integer pixel,
//   q = (a - frame min) / (frame max - frame min)          BaseImage.ground() / normalize(), winston_lutz.py:711-712
//   s = (-q + max q) + min q   over the window (invert)     metrics/image.py:600-607  (low-density BBs: s = q)
//   stretch(s, 0, 1) = ground(normalize(ground(s)) * 1)     metrics/utils.py:118, array_utils.py:141-168
// so every extremum the chain needs is the image of the window's integer minimum / maximum under the same float
// operations, and the workgroup evaluates the chain per pixel with exactly the operations of the separate kernels
// (pl_ground, pl_normalize, pl_invert, pl_scale): no float64 window ever goes through HBM (round 2: nine passes over it).
struct SweepSrc {
  const unsigned short* f16;
  int fh, fw, top, left;
  const double* vmin;
  const double* vmax;
  int invert;
};

__global__ void __launch_bounds__(kSwThreads)
bb_sweep_kernel(const double* __restrict__ sample, const SweepSrc src, int h, int w, const SweepParams prm,
                int32_t* __restrict__ out_count, double* __restrict__ out_xy, int32_t* __restrict__ out_level,
                int32_t* __restrict__ status, int max_runs, int redo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // Up to three passes share this kernel (sweep_launch): the first runs every frame with the run table that keeps the
  // workgroup within a third of a CU's LDS (the 4096-run table of rounds 1-3 left ONE workgroup per CU: four waves on a CU
  // for a kernel that is all latency).  A level with more runs (speckle) ends the frame with status 5; the later passes
  // (redo != 0) take exactly those frames again with larger tables -- a workgroup whose frame is not marked leaves at once.
  if (redo && status[blockIdx.x] != 5) return;
  // ---- dynamic LDS carve-up
  const int npx = h * w;
  unsigned* run_info = reinterpret_cast<unsigned*>(smem);                 // start | end << 8 | row << 16
  unsigned* parent = run_info + max_runs;
  int* t_area = reinterpret_cast<int*>(parent + max_runs);
  int* t_r1 = t_area + max_runs;
  int* t_c0 = t_r1 + max_runs;
  int* t_c1 = t_c0 + max_runs;
  short* s_hx = reinterpret_cast<short*>(t_c1 + max_runs);               // hull work tables: doubled crop coordinates (< 130)
  short* s_hy = s_hx + kSwMaxHullPts;
  short* s_hull_x = s_hy + kSwMaxHullPts;
  short* s_hull_y = s_hull_x + kSwMaxHullPts + 2;
  // crop planes, one byte per crop pixel: bit 0 region mask, bit 1 reached from the crop border, bit 2 region border pixel.
  // A thread only ever rewrites the byte of its OWN pixel (and never bit 0 after the mask is built), neighbours read single
  // bits of it: the same monotone races as three separate planes, a third of the LDS
  unsigned char* crop = reinterpret_cast<unsigned char*>(s_hull_y + kSwMaxHullPts + 2);
  unsigned char* L = crop + kSwMaxCrop * kSwMaxCrop;                      // level map [npx]
  __shared__ int row_cnt[kSwMaxSide + 1];                                  // runs per row, then exclusive prefix
  __shared__ int s_cand[32];
  __shared__ int s_ncand, s_nh, s_inside, s_nruns;
  __shared__ int s_cnt[kSwThreads / PL_WAVE > 4 ? kSwThreads / PL_WAVE : 4];
  __shared__ double s_red[3][kSwThreads / PL_WAVE];
  __shared__ double s_xy[kSwMaxOut][2];
  __shared__ int s_nout, s_first_level, s_status;
  __shared__ unsigned long long s_used;

  const int64_t img = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const double* smp = src.f16 ? nullptr : sample + img * (int64_t)npx;
  // ---- the uint16 source: window extrema, then the constants of the chain (every lane computes the same scalars)
  const unsigned short* f16 = src.f16 ? src.f16 + img * (int64_t)src.fh * src.fw + (int64_t)src.top * src.fw + src.left : nullptr;
  double c_vmin = 0.0, c_gm = 1.0, c_qmin = 0.0, c_qmax = 0.0, c_smin = 0.0, c_mg = 1.0, c_mn2 = 0.0;
  if (f16) {
    int amin = 65535, amax = 0;
    for (int e = tid; e < npx; e += kSwThreads) {
      const int r = e / w, a = (int)f16[(int64_t)r * src.fw + (e - r * w)];
      amin = a < amin ? a : amin;
      amax = a > amax ? a : amax;
    }
    amin = pl_wave_reduce(amin, [](int a, int b) { return a < b ? a : b; });
    amax = pl_wave_reduce(amax, [](int a, int b) { return a > b ? a : b; });
    if (lane == 0) { s_cnt[wv] = amin; s_cand[wv] = amax; }
    __syncthreads();
    for (int k = 0; k < kSwThreads / PL_WAVE; ++k) { amin = s_cnt[k] < amin ? s_cnt[k] : amin; amax = s_cand[k] > amax ? s_cand[k] : amax; }
    __syncthreads();
    c_vmin = src.vmin[img];
    c_gm = src.vmax[img] - c_vmin;                             // ops.normalize(ops.ground(crop, mn=vmin), vmax - vmin)
    c_qmin = ((double)amin - c_vmin) / c_gm;
    c_qmax = ((double)amax - c_vmin) / c_gm;
    const double s_lo = src.invert ? (-c_qmax + c_qmax) + c_qmin : c_qmin;   // the chain's image of the window's extrema
    const double s_hi = src.invert ? (-c_qmin + c_qmax) + c_qmin : c_qmax;
    c_smin = s_lo;
    c_mg = (s_hi - s_lo) + 0.0;                                // max of ground(s)
    c_mn2 = (((s_lo - s_lo) + 0.0) / c_mg) * 1.0;              // min of normalize(ground(s)) * 1
  }
  auto value_at = [&](int r, int c) -> double {                // the stretched sample of window pixel (r, c)
    if (!f16) return smp[r * w + c];
    const double q = ((double)f16[(int64_t)r * src.fw + c] - c_vmin) / c_gm;
    const double sv = src.invert ? (-q + c_qmax) + c_qmin : q;
    const double g = (sv - c_smin) + 0.0;
    return (((g / c_mg) * 1.0) - c_mn2) + 0.0;
  };
  auto M = [&](int e) { return (crop[e] & 1) != 0; };          // crop planes (see the carve-up)
  auto REACH = [&](int e) { return (crop[e] & 2) != 0; };
  const double dp2 = prm.dpmm * prm.dpmm;
  const double pi = 3.141592653589793;
  const double larger = pi * ((prm.radius_mm + prm.tol_mm) * (prm.radius_mm + prm.tol_mm));
  double smaller = pi * ((prm.radius_mm - prm.tol_mm) * (prm.radius_mm - prm.tol_mm));
  if (!(smaller > 2.0)) smaller = 2.0;                      // max((pi*(r-t)**2, 2))
  if (tid == 0) { s_nout = 0; s_first_level = -1; s_status = 0; s_used = 0ull; }
  __syncthreads();
  unsigned long long used = 0ull;

  // ---- level map: L(p) = number of cutoffs below the sample (binary search, exact float64 comparisons)
  // rows are padded to a multiple of four bytes (the row scans below read dwords); padding = level 0 = never foreground
  const int pitch = (w + 3) & ~3;
  for (int e = tid; e < h * pitch; e += kSwThreads) {
    const int r = e / pitch, c = e - r * pitch;
    int lo = 0;
    if (c < w) {
      const double v = value_at(r, c);
      int hi = prm.nlevels;                                 // invariant: v > cut[k] for k < lo, !(v > cut[k]) for k >= hi
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (v > prm.cut[mid]) lo = mid + 1; else hi = mid;
      }
      used |= 1ull << lo;
    }
    L[e] = (unsigned char)lo;
  }
  // which values the level map takes: mask_k = {L > k} differs from mask_(k-1) only when some pixel has L == k, and an
  // unchanged mask yields the same regions -- rejected again, or duplicates of what the earlier level accepted -- so the
  // reference's loop body is a no-op for that level and it is skipped
  used = pl_wave_reduce(used, [](unsigned long long a, unsigned long long b) { return a | b; });
  if (lane == 0) atomicOr(&s_used, used);
  __syncthreads();
  const unsigned long long level_used = s_used;

#if PL_SWEEP_TIMING
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
#endif
  for (int level = 0; level < prm.nlevels; ++level) {
    // (with min_separation == 0 the reference's duplicate test -- distance STRICTLY below the separation -- never fires, and a
    // repeated mask appends the same centroids again until max_number is reached: every level runs then)
    if (level > 0 && prm.min_sep_px > 0 && !((level_used >> level) & 1ull)) continue;
    SW_STAMP(7);
    // ---- A. the row's foreground as a 160-bit mask (lane = row): four level-map bytes per LDS read, "byte > level" for all
    // four at once (bytes <= 64: adding 127 - level sets bit 7 exactly where the byte exceeds the level), the four flags
    // gathered into a nibble.  Runs start where a set bit follows a clear one: popcount.  (Round 2 walked the row byte by
    // byte here and once more in C: 71 % of the kernel after the hull was fixed.)
    unsigned long long fm[3] = {0ull, 0ull, 0ull}, st[3], en[3];
    int my_runs = 0;
    if (tid < h) {
      const unsigned* row32 = reinterpret_cast<const unsigned*>(L + tid * pitch);
      const unsigned add = (unsigned)(127 - level) * 0x01010101u;
#pragma unroll
      for (int wi = 0; wi < 3; ++wi) {
        unsigned long long acc = 0ull;
#pragma unroll 4
        for (int d = 0; d < 16; ++d) {
          const int dd = 16 * wi + d;
          if (4 * dd >= pitch) break;
          const unsigned y = ((row32[dd] + add) >> 7) & 0x01010101u;
          const unsigned nib = (y | (y >> 7) | (y >> 14) | (y >> 21)) & 0xfu;
          acc |= (unsigned long long)nib << (4 * d);
        }
        fm[wi] = acc;
      }
      st[0] = fm[0] & ~(fm[0] << 1);
      st[1] = fm[1] & ~((fm[1] << 1) | (fm[0] >> 63));
      st[2] = fm[2] & ~((fm[2] << 1) | (fm[1] >> 63));
      en[0] = fm[0] & ~((fm[0] >> 1) | (fm[1] << 63));
      en[1] = fm[1] & ~((fm[1] >> 1) | (fm[2] << 63));
      en[2] = fm[2] & ~(fm[2] >> 1);
      my_runs = __popcll(st[0]) + __popcll(st[1]) + __popcll(st[2]);
    }
    // ---- B. exclusive prefix over rows (h <= 160 < 256 lanes): wave scans + the three wave totals
    {
      int inc = my_runs;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
      }
      if (lane == 63 && wv < 4) s_cnt[wv] = inc;
      __syncthreads();
      int base = 0;
      for (int k = 0; k < 4; ++k) base += (k < wv) ? s_cnt[k] : 0;
      if (tid <= kSwMaxSide) row_cnt[tid] = base + inc - my_runs;     // rows >= h hold the total (my_runs = 0 there)
      if (tid == 0) s_ncand = 0;
      if (tid == h) s_nruns = base + inc - my_runs;
    }
    __syncthreads();
    const int nruns = s_nruns;
    if (nruns > max_runs) {                                  // salt-and-pepper level: more runs than the table holds
      if (tid == 0) s_status = 5;
      break;
    }
    if (nruns == 0) continue;                                // uniform on every lane: nothing at this level
    // ---- C. write the runs (k-th start bit pairs with the k-th end bit), initialise the forest and the per-root accumulators
    if (tid < h) {
      int id = row_cnt[tid];
      int ew = 0;
      unsigned long long ecur = en[0];
#pragma unroll
      for (int wi = 0; wi < 3; ++wi) {
        unsigned long long scur = st[wi];
        while (scur) {
          const int start = 64 * wi + __builtin_ctzll(scur);
          scur &= scur - 1ull;
          while (!ecur) { ++ew; ecur = ew == 1 ? en[1] : en[2]; }
          const int end = 64 * ew + __builtin_ctzll(ecur);
          ecur &= ecur - 1ull;
          run_info[id] = (unsigned)start | ((unsigned)end << 8) | ((unsigned)tid << 16);
          parent[id] = (unsigned)id;
          t_area[id] = 0; t_r1[id] = 0; t_c0[id] = 255; t_c1[id] = 0;
          ++id;
        }
      }
    }
    __syncthreads();
    // ---- D. 4-connected merge: runs of row r against runs of row r-1 (both sorted by column)
    if (tid >= 1 && tid < h) {
      int a = row_cnt[tid], a_end = row_cnt[tid + 1];
      int b = row_cnt[tid - 1];
      const int b_end = row_cnt[tid];
      while (a < a_end && b < b_end) {
        const unsigned ia = run_info[a], ib = run_info[b];
        const int sa = ia & 255, ea = (ia >> 8) & 255, sb = ib & 255, eb = (ib >> 8) & 255;
        if (sa <= eb && sb <= ea) sw_unite(parent, (unsigned)a, (unsigned)b);
        if (ea < eb) ++a; else ++b;
      }
    }
    __syncthreads();
    // ---- E. flatten, F. per-root area / bbox
    for (int id = tid; id < nruns; id += kSwThreads) parent[id] = sw_find(parent, (unsigned)id);
    __syncthreads();
    for (int id = tid; id < nruns; id += kSwThreads) {
      const unsigned info = run_info[id];
      const int s = info & 255, e = (info >> 8) & 255, r = (int)(info >> 16);
      const unsigned root = parent[id];
      atomicAdd(&t_area[root], e - s + 1);
      atomicMax(&t_r1[root], r);
      atomicMin(&t_c0[root], s);
      atomicMax(&t_c1[root], e);
    }
    __syncthreads();
    // ---- G. candidate roots: necessary conditions from (area, bbox) only (same as features.hip)
    for (int id = tid; id < nruns; id += kSwThreads) {
      if (parent[id] != (unsigned)id) continue;
      const double area = (double)t_area[id];
      const int r0 = (int)(run_info[id] >> 16), c0 = t_c0[id], r1 = t_r1[id] + 1, c1 = t_c1[id] + 1;   // half-open
      const double bbox_area = (double)(r1 - r0) * (double)(c1 - c0);
      if (r0 == 0 || c0 == 0 || r1 == h || c1 == w) continue;          // clear_border
      if (!(area / dp2 < larger)) continue;                            // filled_area >= area
      if (!(bbox_area / dp2 > smaller)) continue;                      // filled_area <= bbox_area
      const double y = (double)(r1 - r0), x = (double)(c1 - c0);       // is_symmetric (features.py:7-14)
      const double hi = (y * 1.05 > y + 3) ? y * 1.05 : y + 3, lo = (y * 0.95 < y - 3) ? y * 0.95 : y - 3;
      if (x > hi || x < lo) continue;
      if (!(area / bbox_area < pi / 4 * 1.2)) continue;                // is_round upper bound needs filled >= area
      const int slot = atomicAdd(&s_ncand, 1);
      if (slot < 32) s_cand[slot] = id;
    }
    __syncthreads();
    SW_STAMP(0);                                              // runs, merge, flatten, region table, candidate screen
    int ncand = s_ncand;
    if (ncand > 32) { ncand = 32; if (tid == 0) s_status = 2; }
    if (tid == 0)                                             // label order = raster order of the first pixel = root id order
      for (int a = 1; a < ncand; ++a) { int v = s_cand[a], b = a - 1; while (b >= 0 && s_cand[b] > v) { s_cand[b + 1] = s_cand[b]; --b; } s_cand[b + 1] = v; }
    __syncthreads();

    for (int ci = 0; ci < ncand; ++ci) {
      const int k = s_cand[ci];
      const int r0 = (int)(run_info[k] >> 16), c0 = t_c0[k], r1 = t_r1[k] + 1, c1 = t_c1[k] + 1;
      const int ch = r1 - r0, cw = c1 - c0;
      const double area = (double)t_area[k];
      if (ch > kSwMaxCrop || cw > kSwMaxCrop) { if (tid == 0) s_status = 3; continue; }
      const int cpx = ch * cw;
      // region mask of the crop from the runs of its rows
      for (int e = tid; e < cpx; e += kSwThreads) crop[e] = 0;
      __syncthreads();
      for (int rr = tid; rr < ch; rr += kSwThreads) {
        for (int id = row_cnt[r0 + rr]; id < row_cnt[r0 + rr + 1]; ++id) {
          if (parent[id] != (unsigned)k) continue;
          const unsigned info = run_info[id];
          const int s = info & 255, e = (info >> 8) & 255;
          for (int c = s; c <= e; ++c) crop[rr * cw + (c - c0)] = 1;
        }
      }
      __syncthreads();
      SW_STAMP(1);                                            // crop mask
      // ---- filled_area: non-region pixels reachable (8-conn) from the crop border are NOT holes
      for (int e = tid; e < cpx; e += kSwThreads) {
        const int r = e / cw, c = e % cw;
        const bool edge = (r == 0 || c == 0 || r == ch - 1 || c == cw - 1);
        const bool me = M(e);
        bool b = false;
        if (me) b = (r == 0 || !M(e - cw)) || (r == ch - 1 || !M(e + cw)) || (c == 0 || !M(e - 1)) || (c == cw - 1 || !M(e + 1));
        crop[e] = (unsigned char)((me ? 1 : 0) | ((!me && edge) ? 2 : 0) | (b ? 4 : 0));
      }
      __syncthreads();
      for (;;) {
        int changed = 0;
        for (int e = tid; e < cpx; e += kSwThreads) {
          if (crop[e] & 3) continue;
          const int r = e / cw, c = e % cw;
          bool hit = false;
          for (int dr = -1; dr <= 1 && !hit; ++dr)
            for (int dc = -1; dc <= 1; ++dc) {
              const int rr = r + dr, cc = c + dc;
              if ((dr | dc) == 0 || rr < 0 || cc < 0 || rr >= ch || cc >= cw) continue;
              if (REACH(rr * cw + cc)) { hit = true; break; }
            }
          if (hit) { crop[e] = 2; changed = 1; }
        }
        if (!__syncthreads_or(changed)) break;
      }
      SW_STAMP(2);                                            // flood fill
      if (tid < 4) s_cnt[tid] = 0;
      __syncthreads();
      // ---- holes, perimeter codes, weighted moments
      int holes = 0, n1 = 0, n2 = 0, n3 = 0;
      double w0 = 0.0, wr = 0.0, wc = 0.0;
      for (int e = tid; e < cpx; e += kSwThreads) {
        const int r = e / cw, c = e % cw;
        if (!(crop[e] & 3)) ++holes;
        if (crop[e] & 4) {
          auto B = [&](int rr, int cc) { return (rr < 0 || cc < 0 || rr >= ch || cc >= cw) ? 0 : (int)((crop[rr * cw + cc] >> 2) & 1); };
          const int code = 1 + 2 * (B(r - 1, c) + B(r + 1, c) + B(r, c - 1) + B(r, c + 1)) +
                           10 * (B(r - 1, c - 1) + B(r - 1, c + 1) + B(r + 1, c - 1) + B(r + 1, c + 1));
          if (code == 5 || code == 7 || code == 15 || code == 17 || code == 25 || code == 27) ++n1;
          else if (code == 21 || code == 33) ++n2;
          else if (code == 13 || code == 23) ++n3;
        }
        if (M(e)) {
          const double v = value_at(r0 + r, c0 + c);
          w0 += v; wr += v * (double)r; wc += v * (double)c;
        }
      }
      auto addi = [](int a, int b) { return a + b; };
      auto addd = [](double a, double b) { return a + b; };
      holes = pl_wave_reduce(holes, addi); n1 = pl_wave_reduce(n1, addi); n2 = pl_wave_reduce(n2, addi); n3 = pl_wave_reduce(n3, addi);
      w0 = pl_wave_reduce(w0, addd); wr = pl_wave_reduce(wr, addd); wc = pl_wave_reduce(wc, addd);
      if (lane == 0) {
        atomicAdd(&s_cnt[0], holes); atomicAdd(&s_cnt[1], n1); atomicAdd(&s_cnt[2], n2); atomicAdd(&s_cnt[3], n3);
        s_red[0][wv] = w0; s_red[1][wv] = wr; s_red[2][wv] = wc;
      }
      SW_STAMP(3);                                            // holes, perimeter, moments
      // ---- convex hull of the pixels' mid-edge points (the "diamond" hull skimage's convex_hull_image builds).
      // Only the row-extreme pixels matter, and of their points only, for every doubled abscissa X = 2 r - 1 .. 2 r + 1, the
      // lowest and the highest ordinate: X even (= 2 r): 2 cl - 1 and 2 cr + 1; X odd (between rows r and r + 1): min of the two
      // rows' 2 cl, max of their 2 cr.  That is at most 2 (2 ch + 1) points ALREADY SORTED by X.  Round 2 sorted eight points per
      // row by insertion and ran Andrew's chains on ONE lane: 73 % of this kernel's time (phase stopwatch, -DPL_SWEEP_TIMING).
      // Now one wave prunes each chain in parallel: a point whose turn (previous alive, itself, next alive) is not strictly
      // counter-clockwise lies on or inside the chord of two other points and is no hull vertex -- all such points go at once,
      // until none is left: the same strictly convex polygon as the sequential chains, a handful of rounds.
      const int nx = 2 * ch + 1;
      if (nx <= 63) {                                       // wave-uniform: the usual BB blob (ch <= 31 rows)
        for (int r = tid; r < ch; r += kSwThreads) {
          int cl = -1, cr = -1;
          for (int c = 0; c < cw; ++c) if (M(r * cw + c)) { if (cl < 0) cl = c; cr = c; }
          s_hx[r] = (short)cl; s_hy[r] = (short)cr;         // (a connected region has a pixel in every row of its bbox)
        }
        __syncthreads();
        if (wv == 0) {
          const int big = 0x3fffffff;
          int ymin = big, ymax = -big;
          if (lane < nx) {
            if (lane & 1) {                                 // X = lane - 1 even = 2 r
              const int r = (lane - 1) >> 1;
              if (s_hx[r] >= 0) { ymin = 2 * s_hx[r] - 1; ymax = 2 * s_hy[r] + 1; }
            } else {                                        // X odd: between rows r and r + 1
              const int r = (lane >> 1) - 1;
              if (r >= 0 && s_hx[r] >= 0) { ymin = 2 * s_hx[r]; ymax = 2 * s_hy[r]; }
              if (r + 1 < ch && s_hx[r + 1] >= 0) {
                ymin = ymin < 2 * s_hx[r + 1] ? ymin : 2 * s_hx[r + 1];
                ymax = ymax > 2 * s_hy[r + 1] ? ymax : 2 * s_hy[r + 1];
              }
            }
          }
          // chain q = 0: lower (X ascending, lowest ordinates, then the last abscissa's highest); q = 1: upper (X descending,
          // highest ordinates, then the first abscissa's lowest): nx + 1 points each
          int nh = 0;
#pragma unroll 1
          for (int q = 0; q < 2; ++q) {
            const int src = q == 0 ? (lane < nx ? lane : nx - 1) : (lane < nx ? nx - 1 - lane : 0);
            const int ylo = __shfl(ymin, src, 64), yhi = __shfl(ymax, src, 64);
            const int X = src - 1;
            const int Y = q == 0 ? (lane < nx ? ylo : yhi) : (lane < nx ? yhi : ylo);
            const int npts = nx + 1;
            unsigned long long alive = npts >= 64 ? ~0ull : (1ull << npts) - 1ull;
            for (;;) {
              const unsigned long long below = alive & ((1ull << lane) - 1ull);
              const unsigned long long above = lane >= 63 ? 0ull : (alive & ~((2ull << lane) - 1ull));
              const int pi = below ? 63 - __builtin_clzll(below) : 0, ni = above ? __builtin_ctzll(above) : 0;
              const int xp = __shfl(X, pi, 64), yp = __shfl(Y, pi, 64), xn = __shfl(X, ni, 64), yn = __shfl(Y, ni, 64);
              const bool interior = ((alive >> lane) & 1ull) && below && above;
              const unsigned long long gone = __ballot(interior && sw_cross2(xp, yp, X, Y, xn, yn) <= 0);
              if (!gone) break;
              alive &= ~gone;
            }
            // the chain's vertices without its last point (= the first point of the other chain)
            const unsigned long long keep = alive & ~(1ull << (npts - 1));
            if ((keep >> lane) & 1ull) {
              const int at = nh + __popcll(keep & ((1ull << lane) - 1ull));
              s_hull_x[at] = (short)X; s_hull_y[at] = (short)Y;
            }
            nh += __popcll(keep);
          }
          if (lane == 0) { s_nh = nh; s_inside = 0; }
        }
      } else {
      // tall crops: the sequential form (eight points per row, sorted, Andrew's chains on one lane)
      for (int r = tid; r < ch; r += kSwThreads) {
        int cl = -1, cr = -1;
        for (int c = 0; c < cw; ++c) if (M(r * cw + c)) { if (cl < 0) cl = c; cr = c; }
        short* px = s_hx + r * 8; short* py = s_hy + r * 8;
        for (int q = 0; q < 8; ++q) { px[q] = 0x7fff; py[q] = 0; }
        if (cl >= 0) {
          const int xs[2] = {cl, cr};
          for (int q = 0; q < 2; ++q) {
            const int X = 2 * r, Y = 2 * xs[q];
            px[4 * q + 0] = (short)X;       py[4 * q + 0] = (short)(Y - 1);
            px[4 * q + 1] = (short)X;       py[4 * q + 1] = (short)(Y + 1);
            px[4 * q + 2] = (short)(X - 1); py[4 * q + 2] = (short)Y;
            px[4 * q + 3] = (short)(X + 1); py[4 * q + 3] = (short)Y;
          }
        }
      }
      __syncthreads();
      if (tid == 0) {
        const int np = ch * 8;
        for (int a = 1; a < np; ++a) {
          const short vx = s_hx[a], vy = s_hy[a];
          int b = a - 1;
          while (b >= 0 && (s_hx[b] > vx || (s_hx[b] == vx && s_hy[b] > vy))) { s_hx[b + 1] = s_hx[b]; s_hy[b + 1] = s_hy[b]; --b; }
          s_hx[b + 1] = vx; s_hy[b + 1] = vy;
        }
        int n = 0;
        while (n < np && s_hx[n] != 0x7fff) ++n;
        int u = 0;
        for (int a = 0; a < n; ++a) if (a == 0 || s_hx[a] != s_hx[a - 1] || s_hy[a] != s_hy[a - 1]) { s_hx[u] = s_hx[a]; s_hy[u] = s_hy[a]; ++u; }
        n = u;
        int kk = 0;
        for (int a = 0; a < n; ++a) {           // lower chain
          while (kk >= 2 && sw_cross2(s_hull_x[kk - 2], s_hull_y[kk - 2], s_hull_x[kk - 1], s_hull_y[kk - 1], s_hx[a], s_hy[a]) <= 0) --kk;
          s_hull_x[kk] = s_hx[a]; s_hull_y[kk] = s_hy[a]; ++kk;
        }
        const int lower = kk + 1;
        for (int a = n - 2; a >= 0; --a) {      // upper chain
          while (kk >= lower && sw_cross2(s_hull_x[kk - 2], s_hull_y[kk - 2], s_hull_x[kk - 1], s_hull_y[kk - 1], s_hx[a], s_hy[a]) <= 0) --kk;
          s_hull_x[kk] = s_hx[a]; s_hull_y[kk] = s_hy[a]; ++kk;
        }
        s_nh = kk - 1;                          // last point == first point
        s_inside = 0;
      }
      }
      __syncthreads();
      SW_STAMP(4);                                            // hull
      const int nh = s_nh;
      int inside = 0;
      for (int e = tid; e < cpx; e += kSwThreads) {
        const int X = 2 * (e / cw), Y = 2 * (e % cw);
        bool in = true;
        for (int a = 0; a < nh && in; ++a) {
          const int b = (a + 1 == nh) ? 0 : a + 1;
          in = sw_cross2(s_hull_x[a], s_hull_y[a], s_hull_x[b], s_hull_y[b], X, Y) >= 0;
        }
        inside += in ? 1 : 0;
      }
      inside = pl_wave_reduce(inside, addi);
      if (lane == 0) atomicAdd(&s_inside, inside);
      __syncthreads();
      SW_STAMP(5);                                            // pixels inside the hull
      // ---- predicates (pylinac/metrics/features.py) and output
      if (tid == 0) {
        const double filled = area + (double)s_cnt[0];
        const double perim = ((double)s_cnt[1] * 1.0 + (double)s_cnt[2] * 1.4142135623730951) +
                             (double)s_cnt[3] * ((1 + 1.4142135623730951) / 2);
        const double per_mm = perim / prm.dpmm;
        const double bbox_area = (double)ch * (double)cw;
        bool ok = true;
        const double bb_area = filled / dp2;
        ok = ok && (smaller < bb_area && bb_area < larger);                            // is_right_size_bb
        const double ratio = filled / bbox_area;
        ok = ok && (pi / 4 * 1.2 > ratio && ratio > pi / 4 * 0.8);                     // is_round
        ok = ok && (2 * pi * (prm.radius_mm + prm.tol_mm) > per_mm && per_mm > 2 * pi * (prm.radius_mm - prm.tol_mm));
        ok = ok && (area / (double)s_inside > 0.9);                                    // is_solid
        if (ok) {
          double m0 = 0.0, mr = 0.0, mc = 0.0;
          for (int q = 0; q < kSwThreads / PL_WAVE; ++q) { m0 += s_red[0][q]; mr += s_red[1][q]; mc += s_red[2][q]; }
          const double py = mr / m0 + (double)r0, px = mc / m0 + (double)c0;
          // de-duplicate against every point accepted so far, INCLUDING this level's (metrics/utils.py:28-36:
          // `combined_points` aliases `original_points`, so the list being iterated grows)
          bool keep = true;
          for (int q = 0; q < s_nout; ++q) {
            const double dx = px - s_xy[q][0], dy = py - s_xy[q][1];
            if (sqrt(dx * dx + dy * dy) < prm.min_sep_px) { keep = false; break; }
          }
          if (keep) {
            if (s_nout < kSwMaxOut) {
              s_xy[s_nout][0] = px; s_xy[s_nout][1] = py;
              ++s_nout;
              if (s_first_level < 0) s_first_level = level;
            } else {
              s_status = 4;
            }
          }
        }
      }
      __syncthreads();
    }
    if (s_nout >= prm.max_number) break;                     // while ... len(total_features) < max_number
  }
  __syncthreads();
#if PL_SWEEP_TIMING
  if (tid == 0) for (int k = 0; k < 8; ++k) atomicAdd(&pl_sweep_dbg[k], (unsigned long long)tacc[k]);
#endif
  if (tid == 0) {
    out_count[img] = s_nout;
    out_level[img] = s_first_level;
    status[img] = s_status;
    for (int q = 0; q < kSwMaxOut; ++q) {
      out_xy[(img * kSwMaxOut + q) * 2] = q < s_nout ? s_xy[q][0] : 0.0;
      out_xy[(img * kSwMaxOut + q) * 2 + 1] = q < s_nout ? s_xy[q][1] : 0.0;
    }
  }
}

}  // namespace

/* The whole find_features threshold sweep (BB mode) for n windows of h x w float64 samples (already stretched to
 * [0, 1]): d_cutoffs (HOST memory, nlevels <= 64 increasing values: imin + step, imin + 2 step, ... as the reference
 * accumulates them).  Outputs as pl_features_level: d_count int32[n], d_xy float64[n][8][2] (x, y) window coordinates,
 * d_level int32[n] (first level that produced a feature, -1 none), d_status int32[n]: 0 ok, 2 more than 32 candidate
 * regions at a level, 3 a candidate's bbox exceeds 64 pixels, 4 more than 8 features, 5 a level has more than 4096 row runs
 * (the caller then uses the level-by-level path).  Windows up to 160 x 160. */
namespace {
int sweep_launch(const double* d_sample, const SweepSrc& src, int64_t n, int h, int w, double dpmm, double radius_mm, double tol_mm,
                 double min_sep_px, int max_number, const double* h_cutoffs, int nlevels, int32_t* d_count, double* d_xy,
                 int32_t* d_level, int32_t* d_status, void* stream, const char* who) {
  SweepParams prm;
  prm.dpmm = dpmm; prm.radius_mm = radius_mm; prm.tol_mm = tol_mm; prm.min_sep_px = min_sep_px;
  prm.max_number = max_number;
  prm.nlevels = nlevels;
  for (int k = 0; k < kSwMaxLevels; ++k) prm.cut[k] = k < nlevels ? h_cutoffs[k] : 0.0;
  for (int k = 1; k < nlevels; ++k)
    if (!(h_cutoffs[k] > h_cutoffs[k - 1])) { pl_set_error("%s: cutoffs must increase", who); return PL_ERR_INVALID_ARG; }
  const size_t fixed = ((size_t)2 * kSwMaxHullPts + 2 * (kSwMaxHullPts + 2)) * sizeof(short) + (size_t)kSwMaxCrop * kSwMaxCrop +
                       (size_t)h * ((w + 3) & ~3);
  auto lds_for = [&](int runs) { return (size_t)runs * 6 * 4 + fixed; };
  static std::atomic<size_t> attr_lds{0};
  static std::atomic<int> static_lds{-1};
  if (static_lds < 0) {
    hipFuncAttributes fa;
    static_lds = hipFuncGetAttributes(&fa, (const void*)bb_sweep_kernel) == hipSuccess ? (int)fa.sharedSizeBytes : 2048;
    (void)hipGetLastError();
  }
  if (lds_for(kSwMaxRuns) > attr_lds) {
    hipError_t e = hipFuncSetAttribute((const void*)bb_sweep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_for(kSwMaxRuns));
    if (e != hipSuccess) { pl_set_error("%s: LDS attribute: %s", who, hipGetErrorString(e)); return PL_ERR_HIP; }
    attr_lds = lds_for(kSwMaxRuns);
  }
  // Up to three passes over the batch, each taking only the frames the one before left with status 5 (a level with more row
  // runs than its table holds; a workgroup whose frame is not marked leaves at once: 4 us per empty pass):
  //   1. the largest run table that keeps the workgroup within a THIRD of a CU's LDS (896 runs for a 144 x 144 window) --
  //      the kernel is all latency, four waves per workgroup: 1 250 Winston-Lutz frames are 1.6 rounds of 768 slots instead
  //      of 2.4 rounds of 512 (profiles/r05j_wl_sweep_occupancy_ab.txt: 544 -> 442 us with a timing-only build);
  //   2. 1 536 runs, two workgroups per CU (rounds 3-5's first pass);   3. 4 096 runs, one per CU.
  // Measured on 512 frames (round 3, scripts/time_wl_variants.py): 1 536 + 4 096: 0.956 ms clean / 1.157 noisy; 4 096 alone 1.080 / 1.319.
  int first = (int)(((long long)kSwThirdLds - static_lds - (long long)fixed) / 24) & ~63;
  if (first > kSwMidRuns) first = kSwMidRuns;
  if (first < 256) first = kSwMidRuns;                                  // a window too large for three per CU: rounds 3-5's pair
  const int tiers[3] = {first, first < kSwMidRuns ? kSwMidRuns : 0, kSwMaxRuns};
  int redo = 0;
  for (int t = 0; t < 3; ++t) {
    if (tiers[t] == 0) continue;
    hipLaunchKernelGGL(bb_sweep_kernel, dim3((unsigned)n), dim3(kSwThreads), lds_for(tiers[t]), (hipStream_t)stream, d_sample, src, h, w,
                       prm, d_count, d_xy, d_level, d_status, tiers[t], redo);
    redo = 1;
  }
  return pl_check_launch(who);
}
}  // namespace

extern "C" int pl_features_sweep(const double* d_sample, int64_t n, int h, int w, double dpmm, double radius_mm,
                                 double tol_mm, double min_sep_px, int max_number, const double* h_cutoffs, int nlevels,
                                 int32_t* d_count, double* d_xy, int32_t* d_level, int32_t* d_status, void* stream) {
  PL_REQUIRE(d_sample && h_cutoffs && d_count && d_xy && d_level && d_status, "null pointer");
  PL_REQUIRE(n >= 0 && n <= 0x7fffffffLL && h > 0 && w > 0 && h <= kSwMaxSide && w <= kSwMaxSide, "bad shape");
  PL_REQUIRE(nlevels > 0 && nlevels <= kSwMaxLevels && max_number > 0, "bad arguments");
  PL_REQUIRE(dpmm > 0 && radius_mm > 0, "bad physical parameters");
  if (n == 0) return PL_OK;
  const SweepSrc src{nullptr, 0, 0, 0, 0, nullptr, nullptr, 0};
  return sweep_launch(d_sample, src, n, h, w, dpmm, radius_mm, tol_mm, min_sep_px, max_number, h_cutoffs, nlevels, d_count,
                      d_xy, d_level, d_status, stream, "pl_features_sweep");
}

extern "C" int pl_features_sweep_u16(const uint16_t* d_frames, int64_t n, int frame_h, int frame_w, int top, int left, int h,
                                     int w, const double* d_vmin, const double* d_vmax, int invert, double dpmm,
                                     double radius_mm, double tol_mm, double min_sep_px, int max_number,
                                     const double* h_cutoffs, int nlevels, int32_t* d_count, double* d_xy, int32_t* d_level,
                                     int32_t* d_status, void* stream) {
  PL_REQUIRE(d_frames && d_vmin && d_vmax && h_cutoffs && d_count && d_xy && d_level && d_status, "null pointer");
  PL_REQUIRE(n >= 0 && n <= 0x7fffffffLL && h > 0 && w > 0 && h <= kSwMaxSide && w <= kSwMaxSide, "bad shape");
  PL_REQUIRE(frame_h > 0 && frame_w > 0 && top >= 0 && left >= 0 && top + h <= frame_h && left + w <= frame_w,
             "the window must lie inside the frame");
  PL_REQUIRE(nlevels > 0 && nlevels <= kSwMaxLevels && max_number > 0, "bad arguments");
  PL_REQUIRE(dpmm > 0 && radius_mm > 0, "bad physical parameters");
  if (n == 0) return PL_OK;
  const SweepSrc src{d_frames, frame_h, frame_w, top, left, d_vmin, d_vmax, invert ? 1 : 0};
  return sweep_launch(nullptr, src, n, h, w, dpmm, radius_mm, tol_mm, min_sep_px, max_number, h_cutoffs, nlevels, d_count,
                      d_xy, d_level, d_status, stream, "pl_features_sweep_u16");
}

#if PL_SWEEP_TIMING
extern "C" int pl_debug_sweep_timing(unsigned long long* h_out) {
  return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(pl_sweep_dbg), sizeof(pl_sweep_dbg)) == hipSuccess ? 0 : 1;
}
#endif
