// BB / disk finder: one threshold level of pylinac's find_features sweep.
//
// Replaces, per threshold level, the body of the `while cutoff <= imax` loop of
// pylinac/metrics/utils.py:128-180 after `binary = sample > cutoff; label(connectivity=1)`:
//   segmentation.clear_border(labels)            -> labels whose bbox touches the frame are skipped
//   measure.regionprops(labels, intensity=sample) and the predicates of pylinac/metrics/features.py:
//     is_right_size_bb  area_filled / dpmm^2 in (max(pi (r-t)^2, 2), pi (r+t)^2)
//     is_round          filled_area / bbox_area in (0.8, 1.2) * pi/4
//     is_right_circumference  perimeter / dpmm in (2 pi (r-t), 2 pi (r+t))
//     is_symmetric      bbox width vs height
//     is_solid          area / convex_area > 0.9
//   Point(weighted_centroid[1], weighted_centroid[0]) of every region passing ALL predicates,
//   de-duplicated against the points of EARLIER levels (min_separation).
// scikit-image 0.18.3 semantics, restated and pinned in oracle/pylinac_oracle.py:
//   filled_area   holes = non-region pixels of the bbox crop not 8-connected to the crop border
//   perimeter     4-connected erosion border, 3x3 code convolution, weights 1 / sqrt2 / (1+sqrt2)/2
//   convex_area   lattice points inside or on the hull of the mid-edge points (+-0.5) of the
//                 region's hull vertices (exact doubled-integer arithmetic)
//   weighted_centroid  first raw moments of sample*mask over the bbox crop / zeroth moment + bbox
// One workgroup per window: cheap necessary conditions on (area, bbox) select the candidate labels,
// each candidate's bbox crop is then analysed in LDS by the whole workgroup.
#include "pl_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxCrop = 160;          // bbox side limit for the LDS crop analysis
constexpr int kMaxHullPts = 8 * kMaxCrop;
constexpr int kMaxOut = 8;

struct FeatureParams {
  double dpmm, radius_mm, tol_mm, min_sep_px;   // field mode: radius_mm = field width, min_sep_px = field height
  int max_number;
  int border;                                   // field mode: clear_border band (buffer_size + 1)
};

__device__ __forceinline__ long long cross2(int ax, int ay, int bx, int by, int cx, int cy) {
  return (long long)(bx - ax) * (cy - ay) - (long long)(by - ay) * (cx - ax);
}

// FIELD = false: the BB finder above.  FIELD = true: one level of GlobalSizedFieldLocator.calculate
// (pylinac/metrics/image.py:817-897): 8-connected labels of `sample > cutoff` whose bbox keeps clear of the
// border band (segmentation.clear_border(buffer_size=3) before labelling), predicates
// is_right_square_perimeter and is_right_area_square (metrics/features.py:69-101), UNWEIGHTED centroid,
// de-duplication radius max(equivalent_diameter of this level's hits) / dpmm.
template <bool FIELD>
__global__ void __launch_bounds__(kThreads)
features_level_kernel(const double* __restrict__ sample, const int32_t* __restrict__ labels,
                      const int32_t* __restrict__ nlabels, const double* __restrict__ stats, int max_labels,
                      int h, int w, FeatureParams prm, int level, int32_t* __restrict__ done,
                      int32_t* __restrict__ out_count,
                      double* __restrict__ out_xy, int32_t* __restrict__ out_level, int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_cand[32];
  __shared__ int s_ncand;
  __shared__ int s_hx[kMaxHullPts], s_hy[kMaxHullPts];     // hull candidate points (doubled coordinates)
  __shared__ int s_hull_x[kMaxHullPts + 1], s_hull_y[kMaxHullPts + 1];  // monotone-chain stack
  __shared__ int s_nh;
  __shared__ int s_cnt[4];                                  // holes changed flag / n1 / n2 / n3
  __shared__ double s_red[3][kThreads / PL_WAVE];
  __shared__ double s_hit[32][3];                           // field mode: x, y, equivalent diameter of this level's hits
  __shared__ int s_nhit;
  __shared__ int s_inside;
  const int64_t img = blockIdx.x;
  if (done[img]) return;
  const int nl = nlabels[img] < max_labels ? nlabels[img] : max_labels;
  if (nlabels[img] > max_labels && threadIdx.x == 0) status[img] = 1;   // more components than the table holds
  const double* st = stats + img * (int64_t)max_labels * 10;
  const int32_t* lab = labels + img * (int64_t)h * w;
  const double* smp = sample + img * (int64_t)h * w;
  const double dp2 = prm.dpmm * prm.dpmm;
  const double pi = 3.141592653589793;
  const double larger = pi * ((prm.radius_mm + prm.tol_mm) * (prm.radius_mm + prm.tol_mm));
  double smaller = pi * ((prm.radius_mm - prm.tol_mm) * (prm.radius_mm - prm.tol_mm));
  if (!(smaller > 2.0)) smaller = 2.0;                      // max((pi*(r-t)**2, 2))
  if (threadIdx.x == 0) { s_ncand = 0; s_nhit = 0; }
  __syncthreads();
  // ---- candidate labels: necessary conditions from (area, bbox) only -----------------------------
  for (int k = threadIdx.x; k < nl; k += kThreads) {
    const double* s = st + k * 10;
    const double area = s[0];
    if (area < 1.0) continue;
    const int r0 = (int)s[1], c0 = (int)s[2], r1 = (int)s[3], c1 = (int)s[4];
    const double bbox_area = (double)(r1 - r0) * (double)(c1 - c0);
    if constexpr (FIELD) {
      const int b = prm.border;
      if (r0 < b || c0 < b || r1 > h - b || c1 > w - b) continue;    // clear_border(buffer_size = b - 1)
      const double lo_a = (prm.radius_mm - prm.tol_mm) * (prm.min_sep_px - prm.tol_mm);
      const double hi_a = (prm.radius_mm + prm.tol_mm) * (prm.min_sep_px + prm.tol_mm);
      if (!(area / dp2 < hi_a)) continue;                              // filled_area >= area
      if (!(bbox_area / dp2 > lo_a)) continue;                         // filled_area <= bbox_area
    } else {
      if (r0 == 0 || c0 == 0 || r1 == h || c1 == w) continue;          // clear_border
      if (!(area / dp2 < larger)) continue;                            // filled_area >= area
      if (!(bbox_area / dp2 > smaller)) continue;                      // filled_area <= bbox_area
      const double y = (double)(r1 - r0), x = (double)(c1 - c0);       // is_symmetric (features.py:7-14)
      const double hi = (y * 1.05 > y + 3) ? y * 1.05 : y + 3, lo = (y * 0.95 < y - 3) ? y * 0.95 : y - 3;
      if (x > hi || x < lo) continue;
      if (!(area / bbox_area < pi / 4 * 1.2)) continue;                // is_round upper bound needs filled >= area
    }
    const int slot = atomicAdd(&s_ncand, 1);
    if (slot < 32) s_cand[slot] = k;
  }
  __syncthreads();
  int ncand = s_ncand;
  if (ncand > 32) { ncand = 32; if (threadIdx.x == 0) status[img] = 2; }
  // process candidates in label order (the reference iterates regions in label order)
  if (threadIdx.x == 0)
    for (int a = 1; a < ncand; ++a) { int v = s_cand[a], b = a - 1; while (b >= 0 && s_cand[b] > v) { s_cand[b + 1] = s_cand[b]; --b; } s_cand[b + 1] = v; }
  __syncthreads();

  unsigned char* m = smem;                       // [ch][cw] region mask of the crop
  for (int ci = 0; ci < ncand; ++ci) {
    const int k = s_cand[ci];
    const double* s = st + k * 10;
    const int r0 = (int)s[1], c0 = (int)s[2], r1 = (int)s[3], c1 = (int)s[4];
    const int ch = r1 - r0, cw = c1 - c0;
    if (ch > kMaxCrop || cw > kMaxCrop) { if (threadIdx.x == 0) status[img] = 3; continue; }
    unsigned char* reach = m + ch * cw;          // [ch][cw] flood-fill state
    unsigned char* bord = reach + ch * cw;       // [ch][cw] 4-connected erosion border
    const int npx = ch * cw;
    const int label = k + 1;
    for (int e = threadIdx.x; e < npx; e += kThreads) {
      const int r = e / cw, c = e % cw;
      m[e] = (lab[(int64_t)(r0 + r) * w + c0 + c] == label) ? 1 : 0;
    }
    __syncthreads();
    // ---- filled_area: non-region pixels reachable (8-conn) from the crop border are NOT holes -----
    for (int e = threadIdx.x; e < npx; e += kThreads) {
      const int r = e / cw, c = e % cw;
      const bool edge = (r == 0 || c == 0 || r == ch - 1 || c == cw - 1);
      reach[e] = (!m[e] && edge) ? 1 : 0;
      bool b = false;
      if (m[e]) b = (r == 0 || !m[e - cw]) || (r == ch - 1 || !m[e + cw]) || (c == 0 || !m[e - 1]) || (c == cw - 1 || !m[e + 1]);
      bord[e] = b ? 1 : 0;
    }
    __syncthreads();
    for (;;) {
      int changed = 0;
      for (int e = threadIdx.x; e < npx; e += kThreads) {
        if (m[e] || reach[e]) continue;
        const int r = e / cw, c = e % cw;
        bool hit = false;
        for (int dr = -1; dr <= 1 && !hit; ++dr)
          for (int dc = -1; dc <= 1; ++dc) {
            const int rr = r + dr, cc = c + dc;
            if ((dr | dc) == 0 || rr < 0 || cc < 0 || rr >= ch || cc >= cw) continue;
            if (reach[rr * cw + cc]) { hit = true; break; }
          }
        if (hit) { reach[e] = 1; changed = 1; }
      }
      if (!__syncthreads_or(changed)) break;
    }
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    // ---- holes, perimeter codes, weighted moments -------------------------------------------------
    int holes = 0, n1 = 0, n2 = 0, n3 = 0;
    double w0 = 0.0, wr = 0.0, wc = 0.0;
    for (int e = threadIdx.x; e < npx; e += kThreads) {
      const int r = e / cw, c = e % cw;
      if (!m[e] && !reach[e]) ++holes;
      if (bord[e]) {
        auto B = [&](int rr, int cc) { return (rr < 0 || cc < 0 || rr >= ch || cc >= cw) ? 0 : (int)bord[rr * cw + cc]; };
        const int code = 1 + 2 * (B(r - 1, c) + B(r + 1, c) + B(r, c - 1) + B(r, c + 1)) +
                         10 * (B(r - 1, c - 1) + B(r - 1, c + 1) + B(r + 1, c - 1) + B(r + 1, c + 1));
        if (code == 5 || code == 7 || code == 15 || code == 17 || code == 25 || code == 27) ++n1;
        else if (code == 21 || code == 33) ++n2;
        else if (code == 13 || code == 23) ++n3;
      }
      if constexpr (!FIELD) {
        if (m[e]) {
          const double v = smp[(int64_t)(r0 + r) * w + c0 + c];
          w0 += v; wr += v * (double)r; wc += v * (double)c;
        }
      }
    }
    auto addi = [](int a, int b) { return a + b; };
    auto addd = [](double a, double b) { return a + b; };
    holes = pl_wave_reduce(holes, addi); n1 = pl_wave_reduce(n1, addi); n2 = pl_wave_reduce(n2, addi); n3 = pl_wave_reduce(n3, addi);
    w0 = pl_wave_reduce(w0, addd); wr = pl_wave_reduce(wr, addd); wc = pl_wave_reduce(wc, addd);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
      atomicAdd(&s_cnt[0], holes); atomicAdd(&s_cnt[1], n1); atomicAdd(&s_cnt[2], n2); atomicAdd(&s_cnt[3], n3);
      s_red[0][wv] = w0; s_red[1][wv] = wr; s_red[2][wv] = wc;
    }
    if constexpr (!FIELD) {
      // ---- convex hull candidates: mid-edge points of the row-extreme pixels -------------------------
      for (int r = threadIdx.x; r < ch; r += kThreads) {
        int cl = -1, cr = -1;
        for (int c = 0; c < cw; ++c) if (m[r * cw + c]) { if (cl < 0) cl = c; cr = c; }
        int* px = s_hx + r * 8; int* py = s_hy + r * 8;
        // a crop row always holds at least one region pixel?  no (concave shapes): mark unused slots
        for (int q = 0; q < 8; ++q) { px[q] = 0x7fffffff; py[q] = 0; }
        if (cl >= 0) {
          const int xs[2] = {cl, cr};
          for (int q = 0; q < 2; ++q) {
            const int X = 2 * r, Y = 2 * xs[q];
            px[4 * q + 0] = X;     py[4 * q + 0] = Y - 1;
            px[4 * q + 1] = X;     py[4 * q + 1] = Y + 1;
            px[4 * q + 2] = X - 1; py[4 * q + 2] = Y;
            px[4 * q + 3] = X + 1; py[4 * q + 3] = Y;
          }
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        // sort the <= 8*ch points by (x, y) (insertion sort: points arrive nearly sorted), then Andrew's chain
        const int np = ch * 8;
        for (int a = 1; a < np; ++a) {
          const int vx = s_hx[a], vy = s_hy[a];
          int b = a - 1;
          while (b >= 0 && (s_hx[b] > vx || (s_hx[b] == vx && s_hy[b] > vy))) { s_hx[b + 1] = s_hx[b]; s_hy[b + 1] = s_hy[b]; --b; }
          s_hx[b + 1] = vx; s_hy[b + 1] = vy;
        }
        int n = 0;
        while (n < np && s_hx[n] != 0x7fffffff) ++n;
        // drop duplicates
        int u = 0;
        for (int a = 0; a < n; ++a) if (a == 0 || s_hx[a] != s_hx[a - 1] || s_hy[a] != s_hy[a - 1]) { s_hx[u] = s_hx[a]; s_hy[u] = s_hy[a]; ++u; }
        n = u;
        int kk = 0;
        for (int a = 0; a < n; ++a) {           // lower chain
          while (kk >= 2 && cross2(s_hull_x[kk - 2], s_hull_y[kk - 2], s_hull_x[kk - 1], s_hull_y[kk - 1], s_hx[a], s_hy[a]) <= 0) --kk;
          s_hull_x[kk] = s_hx[a]; s_hull_y[kk] = s_hy[a]; ++kk;
        }
        const int lower = kk + 1;
        for (int a = n - 2; a >= 0; --a) {      // upper chain
          while (kk >= lower && cross2(s_hull_x[kk - 2], s_hull_y[kk - 2], s_hull_x[kk - 1], s_hull_y[kk - 1], s_hx[a], s_hy[a]) <= 0) --kk;
          s_hull_x[kk] = s_hx[a]; s_hull_y[kk] = s_hy[a]; ++kk;
        }
        s_nh = kk - 1;                          // last point == first point
      }
      __syncthreads();
      const int nh = s_nh;
      int inside = 0;
      for (int e = threadIdx.x; e < npx; e += kThreads) {
        const int X = 2 * (e / cw), Y = 2 * (e % cw);
        bool in = true;
        for (int a = 0; a < nh && in; ++a) {
          const int b = (a + 1 == nh) ? 0 : a + 1;
          in = cross2(s_hull_x[a], s_hull_y[a], s_hull_x[b], s_hull_y[b], X, Y) >= 0;
        }
        inside += in ? 1 : 0;
      }
      inside = pl_wave_reduce(inside, addi);
      if (threadIdx.x == 0) s_inside = 0;
      __syncthreads();
      if (lane == 0) atomicAdd(&s_inside, inside);
      __syncthreads();
    } else {
      __syncthreads();   // s_cnt / s_red complete
    }
    // ---- predicates (pylinac/metrics/features.py) and output ---------------------------------------
    if (threadIdx.x == 0) {
      const double area = s[0];
      const double filled = area + (double)s_cnt[0];
      const double perim = ((double)s_cnt[1] * 1.0 + (double)s_cnt[2] * 1.4142135623730951) +
                           (double)s_cnt[3] * ((1 + 1.4142135623730951) / 2);
      const double per_mm = perim / prm.dpmm;
      if constexpr (FIELD) {
        const double fw = prm.radius_mm, fh = prm.min_sep_px, ft = prm.tol_mm;
        const double upper = 1.20 * 2 * (fw + ft) + 2 * (fh + ft);   // precedence as written, features.py:75-77
        const double lower = 2 * (fw - ft) + 2 * (fh - ft);
        bool ok = upper > per_mm && per_mm > lower;                                      // is_right_square_perimeter
        const double field_area = filled / dp2;
        ok = ok && ((fw - ft) * (fh - ft) < field_area && field_area < (fw + ft) * (fh + ft));  // is_right_area_square
        if (ok && s_nhit < 32) {
          s_hit[s_nhit][0] = s[6] / area;                 // centroid[1] = mean column
          s_hit[s_nhit][1] = s[5] / area;                 // centroid[0] = mean row
          s_hit[s_nhit][2] = sqrt(4.0 * area / pi);       // equivalent_diameter_area
          ++s_nhit;
        }
      } else {
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
          for (int q = 0; q < kThreads / PL_WAVE; ++q) { m0 += s_red[0][q]; mr += s_red[1][q]; mc += s_red[2][q]; }
          const double py = mr / m0 + (double)r0, px = mc / m0 + (double)c0;
          // de-duplicate against every point accepted so far, INCLUDING this level's (metrics/utils.py:28-36:
          // `combined_points` aliases `original_points`, so the list being iterated grows; candidates come in
          // label order, which is the order regionprops yields them in)
          bool keep = true;
          for (int q = 0; q < out_count[img]; ++q) {
            const double dx = px - out_xy[(img * kMaxOut + q) * 2], dy = py - out_xy[(img * kMaxOut + q) * 2 + 1];
            if (sqrt(dx * dx + dy * dy) < prm.min_sep_px) { keep = false; break; }
          }
          if (keep) {
            const int slot = out_count[img];
            if (slot < kMaxOut) {
              out_xy[(img * kMaxOut + slot) * 2] = px;
              out_xy[(img * kMaxOut + slot) * 2 + 1] = py;
              out_count[img] = slot + 1;
              if (out_level[img] < 0) out_level[img] = level;
            } else {
              status[img] = 4;
            }
          }
        }
      }
    }
    __syncthreads();
  }
  if constexpr (FIELD) {
    if (threadIdx.x == 0 && s_nhit > 0) {
      double sep = 0.0;                                    // max(r.equivalent_diameter_area) / dpmm (image.py:876-879)
      for (int q = 0; q < s_nhit; ++q) sep = s_hit[q][2] > sep ? s_hit[q][2] : sep;
      sep /= prm.dpmm;
      for (int a = 0; a < s_nhit; ++a) {
        bool keep = true;
        for (int q = 0; q < out_count[img]; ++q) {
          const double dx = s_hit[a][0] - out_xy[(img * kMaxOut + q) * 2];
          const double dy = s_hit[a][1] - out_xy[(img * kMaxOut + q) * 2 + 1];
          if (sqrt(dx * dx + dy * dy) < sep) { keep = false; break; }
        }
        if (!keep) continue;
        const int slot = out_count[img];
        if (slot < kMaxOut) {
          out_xy[(img * kMaxOut + slot) * 2] = s_hit[a][0];
          out_xy[(img * kMaxOut + slot) * 2 + 1] = s_hit[a][1];
          out_count[img] = slot + 1;
          if (out_level[img] < 0) out_level[img] = level;
        } else {
          status[img] = 4;
        }
      }
    }
  }
  if (threadIdx.x == 0 && out_count[img] >= prm.max_number) done[img] = 1;
}

}  // namespace

extern "C" int pl_features_level(const double* d_sample, const int32_t* d_labels, const int32_t* d_nlabels,
                                 const double* d_stats, int max_labels, int64_t n, int h, int w, double dpmm,
                                 double radius_mm, double tol_mm, double min_sep_px, int max_number, int level,
                                 int32_t* d_done, int32_t* d_count, double* d_xy,
                                 int32_t* d_level, int32_t* d_status, void* stream) {
  PL_REQUIRE(d_sample && d_labels && d_nlabels && d_stats && d_done && d_count && d_xy && d_level &&
                 d_status, "null pointer");
  PL_REQUIRE(n >= 0 && n <= 0x7fffffffLL && h > 0 && w > 0 && max_labels > 0 && max_number > 0, "bad arguments");
  PL_REQUIRE(dpmm > 0 && radius_mm > 0, "bad physical parameters");
  if (n == 0) return PL_OK;
  FeatureParams prm{dpmm, radius_mm, tol_mm, min_sep_px, max_number, 0};
  const size_t lds = (size_t)3 * kMaxCrop * kMaxCrop;
  static std::atomic<bool> attr{false};
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)features_level_kernel<false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { pl_set_error("pl_features_level: LDS attribute: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
    attr = true;
  }
  hipLaunchKernelGGL(features_level_kernel<false>, dim3((unsigned)n), dim3(kThreads), lds, (hipStream_t)stream,
                     d_sample, d_labels, d_nlabels, d_stats, max_labels, h, w, prm, level, d_done, d_count, d_xy,
                     d_level, d_status);
  return pl_check_launch("pl_features_level");
}

extern "C" int pl_fields_level(const int32_t* d_labels, const int32_t* d_nlabels, const double* d_stats,
                               int max_labels, int64_t n, int h, int w, double dpmm, double field_width_mm,
                               double field_height_mm, double field_tol_mm, int buffer_size, int max_number,
                               int level, int32_t* d_done, int32_t* d_count, double* d_xy, int32_t* d_level,
                               int32_t* d_status, void* stream) {
  PL_REQUIRE(d_labels && d_nlabels && d_stats && d_done && d_count && d_xy && d_level && d_status, "null pointer");
  PL_REQUIRE(n >= 0 && n <= 0x7fffffffLL && h > 0 && w > 0 && max_labels > 0 && max_number > 0, "bad arguments");
  PL_REQUIRE(dpmm > 0 && field_width_mm > 0 && field_height_mm > 0 && buffer_size >= 0, "bad physical parameters");
  if (n == 0) return PL_OK;
  FeatureParams prm{dpmm, field_width_mm, field_tol_mm, field_height_mm, max_number, buffer_size + 1};
  const size_t lds = (size_t)3 * kMaxCrop * kMaxCrop;
  static std::atomic<bool> attr{false};
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)features_level_kernel<true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { pl_set_error("pl_fields_level: LDS attribute: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
    attr = true;
  }
  hipLaunchKernelGGL(features_level_kernel<true>, dim3((unsigned)n), dim3(kThreads), lds, (hipStream_t)stream,
                     (const double*)nullptr, d_labels, d_nlabels, d_stats, max_labels, h, w, prm, level, d_done,
                     d_count, d_xy, d_level, d_status);
  return pl_check_launch("pl_fields_level");
}
