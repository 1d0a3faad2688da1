// CatPhan slice localisation primitives.
//
// Replaces, with the parameters pylinac passes (pylinac/ct.py:381-425 Slice.phantom_roi and
// :3315-3348 get_regions; scikit-image 0.18.3 + scipy semantics):
//   pl_scharr        skimage.filters.scharr(float image): two scipy.ndimage.convolve calls with the
//                    3x3 kernels edge (x) smooth/16 (mode='reflect'; non-zero taps accumulated from 0
//                    in raster order of the flipped kernel), squared, summed, sqrt, / sqrt(2)
//   pl_clip          np.clip(array, lo, hi)                                   (ct.py:400)
//   pl_hist_uniform  np.histogram(values, bins=256) as used by threshold_otsu on float data:
//                    numpy's exact edge-corrected bin assignment, optional pixel mask (draw.disk)
//   pl_compare       edges > thres  (strict)                                   (ct.py:3341)
//   pl_clear_border  skimage.segmentation.clear_border(bw, buffer_size): 8-connected components
//                    that touch the (buffer_size+1)-wide frame border band are removed
//   pl_region_stats  skimage.measure.regionprops raw sums per label: area, bbox, coordinate sums,
//                    intensity-weighted sums (centroid / weighted_centroid / area / bbox)
#include "pl_common.h"

namespace {

constexpr int kThreads = 256;

template <typename T>
__global__ void __launch_bounds__(kThreads)
scharr_kernel(const T* __restrict__ in, double* __restrict__ out, int64_t total, int h, int w) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= total) return;
  const int c = (int)(g % w);
  const int64_t t = g / w;
  const int r = (int)(t % h);
  const T* f = in + (t / h) * (size_t)h * w;
  const int rm = pl_reflect(r - 1, h), rp = pl_reflect(r + 1, h);
  const int cm = pl_reflect(c - 1, w), cp = pl_reflect(c + 1, w);
  auto at = [&](int rr, int cc) { return (double)f[(size_t)rr * w + cc]; };
  const double a = 0.1875, b = 0.625;  // 3/16, 10/16
  // edge along axis 0: flipped kernel rows (-1: -3 -10 -3 ; +1: 3 10 3) / 16, raster order
  double s0 = 0.0;
  s0 = s0 + at(rm, cm) * -a; s0 = s0 + at(rm, c) * -b; s0 = s0 + at(rm, cp) * -a;
  s0 = s0 + at(rp, cm) * a;  s0 = s0 + at(rp, c) * b;  s0 = s0 + at(rp, cp) * a;
  // edge along axis 1: flipped kernel (-3 0 3 ; -10 0 10 ; -3 0 3) / 16, raster order
  double s1 = 0.0;
  s1 = s1 + at(rm, cm) * -a; s1 = s1 + at(rm, cp) * a;
  s1 = s1 + at(r, cm) * -b;  s1 = s1 + at(r, cp) * b;
  s1 = s1 + at(rp, cm) * -a; s1 = s1 + at(rp, cp) * a;
  double o = 0.0;
  o = o + s0 * s0;
  o = o + s1 * s1;
  out[g] = sqrt(o) / 1.4142135623730951;  // np.sqrt(output) / np.sqrt(ndim)
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
clip_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t total, double lo, double hi) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= total) return;
  const T v = in[g];
  const T l = (T)lo, u = (T)hi;
  out[g] = v < l ? l : (v > u ? u : v);
}

// numpy's uniform-bin histogram: estimate the index, then correct it against the actual edges
__global__ void __launch_bounds__(kThreads)
hist_uniform_kernel(const double* __restrict__ in, const uint8_t* __restrict__ mask, int64_t per_frame,
                    const double* __restrict__ edges /* [n][nbins+1] */, int nbins,
                    uint32_t* __restrict__ counts /* [n][nbins], zeroed */) {
  extern __shared__ unsigned lbins[];
  const int64_t frame = blockIdx.y;
  const double* e = edges + frame * (nbins + 1);
  for (int i = threadIdx.x; i < nbins; i += kThreads) lbins[i] = 0;
  __syncthreads();
  const double first = e[0], last = e[nbins];
  const double denom = last - first;
  const double* src = in + frame * per_frame;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < per_frame; i += (int64_t)gridDim.x * kThreads) {
    if (mask && !mask[i]) continue;
    const double v = src[i];
    if (!(v >= first && v <= last)) continue;
    int idx = (int)(((v - first) / denom) * (double)nbins);
    if (idx == nbins) idx -= 1;
    if (idx < 0) idx = 0;
    if (v < e[idx]) idx -= 1;
    else if (v >= e[idx + 1] && idx != nbins - 1) idx += 1;
    atomicAdd(&lbins[idx], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nbins; i += kThreads)
    if (lbins[i]) atomicAdd(&counts[frame * nbins + i], lbins[i]);
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
compare_kernel(const T* __restrict__ in, int64_t total, int64_t per_frame, const double* __restrict__ thr,
               int thr_stride, int op, uint8_t* __restrict__ out) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= total) return;
  const double v = (double)in[g], t = thr[(g / per_frame) * thr_stride];
  const bool r = op == 0 ? (v >= t) : op == 1 ? (v > t) : op == 2 ? (v <= t) : (v < t);
  out[g] = r ? 1 : 0;
}

// ---- clear_border: flag the roots of components that own a pixel in the border band -------------
__global__ void band_flag_kernel(const int* __restrict__ Lall, int64_t total, int h, int w, int ext,
                                 uint8_t* __restrict__ flags) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= total) return;
  const int64_t per_frame = (int64_t)h * w;
  const int i = (int)(g % per_frame);
  const int r = i / w, c = i % w;
  if (r < ext || r >= h - ext || c < ext || c >= w - ext) {
    const int root = Lall[g];
    if (root >= 0) flags[(g / per_frame) * per_frame + root] = 1;
  }
}
__global__ void clear_apply_kernel(const uint8_t* __restrict__ mask, const int* __restrict__ Lall,
                                   const uint8_t* __restrict__ flags, int64_t total, int64_t per_frame,
                                   uint8_t* __restrict__ out) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= total) return;
  const int root = Lall[g];
  out[g] = (root >= 0 && !flags[(g / per_frame) * per_frame + root]) ? (mask[g] ? 1 : 0) : 0;
}

// ---- region sums -------------------------------------------------------------------------------
// stats layout per label (float64 x 10): area, rmin, cmin, rmax+1, cmax+1 (skimage bbox is half-open),
// sum r, sum c, sum w, sum w*r, sum w*c.   Integer quantities are accumulated exactly in a uint64
// side buffer and converted at the end; the weighted sums use float64 atomics (order-dependent in the
// last bits: the reference's moments are compared at 1e-9 relative, SURVEY asks 1e-5).
__global__ void region_init_kernel(unsigned long long* __restrict__ isum, double* __restrict__ wsum, int64_t rows) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= rows) return;
  isum[g * 7 + 0] = 0;                      // area
  isum[g * 7 + 1] = 0xffffffffffffffffull;  // rmin
  isum[g * 7 + 2] = 0xffffffffffffffffull;  // cmin
  isum[g * 7 + 3] = 0;                      // rmax
  isum[g * 7 + 4] = 0;                      // cmax
  isum[g * 7 + 5] = 0;                      // sum r
  isum[g * 7 + 6] = 0;                      // sum c
  wsum[g * 3 + 0] = 0.0; wsum[g * 3 + 1] = 0.0; wsum[g * 3 + 2] = 0.0;
}

// Lanes of a wave that carry the same (frame, label) are reduced across the wave first and ONE lane issues the atomics:
// a CatPhan slice is a single filled disk of 125 000 pixels, and round 1's one-atomic-set-per-pixel version spent 140 ms
// per 200 slices serialising on that label's ten accumulators (profiles/r02d_*): 93 % of the slice localisation.
__global__ void __launch_bounds__(kThreads)
region_accum_kernel(const int32_t* __restrict__ labels, const double* __restrict__ intensity, int64_t total,
                    int h, int w, int max_labels, unsigned long long* __restrict__ isum,
                    double* __restrict__ wsum, int32_t* __restrict__ overflow) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const int64_t per_frame = (int64_t)h * w;
  int lab = 0;
  int64_t frame = 0;
  unsigned long long r = 0, c = 0;
  double v = 0.0;
  if (g < total) {
    lab = labels[g];
    frame = g / per_frame;
    const int i = (int)(g - frame * per_frame);
    r = (unsigned long long)(i / w);
    c = (unsigned long long)(i % w);
    if (lab > max_labels) { overflow[frame] = 1; lab = 0; }
    if (lab < 0) lab = 0;
    if (lab > 0 && intensity) v = intensity[g];
  }
  const long long key = lab > 0 ? frame * (long long)max_labels + (lab - 1) : -1;
  unsigned long long pending = __ballot(key >= 0);
  const int lane = threadIdx.x & 63;
  auto addu = [](unsigned long long a, unsigned long long b) { return a + b; };
  auto minu = [](unsigned long long a, unsigned long long b) { return a < b ? a : b; };
  auto maxu = [](unsigned long long a, unsigned long long b) { return a > b ? a : b; };
  while (pending) {
    const int leader = __builtin_ctzll(pending);
    const long long k = __shfl(key, leader, 64);
    const bool mine = key == k;
    const unsigned long long grp = __ballot(mine);
    const unsigned long long cnt = pl_wave_reduce(mine ? 1ull : 0ull, addu);
    const unsigned long long rmin = pl_wave_reduce(mine ? r : ~0ull, minu), cmin = pl_wave_reduce(mine ? c : ~0ull, minu);
    const unsigned long long rmax = pl_wave_reduce(mine ? r : 0ull, maxu), cmax = pl_wave_reduce(mine ? c : 0ull, maxu);
    const unsigned long long sr = pl_wave_reduce(mine ? r : 0ull, addu), sc = pl_wave_reduce(mine ? c : 0ull, addu);
    if (lane == leader) {
      unsigned long long* s = isum + k * 7;
      atomicAdd(&s[0], cnt);
      atomicMin(&s[1], rmin); atomicMin(&s[2], cmin);
      atomicMax(&s[3], rmax); atomicMax(&s[4], cmax);
      atomicAdd(&s[5], sr); atomicAdd(&s[6], sc);
    }
    if (intensity) {
      // the weighted sums keep float64 atomics: their summation order was never defined (compared at 1e-9, see above);
      // the wave's partial sums are formed in a fixed butterfly order, only the order of the waves' atomics varies
      auto addd = [](double a, double b) { return a + b; };
      const double w0 = pl_wave_reduce(mine ? v : 0.0, addd);
      const double w1 = pl_wave_reduce(mine ? v * (double)r : 0.0, addd);
      const double w2 = pl_wave_reduce(mine ? v * (double)c : 0.0, addd);
      if (lane == leader) {
        double* ws = wsum + k * 3;
        atomicAdd(&ws[0], w0); atomicAdd(&ws[1], w1); atomicAdd(&ws[2], w2);
      }
    }
    pending &= ~grp;
  }
}

// Same sums, EIGHT consecutive pixels of a row per lane (w % 8 == 0): a lane whose eight labels agree folds them in
// registers first, so a region interior costs one wave reduction per 512 pixels instead of one per 64; waves that hold a
// lane with mixed labels present their pixels one position at a time through the same group reduction.
__global__ void __launch_bounds__(kThreads)
region_accum8_kernel(const int32_t* __restrict__ labels, const double* __restrict__ intensity, int64_t total8,
                     int h, int w, int max_labels, unsigned long long* __restrict__ isum,
                     double* __restrict__ wsum, int32_t* __restrict__ overflow) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;   // index of the 8-pixel group
  const int64_t per_frame = (int64_t)h * w;
  const int lane = threadIdx.x & 63;
  int lab[8];
  double val[8];
  int64_t frame = 0;
  unsigned long long r = 0, c0 = 0;
  const bool in_range = g < total8;
  if (in_range) {
    const int64_t p = g * 8;
    frame = p / per_frame;
    const int i = (int)(p - frame * per_frame);
    r = (unsigned long long)(i / w);
    c0 = (unsigned long long)(i % w);
    const int4 a = *reinterpret_cast<const int4*>(labels + p), b = *reinterpret_cast<const int4*>(labels + p + 4);
    lab[0] = a.x; lab[1] = a.y; lab[2] = a.z; lab[3] = a.w; lab[4] = b.x; lab[5] = b.y; lab[6] = b.z; lab[7] = b.w;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (lab[j] > max_labels) { overflow[frame] = 1; lab[j] = 0; }
      if (lab[j] < 0) lab[j] = 0;
      val[j] = (lab[j] > 0 && intensity) ? intensity[p + j] : 0.0;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) { lab[j] = 0; val[j] = 0.0; }
  }
  auto addu = [](unsigned long long a, unsigned long long b) { return a + b; };
  auto minu = [](unsigned long long a, unsigned long long b) { return a < b ? a : b; };
  auto maxu = [](unsigned long long a, unsigned long long b) { return a > b ? a : b; };
  auto addd = [](double a, double b) { return a + b; };
  // one group reduction: every lane contributes (key, cnt, cmin, cmax, sum c, w0, w1 = sum v*r, w2 = sum v*c) of ONE row r
  auto reduce_groups = [&](long long key, unsigned long long cnt, unsigned long long cmin, unsigned long long cmax,
                           unsigned long long sc, double w0, double w2) {
    unsigned long long pending = __ballot(key >= 0);
    while (pending) {
      const int leader = __builtin_ctzll(pending);
      const long long k = __shfl(key, leader, 64);
      const bool mine = key == k;
      const unsigned long long grp = __ballot(mine);
      const unsigned long long n = pl_wave_reduce(mine ? cnt : 0ull, addu);
      const unsigned long long rmin = pl_wave_reduce(mine ? r : ~0ull, minu), rmax = pl_wave_reduce(mine ? r : 0ull, maxu);
      const unsigned long long qmin = pl_wave_reduce(mine ? cmin : ~0ull, minu), qmax = pl_wave_reduce(mine ? cmax : 0ull, maxu);
      const unsigned long long sr = pl_wave_reduce(mine ? r * cnt : 0ull, addu), scs = pl_wave_reduce(mine ? sc : 0ull, addu);
      if (lane == leader) {
        unsigned long long* s = isum + k * 7;
        atomicAdd(&s[0], n);
        atomicMin(&s[1], rmin); atomicMin(&s[2], qmin);
        atomicMax(&s[3], rmax); atomicMax(&s[4], qmax);
        atomicAdd(&s[5], sr); atomicAdd(&s[6], scs);
      }
      if (intensity) {
        const double a0 = pl_wave_reduce(mine ? w0 : 0.0, addd);
        const double a1 = pl_wave_reduce(mine ? w0 * (double)r : 0.0, addd);
        const double a2 = pl_wave_reduce(mine ? w2 : 0.0, addd);
        if (lane == leader) {
          double* ws = wsum + k * 3;
          atomicAdd(&ws[0], a0); atomicAdd(&ws[1], a1); atomicAdd(&ws[2], a2);
        }
      }
      pending &= ~grp;
    }
  };
  bool uni = true;
#pragma unroll
  for (int j = 1; j < 8; ++j) uni = uni && lab[j] == lab[0];
  if (__ballot(!uni) == 0) {          // every lane's eight pixels agree (region interiors, background)
    const long long key = lab[0] > 0 ? frame * (long long)max_labels + (lab[0] - 1) : -1;
    double w0 = 0.0, w2 = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { w0 += val[j]; w2 += val[j] * (double)(c0 + j); }
    reduce_groups(key, 8ull, c0, c0 + 7, 8 * c0 + 28, w0, w2);
  } else {
    // some lane holds more than one label (region borders: two lanes of a row through a disk).  Rounds: every lane folds,
    // in registers, the pixels that carry the label of its first pixel not yet handled, and the wave reduces those folds;
    // lanes of one label are done after the first round, a border lane after one round per label it holds -- two rounds
    // for almost every wave, where presenting the pixels one position at a time took eight
    unsigned todo = 0xffu;                           // bit j: pixel j not yet accumulated (label 0 = background: never)
#pragma unroll
    for (int j = 0; j < 8; ++j) todo &= lab[j] > 0 ? 0xffu : ~(1u << j);
    while (__ballot(todo != 0u) != 0ull) {           // wave-uniform
      const int jf = todo ? __builtin_ctz(todo) : 0;
      int cur = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) cur = (todo && j == jf) ? lab[j] : cur;
      unsigned long long cnt = 0, cmin = ~0ull, cmax = 0, sc = 0;
      double w0 = 0.0, w2 = 0.0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool take = ((todo >> j) & 1u) && lab[j] == cur;
        const unsigned long long c = c0 + j;
        cnt += take ? 1ull : 0ull;
        cmin = take && c < cmin ? c : cmin;
        cmax = take && c > cmax ? c : cmax;
        sc += take ? c : 0ull;
        w0 += take ? val[j] : 0.0;
        w2 += take ? val[j] * (double)c : 0.0;
        todo &= take ? ~(1u << j) : 0xffu;
      }
      const long long key = cnt ? frame * (long long)max_labels + (cur - 1) : -1;
      reduce_groups(key, cnt, cmin, cmax, sc, w0, w2);
    }
  }
}

__global__ void region_finish_kernel(const unsigned long long* __restrict__ isum, const double* __restrict__ wsum,
                                     int64_t rows, double* __restrict__ out) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= rows) return;
  const unsigned long long* s = isum + g * 7;
  double* o = out + g * 10;
  const bool empty = s[0] == 0;
  o[0] = (double)s[0];
  o[1] = empty ? 0.0 : (double)s[1];
  o[2] = empty ? 0.0 : (double)s[2];
  o[3] = empty ? 0.0 : (double)(s[3] + 1);
  o[4] = empty ? 0.0 : (double)(s[4] + 1);
  o[5] = (double)s[5];
  o[6] = (double)s[6];
  o[7] = wsum[g * 3 + 0]; o[8] = wsum[g * 3 + 1]; o[9] = wsum[g * 3 + 2];
}

// ---- exact raw second moments per label (regionprops.orientation / eccentricity / centroid) ------
// m00, m10 = sum r, m01 = sum c, m20 = sum r*r, m02 = sum c*c, m11 = sum r*c in IMAGE coordinates, exact uint64.
// The central moments are translation invariant and are formed from these on the host in exact integer
// arithmetic (n*m20 - m10*m10, ...), so symmetric regions give mu20 == mu02 and mu11 == 0 exactly.
// A wave usually sits inside one region: the lanes that share the first pending label are summed with
// cross-lane butterflies and one lane issues the six atomics.
__global__ void __launch_bounds__(kThreads)
region_moments_kernel(const int32_t* __restrict__ labels, int64_t total, int h, int w, int max_labels,
                      unsigned long long* __restrict__ mom, int32_t* __restrict__ overflow) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const int64_t per_frame = (int64_t)h * w;
  int lab = 0;
  int64_t frame = 0;
  unsigned long long r = 0, c = 0;
  if (g < total) {
    lab = labels[g];
    frame = g / per_frame;
    const int i = (int)(g - frame * per_frame);
    r = (unsigned long long)(i / w);
    c = (unsigned long long)(i % w);
    if (lab > max_labels) { overflow[frame] = 1; lab = 0; }
    if (lab < 0) lab = 0;
  }
  // key = (frame, label): a wave can straddle two frames
  long long key = lab > 0 ? frame * (long long)max_labels + (lab - 1) : -1;
  unsigned long long pending = __ballot(key >= 0);
  const int lane = threadIdx.x & 63;
  while (pending) {
    const int leader = __builtin_ctzll(pending);
    const long long k = __shfl(key, leader, 64);
    const bool mine = key == k;
    const unsigned long long grp = __ballot(mine);
    unsigned long long v[6];
    v[0] = mine ? 1ull : 0ull;
    v[1] = mine ? r : 0ull;
    v[2] = mine ? c : 0ull;
    v[3] = mine ? r * r : 0ull;
    v[4] = mine ? c * c : 0ull;
    v[5] = mine ? r * c : 0ull;
#pragma unroll
    for (int q = 0; q < 6; ++q) v[q] = pl_wave_reduce(v[q], [](unsigned long long a, unsigned long long b) { return a + b; });
    if (lane == leader) {
      unsigned long long* s = mom + k * 6;
#pragma unroll
      for (int q = 0; q < 6; ++q) atomicAdd(&s[q], v[q]);
    }
    pending &= ~grp;
  }
}

// ---- float-image Otsu without the host: np.linspace edges, skimage's class statistics ------------------------------------
// np.linspace(lo, hi, nbins + 1) (numpy/core/function_base.py): step = (hi - lo) / nbins; y = arange * step + lo (two
// roundings: -ffp-contract=off keeps them apart); the last edge is set to hi.  step == 0 takes numpy's other branch
// (y = arange / div * delta + lo): every edge equals lo.
__global__ void linspace_edges_kernel(const double* __restrict__ lo, const double* __restrict__ hi, int nbins, int64_t n,
                                      double* __restrict__ edges /* [n][nbins+1] */) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= n * (nbins + 1)) return;
  const int64_t f = g / (nbins + 1);
  const int i = (int)(g % (nbins + 1));
  const double a = lo[f], b = hi[f];
  const double delta = b - a;
  const double step = delta / (double)nbins;
  double y = step == 0.0 ? ((double)i / (double)nbins) * delta + a : (double)i * step + a;
  if (i == nbins) y = b;
  edges[g] = y;
}

// skimage 0.18.3 threshold_otsu on a ready 256-bin histogram (filters/thresholding.py): centres = (e[:-1] + e[1:]) / 2,
// weight1 = cumsum(counts), weight2 = cumsum(counts[::-1])[::-1], mean1 = cumsum(counts * centres) / weight1,
// mean2 = (cumsum((counts * centres)[::-1]) / weight2[::-1])[::-1], variance12 = weight1[:-1] * weight2[1:] *
// (mean1[:-1] - mean2[1:]) ** 2, first arg-max.  np.cumsum is a sequential float64 loop, so the two cumulative sums of
// counts * centres stay sequential chains (one lane of wave 0 runs the forward one, one lane of wave 1 the reversed one,
// side by side); everything else -- centres, products, the four divisions per bin, the variances -- is one bin per lane.
// (Round 1-3: one lane per FRAME with 4 KB of scratch and 512 dependent divisions: 0.37 ms for 320 slices.)
// lo == hi (constant selection): the threshold is that value.  out = threshold * scale (the reference uses 0.8).
__global__ void __launch_bounds__(256)
otsu_counts_kernel(const uint32_t* __restrict__ counts, const double* __restrict__ edges, int64_t n, double scale,
                   double* __restrict__ thr, double* __restrict__ raw) {
  constexpr int NB = 256;
  __shared__ double s_c[NB], s_p[NB], s_w1[NB], s_s1[NB], s_w2[NB], s_m2[NB], s_var[NB];
  const int64_t f = blockIdx.x;
  const int i = threadIdx.x;
  const double* e = edges + f * (NB + 1);
  const double e_first = e[0], e_last = e[NB];
  const double ci = (double)counts[f * NB + i];
  const double centre = (e[i] + e[i + 1]) / 2.0;
  s_c[i] = ci;
  s_p[i] = ci * centre;
  __syncthreads();
  if (i == 0) {
    double w1 = 0.0, s1 = 0.0;
    for (int k = 0; k < NB; ++k) {
      w1 = w1 + s_c[k];
      s1 = s1 + s_p[k];
      s_w1[k] = w1;
      s_s1[k] = s1;
    }
  } else if (i == PL_WAVE) {
    double aw = 0.0, am = 0.0;
    for (int k = NB - 1; k >= 0; --k) {
      aw = aw + s_c[k];
      am = am + s_p[k];
      s_w2[k] = aw;
      s_m2[k] = am;
    }
  }
  __syncthreads();
  if (i < NB - 1) {
    const double mean1 = s_s1[i] / s_w1[i];
    const double mean2 = s_m2[i + 1] / s_w2[i + 1];
    const double d = mean1 - mean2;
    s_var[i] = (s_w1[i] * s_w2[i + 1]) * (d * d);
  }
  __syncthreads();
  if (i == 0) {
    double otsu;
    if (e_first == e_last) {
      otsu = e_first;
    } else {
      // np.argmax: first maximum; a NaN (empty leading class) is treated as the maximum by numpy
      double best = s_var[0];
      int best_i = 0;
      for (int k = 1; k < NB - 1; ++k) {
        const double var = s_var[k];
        if (var > best || (var != var && best == best)) { best = var; best_i = k; }
      }
      otsu = (e[best_i] + e[best_i + 1]) / 2.0;
    }
    if (raw) raw[f] = otsu;
    thr[f] = otsu * scale;
  }
}

// np.max / np.mean over slices s-k .. s+k of a stack (combine_surrounding_slices, pylinac/ct.py:3351-3386): mode 0 max
// (dtype kept), mode 1 mean (float64, numpy's pairwise-free sequential sum over the 2k+1 slices).  Slices whose window
// leaves the stack are marked invalid by the caller; here the window is clamped so every output is defined.
template <typename T>
__global__ void __launch_bounds__(kThreads)
combine_slices_kernel(const T* __restrict__ in, int64_t n, int64_t per_frame, int k, int mode, int64_t per_volume,
                      T* __restrict__ out_max, double* __restrict__ out_mean) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= n * per_frame) return;
  const int64_t s = g / per_frame, p = g % per_frame;
  // The reference indexes dicomstack[s] for s in range(z - k, z + k + 1) (pylinac/ct.py:3375-3378): a NEGATIVE index wraps to
  // the end of the stack (Python list indexing); an index beyond the last slice raises IndexError -- those slices are
  // reported as invalid by the host layer (ct.ctp528_profiles_batch), here they reuse the last slice so that nothing is
  // read out of bounds.  Elements are visited in the reference's order (np.dstack order: z - k first).
  const int64_t v0 = (s / per_volume) * per_volume, z = s - v0;                    // the slice's own volume
  auto src = [&](int d) {
    int64_t q = z + d;
    if (q < 0) q += per_volume;
    if (q < 0) q = 0;                                                              // k > slices per volume: clamp
    if (q >= per_volume) q = per_volume - 1;
    return in[(v0 + q) * per_frame + p];
  };
  if (mode == 0) {
    T m = src(-k);
    for (int d = -k + 1; d <= k; ++d) { const T v = src(d); m = v > m ? v : m; }
    out_max[g] = m;
  } else {
    double acc = 0.0;
    for (int d = -k; d <= k; ++d) acc += (double)src(d);
    out_mean[g] = acc / (double)(2 * k + 1);
  }
}

}  // namespace

int pl_ccl_roots(const uint8_t* mask, int invert, int64_t n, int h, int w, int conn, int* L, hipStream_t st);

#define PL_CT_TOTAL()                                                                       \
  const int64_t per_frame = (int64_t)h * w, total = n * per_frame;                           \
  PL_REQUIRE(n >= 0 && h > 0 && w > 0, "bad shape");                                        \
  PL_REQUIRE(per_frame <= 0x7fffffffLL && pl_cdiv(total, kThreads) <= 0x7fffffffLL, "too large"); \
  if (n == 0) return PL_OK;                                                                 \
  hipStream_t st = (hipStream_t)stream;                                                     \
  const unsigned blocks = (unsigned)pl_cdiv(total, kThreads);

extern "C" int pl_scharr(const void* in, double* out, int dtype, int64_t n, int h, int w, void* stream) {
  PL_REQUIRE(in && out, "null pointer");
  PL_CT_TOTAL();
  PL_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL(scharr_kernel<T>, dim3(blocks), dim3(kThreads), 0, st, (const T*)in, out, total,
                                       h, w));
  return pl_check_launch("pl_scharr");
}

extern "C" int pl_clip(const void* in, void* out, int dtype, int64_t n, int64_t count, double lo, double hi,
                       void* stream) {
  PL_REQUIRE(in && out, "null pointer");
  PL_REQUIRE(n >= 0 && count > 0 && lo <= hi, "bad arguments");
  if (n == 0) return PL_OK;
  const int64_t total = n * count;
  PL_REQUIRE(pl_cdiv(total, kThreads) <= 0x7fffffffLL, "too large");
  PL_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL(clip_kernel<T>, dim3((unsigned)pl_cdiv(total, kThreads)), dim3(kThreads), 0,
                                       (hipStream_t)stream, (const T*)in, (T*)out, total, lo, hi));
  return pl_check_launch("pl_clip");
}

extern "C" int pl_hist_uniform(const double* in, const uint8_t* d_mask, int64_t n, int64_t count,
                               const double* d_edges, int nbins, uint32_t* d_counts, void* stream) {
  PL_REQUIRE(in && d_edges && d_counts, "null pointer");
  PL_REQUIRE(n >= 0 && n <= 65535 && count > 0 && nbins > 0 && nbins <= 8192, "bad arguments");
  if (n == 0) return PL_OK;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(d_counts, 0, (size_t)n * nbins * sizeof(uint32_t), st);
  if (e != hipSuccess) { pl_set_error("pl_hist_uniform: memset: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
  int bx = (int)pl_cdiv(count, (int64_t)kThreads * 16);
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(hist_uniform_kernel, dim3((unsigned)bx, (unsigned)n), dim3(kThreads), nbins * sizeof(unsigned), st,
                     in, d_mask, count, d_edges, nbins, d_counts);
  return pl_check_launch("pl_hist_uniform");
}

extern "C" int pl_compare(const void* in, int dtype, int64_t n, int64_t count, const double* d_thr,
                          int thr_stride, int op, uint8_t* d_out, void* stream) {
  PL_REQUIRE(in && d_thr && d_out, "null pointer");
  PL_REQUIRE(n >= 0 && count > 0 && op >= 0 && op <= 3 && (thr_stride == 0 || thr_stride == 1), "bad arguments");
  if (n == 0) return PL_OK;
  const int64_t total = n * count;
  PL_REQUIRE(pl_cdiv(total, kThreads) <= 0x7fffffffLL, "too large");
  PL_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL(compare_kernel<T>, dim3((unsigned)pl_cdiv(total, kThreads)), dim3(kThreads), 0,
                                       (hipStream_t)stream, (const T*)in, total, count, d_thr, thr_stride, op, d_out));
  return pl_check_launch("pl_compare");
}

extern "C" int pl_clear_border(const uint8_t* d_mask, uint8_t* d_out, int64_t n, int h, int w, int buffer_size,
                               int32_t* d_work, uint8_t* d_flags, void* stream) {
  PL_REQUIRE(d_mask && d_out && d_work && d_flags, "null pointer");
  PL_REQUIRE(buffer_size >= 0, "negative buffer");
  PL_CT_TOTAL();
  if (int rc = pl_ccl_roots(d_mask, 0, n, h, w, 8, d_work, st)) return rc;  // skimage label default: full connectivity
  hipError_t e = hipMemsetAsync(d_flags, 0, (size_t)total, st);
  if (e != hipSuccess) { pl_set_error("pl_clear_border: memset: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
  hipLaunchKernelGGL(band_flag_kernel, dim3(blocks), dim3(kThreads), 0, st, d_work, total, h, w, buffer_size + 1, d_flags);
  hipLaunchKernelGGL(clear_apply_kernel, dim3(blocks), dim3(kThreads), 0, st, d_mask, d_work, d_flags, total, per_frame,
                     d_out);
  return pl_check_launch("pl_clear_border");
}

extern "C" int pl_region_stats(const int32_t* d_labels, const double* d_intensity, int64_t n, int h, int w,
                               int max_labels, unsigned long long* d_isum, double* d_wsum, double* d_stats,
                               int32_t* d_overflow, void* stream) {
  PL_REQUIRE(d_labels && d_isum && d_wsum && d_stats && d_overflow, "null pointer");
  PL_REQUIRE(max_labels > 0, "max_labels must be positive");
  PL_CT_TOTAL();
  const int64_t rows = n * max_labels;
  hipError_t e = hipMemsetAsync(d_overflow, 0, (size_t)n * sizeof(int32_t), st);
  if (e != hipSuccess) { pl_set_error("pl_region_stats: memset: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
  hipLaunchKernelGGL(region_init_kernel, dim3((unsigned)pl_cdiv(rows, kThreads)), dim3(kThreads), 0, st, d_isum, d_wsum,
                     rows);
  if ((w & 7) == 0 && (reinterpret_cast<uintptr_t>(d_labels) & 15) == 0)
    hipLaunchKernelGGL(region_accum8_kernel, dim3((unsigned)pl_cdiv(total / 8, kThreads)), dim3(kThreads), 0, st, d_labels,
                       d_intensity, total / 8, h, w, max_labels, d_isum, d_wsum, d_overflow);
  else
    hipLaunchKernelGGL(region_accum_kernel, dim3(blocks), dim3(kThreads), 0, st, d_labels, d_intensity, total, h, w,
                       max_labels, d_isum, d_wsum, d_overflow);
  hipLaunchKernelGGL(region_finish_kernel, dim3((unsigned)pl_cdiv(rows, kThreads)), dim3(kThreads), 0, st, d_isum,
                     d_wsum, rows, d_stats);
  return pl_check_launch("pl_region_stats");
}

extern "C" int pl_region_moments(const int32_t* d_labels, int64_t n, int h, int w, int max_labels,
                                 unsigned long long* d_mom, int32_t* d_overflow, void* stream) {
  PL_REQUIRE(d_labels && d_mom && d_overflow, "null pointer");
  PL_REQUIRE(max_labels > 0, "max_labels must be positive");
  PL_CT_TOTAL();
  hipError_t e = hipMemsetAsync(d_overflow, 0, (size_t)n * sizeof(int32_t), st);
  if (e == hipSuccess) e = hipMemsetAsync(d_mom, 0, (size_t)n * max_labels * 6 * sizeof(unsigned long long), st);
  if (e != hipSuccess) { pl_set_error("pl_region_moments: memset: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
  hipLaunchKernelGGL(region_moments_kernel, dim3(blocks), dim3(kThreads), 0, st, d_labels, total, h, w, max_labels,
                     d_mom, d_overflow);
  return pl_check_launch("pl_region_moments");
}

/* np.linspace(lo_i, hi_i, nbins + 1) per frame -> d_edges float64 [n][nbins + 1] (the bin edges np.histogram builds for
 * `bins = nbins` over the range of the selected pixels; feeds pl_hist_uniform) */
extern "C" int pl_linspace_edges(const double* d_lo, const double* d_hi, int nbins, int64_t n, double* d_edges,
                                 void* stream) {
  PL_REQUIRE(d_lo && d_hi && d_edges, "null pointer");
  PL_REQUIRE(n >= 0 && nbins > 0 && nbins <= 8192, "bad arguments");
  if (n == 0) return PL_OK;
  const int64_t total = n * (nbins + 1);
  hipLaunchKernelGGL(linspace_edges_kernel, dim3((unsigned)pl_cdiv(total, kThreads)), dim3(kThreads), 0,
                     (hipStream_t)stream, d_lo, d_hi, nbins, n, d_edges);
  return pl_check_launch("pl_linspace_edges");
}

/* skimage 0.18.3 threshold_otsu from a 256-bin float histogram (counts uint32 [n][256], edges float64 [n][257]) ->
 * d_thr[i] = otsu_i * scale, d_raw[i] = otsu_i (optional).  pylinac/ct.py:3338-3340 uses scale 0.8. */
extern "C" int pl_otsu_from_counts(const uint32_t* d_counts, const double* d_edges, int nbins, int64_t n, double scale,
                                   double* d_thr, double* d_raw, void* stream) {
  PL_REQUIRE(d_counts && d_edges && d_thr, "null pointer");
  PL_REQUIRE(n >= 0 && nbins == 256, "256 bins (skimage's default for float images)");
  if (n == 0) return PL_OK;
  PL_REQUIRE(n <= 0x7fffffffLL, "batch too large");
  hipLaunchKernelGGL(otsu_counts_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, d_counts, d_edges, n, scale,
                     d_thr, d_raw);
  return pl_check_launch("pl_otsu_from_counts");
}

/* combine_surrounding_slices (pylinac/ct.py:3351-3386) for EVERY slice of a stack [n][count] of whole volumes of
 * slices_per_volume slices: mode 0 = np.max (d_out has the input dtype), mode 1 = np.mean (d_out float64).  The window
 * s-k .. s+k is clamped to the slice's own volume. */
extern "C" int pl_combine_slices(const void* in, void* d_out, int dtype, int64_t n, int64_t count, int plusminus, int mode,
                                 int64_t slices_per_volume, void* stream) {
  PL_REQUIRE(in && d_out && in != d_out, "null or aliased pointers");
  PL_REQUIRE(n >= 0 && count > 0 && plusminus >= 0 && (mode == 0 || mode == 1), "bad arguments");
  PL_REQUIRE(slices_per_volume > 0 && n % slices_per_volume == 0, "n must be a whole number of volumes");
  if (n == 0) return PL_OK;
  const int64_t total = n * count;
  PL_REQUIRE(pl_cdiv(total, kThreads) <= 0x7fffffffLL, "too large");
  PL_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL(combine_slices_kernel<T>, dim3((unsigned)pl_cdiv(total, kThreads)), dim3(kThreads), 0,
                                       (hipStream_t)stream, (const T*)in, n, count, plusminus, mode, slices_per_volume,
                                       (T*)d_out, (double*)d_out));
  return pl_check_launch("pl_combine_slices");
}
