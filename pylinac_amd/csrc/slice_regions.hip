// CatPhan slice localisation, the labelling half: ONE workgroup per slice takes the thresholded edge image through
//   bw = edges > thr                                              pylinac/ct.py:3340
//   bw = segmentation.clear_border(bw, buffer_size)              pylinac/ct.py:3342  (8-connected components that own a
//                                                                 pixel of the (buffer_size + 1)-wide border band go)
//   bw = ndimage.binary_fill_holes(bw)                           pylinac/ct.py:3344  (4-connected background components
//                                                                 that do not reach the frame border become foreground)
//   labeled = measure.label(bw);  regionprops(labeled)           pylinac/ct.py:3345-3347  (8-connected; labels numbered in
//                                                                 raster order of each component's first pixel)
// and leaves the region table (area, bbox, coordinate sums) the phantom ROI selection reads (pylinac/ct.py:398-409).
//
// Round 1-3 form: compare + three global union-find labellings on int32 label planes (ccl.hip) + flag / apply passes +
// an atomics region table = ~30 launches and 54 % of config #5's kernel time, all of it pointer chasing through HBM/L2.
// Here the mask of a slice is a BIT PLANE in LDS (512 x 512 -> 32 KB) and a component is a set of horizontal RUNS:
//   * runs come from the bit rows by word arithmetic (starts = w & ~(w << 1 | carry), ends likewise), numbered in raster order;
//   * a union-find over RUN ids in LDS (links point to the smaller id) merges each run with the runs of the row above that it
//     overlaps (by one more pixel either side for 8-connectivity), found by bisection in that row's run list;
//   * clear_border / fill_holes flag the roots whose runs touch the band / the border and clear / set the runs' bits;
//   * the final labelling ranks the roots (a root is the first run of its component in raster order, so the rank IS
//     scikit-image's label) and adds each run's length, row and column sums and extent to its label's row of the table.
// A slice with more runs than the LDS list holds reports status 1 and the host layer repeats it on the general path.
#include "pl_common.h"
#include "edge_exact.h"

// Phase stopwatch (-DPL_SR_TIMING, development builds only: scripts/r06_sr_phases.sh): s_memtime totals per phase, summed over
// the workgroups of every launch since the last read (pl_debug_sr_timing).  0 plane build, 1-3 the three labellings
// (unused), 4 flag + paint passes, 5 region table + results; inside the labellings (clear_border 8-13, fill_holes 14-19,
// final 20-25): runs per row, prefix, run extraction, first-above links, pointer jumping, unions + final flattening.
#ifndef PL_SR_TIMING
#define PL_SR_TIMING 0
#endif
#if PL_SR_TIMING
__device__ unsigned long long pl_sr_dbg[32];
#define SR_STAMP(k) do { if (threadIdx.x == 0) { const long long t_ = clock64(); sr_tacc[k] += t_ - sr_tlast; sr_tlast = t_; } } while (0)
#define SR_TIMING_ARGS , long long* sr_tacc, long long& sr_tlast, int sr_base
#define SR_TIMING_PASS(b) , sr_tacc, sr_tlast, b
#else
#define SR_STAMP(k) do { } while (0)
#define SR_TIMING_ARGS
#define SR_TIMING_PASS(b)
#endif

namespace {

constexpr int kSrThreads = 512;
constexpr int kSrMaxRuns = 3072;
constexpr int kSrMaxLabels = 128;

typedef unsigned long long u64;

__device__ __forceinline__ unsigned sr_find(const unsigned* parent, unsigned i) {
  unsigned p = parent[i];
  while (p != i) {
    i = p;
    p = parent[i];
  }
  return i;
}

__device__ __forceinline__ void sr_unite(unsigned* parent, unsigned a, unsigned b) {
  bool done;
  do {
    a = sr_find(parent, a);
    b = sr_find(parent, b);
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

struct SrLds {
  u64* plane;              // [h][ww] bit rows, bit b of word j = column 64 j + b; bits at or beyond w are 0
  unsigned short* run_s;   // first column of a run
  unsigned short* run_e;   // last column (inclusive)
  unsigned short* run_r;   // row
  unsigned short* aux;     // per-run flag, then the label of a root
  unsigned* parent;
  int* row_off;            // [h + 1] first run id of a row
};

// Runs of the plane (inverted inside the frame when `invert`), numbered in raster order; 4- or 8-connected components over
// them; afterwards parent[id] is the root (smallest run id) of id's component.  -> number of runs, or -1 when they do not fit.
__device__ int sr_label_runs(const SrLds& L, int h, int w, int ww, bool invert, bool conn8, int* s_total SR_TIMING_ARGS) {
  const int tid = threadIdx.x;
  const u64 tail = (w & 63) ? ((1ull << (w & 63)) - 1ull) : ~0ull;
  auto word = [&](int r, int j) -> u64 {
    u64 v = L.plane[r * ww + j];
    if (invert) v = ~v & (j == ww - 1 ? tail : ~0ull);
    return v;
  };
  // runs per row
  for (int r = tid; r < h; r += kSrThreads) {
    int cnt = 0;
    u64 carry = 0;
    for (int j = 0; j < ww; ++j) {
      const u64 v = word(r, j);
      cnt += __popcll(v & ~((v << 1) | carry));
      carry = v >> 63;
    }
    L.row_off[r + 1] = cnt;
  }
  if (tid == 0) L.row_off[0] = 0;
  __syncthreads();
  SR_STAMP(sr_base + 8);
  // inclusive prefix over row_off[1 .. h]: every wave scans 64 rows, the waves' totals chain through LDS (round 6; before,
  // the first wave walked the rows 64 at a time while seven waves waited: a fifth of a labelling)
  {
    __shared__ int s_wtot[kSrThreads / PL_WAVE];
    const int lane = tid & 63, wv = tid >> 6;
    int base = 0;
    for (int c0 = 0; c0 < h; c0 += kSrThreads) {             // (block-uniform trip count)
      const int r = c0 + tid;
      int v = r < h ? L.row_off[r + 1] : 0;
#pragma unroll
      for (int o = 1; o < PL_WAVE; o <<= 1) {
        const int u = __shfl_up(v, o, PL_WAVE);
        if (lane >= o) v += u;
      }
      if (lane == PL_WAVE - 1) s_wtot[wv] = v;
      __syncthreads();
      int add = base, tot = 0;
#pragma unroll
      for (int k = 0; k < kSrThreads / PL_WAVE; ++k) {
        add += k < wv ? s_wtot[k] : 0;
        tot += s_wtot[k];
      }
      if (r < h) L.row_off[r + 1] = add + v;
      base += tot;
      __syncthreads();
    }
    if (tid == 0) *s_total = base;
  }
  __syncthreads();
  const int nruns = *s_total;
  SR_STAMP(sr_base + 9);
  if (nruns > kSrMaxRuns) return -1;
  // the runs themselves: the k-th start and the k-th end of a row belong together
  for (int r = tid; r < h; r += kSrThreads) {
    int ids = L.row_off[r], ide = ids;
    u64 carry = 0;
    u64 v = word(r, 0);
    for (int j = 0; j < ww; ++j) {
      const u64 nxt = j + 1 < ww ? word(r, j + 1) : 0ull;
      u64 st = v & ~((v << 1) | carry);
      u64 en = v & ~((v >> 1) | (nxt << 63));
      while (st) {
        const int b = __ffsll((long long)st) - 1;
        L.run_s[ids] = (unsigned short)(j * 64 + b);
        L.run_r[ids] = (unsigned short)r;
        L.parent[ids] = (unsigned)ids;
        L.aux[ids] = 0;
        ++ids;
        st &= st - 1;
      }
      while (en) {
        const int b = __ffsll((long long)en) - 1;
        L.run_e[ide++] = (unsigned short)(j * 64 + b);
        en &= en - 1;
      }
      carry = v >> 63;
      v = nxt;
    }
  }
  __syncthreads();
  SR_STAMP(sr_base + 10);
  // ---- components over the runs.  Rounds 1-3 united every run with every overlapping run of the row above through an
  // atomic union-find whose links point to the smaller id: a tall component (the phantom's outline is 450 rows of one or two
  // runs) became a 450-deep chain that every find walked link by link, three labellings per slice.  Now:
  //   1. every run points at the FIRST run of the row above that it overlaps (no atomics): a forest, each tree reaching up;
  //   2. pointer jumping flattens the forest in log2(depth) sweeps (a racing read sees an ancestor either way);
  //   3. only runs that overlap MORE than one run above -- where two branches meet -- go through the atomic union, now
  //      between roots; 4. one more flattening.  The root of a component is still its first run in raster order: that run
  //      has nothing above it, so it is a forest root, and unions keep the smaller id.
  const int reach = conn8 ? 1 : 0;
  for (int id = tid; id < nruns; id += kSrThreads) {
    const int r = L.run_r[id];
    unsigned first_above = (unsigned)id;
    bool more = false;
    if (r > 0) {
      const int s = (int)L.run_s[id] - reach, e = (int)L.run_e[id] + reach;
      int lo = L.row_off[r - 1];
      const int end = L.row_off[r];
      int hi = end;
      while (lo < hi) {                       // first run of the row above whose end is >= s
        const int mid = (lo + hi) >> 1;
        if ((int)L.run_e[mid] < s) lo = mid + 1; else hi = mid;
      }
      if (lo < end && (int)L.run_s[lo] <= e) {
        first_above = (unsigned)lo;
        more = lo + 1 < end && (int)L.run_s[lo + 1] <= e;
      }
    }
    L.parent[id] = first_above;
    L.aux[id] = more ? 1 : 0;
  }
  __syncthreads();
  SR_STAMP(sr_base + 11);
  for (;;) {
    int changed = 0;
    for (int id = tid; id < nruns; id += kSrThreads) {
      const unsigned p = L.parent[id];
      const unsigned pp = L.parent[p];
      if (pp != p) { L.parent[id] = pp; changed = 1; }
    }
    if (!__syncthreads_or(changed)) break;
  }
  SR_STAMP(sr_base + 12);
  for (int id = tid; id < nruns; id += kSrThreads) {
    if (!L.aux[id]) continue;
    const int r = L.run_r[id];
    const int e = (int)L.run_e[id] + reach;
    const int end = L.row_off[r];
    // the runs after the first overlapping one: L.parent[id] no longer names it, so find it again
    const int s = (int)L.run_s[id] - reach;
    int lo = L.row_off[r - 1], hi = end;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((int)L.run_e[mid] < s) lo = mid + 1; else hi = mid;
    }
    for (int k = lo + 1; k < end && (int)L.run_s[k] <= e; ++k) sr_unite(L.parent, (unsigned)id, (unsigned)k);
  }
  __syncthreads();
  for (int id = tid; id < nruns; id += kSrThreads) {
    const unsigned root = sr_find(L.parent, (unsigned)id);
    // writing a root's own entry never changes it; other entries only ever move closer to the root
    L.parent[id] = root;
    L.aux[id] = 0;
  }
  __syncthreads();
  SR_STAMP(sr_base + 13);
  return nruns;
}

__host__ __device__ inline size_t sr_lds_bytes(int h, int w) {
  const int ww = (w + 63) >> 6;
  return (size_t)h * ww * 8 + (size_t)kSrMaxRuns * 4 + (size_t)(h + 1 + ((h + 1) & 1)) * 4 + (size_t)kSrMaxRuns * 2 * 4;
}
__host__ __device__ inline size_t es_scratch_doubles(int rad) { return (size_t)(2 * rad + 1) * (2 * rad + 1) + (2 * rad + 1); }

// set (value = true) or clear the bits of run `id` in the plane
__device__ __forceinline__ void sr_paint(const SrLds& L, int ww, int id, bool value) {
  const int r = L.run_r[id], s = L.run_s[id], e = L.run_e[id];
  for (int j = s >> 6; j <= (e >> 6); ++j) {
    const int b0 = j == (s >> 6) ? (s & 63) : 0, b1 = j == (e >> 6) ? (e & 63) : 63;
    const u64 m = (b1 == 63 ? ~0ull : ((1ull << (b1 + 1)) - 1ull)) & ~((1ull << b0) - 1ull);
    if (value) atomicOr(&L.plane[r * ww + j], m); else atomicAnd(&L.plane[r * ww + j], ~m);
  }
}

// what the float32 form needs beside the plane: the slices and taps the plane was made from (pixels whose float32 neighbours
// lie on both sides of the threshold are recomputed exactly), and the phantom-ROI selection of Slice.phantom_roi
struct SrEdgeArgs {
  const void* raw;            // int16 / uint16 slices [n][h][w]
  int raw_is_signed;
  const double* wts;          // device float64 [2 rad + 1]
  int rad;
  double catphan_size;        // > 0 with roi
  const double* rawmax;       // [n] max of the raw Scharr magnitude (the "no edges" test)
  double* roi;                // [n][8] or NULL
  unsigned bracket;           // the plane lies within this many float32 bit patterns of the exact value (>= 1)
};

// The float32 plane -> bit plane for frames whose rows are WW whole 64-pixel words (w = 64 WW: 256, 512 and 1024 wide CT
// slices): a wave takes 32 / WW consecutive rows per trip, and every one of its 32 loads is `scalar row base + lane + a
// compile-time offset` -- no vector address arithmetic between the loads (r06e: the generic loop below spent seventeen
// instructions, one of them a quarter-rate multiply, between two loads, so a wave's 32 loads trickled out over a memory
// latency instead of being in flight together; the build ran at 3 TB/s on 16 waves per CU).  Marks undecided words in `und`
// exactly like the generic loop; -> whether any was seen.
template <int WW>
__device__ __forceinline__ bool sr_build_rows(const float* __restrict__ src, int h, int r0, unsigned f2, unsigned lowb, unsigned span,
                                              u64* __restrict__ plane, u64* __restrict__ und, int lane) {
  constexpr int R = 32 / WW, W = 64 * WW;
  float v[32];
#pragma unroll
  for (int u = 0; u < 32; ++u) {
    const int rr = u / WW, jj = u % WW;
    const int rl = r0 + rr < h ? r0 + rr : h - 1;          // wave-uniform: rows past the frame re-read its last row
    v[u] = src[(unsigned)rl * (unsigned)W + (unsigned)(jj * 64) + (unsigned)lane];
  }
  bool any_und = false;
#pragma unroll
  for (int u = 0; u < 32; ++u) {
    const int rr = u / WW, jj = u % WW;
    if (r0 + rr >= h) continue;                            // wave-uniform
    const unsigned vb = __float_as_uint(v[u]);
    const u64 m = __ballot(vb >= f2);
    const u64 todo = __ballot(((vb - lowb) < span) & (vb != 0u));
    if (lane == 0) {
      plane[(r0 + rr) * WW + jj] = m;
      und[u] = todo;
    }
    any_und |= todo != 0;
  }
  (void)R;
  return any_und;
}

template <typename T>
__global__ void __launch_bounds__(kSrThreads)
mask_regions_kernel(const T* __restrict__ in, const double* __restrict__ thr, int h, int w, int clear_ext, int fill,
                    int max_labels, double* __restrict__ table /* [n][max_labels][7] or NULL */, int32_t* __restrict__ count,
                    int32_t* __restrict__ status, uint8_t* __restrict__ out_mask /* optional [n][h][w] */, SrEdgeArgs ea) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ww = (w + 63) >> 6;
  SrLds L;
  L.plane = reinterpret_cast<u64*>(smem);
  L.parent = reinterpret_cast<unsigned*>(L.plane + (size_t)h * ww);
  L.row_off = reinterpret_cast<int*>(L.parent + kSrMaxRuns);
  L.run_s = reinterpret_cast<unsigned short*>(L.row_off + h + 1 + ((h + 1) & 1));
  L.run_e = L.run_s + kSrMaxRuns;
  L.run_r = L.run_e + kSrMaxRuns;
  L.aux = L.run_r + kSrMaxRuns;
  __shared__ int s_total, s_nlab;
  __shared__ unsigned t_area[kSrMaxLabels];
  __shared__ int t_r0[kSrMaxLabels], t_c0[kSrMaxLabels], t_r1[kSrMaxLabels], t_c1[kSrMaxLabels];
  __shared__ u64 t_sr[kSrMaxLabels], t_sc[kSrMaxLabels];

  const int64_t f = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#if PL_SR_TIMING
  long long sr_tacc[32] = {0}, sr_tlast = clock64();
#endif
  const T* src = in + f * (int64_t)h * w;
  const double t = thr ? thr[f] : 0.0;
  // ---- the bit plane: a wave turns 64 consecutive pixels of a row into one word; thirty-two words' loads are in flight per
  // wave (one load per word and iteration left every wave waiting out a full memory round trip per 256 bytes; the build is
  // a stream whose rate is bytes in flight / latency)
  const int nwords = h * ww;
  constexpr int U = sizeof(T) == 4 ? 32 : 16;
  // float32 planes: words with a pixel the float32 value cannot decide are noted here (the run tables, which nothing uses
  // before the labelling starts, lend the space) and settled after the unrolled body by the exact recomputation
  u64* und = reinterpret_cast<u64*>(L.parent) + (size_t)wv * U;
  double* scratch = reinterpret_cast<double*>(reinterpret_cast<u64*>(L.parent) + (size_t)(kSrThreads / PL_WAVE) * U) +
                    (size_t)wv * es_scratch_doubles(sizeof(T) == 4 ? ea.rad : 0);
  // Round 6 (phase stopwatch, profiles/r06b_sr_phases.txt: the build was 78 % of this kernel, and vector-instruction bound:
  // two float32 -> float64 conversions, two float64 compares and an integer division per pixel word).  A float32 plane is
  // now compared in its BIT domain.  The exact value lies between the two float32 neighbours of the stored value v (edge
  // values are >= 0, bit patterns order like the values): with a = the largest float32 <= t,
  //     prev(v) > t  <=>  bits(v) >= bits(a) + 2        -> foreground, decided
  //     next(v) > t >= prev(v)  <=>  bits(v) - bits(a) in {0, 1}   (v != 0: a stored 0 is an exact 0)   -> undecided
  // one v_cmp per mask, whose result IS the wave's ballot.  t < 0: everything is foreground; t NaN / beyond FLT_MAX: nothing.
  // With a bracket of B patterns (ea.bracket; 1 = a plane that stores RN32 of the exact value):
  //     foreground, decided:  bits(v) - B >= bits(a) + 1           <=>  bits(v) >= f2 = bits(a) + B + 1
  //     undecided:            bits(v) + B >= bits(a) + 1 otherwise  <=>  lowb = bits(a) + 1 - B <= bits(v) < f2
  unsigned f2 = 0xffffffffu, lowb = 0xffffffffu;          // "nothing is foreground, nothing undecided"
  if constexpr (sizeof(T) == 4) {
    const unsigned B = ea.bracket ? ea.bracket : 1u;
    if (t < 0.0) { f2 = 0u; lowb = 0u; }                   // everything is foreground
    else if (t < 3.0e38) {
      float a = (float)t;                                  // RN; step down when it rounded up
      if ((double)a > t) a = __uint_as_float(__float_as_uint(a) - 1u);
      const unsigned fa = __float_as_uint(a);
      f2 = fa + B + 1u;
      lowb = fa + 1u > B ? fa + 1u - B : 1u;
    }
  }
  const unsigned span = f2 - lowb;
  bool rows_done = false;
  if constexpr (sizeof(T) == 4) {
    if (w == ww * 64 && (ww == 4 || ww == 8 || ww == 16)) {
      rows_done = true;
      const int rows_per = 32 / ww;
      for (int r0 = wv * rows_per; r0 < h; r0 += (kSrThreads / PL_WAVE) * rows_per) {
        const float* fsrc = reinterpret_cast<const float*>(src);
        const bool any_und = ww == 8 ? sr_build_rows<8>(fsrc, h, r0, f2, lowb, span, L.plane, und, lane)
                             : (ww == 4 ? sr_build_rows<4>(fsrc, h, r0, f2, lowb, span, L.plane, und, lane)
                                        : sr_build_rows<16>(fsrc, h, r0, f2, lowb, span, L.plane, und, lane));
        if (any_und) {                                     // a handful of pixels per thousand slices
          pl_wave_sync();
          const int q0 = r0 * ww;
#pragma unroll 1
          for (int u = 0; u < U && q0 + u < nwords; ++u) {
            u64 todo = und[u];
            const int q = q0 + u, r = q / ww, j = q - r * ww;
            u64 add = 0;
            while (todo) {
              const int l = __builtin_ctzll(todo);
              todo &= todo - 1;
              const int64_t off = f * (int64_t)h * w;
              const double ev = ea.raw_is_signed
                                    ? es_exact_wave(static_cast<const short*>(ea.raw) + off, h, w, r, j * 64 + l, ea.wts, ea.rad, scratch)
                                    : es_exact_wave(static_cast<const unsigned short*>(ea.raw) + off, h, w, r, j * 64 + l, ea.wts, ea.rad, scratch);
              if (ev > t) add |= 1ull << l;
            }
            if (lane == 0 && add) L.plane[q] |= add;
          }
          pl_wave_sync();
        }
      }
    }
  }
  for (int q0 = wv * U; q0 < nwords && !rows_done; q0 += (kSrThreads / PL_WAVE) * U) {
    T v[U];
    bool inside[U];
    // (row, word) of the first word by ONE scalar division; the unrolled body steps them
    int r = __builtin_amdgcn_readfirstlane(q0 / ww), j = __builtin_amdgcn_readfirstlane(q0 - (q0 / ww) * ww);
    {
      int rr = r, jj = j;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool live = q0 + u < nwords;                 // wave-uniform
        const int c = jj * 64 + lane;
        inside[u] = live & (c < w);
        const int rl = live ? rr : h - 1, cl = c < w ? c : w - 1;
        v[u] = src[(unsigned)rl * (unsigned)w + (unsigned)cl];
        if (++jj == ww) { jj = 0; ++rr; }
      }
    }
    bool any_und = false;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = q0 + u;
      if (q >= nwords) continue;                           // wave-uniform
      u64 m;
      if constexpr (sizeof(T) == 4) {
        const unsigned vb = __float_as_uint(v[u]);
        m = __ballot(inside[u] & (vb >= f2));
        const u64 todo = __ballot(inside[u] & ((vb - lowb) < span) & (vb != 0u));
        if (lane == 0) und[u] = todo;
        any_und |= todo != 0;
      } else {
        m = __ballot(inside[u] & (thr ? ((double)v[u] > t) : (v[u] != (T)0)));
      }
      if (lane == 0) L.plane[q] = m;
    }
    if constexpr (sizeof(T) == 4) {
      if (any_und) {                                       // a handful of pixels per thousand slices
        pl_wave_sync();
#pragma unroll 1
        for (int u = 0; u < U && q0 + u < nwords; ++u) {
          u64 todo = und[u];
          const int q = q0 + u, r = q / ww, j = q - r * ww;
          u64 add = 0;
          while (todo) {
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int64_t off = f * (int64_t)h * w;
            const double ev = ea.raw_is_signed
                                  ? es_exact_wave(static_cast<const short*>(ea.raw) + off, h, w, r, j * 64 + l, ea.wts, ea.rad, scratch)
                                  : es_exact_wave(static_cast<const unsigned short*>(ea.raw) + off, h, w, r, j * 64 + l, ea.wts, ea.rad, scratch);
            if (ev > t) add |= 1ull << l;
          }
          if (lane == 0 && add) L.plane[q] |= add;
        }
        pl_wave_sync();
      }
    }
  }
  __syncthreads();
  SR_STAMP(0);
  int st = 0;
  // ---- clear_border: 8-connected components with a pixel in the border band.  A band without a single foreground pixel
  // (the usual CT slice: air all around the phantom and the couch) has nothing to clear: the labelling that would find
  // that out -- a third of this kernel's labelling work -- is skipped after one look at the band's words (round 6)
  bool band_empty = false;
  if (clear_ext > 0 && clear_ext <= 64 && w >= 2 * clear_ext) {
    const u64 low = clear_ext == 64 ? ~0ull : ((1ull << clear_ext) - 1ull);
    const int last_bits = w - (ww - 1) * 64;                 // valid bits of a row's last word (1 .. 64)
    int hit = 0;
    for (int r = tid; r < h; r += kSrThreads) {
      const u64* row = L.plane + (size_t)r * ww;
      if (r < clear_ext || r >= h - clear_ext) {
        for (int j = 0; j < ww; ++j) hit |= row[j] != 0ull;
      } else {
        hit |= (row[0] & low) != 0ull;
        // the last clear_ext columns: bits last_bits - clear_ext .. last_bits - 1 of the last word, and when they do not fit
        // there, the top bits of the word before it
        const int from = last_bits - clear_ext;
        if (from >= 0) {
          hit |= ((row[ww - 1] >> from) & low) != 0ull;
        } else {
          hit |= row[ww - 1] != 0ull;
          hit |= (row[ww - 2] >> (64 + from)) != 0ull;
        }
      }
    }
    band_empty = __syncthreads_or(hit) == 0;
  }
  if (clear_ext > 0 && !band_empty) {
    const int nr = sr_label_runs(L, h, w, ww, false, true, &s_total SR_TIMING_PASS(0));
    if (nr < 0) st = 1;
    else {
      for (int id = tid; id < nr; id += kSrThreads) {
        const int r = L.run_r[id], s = L.run_s[id], e = L.run_e[id];
        if (r < clear_ext || r >= h - clear_ext || s < clear_ext || e >= w - clear_ext) L.aux[L.parent[id]] = 1;
      }
      __syncthreads();
      for (int id = tid; id < nr; id += kSrThreads)
        if (L.aux[L.parent[id]]) sr_paint(L, ww, id, false);
      __syncthreads();
    }
    SR_STAMP(4);
  }
  // ---- binary_fill_holes: 4-connected background components away from the frame border
  if (fill && st == 0) {
    const int nr = sr_label_runs(L, h, w, ww, true, false, &s_total SR_TIMING_PASS(6));
    if (nr < 0) st = 1;
    else {
      for (int id = tid; id < nr; id += kSrThreads) {
        const int r = L.run_r[id], s = L.run_s[id], e = L.run_e[id];
        if (r == 0 || r == h - 1 || s == 0 || e == w - 1) L.aux[L.parent[id]] = 1;
      }
      __syncthreads();
      for (int id = tid; id < nr; id += kSrThreads)
        if (!L.aux[L.parent[id]]) sr_paint(L, ww, id, true);
      __syncthreads();
    }
    SR_STAMP(4);
  }
  // ---- label (8-connected) + region table
  int nlab = 0;
  if (st == 0) {
    const int nr = sr_label_runs(L, h, w, ww, false, true, &s_total SR_TIMING_PASS(12));
    if (nr < 0) st = 1;
    else {
      for (int k = tid; k < kSrMaxLabels; k += kSrThreads) {
        t_area[k] = 0; t_r0[k] = 0x7fffffff; t_c0[k] = 0x7fffffff; t_r1[k] = -1; t_c1[k] = -1; t_sr[k] = 0; t_sc[k] = 0;
      }
      // rank of the roots in run order = the label (0-based) ; aux[root] = min(rank, 0xffff)
      if (tid < PL_WAVE) {
        int base = 0;
        for (int i0 = 0; i0 < nr; i0 += PL_WAVE) {
          const int id = i0 + tid;
          const bool root = id < nr && L.parent[id] == (unsigned)id;
          const u64 b = __ballot(root);
          if (root) {
            const int k = base + __popcll(b & ((1ull << tid) - 1ull));
            L.aux[id] = (unsigned short)(k < 0xffff ? k : 0xffff);
          }
          base += __popcll(b);
        }
        if (tid == 0) s_nlab = base;
      }
      __syncthreads();
      nlab = s_nlab;
      for (int id = tid; id < nr; id += kSrThreads) {
        const int k = L.aux[L.parent[id]];
        if (k >= max_labels) continue;
        const int r = L.run_r[id], s = L.run_s[id], e = L.run_e[id];
        const unsigned len = (unsigned)(e - s + 1);
        atomicAdd(&t_area[k], len);
        atomicAdd(&t_sr[k], (u64)r * len);
        atomicAdd(&t_sc[k], (u64)(s + e) * len / 2);
        atomicMin(&t_r0[k], r); atomicMax(&t_r1[k], r);
        atomicMin(&t_c0[k], s); atomicMax(&t_c1[k], e);
      }
      __syncthreads();
    }
  }
  // ---- results
  if (table) {
    double* tab = table + f * (int64_t)max_labels * 7;
    for (int k = tid; k < max_labels; k += kSrThreads) {
      const bool have = st == 0 && k < nlab;
      tab[k * 7 + 0] = have ? (double)t_area[k] : 0.0;
      tab[k * 7 + 1] = have ? (double)t_r0[k] : 0.0;
      tab[k * 7 + 2] = have ? (double)t_c0[k] : 0.0;
      tab[k * 7 + 3] = have ? (double)(t_r1[k] + 1) : 0.0;      // half-open like regionprops' bbox
      tab[k * 7 + 4] = have ? (double)(t_c1[k] + 1) : 0.0;
      tab[k * 7 + 5] = have ? (double)t_sr[k] : 0.0;
      tab[k * 7 + 6] = have ? (double)t_sc[k] : 0.0;
    }
  }
  if (tid == 0) {
    count[f] = st == 0 ? nlab : 0;
    status[f] = st;
    if (ea.roi) {
      // Slice.phantom_roi (pylinac/ct.py:381-425): the region whose filled area (= area after binary_fill_holes) is closest
      // to the phantom's, first one on ties (sorted() is stable), must lie within a factor 1.3 of it.  status: 0 ok,
      // 1 no edges (np.max(edges) < 0.1), 2 no region, 3 wrong size, 4 more labels than the table holds, 5 = this kernel's
      // run list overflowed (the caller repeats the slice on the general path)
      double* o = ea.roi + f * 8;
      const double nan = __longlong_as_double(0x7ff8000000000000LL);
      const int num = nlab < max_labels ? nlab : max_labels;
      int best = 0;
      double best_d = __longlong_as_double(0x7ff0000000000000LL), fk = 0.0;
      for (int k = 0; k < num; ++k) {
        const double d = fabs((double)t_area[k] - ea.catphan_size);
        if (d < best_d) { best_d = d; best = k; }
      }
      if (num > 0) fk = (double)t_area[best];
      int code = 0;
      if (num > 0 && (ea.catphan_size * 1.3 < fk || fk < ea.catphan_size / 1.3)) code = 3;
      if (nlab > max_labels) code = 4;
      if (num < 1) code = 2;
      if (ea.rawmax[f] < 0.1) code = 1;
      if (st != 0) code = 5;
      o[0] = (double)code;
      const bool ok = code == 0;
      o[1] = ok ? (double)(best + 1) : nan;
      o[2] = ok ? fk : nan;
      o[3] = ok ? (double)t_sr[best] / fk : nan;
      o[4] = ok ? (double)t_sc[best] / fk : nan;
      o[5] = ok ? (double)t_r0[best] : nan;
      o[6] = ok ? (double)t_c0[best] : nan;
      o[7] = ok ? (double)(t_r1[best] + 1) : nan;
    }
  }
#if PL_SR_TIMING
  SR_STAMP(5);
  if (tid == 0) for (int k = 0; k < 32; ++k) atomicAdd(&pl_sr_dbg[k], (unsigned long long)sr_tacc[k]);
#endif
  if (out_mask && st == 0) {
    uint8_t* om = out_mask + f * (int64_t)h * w;
    for (int q = wv; q < nwords; q += kSrThreads / PL_WAVE) {
      const int r = q / ww, j = q - r * ww;
      const int c = j * 64 + lane;
      if (c < w) om[(int64_t)r * w + c] = (uint8_t)((L.plane[q] >> lane) & 1ull);
    }
  }
}

}  // namespace

extern "C" int pl_mask_regions_fits(int h, int w, int max_labels) {
  return h > 0 && w > 0 && h <= 4096 && w <= 4096 && max_labels > 0 && max_labels <= kSrMaxLabels &&
         sr_lds_bytes(h, w) <= 150 * 1024;
}

extern "C" int pl_mask_regions(const void* in, int dtype, const double* d_thr, int64_t n, int h, int w, int clear_border_ext,
                               int fill_holes, int max_labels, double* d_table, int32_t* d_count, int32_t* d_status,
                               uint8_t* d_out_mask, void* stream) {
  PL_REQUIRE(in && d_table && d_count && d_status, "null pointer");
  PL_REQUIRE(dtype == PL_F64 || dtype == PL_U8, "float64 frames with a threshold, or uint8 masks");
  PL_REQUIRE(dtype == PL_U8 || d_thr, "float64 frames need per-frame thresholds");
  PL_REQUIRE(n >= 0 && n <= 0x7fffffffLL && clear_border_ext >= 0, "bad arguments");
  PL_REQUIRE(pl_mask_regions_fits(h, w, max_labels), "frame or label table too large for the LDS form (pl_mask_regions_fits)");
  if (n == 0) return PL_OK;
  const size_t lds = sr_lds_bytes(h, w);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PL_F64) {
    static std::atomic<size_t> attr{0};
    if (lds > attr) {
      hipError_t e = hipFuncSetAttribute((const void*)mask_regions_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { pl_set_error("pl_mask_regions: LDS attribute: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
      attr = lds;
    }
    hipLaunchKernelGGL(mask_regions_kernel<double>, dim3((unsigned)n), dim3(kSrThreads), lds, st, (const double*)in, d_thr, h, w,
                       clear_border_ext, fill_holes, max_labels, d_table, d_count, d_status, d_out_mask, SrEdgeArgs{});
  } else {
    static std::atomic<size_t> attr{0};
    if (lds > attr) {
      hipError_t e = hipFuncSetAttribute((const void*)mask_regions_kernel<uint8_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { pl_set_error("pl_mask_regions: LDS attribute: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
      attr = lds;
    }
    hipLaunchKernelGGL(mask_regions_kernel<uint8_t>, dim3((unsigned)n), dim3(kSrThreads), lds, st, (const uint8_t*)in, nullptr, h, w,
                       clear_border_ext, fill_holes, max_labels, d_table, d_count, d_status, d_out_mask, SrEdgeArgs{});
  }
  return pl_check_launch("pl_mask_regions");
}

/* pl_mask_regions on the float32 plane of pl_edge_plane, with the phantom ROI chosen in the same launch */
extern "C" int pl_edge_regions_ex(const float* d_plane, const void* in_raw, int dtype, const double* d_weights, int radius,
                                  const double* d_thr, int64_t n, int h, int w, int clear_border_ext, int fill_holes, int max_labels,
                                  double* d_table, int32_t* d_count, int32_t* d_status, uint8_t* d_out_mask, double catphan_size,
                                  const double* d_rawmax, double* d_roi, int bracket, void* stream) {
  PL_REQUIRE(bracket >= 1 && bracket <= 4096, "bracket: 1 .. 4096 float32 bit patterns");
  PL_REQUIRE(d_plane && in_raw && d_weights && d_thr && d_count && d_status, "null pointer");
  PL_REQUIRE(dtype == PL_I16 || dtype == PL_U16, "int16 / uint16 slices");
  PL_REQUIRE(radius >= 1 && radius <= 8, "radius 1..8");
  PL_REQUIRE(n >= 0 && n <= 0x7fffffffLL && clear_border_ext >= 0, "bad arguments");
  PL_REQUIRE(!d_roi || (d_rawmax && catphan_size > 0), "the ROI selection needs the phantom size and the raw edge maxima");
  PL_REQUIRE(pl_mask_regions_fits(h, w, max_labels), "frame or label table too large for the LDS form (pl_mask_regions_fits)");
  if (n == 0) return PL_OK;
  const size_t lds = sr_lds_bytes(h, w);
  static_assert((kSrThreads / PL_WAVE) * (((2 * 8 + 1) * (2 * 8 + 1) + 2 * 8 + 1) * sizeof(double) + 32 * 8) <= (size_t)kSrMaxRuns * 12,
                "the exact recomputation's scratch (radius 8, every wave) and the undecided-word notes must fit the run tables they borrow");
  static std::atomic<size_t> attr{0};
  if (lds > attr) {
    hipError_t e = hipFuncSetAttribute((const void*)mask_regions_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { pl_set_error("pl_edge_regions: LDS attribute: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
    attr = lds;
  }
  SrEdgeArgs ea{in_raw, dtype == PL_I16, d_weights, radius, catphan_size, d_rawmax, d_roi, (unsigned)bracket};
  hipLaunchKernelGGL(mask_regions_kernel<float>, dim3((unsigned)n), dim3(kSrThreads), lds, (hipStream_t)stream, d_plane, d_thr, h, w,
                     clear_border_ext, fill_holes, max_labels, d_table, d_count, d_status, d_out_mask, ea);
  return pl_check_launch("pl_edge_regions");
}

#if PL_SR_TIMING
extern "C" int pl_debug_sr_timing(unsigned long long* h_out) {
  if (hipMemcpyFromSymbol(h_out, HIP_SYMBOL(pl_sr_dbg), sizeof(pl_sr_dbg)) != hipSuccess) return 1;
  unsigned long long zero[32] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(pl_sr_dbg), zero, sizeof(zero)) == hipSuccess ? 0 : 1;
}
#endif

/* pl_edge_regions_ex for a plane that stores RN32 of the exact value (bracket 1): pl_edge_plane's */
extern "C" int pl_edge_regions(const float* d_plane, const void* in_raw, int dtype, const double* d_weights, int radius,
                               const double* d_thr, int64_t n, int h, int w, int clear_border_ext, int fill_holes, int max_labels,
                               double* d_table, int32_t* d_count, int32_t* d_status, uint8_t* d_out_mask, double catphan_size,
                               const double* d_rawmax, double* d_roi, void* stream) {
  return pl_edge_regions_ex(d_plane, in_raw, dtype, d_weights, radius, d_thr, n, h, w, clear_border_ext, fill_holes, max_labels,
                            d_table, d_count, d_status, d_out_mask, catphan_size, d_rawmax, d_roi, 1, stream);
}
