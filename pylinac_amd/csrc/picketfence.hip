// Picket-fence per-image measurement (BASELINE config #3).
//
// Replaces the numpy/scipy work inside PicketFence.analyze for UP_DOWN pickets
// (pylinac/picketfence.py:745-803) on frames that went through the constructor's
// ground()/normalize() (picketfence.py:322-323) -- WITHOUT materialising the float64 frame: every
// pixel value is the float64 quotient q = (a - sub_i) / div_i of the integer frame, evaluated
// wherever the reference would read the normalised image, so all results stay bit-identical.
//
//   pl_scaled_colmean   np.mean(image, 0)                         picketfence.py:747
//   pl_pf_pickets       FWXM picket centres int(round(l + (r-l)/2)), their heights, and the
//                       picket spacing np.median(np.diff(np.sort(idx)))   profile.py:2165-2170,
//                                                                          picketfence.py:766-767
//   pl_pf_windows       per (frame, leaf, picket): the window of _get_mlc_window (:859-886), the
//                       _is_mlc_peak_in_window test (:847-857: np.std(axis=1) with numpy's pairwise
//                       summation, np.max, np.median), np.median(window, axis=0), then
//                       ground + normalize-to-max of that profile (FWXMProfilePhysical(ground=True,
//                       normalization=MAX), :1609-1614)
//   pl_pf_positions     centre + max(approx_idx - spacing/2, 0)    picketfence.py:1624-1627
// One wave per window; the window's integer pixels are staged in LDS once.
#include "pl_common.h"
#include "peaks_device.h"

namespace {

#ifndef PL_PF_VARIANT
#define PL_PF_VARIANT 0         // 1 / 2 / 3: stopwatch builds that skip a stage of pf_windows_kernel (WRONG results; scripts/)
#endif
constexpr int kThreads = 256;
constexpr int kMaxRows = 48;    // window rows  (leaf width in pixels)
constexpr int kMaxCols = 128;   // window cols  (picket spacing in pixels); numpy pairwise-sum single block

__global__ void __launch_bounds__(kThreads)
scaled_colmean_kernel(const unsigned short* __restrict__ in, int h, int w, int col_tiles,
                      const double* __restrict__ sub, const double* __restrict__ div, double* __restrict__ out) {
  const int ct = blockIdx.x % col_tiles;
  const size_t frame = blockIdx.x / col_tiles;
  const int c = ct * kThreads + threadIdx.x;
  if (c >= w) return;
  const unsigned short* p = in + frame * (size_t)h * w + c;
  const PlQuot k = pl_quot_make(sub[frame], div[frame]);
  double acc = 0.0;
  for (int r = 0; r < h; ++r) acc = acc + pl_quot(k, (double)p[(size_t)r * w]);   // numpy adds row by row
  out[frame * w + c] = acc / (double)h;
}

// width % 4 == 0: a lane owns FOUR adjacent columns (one 8-byte load per row: 512 B per wave and row instead of 128) and
// keeps sixteen rows of loads in flight; the four sums are independent chains, each still row by row like numpy's.
// Round 1-3's one-column lanes reached 1 TB/s (a dependent float64 division per row and 128-byte wave loads).
// COLS = 2 (round 4): twice the waves (256 frames x 1024 columns are 1024 waves at four columns per lane -- ONE per SIMD, which
// then alternates between waiting for its loads and dividing) with twice the rows in flight.
#ifndef PL_COLMEAN_COLS
#define PL_COLMEAN_COLS 2
#endif
#ifndef PL_COLMEAN_ROWS
#define PL_COLMEAN_ROWS 32
#endif
template <int COLS, int U>
__global__ void __launch_bounds__(kThreads)
scaled_colmeanv_kernel(const unsigned short* __restrict__ in, int h, int w, int64_t total_groups,
                       const double* __restrict__ sub, const double* __restrict__ div, double* __restrict__ out) {
  static_assert(COLS == 1 || COLS == 2 || COLS == 4, "one, two or four columns per lane");
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= total_groups) return;
  const int gpr = w / COLS;
  const size_t frame = (size_t)(g / gpr);
  const int c = (int)(g % gpr) * COLS;
  const unsigned short* p = in + frame * (size_t)h * w + c;
  const PlQuot k = pl_quot_make(sub[frame], div[frame]);
  double a[COLS];
#pragma unroll
  for (int j = 0; j < COLS; ++j) a[j] = 0.0;
  struct Px { unsigned w[COLS > 1 ? COLS / 2 : 1]; };
  auto load = [&](int r) {
    Px v;
    if constexpr (COLS == 1) {
      v.w[0] = p[(size_t)r * w];
    } else if constexpr (COLS == 4) {
      const uint2 q = *reinterpret_cast<const uint2*>(p + (size_t)r * w);
      v.w[0] = q.x; v.w[1] = q.y;
    } else {
      v.w[0] = *reinterpret_cast<const unsigned*>(p + (size_t)r * w);
    }
    return v;
  };
  auto add = [&](const Px& v) {
    if constexpr (COLS == 1) a[0] = a[0] + pl_quot(k, (double)v.w[0]);
#pragma unroll
    for (int j = 0; j < COLS / 2; ++j) {
      a[2 * j] = a[2 * j] + pl_quot(k, (double)(v.w[j] & 0xffffu));
      a[2 * j + 1] = a[2 * j + 1] + pl_quot(k, (double)(v.w[j] >> 16));
    }
  };
  int r = 0;
  for (; r + U <= h; r += U) {
    Px v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = load(r + u);
#pragma unroll
    for (int u = 0; u < U; ++u) add(v[u]);
  }
  for (; r < h; ++r) add(load(r));
  double* o = out + frame * (size_t)w + c;
#pragma unroll
  for (int j = 0; j < COLS; ++j) o[j] = a[j] / (double)h;
}

// np.mean(image, 1) of the normalised frame (LEFT_RIGHT pickets, picketfence.py:749): along the CONTIGUOUS axis numpy sums
// pairwise (numpy/_core/src/umath/loops_utils.h.src): a row of more than 128 values is halved recursively (first half rounded
// down to a multiple of 8) into leaf blocks of at most 128, each summed with eight running partial sums combined as a tree,
// and the leaves are added up along the recursion tree.  The tree depends on the row length only, so the HOST lays it out
// once (leaf starts / lengths and a postfix program: k >= 0 pushes leaf k's sum, -1 adds the two on top); a wave stages its
// row in LDS, eight lanes sum each leaf (lane j = partial sum j), one lane runs the program.
constexpr int kRmMaxLeaves = 256;

__global__ void __launch_bounds__(kThreads)
scaled_rowmean_kernel(const unsigned short* __restrict__ in, int h, int w, int64_t total_rows, const double* __restrict__ sub,
                      const double* __restrict__ div, const int32_t* __restrict__ leaf_start, const int32_t* __restrict__ leaf_len,
                      int nleaves, const int32_t* __restrict__ program, int nprog, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rm_lds[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t row = (int64_t)blockIdx.x * (kThreads / PL_WAVE) + wv;
  if (row >= total_rows) return;
  const size_t wave_bytes = (((size_t)w * 2 + 15) & ~(size_t)15) + (size_t)(kRmMaxLeaves + 16) * sizeof(double);
  unsigned short* srow = reinterpret_cast<unsigned short*>(rm_lds + (size_t)wv * wave_bytes);
  double* ssum = reinterpret_cast<double*>(rm_lds + (size_t)wv * wave_bytes + (((size_t)w * 2 + 15) & ~(size_t)15));
  const int64_t frame = row / h;
  const unsigned short* p = in + row * (int64_t)w;
  for (int c = lane; c < w; c += PL_WAVE) srow[c] = p[c];
  pl_wave_sync();
  const PlQuot k = pl_quot_make(sub[frame], div[frame]);
  auto at = [&](int c) { return pl_quot(k, (double)srow[c]); };
  const int j = lane & 7, g = lane >> 3;
  for (int k0 = 0; k0 < nleaves; k0 += 8) {                  // wave-uniform trip count
    const int kk = k0 + g;
    const bool act = kk < nleaves;
    const int s0 = act ? leaf_start[kk] : 0, len = act ? leaf_len[kk] : 8;
    double res;
    if (len < 8) {                                           // only rows shorter than eight pixels
      res = 0.0;
      if (j == 0) for (int i = 0; i < len; ++i) res = res + at(s0 + i);
    } else {
      const int nmain = len - (len % 8);
      double acc = at(s0 + j);
      for (int i = 8; i < nmain; i += 8) acc = acc + at(s0 + i + j);
      acc = acc + __shfl_xor(acc, 1, 64);
      acc = acc + __shfl_xor(acc, 2, 64);
      acc = acc + __shfl_xor(acc, 4, 64);
      res = acc;
      if (j == 0) for (int i = nmain; i < len; ++i) res = res + at(s0 + i);
    }
    if (act && j == 0) ssum[kk] = res;
  }
  pl_wave_sync();
  if (lane == 0) {
    double* stack = ssum + kRmMaxLeaves;                     // 16 entries: the recursion is log2(w / 64) + 1 deep
    int sp = 0;
    for (int i = 0; i < nprog; ++i) {
      const int op = program[i];
      if (op >= 0) stack[sp++] = ssum[op];
      else { --sp; stack[sp - 1] = stack[sp - 1] + stack[sp]; }
    }
    out[row] = stack[0] / (double)w;
  }
}

__global__ void pf_pickets_kernel(const int32_t* __restrict__ count, const double* __restrict__ props, int cap,
                                  const double* __restrict__ prof, int w, int64_t n,
                                  int32_t* __restrict__ pk_idx, double* __restrict__ pk_val,
                                  double* __restrict__ spacing) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = count[i];
  const double* p = props + i * 6 * (int64_t)cap;
  int* idx = pk_idx + i * cap;
  double* val = pk_val + i * cap;
  for (int k = 0; k < c; ++k) {
    const double lt = p[4 * cap + k], rt = p[5 * cap + k];
    const int id = (int)rint(lt + (rt - lt) / 2);      // python round(): half to even
    idx[k] = id;
    val[k] = (id >= 0 && id < w) ? prof[i * w + id] : __longlong_as_double(0x7ff8000000000000LL);
  }
  // np.median(np.diff(np.sort(idx))) on a private sorted copy (cap is small)
  double sp = __longlong_as_double(0x7ff8000000000000LL);
  if (c >= 2 && c <= 64) {
    int s[64];
    for (int k = 0; k < c; ++k) s[k] = idx[k];
    for (int a = 1; a < c; ++a) { int v = s[a], b = a - 1; while (b >= 0 && s[b] > v) { s[b + 1] = s[b]; --b; } s[b + 1] = v; }
    int d[64];
    for (int k = 0; k + 1 < c; ++k) d[k] = s[k + 1] - s[k];
    const int m = c - 1;
    for (int a = 1; a < m; ++a) { int v = d[a], b = a - 1; while (b >= 0 && d[b] > v) { d[b + 1] = d[b]; --b; } d[b + 1] = v; }
    sp = (m & 1) ? (double)d[m / 2] : ((double)d[m / 2 - 1] + (double)d[m / 2]) / 2.0;
  }
  spacing[i] = sp;
}

// numpy's pairwise_sum for one contiguous block of n <= 128 float64 values (numpy/_core/src/umath/
// loops_utils.h.src): n < 8 plain loop, otherwise 8 running partial sums combined as a tree.
template <typename F>
__device__ __forceinline__ double pairwise_sum_block(int n, F at) {
  if (n < 8) {
    double res = 0.0;
    for (int i = 0; i < n; ++i) res = res + at(i);
    return res;
  }
  double r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = at(j);
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = r[j] + at(i + j);
  }
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res = res + at(i);
  return res;
}

// Middle order statistics of a window column (n <= N rows, pitch `pitch` in LDS) by Batcher's odd-even merge sort on N
// registers: lo = sorted[N/2 - 1], hi = sorted[N/2] of the column padded with floor((N - n) / 2) values of -1 in front and
// INT_MAX behind.  For even n these are the two middle values, for odd n `lo` is THE median.
template <int N>
__device__ __forceinline__ void pf_column_median(const unsigned short* col, int pitch, int n, int& lo, int& hi) {
  int v[N];
  const int pad_lo = (N - n) >> 1;
#pragma unroll
  for (int a = 0; a < N; ++a) {
    const int r = a - pad_lo;                        // wave-uniform; the read itself is unconditional (clamped row)
    const int rc = r < 0 ? 0 : (r < n ? r : n - 1);
    const int x = (int)col[rc * pitch];
    v[a] = r < 0 ? -1 : (r < n ? x : 0x7fffffff);
  }
#pragma unroll
  for (int p = 1; p < N; p <<= 1)
#pragma unroll
    for (int k = p; k >= 1; k >>= 1)
#pragma unroll
      for (int j = k % p; j + k < N; j += 2 * k)
#pragma unroll
        for (int i = 0; i < k; ++i)
          if (i + j + k < N && (i + j) / (2 * p) == (i + j + k) / (2 * p)) {
            const int x = v[i + j], y = v[i + j + k];
            v[i + j] = x < y ? x : y;
            v[i + j + k] = x < y ? y : x;
          }
  lo = v[N / 2 - 1];
  hi = v[N / 2];
}

__global__ void __launch_bounds__(kThreads)
pf_windows_kernel(const unsigned short* __restrict__ in, int h, int w, const double* __restrict__ sub,
                  const double* __restrict__ div, const int32_t* __restrict__ pk_count,
                  const int32_t* __restrict__ pk_idx, const double* __restrict__ pk_val, int cap,
                  const double* __restrict__ spacing, const int32_t* __restrict__ leaf_top,
                  const int32_t* __restrict__ leaf_bottom, int nleaves, double height_threshold,
                  double edge_threshold, int lmax, double* __restrict__ prof_out, int32_t* __restrict__ len_out,
                  double* __restrict__ offset_out, int32_t* __restrict__ status_out, int64_t total_windows, int rows_cap,
                  int lr, int wave_bytes, pl_peak_params fw, double* __restrict__ rec_out, int exact_std) {
  // dynamic LDS: per wave `rows_cap` x 128 window pixels, then `rows_cap` row deviations.  rows_cap is the tallest leaf window
  // the CALLER will ask for (the leaf geometry is host knowledge): the fixed 48-row capacity of rounds 1-3 kept three
  // workgroups on a CU where a 26-row bank leaves room for five -- the kernel is one long dependent chain per wave and
  // lives on the number of waves that hide it
  extern __shared__ __attribute__((aligned(16))) unsigned char pf_lds[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // per wave: [rows_cap row deviations][rows_cap x 128 window pixels]; once the column medians are in registers the same
  // bytes hold the FWXM search (rec_out): [128 profile samples][find_peaks tables]
  unsigned char* const wave_lds = pf_lds + (size_t)wv * wave_bytes;
  double* const s_std_w_ = reinterpret_cast<double*>(wave_lds);
  unsigned short* const sw_ = reinterpret_cast<unsigned short*>(wave_lds + (size_t)rows_cap * sizeof(double));
  __shared__ Scan pk_scan[kThreads / PL_WAVE];
  __shared__ double pk_red[kThreads / PL_WAVE][2 * (kThreads / PL_WAVE)];
  __shared__ int pk_cnt[kThreads / PL_WAVE];
  __shared__ int32_t o_cnt[kThreads / PL_WAVE], o_st[kThreads / PL_WAVE], o_idx[kThreads / PL_WAVE], o_lb[kThreads / PL_WAVE],
      o_rb[kThreads / PL_WAVE];
  __shared__ double o_p[kThreads / PL_WAVE][6];
  // window index in 32 bits (the launcher refuses more: the profiles of 2^31 windows would be 2 TB), wave-uniform: scalar
  const unsigned win = blockIdx.x * (unsigned)(kThreads / PL_WAVE) + (unsigned)wv;
  if ((int64_t)win >= total_windows) return;
  const int pi = (int)(win % (unsigned)cap);
  const int li = (int)((win / (unsigned)cap) % (unsigned)nleaves);
  const int64_t frame = win / ((unsigned)cap * (unsigned)nleaves);
  int status = 0;   // 0 valid, 1 no such picket, 2 failed _is_mlc_peak_in_window, 3 window too large / empty
  double* pout = prof_out ? prof_out + (size_t)win * lmax : nullptr;
  const double qnan = __longlong_as_double(0x7ff8000000000000LL);
  auto leave = [&](int code, double off_) {                // no measurement for this window
    if (lane == 0) {
      status_out[win] = code;
      if (len_out) len_out[win] = 0;
      if (offset_out) offset_out[win] = off_;
      if (rec_out) { rec_out[(size_t)win * 3] = qnan; rec_out[(size_t)win * 3 + 1] = qnan; rec_out[(size_t)win * 3 + 2] = qnan; }
    }
  };
  if (pi >= pk_count[frame]) {
    leave(1, 0.0);
    return;
  }
  const double approx = (double)pk_idx[frame * cap + pi];
  const double sp = spacing[frame];
  const int top = __builtin_amdgcn_readfirstlane(leaf_top[li]), bottom = __builtin_amdgcn_readfirstlane(leaf_bottom[li]);
  int left = (int)(approx - sp / 2);            // python int(): truncation toward zero
  if (left < 0) left = 0;
  int right = (int)(approx + sp / 2);
  const int travel_len = lr ? h : w;               // LEFT_RIGHT: the pickets run along the rows (picketfence.py:877-884)
  if (right > travel_len) right = travel_len;
  left = __builtin_amdgcn_readfirstlane(left);     // the window geometry is the wave's: keep it in scalar registers
  right = __builtin_amdgcn_readfirstlane(right);
  const int nrows = bottom - top, ncols = right - left;
  const double off = (approx - sp / 2 > 0.0) ? (approx - sp / 2) : 0.0;   // max(approx_idx - spacing/2, 0)
  if (nrows <= 0 || ncols <= 2 || nrows > rows_cap || ncols > kMaxCols || !(sp == sp)) {
    leave(3, off);
    return;
  }
  const unsigned short* f = in + frame * (size_t)h * w;
  unsigned short* sw = sw_;
  double* const s_std_w = s_std_w_;
  // the window into LDS, its maximum on the way (element e = lane, lane + 64, ..: row / column advance by carry, no division).
  // EIGHT loads are issued before the first of them is consumed: one load per trip made the wave pay the full memory
  // latency for every 64 pixels (eight dependent round trips for a 12 x 38 window)
  int vmax = 0;
  {
    int r = lane / ncols, c = lane - r * ncols;
    const int dr = PL_WAVE / ncols, dc = PL_WAVE - dr * ncols;
    const unsigned stride_across = lr ? 1u : (unsigned)w, stride_along = lr ? (unsigned)w : 1u;   // scalars: no branch per load
    const int total = nrows * ncols;
    for (int e0 = lane; e0 < total + lane; e0 += 8 * PL_WAVE) {      // wave-uniform trip count
      unsigned short v[8];
      int rr = r, cc = c;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        // window element (rr, cc) = (position across the leaf, position along the leaf's travel): UP_DOWN reads image row
        // top + rr, LEFT_RIGHT image column top + rr -- the LDS copy is the TRANSPOSED window then, and everything below
        // (max, column median = np.median(window, axis=1), the FWXM profile) is the UP_DOWN code.  The load is UNCONDITIONAL
        // (an element past the window's end re-reads its last row; only the LDS store below looks at the bound): behind a
        // per-lane branch every load cost twenty instructions of exec-mask bookkeeping and 64-bit address arithmetic
        // (r05c ISA); here it is a 32-bit element offset from the frame's scalar base
        const int rrc = rr < nrows ? rr : nrows - 1;
        v[k] = f[(unsigned)(top + rrc) * stride_across + (unsigned)(left + cc) * stride_along];
        rr += dr;
        cc += dc;
        if (cc >= ncols) { cc -= ncols; ++rr; }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (e0 + k * PL_WAVE < total) {
          sw[e0 + k * PL_WAVE] = v[k];
          vmax = max(vmax, (int)v[k]);
        }
      }
      r = rr;
      c = cc;
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  const PlQuot kq = pl_quot_make(sub[frame], div[frame]);
  auto q = [&](int r, int c) { return pl_quot(kq, (double)sw[r * ncols + c]); };

  // np.max(window) > height_threshold * picket_peak_val   (q is monotone in the integer pixel)
  vmax = pl_wave_reduce_idem(vmax, [](int a, int b) { return a > b ? a : b; });
  const bool above = pl_quot(kq, (double)vmax) > height_threshold * pk_val[frame * cap + pi];

  // np.std(window, axis=1) with numpy's pairwise summation order (one block of ncols <= 128 values: eight running sums
  // r[j] over elements j, j+8, ..., combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the n % 8 tail).  EIGHT lanes
  // share a row: lane j of the group owns chain r[j], the tree is three xor-shuffles (float addition commutes, so both
  // partners hold the same sum), one lane adds the tail.  Same operations in the same order as one lane doing it all,
  // at an eighth of the float64 divisions per lane (round 1: one lane per row, 12 of 64 lanes busy).
  // ---- is_not_at_edge = max(std) < edge_threshold * np.median(std), std = np.std(window, axis) per leaf pixel row.
  // The DECISION does not need numpy's float64 deviations: std(q) = std(a) / div for the integer pixels a, and with
  // V_r = n * sum(a^2) - (sum a)^2 (an exact integer below 2^47) the test reads  max_r sqrt(V_r) < thr * median_r sqrt(V_r)
  // -- n and div cancel.  numpy's values (pairwise or sequential sums of rounded quotients, two-pass deviations) differ from
  // the exact ones by far less than 1e-7 relative + 1e-11 * div * n absolute in these units (a row that is not constant has
  // sqrt(V) >= 1.4), so the test is taken from the integer moments whenever it holds or fails by more than that margin, and
  // only a window inside the margin (or with `exact_std`) evaluates numpy's sequence below.  r05a stopwatch: the float64
  // stage was 180 of the kernel's 485 us.
  bool decided = false, not_edge = false;
  if (!exact_std) {
    const int j = lane & 7, gr = lane >> 3;
    for (int row0 = 0; row0 < nrows; row0 += 8) {           // wave-uniform trip count
      const int r = row0 + gr;
      const bool act = r < nrows;
      const unsigned short* rowp = sw + (act ? r : 0) * ncols;
      unsigned s1 = 0;
      unsigned long long s2 = 0;
      for (int c = j; c < ncols; c += 8) {
        const unsigned a = rowp[c];
        s1 += a;
        s2 += (unsigned long long)(a * a);                   // a < 2^16: the product fits 32 bits
      }
      s1 = pl_group8_sum(s1);
      s2 = pl_group8_sum(s2);
      if (act && j == 0) {
        const unsigned long long v = (unsigned long long)ncols * s2 - (unsigned long long)s1 * (unsigned long long)s1;
        s_std_w[r] = sqrt((double)v);                        // V < 2^47: the conversion is exact
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const bool act = lane < nrows;
    const double va = act ? s_std_w[lane] : 0.0;
    const double smax_i = pl_wave_reduce_idem(act ? va : -1.0, [](double x, double y) { return x > y ? x : y; });
    int rank = 0;
    for (int b2 = 0; b2 < nrows; ++b2) {
      const double vb = s_std_w[b2];
      rank += (vb < va || (vb == va && b2 < lane)) ? 1 : 0;
    }
    const int k_hi = nrows / 2, k_lo = (nrows & 1) ? k_hi : k_hi - 1;
    const unsigned long long m_lo = __ballot(act && rank == k_lo), m_hi = __ballot(act && rank == k_hi);
    const double v_lo = __shfl(va, __builtin_ctzll(m_lo | (1ull << 63)), 64);
    const double v_hi = __shfl(va, __builtin_ctzll(m_hi | (1ull << 63)), 64);
    const double med_i = (nrows & 1) ? v_hi : (v_lo + v_hi) / 2.0;
    const double tau = 1.0e-11 * div[frame] * (double)ncols, eta = 1.0e-7;
    const double max_hi = smax_i * (1.0 + eta) + tau, max_lo = smax_i * (1.0 - eta) - tau;
    const double med_hi = med_i * (1.0 + eta) + tau, med_lo = med_i * (1.0 - eta) - tau;
    if (max_hi < edge_threshold * med_lo) { decided = true; not_edge = true; }
    else if (max_lo >= edge_threshold * med_hi) { decided = true; not_edge = false; }
    __builtin_amdgcn_wave_barrier();                         // (the exact stage below rewrites s_std_w)
  }
  if (!decided) {
#if PL_PF_VARIANT == 1                                     // stopwatch: no deviation stage (wrong edge test)
  if (lane < nrows) s_std_w[lane] = 1.0;
#else
  if (lr) {
    // LEFT_RIGHT: np.std(window, axis=0) reduces over the window's ROWS (the travel direction), which numpy adds up one row
    // after the other -- no pairwise blocks on a non-contiguous reduction axis: a plain left-to-right sum per leaf pixel
    for (int a = lane; a < nrows; a += PL_WAVE) {
      double sum = 0.0;
      for (int b = 0; b < ncols; ++b) sum = sum + q(a, b);
      const double mean = sum / (double)ncols;
      double ss = 0.0;
      for (int b = 0; b < ncols; ++b) { const double x = q(a, b) - mean; ss = ss + x * x; }
      s_std_w[a] = sqrt(ss / (double)ncols);
    }
  } else
  {
    const int j = lane & 7, gr = lane >> 3;                 // chain index, row inside the pass of 8 rows
    const int nmain = ncols - (ncols % 8);
    for (int row0 = 0; row0 < nrows; row0 += 8) {           // wave-uniform trip count
      const int r = row0 + gr;
      const bool act = r < nrows;
      const int rr = act ? r : 0;
      auto block_sum = [&](auto at) -> double {            // every lane of the wave executes the shuffles
        double res;
        if (ncols < 8) {
          res = 0.0;
          if (j == 0) for (int i = 0; i < ncols; ++i) res = res + at(i);
        } else {
          double acc = at(j);
          for (int i = 8; i < nmain; i += 8) acc = acc + at(i + j);
          acc = acc + __shfl_xor(acc, 1, 64);
          acc = acc + __shfl_xor(acc, 2, 64);
          acc = acc + __shfl_xor(acc, 4, 64);
          res = acc;
          if (j == 0) for (int i = nmain; i < ncols; ++i) res = res + at(i);
        }
        return __shfl(res, lane & ~7, 64);                  // the group's lane 0 holds the total
      };
      const double mean = block_sum([&](int c) { return q(rr, c); }) / (double)ncols;
      const double ss = block_sum([&](int c) { const double x = q(rr, c) - mean; return x * x; });
      if (act && j == 0) s_std_w[r] = sqrt(ss / (double)ncols);
    }
  }
#endif
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  // max(std) < edge_threshold * np.median(std): lane a ranks std[a] (nrows <= 48 <= 64 lanes)
  double smax, med;
  {
    const bool act = lane < nrows;
    const double va = act ? s_std_w[lane] : 0.0;
    smax = pl_wave_reduce_idem(act ? va : -1.0, [](double x, double y) { return x > y ? x : y; });   // std >= 0
    int rank = 0;
    for (int b2 = 0; b2 < nrows; ++b2) {
      const double vb = s_std_w[b2];
      rank += (vb < va || (vb == va && b2 < lane)) ? 1 : 0;
    }
    const int k_hi = nrows / 2, k_lo = (nrows & 1) ? k_hi : k_hi - 1;
    const unsigned long long m_lo = __ballot(act && rank == k_lo), m_hi = __ballot(act && rank == k_hi);
    const double v_lo = __shfl(va, __builtin_ctzll(m_lo | (1ull << 63)), 64);
    const double v_hi = __shfl(va, __builtin_ctzll(m_hi | (1ull << 63)), 64);
    med = (nrows & 1) ? v_hi : (v_lo + v_hi) / 2.0;
  }
  not_edge = smax < edge_threshold * med;
  }
  if (!(above && not_edge)) status = 2;

  // np.median(window, axis=0): lane c (and c+64) selects the middle order statistic(s) of its column; the profile stays in
  // registers (ncols <= 128: two columns per lane) through ground() and normalize() and is stored once -- round 2 wrote it
  // to global memory and read it back three times, each a full store -> load round trip
  double pvr[2] = {0.0, 0.0};
#pragma unroll
  for (int slot = 0; slot < 2; ++slot) {
    const int c = lane + slot * PL_WAVE;
    if (c >= ncols) continue;
    const int k_hi = nrows / 2, k_lo = (nrows & 1) ? k_hi : k_hi - 1;
    int v_lo = 0, v_hi = 0;
#if PL_PF_VARIANT == 2                                     // stopwatch: no column median (row 0 instead)
    v_lo = v_hi = sw[c];
#else
    if (nrows <= 32) {                               // wave-uniform; every leaf of the Millennium / HD / Agility banks at EPID scale
      // A sorting network on the column in registers.  The column is padded to N values with floor((N - n) / 2) values below
      // every pixel and the rest above: the middle order statistics then sit at the FIXED positions N/2 - 1 and N/2 (n even)
      // or N/2 - 1 alone (n odd), so the compiler drops every compare-exchange the two outputs do not depend on.  Rounds 1-3
      // ranked by counting (n^2 LDS reads beyond 16 rows: 20 us for a 26-row leaf).
      if (nrows <= 16) pf_column_median<16>(sw + c, ncols, nrows, v_lo, v_hi);
      else pf_column_median<32>(sw + c, ncols, nrows, v_lo, v_hi);
      if (nrows & 1) v_hi = v_lo;
    } else {
      for (int a = 0; a < nrows; ++a) {
        const int va = sw[a * ncols + c];
        int rank = 0;
        for (int b = 0; b < nrows; ++b) {
          const int vb = sw[b * ncols + c];
          rank += (vb < va || (vb == va && b < a)) ? 1 : 0;
        }
        if (rank == k_lo) v_lo = va;
        if (rank == k_hi) v_hi = va;
      }
    }
#endif
    const double qh = pl_quot(kq, (double)v_hi);
    pvr[slot] = (nrows & 1) ? qh : (pl_quot(kq, (double)v_lo) + qh) / 2.0;
  }
  // ground (values - min) then normalize (/ max of the grounded profile)
  const bool has0 = lane < ncols, has1 = lane + PL_WAVE < ncols;
  double mn = __longlong_as_double(0x7ff0000000000000LL);
  if (has0) mn = pvr[0] < mn ? pvr[0] : mn;
  if (has1) mn = pvr[1] < mn ? pvr[1] : mn;
  mn = pl_wave_reduce_idem(mn, [](double a, double b) { return a < b ? a : b; });
  double mx = __longlong_as_double((long long)0xfff0000000000000ULL);
  if (has0) { const double g = pvr[0] - mn; mx = g > mx ? g : mx; }
  if (has1) { const double g = pvr[1] - mn; mx = g > mx ? g : mx; }
  mx = pl_wave_reduce_idem(mx, [](double a, double b) { return a > b ? a : b; });
  const double p0 = (pvr[0] - mn) / mx, p1 = (pvr[1] - mn) / mx;
  if (prof_out) {
    if (has0) pout[lane] = p0;
    if (has1) pout[lane + PL_WAVE] = p1;
  }
  if (lane == 0) {
    status_out[win] = status;
    if (len_out) len_out[win] = ncols;
    if (offset_out) offset_out[win] = off;
  }
  if (!rec_out) return;
  // ---- the FWXM search on the profile just built (FWXMProfilePhysical.field_edge_idx / center_idx, profile.py:602-611,
  // 322-327: find_peaks(fwxm_height, max_number = 1), left_ips / right_ips of the most prominent peak) -- rounds 1-3 wrote the
  // profile to a [windows][128] float64 table (0.5 GB per 512 frames) for a second launch to read
  double c_pos = qnan, l_pos = qnan, r_pos = qnan;
#if PL_PF_VARIANT == 3                                     // stopwatch: no FWXM search
  if (status == 0) c_pos = l_pos = r_pos = p0 + p1;
  if (false) {
#else
  if (status == 0) {                                       // wave-uniform
#endif
    pl_wave_sync();                                        // every lane is done with the window pixels: the bytes change hands
    double* s_prof = reinterpret_cast<double*>(wave_lds);
    if (has0) s_prof[lane] = p0;
    if (has1) s_prof[lane + PL_WAVE] = p1;
    pl_wave_sync();
    constexpr int kMaxc = kMaxCols / 2 + 1;
    const PeakLds L{wave_lds + kMaxCols * sizeof(double), &pk_scan[wv], pk_red[wv], &pk_cnt[wv]};
    find_peaks_profile<true, PL_WAVE>(s_prof, ncols, fw.region_lo, fw.region_hi, fw, 1, kMaxc, L, lane, &o_cnt[wv], &o_idx[wv],
                                      &o_lb[wv], &o_rb[wv], o_p[wv], &o_st[wv]);
    pl_wave_sync();
    if (o_cnt[wv] > 0) {
      const double l = o_p[wv][4], r = o_p[wv][5];
      c_pos = (fabs(r - l) / 2 + l) + off;                 // center_idx + max(approx_idx - spacing / 2, 0)
      l_pos = l + off;
      r_pos = r + off;
    }
  }
  if (lane == 0) { rec_out[(size_t)win * 3] = c_pos; rec_out[(size_t)win * 3 + 1] = l_pos; rec_out[(size_t)win * 3 + 2] = r_pos; }
}

__global__ void pf_positions_kernel(const int32_t* __restrict__ status, const double* __restrict__ fwxm /*[M][8]*/,
                                    const double* __restrict__ offset, int64_t m, double* __restrict__ pos) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  pos[i] = (status[i] == 0 && fwxm[i * 8] > 0.0) ? fwxm[i * 8 + 6] + offset[i] : nan;
}

}  // namespace

extern "C" int pl_scaled_colmean(const uint16_t* in, int64_t n, int h, int w, const double* d_sub,
                                 const double* d_div, double* d_out, void* stream) {
  PL_REQUIRE(in && d_sub && d_div && d_out, "null pointer");
  PL_REQUIRE(n >= 0 && h > 0 && w > 0, "bad shape");
  if (n == 0) return PL_OK;
  const int col_tiles = (int)pl_cdiv(w, kThreads);
  PL_REQUIRE(n * col_tiles <= 0x7fffffffLL, "batch too large");
  if ((w & 3) == 0 && (reinterpret_cast<uintptr_t>(in) & 7) == 0) {
    const int64_t groups = n * (int64_t)(w / PL_COLMEAN_COLS);
    PL_REQUIRE(pl_cdiv(groups, kThreads) <= 0x7fffffffLL, "batch too large");
    hipLaunchKernelGGL((scaled_colmeanv_kernel<PL_COLMEAN_COLS, PL_COLMEAN_ROWS>), dim3((unsigned)pl_cdiv(groups, kThreads)),
                       dim3(kThreads), 0, (hipStream_t)stream, in, h, w, groups, d_sub, d_div, d_out);
    return pl_check_launch("pl_scaled_colmean");
  }
  hipLaunchKernelGGL(scaled_colmean_kernel, dim3((unsigned)(n * col_tiles)), dim3(kThreads), 0, (hipStream_t)stream,
                     in, h, w, col_tiles, d_sub, d_div, d_out);
  return pl_check_launch("pl_scaled_colmean");
}

extern "C" int pl_pf_pickets(const int32_t* d_count, const double* d_props, int cap, const double* d_prof, int w,
                             int64_t n, int32_t* d_pk_idx, double* d_pk_val, double* d_spacing, void* stream) {
  PL_REQUIRE(d_count && d_props && d_prof && d_pk_idx && d_pk_val && d_spacing, "null pointer");
  PL_REQUIRE(n >= 0 && cap > 0 && cap <= 64 && w > 0, "bad shape (cap <= 64)");
  if (n == 0) return PL_OK;
  hipLaunchKernelGGL(pf_pickets_kernel, dim3((unsigned)pl_cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, d_count,
                     d_props, cap, d_prof, w, n, d_pk_idx, d_pk_val, d_spacing);
  return pl_check_launch("pl_pf_pickets");
}

extern "C" int pl_pf_windows_rows(const uint16_t* in, int64_t n, int h, int w, const double* d_sub, const double* d_div,
                                  const int32_t* d_pk_count, const int32_t* d_pk_idx, const double* d_pk_val, int cap,
                                  const double* d_spacing, const int32_t* d_leaf_top, const int32_t* d_leaf_bottom,
                                  int nleaves, int max_rows, double height_threshold, double edge_threshold, int lmax,
                                  double* d_prof, int32_t* d_len, double* d_offset, int32_t* d_status, void* stream);

extern "C" int pl_pf_windows(const uint16_t* in, int64_t n, int h, int w, const double* d_sub, const double* d_div,
                             const int32_t* d_pk_count, const int32_t* d_pk_idx, const double* d_pk_val, int cap,
                             const double* d_spacing, const int32_t* d_leaf_top, const int32_t* d_leaf_bottom,
                             int nleaves, double height_threshold, double edge_threshold, int lmax,
                             double* d_prof, int32_t* d_len, double* d_offset, int32_t* d_status, void* stream) {
  return pl_pf_windows_rows(in, n, h, w, d_sub, d_div, d_pk_count, d_pk_idx, d_pk_val, cap, d_spacing, d_leaf_top, d_leaf_bottom,
                            nleaves, kMaxRows, height_threshold, edge_threshold, lmax, d_prof, d_len, d_offset, d_status, stream);
}

// per-wave LDS of pf_windows_kernel: the window block, or (fused FWXM search) the profile + find_peaks tables, whichever is larger
static size_t pf_wave_bytes(int rows_cap, bool fused) {
  size_t wb = (size_t)rows_cap * (sizeof(double) + (size_t)kMaxCols * sizeof(unsigned short));
  if (fused) {
    constexpr int maxc = kMaxCols / 2 + 1;
    size_t sb = (size_t)kMaxCols * sizeof(double) + (((size_t)maxc * (8 + 8 + 4 * 4) + 8 + (size_t)kMaxCols * 8 + 15) & ~(size_t)15);
    if (sb > wb) wb = sb;
  }
  return (wb + 15) & ~(size_t)15;
}

// max_rows: the tallest leaf window (bottom - top) of the call, 1 .. 48 -- windows taller than that get status 3
extern "C" int pl_pf_windows_rows(const uint16_t* in, int64_t n, int h, int w, const double* d_sub, const double* d_div,
                                  const int32_t* d_pk_count, const int32_t* d_pk_idx, const double* d_pk_val, int cap,
                                  const double* d_spacing, const int32_t* d_leaf_top, const int32_t* d_leaf_bottom,
                                  int nleaves, int max_rows, double height_threshold, double edge_threshold, int lmax,
                                  double* d_prof, int32_t* d_len, double* d_offset, int32_t* d_status, void* stream) {
  PL_REQUIRE(max_rows >= 1 && max_rows <= kMaxRows, "max_rows 1..48");
  PL_REQUIRE(in && d_sub && d_div && d_pk_count && d_pk_idx && d_pk_val && d_spacing && d_leaf_top && d_leaf_bottom &&
                 d_prof && d_len && d_offset && d_status, "null pointer");
  PL_REQUIRE(n >= 0 && h > 0 && w > 0 && cap > 0 && nleaves > 0 && lmax >= kMaxCols, "bad shape (lmax >= 128)");
  if (n == 0) return PL_OK;
  const int64_t total = n * (int64_t)nleaves * cap;
  const int64_t blocks = pl_cdiv(total, kThreads / PL_WAVE);
  PL_REQUIRE(total <= 0x7fffffffLL, "batch too large");
  const int rows_cap = (max_rows + 1) & ~1;          // even: the window planes stay 4-byte aligned behind the row deviations
  const size_t wb = pf_wave_bytes(rows_cap, false);
  hipLaunchKernelGGL(pf_windows_kernel, dim3((unsigned)blocks), dim3(kThreads), wb * (kThreads / PL_WAVE), (hipStream_t)stream, in, h, w,
                     d_sub, d_div, d_pk_count, d_pk_idx, d_pk_val, cap, d_spacing, d_leaf_top, d_leaf_bottom, nleaves,
                     height_threshold, edge_threshold, lmax, d_prof, d_len, d_offset, d_status, total, rows_cap, 0, (int)wb,
                     pl_peak_params{}, nullptr, 0);
  return pl_check_launch("pl_pf_windows");
}

/* windows + FWXM positions in one launch, either orientation: see pylinac_hip.h */
extern "C" int pl_pf_measure(const uint16_t* in, int64_t n, int h, int w, int orientation, const double* d_sub, const double* d_div,
                             const int32_t* d_pk_count, const int32_t* d_pk_idx, const double* d_pk_val, int cap,
                             const double* d_spacing, const int32_t* d_leaf_lo, const int32_t* d_leaf_hi, int nleaves,
                             int max_rows, double height_threshold, double edge_threshold, int exact_deviation,
                             const pl_peak_params* fwxm_params, double* d_rec, int32_t* d_status, double* d_prof, int lmax,
                             void* stream) {
  PL_REQUIRE(max_rows >= 1 && max_rows <= kMaxRows, "max_rows 1..48");
  PL_REQUIRE(orientation == 0 || orientation == 1, "orientation 0 (UP_DOWN) or 1 (LEFT_RIGHT)");
  PL_REQUIRE(in && d_sub && d_div && d_pk_count && d_pk_idx && d_pk_val && d_spacing && d_leaf_lo && d_leaf_hi && fwxm_params &&
                 d_rec && d_status, "null pointer");
  PL_REQUIRE(n >= 0 && h > 0 && w > 0 && cap > 0 && nleaves > 0 && (!d_prof || lmax >= kMaxCols), "bad shape (lmax >= 128)");
  PL_REQUIRE(fwxm_params->distance >= 1, "distance must be >= 1");
  if (n == 0) return PL_OK;
  const int64_t total = n * (int64_t)nleaves * cap;
  const int64_t blocks = pl_cdiv(total, kThreads / PL_WAVE);
  PL_REQUIRE(total <= 0x7fffffffLL, "batch too large");
  const int rows_cap = (max_rows + 1) & ~1;
  const size_t wb = pf_wave_bytes(rows_cap, true);
  hipLaunchKernelGGL(pf_windows_kernel, dim3((unsigned)blocks), dim3(kThreads), wb * (kThreads / PL_WAVE), (hipStream_t)stream, in, h, w,
                     d_sub, d_div, d_pk_count, d_pk_idx, d_pk_val, cap, d_spacing, d_leaf_lo, d_leaf_hi, nleaves,
                     height_threshold, edge_threshold, lmax, d_prof, (int32_t*)nullptr, (double*)nullptr, d_status, total, rows_cap,
                     orientation, (int)wb, *fwxm_params, d_rec, exact_deviation ? 1 : 0);
  return pl_check_launch("pl_pf_measure");
}

extern "C" int pl_pf_positions(const int32_t* d_status, const double* d_fwxm, const double* d_offset, int64_t m,
                               double* d_pos, void* stream) {
  PL_REQUIRE(d_status && d_fwxm && d_offset && d_pos, "null pointer");
  PL_REQUIRE(m >= 0, "bad shape");
  if (m == 0) return PL_OK;
  hipLaunchKernelGGL(pf_positions_kernel, dim3((unsigned)pl_cdiv(m, 256)), dim3(256), 0, (hipStream_t)stream, d_status,
                     d_fwxm, d_offset, m, d_pos);
  return pl_check_launch("pl_pf_positions");
}

/* np.mean(q, 1) -> d_out float64 [n][h] in numpy's pairwise order; the summation tree (leaves, postfix program) of a row of
 * w values is laid out by the caller: see scaled_rowmean_kernel */
extern "C" int pl_scaled_rowmean(const uint16_t* in, int64_t n, int h, int w, const double* d_sub, const double* d_div,
                                 const int32_t* d_leaf_start, const int32_t* d_leaf_len, int nleaves, const int32_t* d_program,
                                 int nprog, double* d_out, void* stream) {
  PL_REQUIRE(in && d_sub && d_div && d_leaf_start && d_leaf_len && d_program && d_out, "null pointer");
  PL_REQUIRE(n >= 0 && h > 0 && w > 0 && w <= 16384, "bad shape (rows of at most 16384 pixels)");
  PL_REQUIRE(nleaves >= 1 && nleaves <= kRmMaxLeaves && nprog == 2 * nleaves - 1, "a summation tree of 1..256 leaves");
  if (n == 0) return PL_OK;
  const int64_t rows = n * (int64_t)h;
  PL_REQUIRE(pl_cdiv(rows, kThreads / PL_WAVE) <= 0x7fffffffLL, "batch too large");
  const size_t wave_bytes = (((size_t)w * 2 + 15) & ~(size_t)15) + (size_t)(kRmMaxLeaves + 16) * sizeof(double);
  const size_t lds = wave_bytes * (kThreads / PL_WAVE);
  static std::atomic<size_t> attr{0};
  if (lds > 64 * 1024 && lds > attr) {
    hipError_t e = hipFuncSetAttribute((const void*)scaled_rowmean_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { pl_set_error("pl_scaled_rowmean: LDS attribute: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
    attr = lds;
  }
  hipLaunchKernelGGL(scaled_rowmean_kernel, dim3((unsigned)pl_cdiv(rows, kThreads / PL_WAVE)), dim3(kThreads), lds, (hipStream_t)stream,
                     in, h, w, rows, d_sub, d_div, d_leaf_start, d_leaf_len, nleaves, d_program, nprog, d_out);
  return pl_check_launch("pl_scaled_rowmean");
}
