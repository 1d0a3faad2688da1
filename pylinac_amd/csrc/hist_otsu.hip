// Per-frame exact 16-bit histogram, Otsu threshold and order statistics
// (SURVEY.md section 8 rows a4 and a6).
//
// Replaces:
//   skimage.filters.threshold_otsu on integer images  (pylinac/ct.py:3323,3338-3340, acr.py:1409;
//       skimage 0.18.3 filters/thresholding.py + exposure.histogram: one bin per integer value in
//       [min,max], float64 cumulative class statistics, FIRST argmax of
//       w1[:-1]*w2[1:]*(m1[:-1]-m2[1:])**2, returns the bin centre),
//   np.percentile order statistics  (pylinac/core/image.py:899-926, picketfence.py:229-238,
//       winston_lutz.py:775,1109-1133): exact k-th smallest values, the linear interpolation
//       between the two neighbours is two flops done by the host layer.
//
// Histogram: a 65536-bin uint32 table is 256 KiB -- larger than LDS (160 KiB) -- and global
// atomics would serialise on L2.  Each frame is therefore histogrammed by PARTS=2 workgroups,
// workgroup p owning bins [32768p, 32768p+32768) in 128 KiB of LDS and streaming the whole frame
// (16-byte loads; the two readers of a frame are placed on one XCD so the second read is an L2 hit),
// one LDS atomic per in-range pixel.  Every bin is written exactly once with a plain coalesced store,
// so the table needs no zero-fill and no global atomics.
//
// Otsu / order statistics: one 1024-lane workgroup per frame; lane t owns bins [64t, 64t+64);
// exact integer prefix sums (uint64 counts, int64 value sums -- identical to numpy's float64
// cumsum because every partial sum is an integer < 2^53), then the float64 variance expression in
// skimage's operation order and a (value, first-index) arg-max reduction.
#include "pl_common.h"
#include "median3_rows.h"

// rows the fused median + Otsu kernel keeps in flight per lane (1024 threads: 128 VGPRs)
#ifndef PL_OTSU_AHEAD
#define PL_OTSU_AHEAD 2
#endif

namespace {

constexpr int kHistThreads = 1024;

// PARTS workgroups per frame, workgroup p owns bins [65536/PARTS * p, ...) in LDS and streams the whole frame.
// RUNS: merge consecutive equal in-range values of a lane's stream into one LDS atomic (fewer atomics, more VALU).
template <int PARTS, bool RUNS>
__global__ void __launch_bounds__(kHistThreads)
hist16_kernel(const unsigned short* __restrict__ in, int64_t n, int64_t count, unsigned flip,
              uint32_t* __restrict__ hist, const int32_t* __restrict__ only /* NULL, or per-frame: run when != 0 */,
              unsigned virtual_blocks) {
  constexpr int kBins = 65536 / PARTS;
  constexpr int kShift = PARTS == 4 ? 14 : 15;
  extern __shared__ unsigned bins[];  // kBins
  // gated launches are persistent (see median3_oct_kernel): leave at once when no frame is flagged
  if (only) {
    int any = 0;
    for (int64_t i = threadIdx.x; i < n; i += kHistThreads) any |= only[i];
    if (!__syncthreads_or(any)) return;
  }
  for (unsigned vb = blockIdx.x; vb < virtual_blocks; vb += gridDim.x) {
  // 8 frames x PARTS parts per group; the parts of one frame share block % 8 (same XCD)
  const unsigned within = vb % (8 * PARTS);
  const int64_t frame = (int64_t)(vb / (8 * PARTS)) * 8 + (within & 7);
  const unsigned part = within >> 3;
  if (frame >= n) continue;
  if (only && !only[frame]) continue;
  __syncthreads();                                   // the previous virtual block's copy-out is done with the bins
  for (int i = threadIdx.x; i < kBins; i += kHistThreads) bins[i] = 0;
  __syncthreads();

  const unsigned short* src = in + frame * count;
  unsigned prev = 0xffffffffu, run = 0;
  auto tally = [&](unsigned key) {
    key ^= flip;
    if constexpr (RUNS) {
      const unsigned b = ((key >> kShift) == part) ? (key & (kBins - 1)) : 0xffffffffu;
      if (b == prev) {
        ++run;
      } else {
        if (prev != 0xffffffffu) atomicAdd(&bins[prev], run);
        prev = b;
        run = 1;
      }
    } else {
      if ((key >> kShift) == part) atomicAdd(&bins[key & (kBins - 1)], 1u);
    }
  };
  if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    const int64_t nvec = count / 8;
    const uint4* vsrc = reinterpret_cast<const uint4*>(src);
    // Flat regions (the zero background of a Winston-Lutz frame, the air around a phantom) send all 64 lanes to ONE
    // bin: 64-way same-address serialisation per pixel (measured 1.76 ms per 256 frames on WL frames against 0.16 ms on
    // EPID content).  When the whole wave holds one value in this vector (8 pixels per lane) a single lane adds the
    // whole count; the test costs a handful of instructions per eight pixels.
    auto tally4 = [&](uint4 q) {
      const unsigned first = q.x & 0xffffu;
      const unsigned splat = first | (first << 16);
      const bool lane_flat = q.x == splat && q.y == splat && q.z == splat && q.w == splat;
      const unsigned wave_first = __builtin_amdgcn_readfirstlane(splat);
      const unsigned long long active = __ballot(1);                  // taken by ALL active lanes, before any lane-only branch
      if (__ballot(!(lane_flat && splat == wave_first)) == 0) {       // wave-uniform branch
        if constexpr (!RUNS) {
          const unsigned key = (wave_first & 0xffffu) ^ flip;
          if ((key >> kShift) == part && (threadIdx.x & 63) == __builtin_ctzll(active))
            atomicAdd(&bins[key & (kBins - 1)], 8u * (unsigned)__popcll(active));
          return;
        }
      }
      const unsigned wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        tally(wds[k] & 0xffffu);
        tally(wds[k] >> 16);
      }
    };
    int64_t v = threadIdx.x;
    constexpr int U = 4;  // independent 16-byte loads in flight per lane
    for (; v + (int64_t)(U - 1) * kHistThreads < nvec; v += (int64_t)U * kHistThreads) {
      uint4 q[U];
#pragma unroll
      for (int k = 0; k < U; ++k) q[k] = vsrc[v + (int64_t)k * kHistThreads];
#pragma unroll
      for (int k = 0; k < U; ++k) tally4(q[k]);
    }
    for (; v < nvec; v += kHistThreads) tally4(vsrc[v]);
    for (int64_t i = nvec * 8 + threadIdx.x; i < count; i += kHistThreads) tally(src[i]);
  } else {
    for (int64_t i = threadIdx.x; i < count; i += kHistThreads) tally(src[i]);
  }
  if constexpr (RUNS) {
    if (prev != 0xffffffffu) atomicAdd(&bins[prev], run);
  }
  __syncthreads();
  uint32_t* dst = hist + frame * 65536 + (size_t)part * kBins;
  for (int i = threadIdx.x; i < kBins; i += kHistThreads) dst[i] = bins[i];
  }
}

template <int PARTS, bool RUNS>
int launch_hist16(const unsigned short* in, int64_t n, int64_t count, unsigned flip, uint32_t* hist, hipStream_t st,
                  const int32_t* only = nullptr) {
  const size_t lds = (size_t)(65536 / PARTS) * sizeof(unsigned);
  static std::atomic<bool> attr{false};
  if (!attr) {
    if (hipFuncSetAttribute((const void*)hist16_kernel<PARTS, RUNS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      return -1;
    }
    attr = true;
  }
  const int64_t blocks = pl_cdiv(n, 8) * 8 * PARTS;
  if (blocks > 0x7fffffffLL) return -1;
  const unsigned grid = only ? (unsigned)(blocks < 256 ? blocks : 256) : (unsigned)blocks;
  hipLaunchKernelGGL((hist16_kernel<PARTS, RUNS>), dim3(grid), dim3(kHistThreads), lds, st, in, n, count,
                     flip, hist, only, (unsigned)blocks);
  return 0;
}

// ---- block-wide exclusive scan over 1024 lanes (16 waves) for a pair of 64-bit integers --------
struct Pair { unsigned long long c; long long s; };

__device__ __forceinline__ Pair block_exclusive_scan(Pair v, Pair* total, Pair* wave_tot /*[16]*/) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  Pair inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    unsigned long long c = __shfl_up(inc.c, o, 64);
    long long s = __shfl_up(inc.s, o, 64);
    if (lane >= o) { inc.c += c; inc.s += s; }
  }
  if (lane == 63) wave_tot[wv] = inc;
  __syncthreads();
  Pair base = {0, 0};
  Pair tot = {0, 0};
  for (int k = 0; k < kHistThreads / 64; ++k) {
    if (k < wv) { base.c += wave_tot[k].c; base.s += wave_tot[k].s; }
    tot.c += wave_tot[k].c; tot.s += wave_tot[k].s;
  }
  __syncthreads();
  *total = tot;
  Pair ex = {base.c + inc.c - v.c, base.s + inc.s - v.s};
  return ex;
}

__global__ void __launch_bounds__(kHistThreads)
otsu_kernel(const uint32_t* __restrict__ hist, int bias, int32_t* __restrict__ thr,
            int32_t* __restrict__ vmin, int32_t* __restrict__ vmax, const int32_t* __restrict__ only) {
  __shared__ Pair wave_tot[kHistThreads / 64];
  __shared__ int s_lo[kHistThreads / 64], s_hi[kHistThreads / 64];
  __shared__ double s_var[kHistThreads / 64];
  __shared__ int s_idx[kHistThreads / 64];
  const int64_t frame = blockIdx.x;
  if (only && !only[frame]) return;
  const uint32_t* hh = hist + frame * 65536;
  const int b0 = threadIdx.x * 64;
  const uint4* p = reinterpret_cast<const uint4*>(hh + b0);
  Pair mine = {0, 0};
  int lo = 1 << 30, hi = -1;
  for (int k4 = 0; k4 < 16; ++k4) {
    const uint4 q = p[k4];
    const uint32_t c[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int b = b0 + 4 * k4 + j;
      mine.c += c[j];
      mine.s += (long long)c[j] * (long long)(b - bias);
      if (c[j]) { if (lo == (1 << 30)) lo = b; hi = b; }
    }
  }
  // first / last occupied bin of the frame
  lo = pl_wave_reduce(lo, [](int a, int b) { return a < b ? a : b; });
  hi = pl_wave_reduce(hi, [](int a, int b) { return a > b ? a : b; });
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { s_lo[wv] = lo; s_hi[wv] = hi; }
  Pair total;
  Pair ex = block_exclusive_scan(mine, &total, wave_tot);  // contains __syncthreads
  for (int k = 0; k < kHistThreads / 64; ++k) { lo = s_lo[k] < lo ? s_lo[k] : lo; hi = s_hi[k] > hi ? s_hi[k] : hi; }

  double best = -1.0;
  int best_k = 1 << 30;
  // (float64 running sums of integers below 2^53: exact, see otsu16_window_kernel)
  double dw1 = (double)ex.c, ds1 = (double)ex.s;
  const double dW = (double)total.c, dS = (double)total.s;
  for (int k4 = 0; k4 < 16; ++k4) {   // second sweep re-reads the 256 B from L2 (no spills)
    const uint4 q = p[k4];
    const uint32_t c[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int b = b0 + 4 * k4 + j;
      // an empty bin leaves both classes as they were: its variance equals the previous non-empty bin's (>= lo, evaluated) and
      // can never be strictly greater -- skipped, with its two float64 divisions
      if (c[j] != 0u) {
        dw1 += (double)c[j];
        ds1 = fma((double)c[j], (double)(b - bias), ds1);
        if (b >= lo && b < hi) {
          const double dw2 = dW - dw1;
          const double m1 = ds1 / dw1;
          const double m2 = (dS - ds1) / dw2;
          const double d = m1 - m2;
          const double var = (dw1 * dw2) * (d * d);
          if (var > best) { best = var; best_k = b; }
        }
      }
    }
  }
  // arg-max with first-index tie rule
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    double ov = __shfl_xor(best, o, 64);
    int ok = __shfl_xor(best_k, o, 64);
    if (ov > best || (ov == best && ok < best_k)) { best = ov; best_k = ok; }
  }
  if (lane == 0) { s_var[wv] = best; s_idx[wv] = best_k; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < kHistThreads / 64; ++k)
      if (s_var[k] > best || (s_var[k] == best && s_idx[k] < best_k)) { best = s_var[k]; best_k = s_idx[k]; }
    // constant image: skimage returns that value (thresholding.py: np.all(image == first_pixel))
    thr[frame] = (lo == hi) ? (lo - bias) : (best_k - bias);
    if (vmin) vmin[frame] = lo - bias;
    if (vmax) vmax[frame] = hi - bias;
  }
}

// ---- single-pass Otsu: histogram in LDS + class statistics, one workgroup per frame -------------------------------------
// A 152 KiB LDS window holds 38 912 consecutive bins.  Where the frame's values fit such a window the frame is read ONCE
// (plus a 1/16 row sample), every pixel is one LDS atomic, and the Otsu scan runs on the LDS bins -- no 256 KiB table per
// frame in HBM, no second reader, no second launch (256 x 1024^2: hist16 0.158 ms + otsu 0.064 ms before).
// Window placement: the caller's bounds (lo_hint <= every value <= hi_hint) when given, else the extrema of every 16th
// row, centred in the window (filtered frames are smooth: the sample misses the true extrema by a few counts, the
// slack on either side is hundreds to thousands).  A pixel outside the window, or a range wider than the window,
// sets flag[frame] = 1: the frame is then left to the two-kernel path, launched right behind and gated per frame by that flag.
#ifndef PL_OTSU_VARIANT
#define PL_OTSU_VARIANT 0
#endif
constexpr int kWinBins = 38912;   // 152 KiB
struct OtsuScratch {
  Pair wave_tot[kHistThreads / 64];
  double s_var[kHistThreads / 64];
  int s_lo[kHistThreads / 64], s_hi[kHistThreads / 64], s_idx[kHistThreads / 64];
  int any;
};
constexpr int kOtsuScratchAt = ((kWinBins + 1) * 4 + 15) & ~15;
constexpr size_t kOtsuLds = kOtsuScratchAt + sizeof(OtsuScratch);

// MED3: the histogram is that of the 3x3 MEDIAN of the frame (h x w, geometry of pl_median3_rows_covers), computed on the
// fly by pl_median3_rows -- the median plane is never written (the window is still placed from a sample of the raw frame: a
// median lies between the extrema of its window).  T = short / unsigned short decides how the median compares.
template <typename T, bool MED3>
__global__ void __launch_bounds__(kHistThreads)
otsu16_window_kernel(const unsigned short* __restrict__ in, int64_t count, int h, int w, unsigned flip, int bias,
                     const int32_t* __restrict__ lo_hint, const int32_t* __restrict__ hi_hint, int32_t* __restrict__ thr,
                     int32_t* __restrict__ vmin, int32_t* __restrict__ vmax, int32_t* __restrict__ flag, int parts,
                     uint32_t* __restrict__ merge /* parts > 1: [n][65536] zeroed; flag zeroed too */) {
  // ALL of the kernel's LDS is the dynamic block: the bins start at LDS address 0, so a bin's byte offset IS its address
  // (with static arrays in front the compiler spent one add per pixel on the base), the reduction scratch sits behind them
  extern __shared__ __attribute__((aligned(16))) unsigned bins[];  // kWinBins + 1, then OtsuScratch
  OtsuScratch& scr = *reinterpret_cast<OtsuScratch*>(reinterpret_cast<unsigned char*>(bins) + kOtsuScratchAt);
  Pair* const wave_tot = scr.wave_tot;
  int* const s_lo = scr.s_lo;
  int* const s_hi = scr.s_hi;
  double* const s_var = scr.s_var;
  int* const s_idx = scr.s_idx;
  // parts > 1 (small batches: one workgroup per frame would leave most of the chip idle): `parts` workgroups share a frame,
  // each tallies a band of its rows into its own LDS window (placed identically: same hints / same sample), the windows are
  // merged into the frame's table in global memory by returning device-scope atomics, and the part that arrives last runs the
  // scan on the merged counts.  No fences: every access to the table is a device-scope atomic or a device-coherent load.
  const int64_t frame = blockIdx.x / (unsigned)parts;
  const int part = (int)(blockIdx.x % (unsigned)parts);
  const unsigned short* src = in + frame * count;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int klo, khi;   // window = keys [klo, khi] (biased domain)
  if (lo_hint) {
    klo = lo_hint[frame] + bias;
    khi = hi_hint[frame] + bias;
  } else {
    // extrema of a 1/16 sample: blocks of 1024 pixels (128 x 16 bytes), every 16th block
    int mn = 1 << 30, mx = -1;
    auto see = [&](unsigned key) {
      const int k = (int)(key ^ flip);
      mn = k < mn ? k : mn;
      mx = k > mx ? k : mx;
    };
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0 && count >= 8) {
      const int64_t nvec = count / 8;
      const uint4* vsrc = reinterpret_cast<const uint4*>(src);
      for (int64_t blk = threadIdx.x >> 7; blk * 2048 < nvec; blk += kHistThreads >> 7) {
        const int64_t idx = blk * 2048 + (threadIdx.x & 127);
        if (idx < nvec) {
          const uint4 q = vsrc[idx];
          const unsigned wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) { see(wds[k] & 0xffffu); see(wds[k] >> 16); }
        }
      }
    } else {
      for (int64_t i = threadIdx.x; i < count; i += 16LL * kHistThreads) see(src[i]);
    }
    mn = pl_wave_reduce(mn, [](int a, int b) { return a < b ? a : b; });
    mx = pl_wave_reduce(mx, [](int a, int b) { return a > b ? a : b; });
    if (lane == 0) { s_lo[wv] = mn; s_hi[wv] = mx; }
    __syncthreads();
    for (int k = 0; k < kHistThreads / 64; ++k) { mn = s_lo[k] < mn ? s_lo[k] : mn; mx = s_hi[k] > mx ? s_hi[k] : mx; }
    __syncthreads();
    if (mx < mn) { mn = 0; mx = 0; }
    const int slack = kWinBins - (mx - mn + 1);
    klo = mn - (slack > 0 ? slack / 2 : 0);
    if (klo < 0) klo = 0;
    khi = klo + kWinBins - 1;
    if (khi > 65535) { khi = 65535; klo = khi - kWinBins + 1; }
    if (slack < 0) khi = klo + kWinBins;   // too wide: fails the range test below
  }
  const int range = khi - klo + 1;
  // (MED3 addresses the bins by absolute LDS address: they must start the workgroup's LDS -- else the two-kernel path)
  if (range > kWinBins || range <= 0 || klo < 0 || khi > 65535 || (MED3 && pl_lds_base(bins) != 0u)) {
    if (threadIdx.x == 0) flag[frame] = 1;                   // (every part of a frame decides the same)
    return;
  }
  for (int i = threadIdx.x; i <= range; i += kHistThreads) bins[i] = 0;      // + the spare bin at index `range`
  if (threadIdx.x == 0) scr.any = 0;
  __syncthreads();

  int outside = 0;
  auto tally = [&](unsigned key) {
    const unsigned b = (key ^ flip) - (unsigned)klo;
    if (b < (unsigned)range) atomicAdd(&bins[b], 1u);
    else outside = 1;
  };
  if (MED3) {
    // every wave walks (column block of 512, row group of 32) items; all 64 lanes take part in the median's cross-lane moves,
    // lanes beyond the frame's width tally nothing
    constexpr int kRows = 32;                        // rows per item: two halo rows are re-read per item
    constexpr int kSBias = ((T)-1 < (T)0) ? 32768 : 0;          // median (sign-extended for int16) -> key of the biased domain
    const unsigned kbase4 = 4u * (unsigned)(kSBias - klo);
#if PL_OTSU_VARIANT & 1
    unsigned dummy = 0;
#endif
    const int col_waves = (w / 8 + PL_WAVE - 1) / PL_WAVE, row_groups = (h + kRows - 1) / kRows;
    const int rg_lo = (int)((int64_t)row_groups * part / parts), rg_hi = (int)((int64_t)row_groups * (part + 1) / parts);
    for (int item = rg_lo * col_waves + __builtin_amdgcn_readfirstlane(wv); item < col_waves * rg_hi; item += kHistThreads / 64) {   // scalar
      const int c0 = ((item % col_waves) * PL_WAVE + lane) * 8;
      const bool on = c0 < w;
      // per item: which lanes hold columns of the frame (all of them unless w / 8 is not a multiple of 64)
      const unsigned long long act = __ballot(on);
      if (act == 0ull) continue;
      const int first_on = __builtin_ctzll(act);
      const unsigned cap4 = on ? 4u * (unsigned)range : 0u;      // lanes beyond the frame add 0 to bin 0
      const unsigned inc = on ? 1u : 0u;
      pl_median3_rows<T, kRows, PL_OTSU_AHEAD>(reinterpret_cast<const T*>(src), h, w, c0, lane, (item / col_waves) * kRows,
                             [&](int, const int (&m)[8]) {
        // one value in the whole wave (saturated / constant neighbourhoods): ONE atomic of 8 x the lane count -- 512 atomics
        // on one address would serialise
        unsigned spread = 0;
#pragma unroll
        for (int k = 1; k < 8; ++k) spread |= (unsigned)(m[k] ^ m[0]);
        const int wave_first = __builtin_amdgcn_readlane(m[0], first_on);
        if (__ballot(on && (spread | (unsigned)(m[0] ^ wave_first)) != 0u) == 0ull) {
          const unsigned b = (unsigned)(wave_first + kSBias - klo);
          if (b < (unsigned)range) {
            if (lane == first_on) atomicAdd(&bins[b], 8u * (unsigned)__popcll(act));
          } else {
            outside = 1;
          }
          return;
        }
        // branch-free tally: a value outside the window lands in the spare bin at index `range` (checked after the pass).
        // Per pixel: scale (x 4: the byte offset of the bin IS its LDS address) and subtract the window's base in one
        // instruction, clamp, one LDS atomic
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const unsigned b4 = ((unsigned)m[k] << 2) + kbase4;
          const unsigned a4 = b4 < cap4 ? b4 : cap4;
#if PL_OTSU_VARIANT & 1    // stopwatch only: everything but the LDS atomics
          dummy += a4;
#else
          pl_lds_add_abs(a4, inc);
#endif
        }
      });
    }
#if PL_OTSU_VARIANT & 1
    if (dummy == 0x12345678u) bins[0] = 1;
#endif
  } else if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    const int64_t nvec = count / 8;
    const uint4* vsrc = reinterpret_cast<const uint4*>(src);
    auto tally4 = [&](uint4 q) {
      // a wave that holds ONE value in this vector adds its whole count with one atomic (see hist16_kernel)
      const unsigned first = q.x & 0xffffu;
      const unsigned splat = first | (first << 16);
      const bool lane_flat = q.x == splat && q.y == splat && q.z == splat && q.w == splat;
      const unsigned wave_first = __builtin_amdgcn_readfirstlane(splat);
      const unsigned long long active = __ballot(1);                  // taken by ALL active lanes, before any lane-only branch
      if (__ballot(!(lane_flat && splat == wave_first)) == 0) {       // wave-uniform branch
        const unsigned b = ((wave_first & 0xffffu) ^ flip) - (unsigned)klo;
        if (b < (unsigned)range) {
          if ((threadIdx.x & 63) == __builtin_ctzll(active)) atomicAdd(&bins[b], 8u * (unsigned)__popcll(active));
        } else {
          outside = 1;
        }
        return;
      }
      const unsigned wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        tally(wds[k] & 0xffffu);
        tally(wds[k] >> 16);
      }
    };
    const int64_t v_lo = nvec * part / parts, v_hi = nvec * (part + 1) / parts;      // this part's share of the vectors
    int64_t v = v_lo + threadIdx.x;
    constexpr int U = 4;  // independent 16-byte loads in flight per lane
    for (; v + (int64_t)(U - 1) * kHistThreads < v_hi; v += (int64_t)U * kHistThreads) {
      uint4 q[U];
#pragma unroll
      for (int k = 0; k < U; ++k) q[k] = vsrc[v + (int64_t)k * kHistThreads];
#pragma unroll
      for (int k = 0; k < U; ++k) tally4(q[k]);
    }
    for (; v < v_hi; v += kHistThreads) tally4(vsrc[v]);
    if (part == parts - 1)
      for (int64_t i = nvec * 8 + threadIdx.x; i < count; i += kHistThreads) tally(src[i]);
  } else {
    const int64_t i_lo = count * part / parts, i_hi = count * (part + 1) / parts;
    for (int64_t i = i_lo + threadIdx.x; i < i_hi; i += kHistThreads) tally(src[i]);
  }
  // a pixel fell outside the window (or outside the caller's bounds): two-kernel path.  (__syncthreads_or would bring 256
  // bytes of static LDS in front of the bins.)  The branch-free tally counts out-of-window pixels in the spare bin.
  if (outside) scr.any = 1;
  __syncthreads();
  const bool spilled = scr.any != 0 || (MED3 && bins[range] != 0u);   // the same for every thread
  if (parts == 1) {
    if (spilled) {
      if (threadIdx.x == 0) flag[frame] = 1;
      return;
    }
    if (threadIdx.x == 0) flag[frame] = 0;
  } else {
    uint32_t* table = merge + frame * 65536;
    unsigned seen = 0;
    // RETURNING exchange, its value folded into `seen`: the wave waits for the flag write to be performed before it reaches
    // the barrier and thread 0 takes the arrival ticket, so the last part cannot read a stale 0
    if (spilled && threadIdx.x == 0) seen |= atomicExch(&flag[frame], 1) == 0x7fffffff ? 1u : 0u;
    if (!spilled)
      for (int i = threadIdx.x; i < range; i += kHistThreads) {
        const unsigned c = bins[i];
        if (c) seen |= atomicAdd(&table[i], c) == 0xffffffffu ? 1u : 0u;   // RETURNING: the wave waits for its adds
      }
    // (a count cannot reach 2^32 - 1: `seen` stays 0; it exists so that the barrier below consumes the returned values.  Not
    // __syncthreads_or: that brings static LDS in front of the bins, which must start at LDS address 0)
    if (seen) scr.any = 2;
    __syncthreads();
    if (threadIdx.x == 0) scr.any = atomicAdd(&table[65534], scr.any == 2 ? 0u : 1u) == (unsigned)(parts - 1) ? 1 : 0;
    __syncthreads();
    if (scr.any == 0) return;                                // not the last part of this frame
    if (__hip_atomic_load(&flag[frame], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;   // a part spilled: fallback
    for (int i = threadIdx.x; i < range; i += kHistThreads)
      bins[i] = __hip_atomic_load(&table[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  }

  // ---- Otsu on the window: lane t owns window bins [t*per, t*per + per); same integer prefix sums, float64 expression
  // and first-index arg-max as otsu_kernel
  // an ODD number of bins per lane: lane t's j-th bin sits in bank (per * t + j) mod 32, and with the even 38 of a full window
  // the 64 lanes of a read met in 16 banks (four-way conflicts on each of the 2 x 38 reads); 39 spreads them over all 32
#ifdef PL_OTSU_EVEN_PER
  const int per = (range + kHistThreads - 1) / kHistThreads;
#else
  const int per = ((range + kHistThreads - 1) / kHistThreads) | 1;
#endif
  const int b0 = threadIdx.x * per;
  const int b1 = b0 + per < range ? b0 + per : range;
  Pair mine = {0, 0};
  int lo = 1 << 30, hi = -1;
  for (int b = b0; b < b1; ++b) {
    const unsigned c = bins[b];
    mine.c += c;
    mine.s += (long long)c * (long long)(b + klo - bias);
    if (c) { if (lo == (1 << 30)) lo = b; hi = b; }
  }
  lo = pl_wave_reduce(lo, [](int a, int b) { return a < b ? a : b; });
  hi = pl_wave_reduce(hi, [](int a, int b) { return a > b ? a : b; });
  if (lane == 0) { s_lo[wv] = lo; s_hi[wv] = hi; }
  Pair total;
  Pair ex = block_exclusive_scan(mine, &total, wave_tot);  // contains __syncthreads
  for (int k = 0; k < kHistThreads / 64; ++k) { lo = s_lo[k] < lo ? s_lo[k] : lo; hi = s_hi[k] > hi ? s_hi[k] : hi; }

  double best = -1.0;
  int best_k = 1 << 30;
  // the running class sums as float64: every value is an integer below 2^53 (a frame's pixel count and its sum of 16-bit
  // keys), so `dw1 += c` and the fused `c * key + ds1` are EXACT and equal (double)w1 / (double)s1 of the integer prefix
  // sums bit for bit -- without the 64-bit multiply-add and the four 64-bit integer -> float64 conversions per bin
  double dw1 = (double)ex.c, ds1 = (double)ex.s;
  const double dW = (double)total.c, dS = (double)total.s;
  for (int b = b0; b < b1; ++b) {
    const unsigned c = bins[b];
    if (c != 0u) {                                // (an empty bin changes nothing)
      dw1 += (double)c;
      ds1 = fma((double)c, (double)(b + klo - bias), ds1);
      if (b >= lo && b < hi) {                    // empty bins: see otsu_kernel
        const double dw2 = dW - dw1;
        const double m1 = ds1 / dw1;
        const double m2 = (dS - ds1) / dw2;
        const double d = m1 - m2;
        const double var = (dw1 * dw2) * (d * d);
        if (var > best) { best = var; best_k = b; }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    double ov = __shfl_xor(best, o, 64);
    int ok = __shfl_xor(best_k, o, 64);
    if (ov > best || (ov == best && ok < best_k)) { best = ov; best_k = ok; }
  }
  if (lane == 0) { s_var[wv] = best; s_idx[wv] = best_k; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < kHistThreads / 64; ++k)
      if (s_var[k] > best || (s_var[k] == best && s_idx[k] < best_k)) { best = s_var[k]; best_k = s_idx[k]; }
    // constant image: skimage returns that value (thresholding.py: np.all(image == first_pixel))
    thr[frame] = (lo == hi) ? (lo + klo - bias) : (best_k + klo - bias);
    if (vmin) vmin[frame] = lo + klo - bias;
    if (vmax) vmax[frame] = hi + klo - bias;
  }
}

// ---- single-pass Otsu for frames the 38 912-bin window cannot hold: ALL 65 536 bins in LDS as packed 16-bit counters -------
// (a frame stretched to the full 16-bit range fell off a cliff -- gated median plane + two-part histogram + scan,
// 0.226 -> 0.489 ms per 256 frames; three one-read alternatives built on 32-bit bins were slower still.)
// 65 536 counters of 16 bits are 128 KiB: two per dword (values v and v + 32 768).  FIRST ATTEMPT (round 4, second half):
// plain 16-bit fields and non-returning adds, exactness CHECKED afterwards through the decoded total (full_tally_pass below);
// wave-wide flat neighbourhoods go to a small side list of 32-bit counts.  SECOND ATTEMPT, only for a frame whose total did
// not add up: one returning LDS atomic per pixel on fields of 15 bits; bit 15 is a GUARD: the add that finds the field at
// 0x7fff sets the guard (no carry can reach the neighbouring field: that would take 32 768 further adds before the fix
// below), and exactly that lane -- it alone saw 0x7fff come back -- subtracts 0x8000 again and notes the key in a short LDS
// list: every entry stands for 32 768 pixels of that value.  Adds that land between the two steps are preserved (the fix is
// a subtraction, not a store).  The Otsu scan reads the fields and adds the noted counts: same integer prefix sums, float64
// expression and first-index arg-max as otsu_kernel.
// One workgroup per frame, gated by the window kernel's flag; medians on the fly like the window kernel (MED3).
// Measured on 256 stretched 1024 x 1024 frames (profiles/r04_otsu_full_range_variants.txt): guard form 0.292-0.302 ms for the
// stage (window attempt included) -> 0.245 with the unguarded first attempt -> 0.225 with the bank-conflict-free scan.
constexpr int kFullEvents = 2048;                            // overflow notes: frames of up to 2^26 pixels
constexpr int kFullBulk = 64;                                // values that whole waves hold (flat background / saturation)
struct FullScratch {
  Pair wave_tot[kHistThreads / 64];
  double s_var[kHistThreads / 64];
  int s_lo[kHistThreads / 64], s_hi[kHistThreads / 64], s_idx[kHistThreads / 64];
  int n_events, n_bulk;
  unsigned bulk_key[kFullBulk], bulk_cnt[kFullBulk];
  unsigned short event_key[kFullEvents];
};
constexpr int kFullBinsBytes = 65536 * 2;
constexpr size_t kFullLds = kFullBinsBytes + sizeof(FullScratch);

// One pass over the frame into the packed fields.
//   GUARD = false (first attempt): fields of 16 bits, NON-returning adds (the returning add of the guarded form and the test
//     of what came back cost a third of the kernel), a wave's bulk count of a flat neighbourhood -- the one thing that
//     routinely exceeds 65 535 per value -- into a 64-entry side list of 32-bit counts.  A field that overflows all the same
//     carries into its neighbour or out of the dword; either way the decoded total comes out SMALLER than the pixel count
//     (-65 535 or -65 536 per overflow, never compensated), which the caller tests: exactness is checked, not assumed.
//   GUARD = true (the frame whose total did not add up: > 65 535 pixels of one value outside flat neighbourhoods): the
//     15-bit fields with the guard bit described above.
template <typename T, bool MED3, bool GUARD>
__device__ __forceinline__ void full_tally_pass(const unsigned short* __restrict__ src, int64_t count, int h, int w, unsigned flip,
                                                unsigned* __restrict__ bins, FullScratch& scr) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // key = value in the biased domain (0 .. 65535), n = how many pixels of it (1, or a whole wave's worth <= 512: the guard
  // bit leaves room for 32 767 more before a carry, far more than every wave's bulk add landing between the two steps)
  auto tally_n = [&](unsigned key, unsigned n) {
    // values v and v + 32 768 share a dword (not v and v + 1: neighbouring pixels hold neighbouring values, and two lanes
    // on one dword serialise in the LDS atomic unit)
    const unsigned sh = (key >> 15) << 4, at = key & 0x7fffu;
    if (!GUARD) {                // four vector instructions and the atomic; the bins start the workgroup's LDS (kernel's test)
#if PL_OTSU_VARIANT & 1            // stopwatch only: everything but the LDS atomics
      asm volatile("" ::"v"((key << 2) & 0x1fffcu), "v"(n * (1u + ((key >> 15) & 1u) * 0xffffu)));
#else
      pl_lds_add_abs((key << 2) & 0x1fffcu, n * (1u + ((key >> 15) & 1u) * 0xffffu));
#endif
      return;
    }
    const unsigned old = atomicAdd(&bins[at], n << sh);
    const unsigned f = (old >> sh) & 0xffffu;
    if (f < 0x8000u && f + n >= 0x8000u) {                   // this add took the field across the guard: fold 32 768 away
      atomicSub(&bins[at], 0x8000u << sh);
      const int slot = atomicAdd(&scr.n_events, 1);
      if (slot < kFullEvents) scr.event_key[slot] = (unsigned short)key;
    }
  };
  auto tally = [&](unsigned key) { tally_n(key, 1u); };
  // a flat wave's count (called by the whole, converged wave; `key`, `n` wave-uniform)
  auto tally_bulk = [&](unsigned key, unsigned n, int first_on) {
    if (GUARD) {
      if (lane == first_on) tally_n(key, n);
      return;
    }
    const int have = scr.n_bulk < kFullBulk ? scr.n_bulk : kFullBulk;
    const unsigned long long hit = __ballot(lane < have && scr.bulk_key[lane] == key);
    if (lane != first_on) return;
    if (hit) { atomicAdd(&scr.bulk_cnt[__builtin_ctzll(hit)], n); return; }
    const int slot = atomicAdd(&scr.n_bulk, 1);              // (two waves may open the same value twice: both entries count)
    if (slot < kFullBulk) { atomicAdd(&scr.bulk_cnt[slot], n); scr.bulk_key[slot] = key; }
    else tally_n(key, n);                                    // list full: the field (an overflow there shows in the total)
  };
  if (MED3) {
    constexpr int kRows = 32;
    constexpr int kSBias = ((T)-1 < (T)0) ? 32768 : 0;
    const int col_waves = (w / 8 + PL_WAVE - 1) / PL_WAVE, row_groups = (h + kRows - 1) / kRows;
    for (int item = __builtin_amdgcn_readfirstlane(wv); item < col_waves * row_groups; item += kHistThreads / 64) {
      const int c0 = ((item % col_waves) * PL_WAVE + lane) * 8;
      const bool on = c0 < w;
      const unsigned long long act = __ballot(on);
      if (act == 0ull) continue;
      const int first_on = __builtin_ctzll(act);
      pl_median3_rows<T, kRows, PL_OTSU_AHEAD>(reinterpret_cast<const T*>(src), h, w, c0, lane, (item / col_waves) * kRows,
                             [&](int, const int (&m)[8]) {
        // one value in the whole wave (saturated / constant neighbourhoods): one add of the wave's count
        unsigned spread = 0;
#pragma unroll
        for (int k = 1; k < 8; ++k) spread |= (unsigned)(m[k] ^ m[0]);
        const int wave_first = __builtin_amdgcn_readlane(m[0], first_on);
        if (__ballot(on && (spread | (unsigned)(m[0] ^ wave_first)) != 0u) == 0ull) {
          tally_bulk((unsigned)(wave_first + kSBias), 8u * (unsigned)__popcll(act), first_on);
          return;
        }
        if (!on) return;
#pragma unroll
        for (int k = 0; k < 8; ++k) tally((unsigned)(m[k] + kSBias));
      });
    }
  } else {
    for (int64_t i = threadIdx.x; i < count; i += kHistThreads) tally((unsigned)src[i] ^ flip);
  }
}

template <typename T, bool MED3>
__global__ void __launch_bounds__(kHistThreads)
otsu16_full_kernel(const unsigned short* __restrict__ in, int64_t count, int h, int w, unsigned flip, int bias,
                   int32_t* __restrict__ thr, int32_t* __restrict__ vmin, int32_t* __restrict__ vmax,
                   int32_t* __restrict__ flag) {
  extern __shared__ __attribute__((aligned(16))) unsigned bins[];  // 32 768 dwords = 65 536 fields, then FullScratch
  FullScratch& scr = *reinterpret_cast<FullScratch*>(reinterpret_cast<unsigned char*>(bins) + kFullBinsBytes);
  const int64_t frame = blockIdx.x;
#ifndef PL_OTSU_FULL_ALWAYS
  if (flag[frame] == 0) return;                              // the window kernel finished this frame
#endif
  const unsigned short* src = in + frame * count;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  auto clear = [&]() {
    for (int i = threadIdx.x; i < 32768; i += kHistThreads) bins[i] = 0;
    if (threadIdx.x < kFullBulk) { scr.bulk_key[threadIdx.x] = 0xffffffffu; scr.bulk_cnt[threadIdx.x] = 0; }
    if (threadIdx.x == 0) { scr.n_events = 0; scr.n_bulk = 0; }
    __syncthreads();
  };

  // ---- Otsu on the 65 536 bins: lane t owns bins [64 t, 64 t + 64)
  const int b0 = threadIdx.x * 64;
  int n_events = 0, n_bulk = 0, my_notes = 0;
  auto count_of = [&](int b) {
    unsigned c = (bins[b & 0x7fff] >> ((b >> 15) << 4)) & 0xffffu;
    if (my_notes) {                                          // rare, so a lane without any skips the lookups
      for (int e = 0; e < n_events; ++e) c += (int)scr.event_key[e] == b ? 32768u : 0u;
      for (int e = 0; e < n_bulk; ++e) c += (int)scr.bulk_key[e] == b ? scr.bulk_cnt[e] : 0u;
    }
    return c;
  };
  // Lane t reads ITS 64 fields starting at the t-th (cyclically): at every step the 64 lanes of a wave are on 64 different
  // banks.  In value order -- every lane at its first field -- they would all be on ONE bank (a lane's fields are 64 dwords
  // apart): 2 x 64 reads of 64 cycles each per wave, a fifth of this kernel's time before round 4 noticed.  The sums do not
  // care about the order; the running prefix of the second pass starts behind the wrapped part (`wrapped`) and is reset
  // where the walk wraps.
  Pair mine, total, ex, wrapped;
  int lo, hi;
  auto totals = [&]() {                                      // contains __syncthreads
    n_events = scr.n_events < kFullEvents ? scr.n_events : kFullEvents;   // (more: a frame beyond 2^26 pixels, refused by the launcher)
    n_bulk = scr.n_bulk < kFullBulk ? scr.n_bulk : kFullBulk;
    my_notes = 0;
    for (int e = 0; e < n_events; ++e) my_notes += ((int)scr.event_key[e] >> 6) == (int)threadIdx.x ? 1 : 0;
    for (int e = 0; e < n_bulk; ++e) my_notes += (scr.bulk_key[e] >> 6) == threadIdx.x ? 1 : 0;
    mine = {0, 0};
    wrapped = {0, 0};
    lo = 1 << 30, hi = -1;
    for (int j = 0; j < 64; ++j) {
      const int b = b0 + ((j + lane) & 63);
      const unsigned c = count_of(b);
      Pair& acc = j + lane < 64 ? mine : wrapped;
      acc.c += c;
      acc.s += (long long)c * (long long)(b - bias);
      if (c) { lo = b < lo ? b : lo; hi = b > hi ? b : hi; }
    }
    mine.c += wrapped.c;
    mine.s += wrapped.s;
    lo = pl_wave_reduce(lo, [](int a, int b) { return a < b ? a : b; });
    hi = pl_wave_reduce(hi, [](int a, int b) { return a > b ? a : b; });
    if (lane == 0) { scr.s_lo[wv] = lo; scr.s_hi[wv] = hi; }
    ex = block_exclusive_scan(mine, &total, scr.wave_tot);
    for (int k = 0; k < kHistThreads / 64; ++k) { lo = scr.s_lo[k] < lo ? scr.s_lo[k] : lo; hi = scr.s_hi[k] > hi ? scr.s_hi[k] : hi; }
  };

  bool exact = false;
  if (pl_lds_base(bins) == 0u) {                             // (the fast pass addresses the bins by absolute LDS address)
    clear();
    full_tally_pass<T, MED3, false>(src, count, h, w, flip, bins, scr);
    __syncthreads();
    totals();
    exact = total.c == (unsigned long long)count;
    __syncthreads();
  }
  if (!exact) {                                              // a 16-bit field overflowed: once more with the guard bit
    clear();
    full_tally_pass<T, MED3, true>(src, count, h, w, flip, bins, scr);
    __syncthreads();
    totals();
  }
  double best = -1.0;
  int best_k = 1 << 30;
  // (float64 running sums of integers below 2^53: exact, see otsu16_window_kernel)
  double dw1 = (double)(ex.c + wrapped.c);                   // the prefix in front of field `lane`, where the walk starts
  double ds1 = (double)(ex.s + wrapped.s);
  const double dW = (double)total.c, dS = (double)total.s;
  for (int j = 0; j < 64; ++j) {
    if (j + lane == 64) { dw1 = (double)ex.c; ds1 = (double)ex.s; }   // wrapped to the lane's first field
    const int b = b0 + ((j + lane) & 63);
    const unsigned c = count_of(b);
    if (c != 0u) {
      dw1 += (double)c;
      ds1 = fma((double)c, (double)(b - bias), ds1);
      if (b >= lo && b < hi) {                    // empty bins: see otsu_kernel
        const double dw2 = dW - dw1;
        const double m1 = ds1 / dw1;
        const double m2 = (dS - ds1) / dw2;
        const double d = m1 - m2;
        const double var = (dw1 * dw2) * (d * d);
        if (var > best || (var == best && b < best_k)) { best = var; best_k = b; }   // first index of the maximum
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    double ov = __shfl_xor(best, o, 64);
    int ok = __shfl_xor(best_k, o, 64);
    if (ov > best || (ov == best && ok < best_k)) { best = ov; best_k = ok; }
  }
  if (lane == 0) { scr.s_var[wv] = best; scr.s_idx[wv] = best_k; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < kHistThreads / 64; ++k)
      if (scr.s_var[k] > best || (scr.s_var[k] == best && scr.s_idx[k] < best_k)) { best = scr.s_var[k]; best_k = scr.s_idx[k]; }
    thr[frame] = (lo == hi) ? (lo - bias) : (best_k - bias);
    if (vmin) vmin[frame] = lo - bias;
    if (vmax) vmax[frame] = hi - bias;
  }
}

__global__ void __launch_bounds__(kHistThreads)
order_stats_kernel(const uint32_t* __restrict__ hist, int bias, const int64_t* __restrict__ ranks,
                   int nranks, int32_t* __restrict__ out) {
  __shared__ Pair wave_tot[kHistThreads / 64];
  const int64_t frame = blockIdx.x;
  const uint32_t* hh = hist + frame * 65536;
  const int b0 = threadIdx.x * 64;
  Pair mine = {0, 0};
  for (int k = 0; k < 64; ++k) mine.c += hh[b0 + k];
  Pair total;
  Pair ex = block_exclusive_scan(mine, &total, wave_tot);
  for (int q = 0; q < nranks; ++q) {
    long long r = ranks[q];
    if (r < 0) r = 0;
    if ((unsigned long long)r >= total.c) r = (long long)total.c - 1;
    if ((unsigned long long)r >= ex.c && (unsigned long long)r < ex.c + mine.c) {
      unsigned long long acc = ex.c;
      for (int k = 0; k < 64; ++k) {
        acc += hh[b0 + k];
        if ((unsigned long long)r < acc) { out[frame * nranks + q] = b0 + k - bias; break; }
      }
    }
  }
}


// ---- exact 65 536-bin histogram in ONE read of the frame: two LDS windows + global atomics for the rest -----------------
// hist16_kernel keeps half (or a quarter) of the bins in LDS and reads the frame once per part.  Radiographs are bimodal --
// an unexposed background and an exposed field, few pixels in between -- so here ONE workgroup per frame holds TWO windows
// of 19 456 consecutive bins (2 x 76 KiB), placed at the two ends of the value range seen in a 1/16 row sample (one
// contiguous 38 912-bin window when the range is that narrow); a pixel that falls into neither goes to the frame's table
// in HBM with a global atomic (exact whatever the sample missed; only slower when many pixels do).  A wave that holds a
// single value adds its whole count with one atomic (hist16_kernel's flat test); beyond that every wave PEELS a hot value:
// lanes whose pixel equals the wave's current guess L count it in a register instead of the LDS atomic unit -- the clipped
// dark noise of a real detector puts half of the background on one value, 30-way same-address serialisation per atomic
// instruction (the noisy Winston-Lutz frames: 0.90 ms per 256 frames with hist16_kernel<2> against 0.47 without noise).
// The guess is replaced by the first lane's pixel whenever it attracted less than 1/16 of the last 64 pixels per lane.
// Two instantiations.  19 456 bins per window (152 KiB: one workgroup per CU) for frames nobody knows anything about: two
// windows then cover 59 % of the 16-bit range.  9 728 (76 KiB: TWO workgroups per CU, one's prologue / epilogue and atomic
// stalls under the other's loads) for pl_hist16_wl, whose frames are Winston-Lutz images -- a flat background, a flat field
// and a few thousand penumbra pixels: r06x2, 1 250 frames: with dark-current noise 2.14 -> 1.83 ms per pass, noise-free
// unchanged (1.51); at 256 frames, one workgroup per CU either way, the small windows are 4 % slower: taken only for batches
// of more frames than the chip has CUs.
constexpr int kTwBinsWide = 19456, kTwBinsWl = 9728;
#ifndef PL_TW_TIMING
#define PL_TW_TIMING 0   // 1: thread 0 leaves wall_clock64() stamps of the phases in bins 65520.. of the frame's table (timing builds only)
#endif
#if PL_TW_TIMING
#define TW_STAMP(k) do { __syncthreads(); if (threadIdx.x == 0) tw_stamp[k] = wall_clock64(); } while (0)
#else
#define TW_STAMP(k) do { } while (0)
#endif
struct TwScratch { int s_lo[kHistThreads / 64], s_hi[kHistThreads / 64]; };
constexpr int tw_scratch_at(int bins) { return (2 * bins + 1 + PL_WAVE + 3) / 4 * 16; }   // two windows, the spare bin, one dummy bin per lane
constexpr size_t tw_lds(int bins) { return tw_scratch_at(bins) + sizeof(TwScratch); }

template <int kTwBins, int WAVES_PER_EU>
__global__ void __launch_bounds__(kHistThreads, WAVES_PER_EU)
hist16_two_window_kernel(const unsigned short* __restrict__ in, int64_t count, unsigned flip, uint32_t* __restrict__ hist,
                         unsigned short* __restrict__ tile_max /* optional [n][ceil(count / 512)] */, int eh, int ew, int ews,
                         int32_t* __restrict__ edge_min, int32_t* __restrict__ edge_max /* optional: pl_hist16_wl */,
                         const int64_t* __restrict__ ranks, int nranks, int32_t* __restrict__ stats /* optional: pl_hist16_wl */) {
  extern __shared__ __attribute__((aligned(16))) unsigned bins[];  // 2 * kTwBins, then TwScratch
  TwScratch& scr = *reinterpret_cast<TwScratch*>(reinterpret_cast<unsigned char*>(bins) + tw_scratch_at(kTwBins));
  const int64_t frame = blockIdx.x;
  const unsigned short* src = in + frame * count;
  uint32_t* row = hist + frame * 65536;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool vec = (reinterpret_cast<uintptr_t>(src) & 15) == 0 && count >= 8;
  const int64_t nvec = vec ? count / 8 : 0;
  const uint4* vsrc = reinterpret_cast<const uint4*>(src);

#if PL_TW_TIMING
  unsigned long long tw_stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  TW_STAMP(0);
  for (int i = threadIdx.x; i < 2 * kTwBins; i += kHistThreads) bins[i] = 0;
  if (edge_min) {
    // min / max over the four `ews`-wide edge strips of the eh x ew frame (WLBaseImage._clean_edges' edge test,
    // pylinac/winston_lutz.py:1109-1133; pl_edge_minmax's loops): a few scattered loads per thread, all issued before the
    // first is looked at -- as a kernel of its own (one 256-thread workgroup per frame walking them sixteen deep) this was
    // 30 us per 512 frames of pure latency (r05z)
    int mn = 0x7fffffff, mx = -0x7fffffff - 1;
    auto value = [&](unsigned short raw) { return flip ? (int)(short)raw : (int)raw; };
    auto see = [&](int v) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; };
    const int band = ews < eh ? ews : eh, cb = ews < ew ? ews : ew;
    constexpr int E = 4;
    for (int e0 = threadIdx.x; e0 < band * ew; e0 += E * kHistThreads) {        // top and bottom strips
      unsigned short a[E], b[E];
#pragma unroll
      for (int k = 0; k < E; ++k) {
        const int e = e0 + k * kHistThreads < band * ew ? e0 + k * kHistThreads : e0;
        a[k] = src[e];
        b[k] = src[(int64_t)(eh - band) * ew + e];
      }
#pragma unroll
      for (int k = 0; k < E; ++k) { see(value(a[k])); see(value(b[k])); }
    }
    for (int e0 = threadIdx.x; e0 < eh * cb; e0 += E * kHistThreads) {          // left and right strips
      unsigned short a[E], b[E];
#pragma unroll
      for (int k = 0; k < E; ++k) {
        const int e = e0 + k * kHistThreads < eh * cb ? e0 + k * kHistThreads : e0;
        const int r = e / cb, c = e - r * cb;
        a[k] = src[(int64_t)r * ew + c];
        b[k] = src[(int64_t)r * ew + (ew - cb) + c];
      }
#pragma unroll
      for (int k = 0; k < E; ++k) { see(value(a[k])); see(value(b[k])); }
    }
    mn = pl_wave_reduce(mn, [](int a, int b) { return a < b ? a : b; });
    mx = pl_wave_reduce(mx, [](int a, int b) { return a > b ? a : b; });
    if (lane == 0) { scr.s_lo[wv] = mn; scr.s_hi[wv] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int k = 0; k < kHistThreads / 64; ++k) { mn = scr.s_lo[k] < mn ? scr.s_lo[k] : mn; mx = scr.s_hi[k] > mx ? scr.s_hi[k] : mx; }
      edge_min[frame] = mn;
      edge_max[frame] = mx;
    }
    __syncthreads();                                                  // (the scratch is used again for the sample's extrema)
  }
  TW_STAMP(1);
  // tile maxima (pl_hist16_tiles): tile t = pixels [512 t, 512 t + 512) = the 64 vectors ONE wave load of the main loop
  // fetches; its largest key (biased domain) lets a later pass skip every tile that cannot hold a pixel above its
  // threshold (pl_field_cax_tiles).  Tiles the main loop does not cover keep 0xffff ("look inside").
  const int64_t ntiles = (count + 511) / 512;
  unsigned short* const tmax = tile_max ? tile_max + frame * ntiles : nullptr;
  if (tmax)
    for (int64_t i = threadIdx.x; i < ntiles; i += kHistThreads) tmax[i] = (unsigned short)0xffffu;
  // extrema of a 1/16 sample (blocks of 1024 pixels, every 16th block), as otsu16_window_kernel
  int mn = 1 << 30, mx = -1;
  auto see = [&](unsigned key) {
    const int k = (int)(key ^ flip);
    mn = k < mn ? k : mn;
    mx = k > mx ? k : mx;
  };
  if (vec) {
    for (int64_t blk = threadIdx.x >> 7; blk * 2048 < nvec; blk += kHistThreads >> 7) {
      const int64_t idx = blk * 2048 + (threadIdx.x & 127);
      if (idx < nvec) {
        const uint4 q = vsrc[idx];
        const unsigned wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { see(wds[k] & 0xffffu); see(wds[k] >> 16); }
      }
    }
  } else {
    for (int64_t i = threadIdx.x; i < count; i += 16LL * kHistThreads) see(src[i]);
  }
  mn = pl_wave_reduce(mn, [](int a, int b) { return a < b ? a : b; });
  mx = pl_wave_reduce(mx, [](int a, int b) { return a > b ? a : b; });
  if (lane == 0) { scr.s_lo[wv] = mn; scr.s_hi[wv] = mx; }
  __syncthreads();
  for (int k = 0; k < kHistThreads / 64; ++k) { mn = scr.s_lo[k] < mn ? scr.s_lo[k] : mn; mx = scr.s_hi[k] > mx ? scr.s_hi[k] : mx; }
  if (mx < mn) { mn = 0; mx = 0; }
  TW_STAMP(2);
  // window bases (biased key domain): contiguous [w0, w0 + 2 W) around a narrow range, else one window at each end
  int w0, w1;
  if (mx - mn + 1 <= 2 * kTwBins) {
    w0 = mn - (2 * kTwBins - (mx - mn + 1)) / 2;
    w0 = w0 < 0 ? 0 : (w0 > 65536 - 2 * kTwBins ? 65536 - 2 * kTwBins : w0);
    w1 = w0 + kTwBins;
  } else {
    w0 = mn - 256 < 0 ? 0 : mn - 256;                // the sample misses the true extrema by a little
    w1 = mx + 256 > 65535 ? 65536 - kTwBins : mx + 257 - kTwBins;
    if (w1 < w0 + kTwBins) w1 = w0 + kTwBins;        // (cannot happen for a range wider than two windows; keeps them disjoint)
  }
  const unsigned uw0 = (unsigned)w0, uw1 = (unsigned)w1;
  // the table's bins OUTSIDE the windows start as zeros (they only ever see the global atomics below; the windows' bins are
  // stored whole at the end): quads of four bins, a quad that lies entirely inside a window is skipped
  for (int i = threadIdx.x; i < 65536 / 4; i += kHistThreads) {
    const unsigned b = 4u * (unsigned)i;
    const bool inside = (b - uw0 < (unsigned)kTwBins && b + 3u - uw0 < (unsigned)kTwBins) ||
                        (b - uw1 < (unsigned)kTwBins && b + 3u - uw1 < (unsigned)kTwBins);
    if (!inside) reinterpret_cast<uint4*>(row)[i] = uint4{0u, 0u, 0u, 0u};
  }
  __syncthreads();                                   // the zeroed table is in place before any global atomic
  TW_STAMP(3);

  unsigned hot = 0;                                  // this lane's pixels that equalled the wave's guess since the last flush
  unsigned guess = 0xffffffffu;                      // wave-uniform key (biased domain), none yet
  auto add_key = [&](unsigned key, unsigned n) {     // n pixels of one key, from one lane (the rare paths)
    const unsigned b0 = key - uw0, b1 = key - uw1;
    if (b0 < (unsigned)kTwBins) atomicAdd(&bins[b0], n);
    else if (b1 < (unsigned)kTwBins) atomicAdd(&bins[kTwBins + b1], n);
    else atomicAdd(row + key, n);
  };
  auto flush = [&]() {                               // the wave's count of its guess -> one atomic
    const unsigned tot = pl_wave_reduce(hot, [](unsigned a, unsigned b) { return a + b; });
    if (tot != 0u && lane == 0) add_key(guess, tot);
    hot = 0;
    return tot;
  };
  // Branch-free per pixel: ONE LDS atomic at an absolute LDS byte address -- the pixel's bin in window 0 or 1, a spare bin for
  // pixels outside both (those are re-walked with global atomics when a vector has any), a per-lane dummy bin for pixels that
  // equal the wave's guess (an add of 0 to the guess's bin would still queue on that address).  Twelve vector instructions.
  const unsigned base = pl_lds_base(bins);
  const unsigned k0 = base - 4u * uw0, k1 = base + 4u * (unsigned)kTwBins - 4u * uw1;
  const unsigned spare_addr = base + 4u * (unsigned)(2 * kTwBins), dummy_addr = spare_addr + 4u + 4u * (unsigned)lane;
  auto tally = [&](unsigned raw, unsigned long long& outside) {
    const unsigned key = raw ^ flip;
    const bool in0 = key - uw0 < (unsigned)kTwBins, in1 = key - uw1 < (unsigned)kTwBins, is_hot = key == guess;
    unsigned addr = in0 ? (key << 2) + k0 : (in1 ? (key << 2) + k1 : spare_addr);
    addr = is_hot ? dummy_addr : addr;
    hot += is_hot ? 1u : 0u;
    pl_lds_add_abs(addr, 1u);
    outside |= __ballot(!(in0 || in1 || is_hot));
  };
  auto tally4 = [&](uint4 q) {
    // a wave that holds ONE value in this vector adds its whole count with one atomic (see hist16_kernel)
    const unsigned first = q.x & 0xffffu;
    const unsigned splat = first | (first << 16);
    const bool lane_flat = q.x == splat && q.y == splat && q.z == splat && q.w == splat;
    const unsigned wave_first = __builtin_amdgcn_readfirstlane(splat);
    const unsigned long long active = __ballot(1);                  // taken by ALL active lanes, before any lane-only branch
    if (__ballot(!(lane_flat && splat == wave_first)) == 0) {       // wave-uniform branch
      if ((threadIdx.x & 63) == __builtin_ctzll(active)) add_key((wave_first & 0xffffu) ^ flip, 8u * (unsigned)__popcll(active));
      return;
    }
    const unsigned wds[4] = {q.x, q.y, q.z, q.w};
    unsigned long long outside = 0ull;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      tally(wds[k] & 0xffffu, outside);
      tally(wds[k] >> 16, outside);
    }
    if (outside != 0ull) {                                          // wave-uniform, rare: the pixels between the windows
#pragma unroll 1
      for (int k = 0; k < 8; ++k) {
        const unsigned key = ((k & 1) ? wds[k >> 1] >> 16 : wds[k >> 1] & 0xffffu) ^ flip;
        if (key - uw0 >= (unsigned)kTwBins && key - uw1 >= (unsigned)kTwBins && key != guess) atomicAdd(row + key, 1u);
      }
    }
  };
  auto tally1 = [&](unsigned raw) {                                 // tail pixels of a frame whose size is not a multiple of 8
    const unsigned key = raw ^ flip;
    if (key == guess) ++hot; else add_key(key, 1u);
  };
  int64_t v = threadIdx.x;
  constexpr int U = 8;  // independent 16-byte loads in flight per lane (one workgroup per CU: 128 KiB in flight)
  // the trip count is the WAVE's (its last lane decides): the wave-level steps inside never run under divergence
  for (; v - lane + (PL_WAVE - 1) + (int64_t)(U - 1) * kHistThreads < nvec; v += (int64_t)U * kHistThreads) {
    uint4 q[U];
#pragma unroll
    for (int k = 0; k < U; ++k) q[k] = vsrc[v + (int64_t)k * kHistThreads];
    if (tmax) {                                                     // wave-uniform
      const unsigned f2 = flip | (flip << 16);
#pragma unroll
      for (int k = 0; k < U; ++k) {
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        auto pmax = [](unsigned a, unsigned b) {
          const us2 r = __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b));
          return __builtin_bit_cast(unsigned, r);
        };
        const unsigned m2 = pmax(pmax(q[k].x ^ f2, q[k].y ^ f2), pmax(q[k].z ^ f2, q[k].w ^ f2));
        int m = (int)((m2 & 0xffffu) > (m2 >> 16) ? (m2 & 0xffffu) : (m2 >> 16));
        m = pl_wave_reduce_idem(m, [](int a, int b) { return a > b ? a : b; });
        if (lane == 0) tmax[(v - lane + (int64_t)k * kHistThreads) >> 6] = (unsigned short)m;
      }
    }
#pragma unroll
    for (int k = 0; k < U; ++k) tally4(q[k]);
    // 64 pixels per lane later: keep a guess that attracted >= 1/16 of them, else try the first lane's latest pixel
    const unsigned got = flush();
    if (got < 256u) guess = __builtin_amdgcn_readfirstlane((q[U - 1].w >> 16) ^ flip);
  }
  for (; v < nvec; v += kHistThreads) {              // fewer than 8192 vectors are left: no wave-level steps under divergence
    const uint4 q = vsrc[v];
    const unsigned wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { tally1(wds[k] & 0xffffu); tally1(wds[k] >> 16); }
  }
  for (int64_t i = nvec * 8 + threadIdx.x; i < count; i += kHistThreads) tally1(src[i]);
  flush();
  __syncthreads();
  TW_STAMP(4);
  if (ranks) {
    // pl_hist16_wl: the order statistics are taken HERE, from the windows while they are still in LDS (order_stats_kernel's
    // selection: thread t owns bins [64 t, 64 t + 64), exclusive scan of the counts, the bin where the running count passes
    // the rank) -- the windows are never stored and no second launch reads the 256 KiB table back (r05z: 55 us per 512
    // frames).  Bins outside the windows come from the frame's table (zeros + what the global atomics added: device-coherent
    // loads).  The lanes of a wave start at different bins of their 64 (the sum does not care): a common start is a 64-way
    // bank conflict on every read.
    __shared__ Pair wave_tot[kHistThreads / 64];
    const int b0 = threadIdx.x * 64;
    const unsigned d0 = (unsigned)b0 - uw0, d1 = (unsigned)b0 - uw1;
    const bool all0 = d0 < (unsigned)kTwBins && d0 + 63u < (unsigned)kTwBins;
    const bool all1 = d1 < (unsigned)kTwBins && d1 + 63u < (unsigned)kTwBins;
    // the table's bins were written by this workgroup's own stores and L2 atomics (complete at the barrier above); the
    // invalidate makes the plain vector loads below see L2, not a line of this CU's L1
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // half of a thread's 64 counts, in bin order, in registers.  Threads whose bins lie in a window read LDS; the others take
    // the table's bytes with eight loads issued together (r05f walked them one dependent load at a time, and the bins outside
    // the windows are 40 % of the table: +140 us per 512 frames) and patch in whatever part of a window they straddle
    auto counts = [&](int half, unsigned (&val)[32]) {
      if (all0 || all1) {
        const unsigned* p = (all0 ? bins + d0 : bins + kTwBins + d1) + 32 * half;
#pragma unroll
        for (int k = 0; k < 32; ++k) val[k] = p[k];
      } else {
        const uint4* g = reinterpret_cast<const uint4*>(row + b0 + 32 * half);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint4 q = g[j];
          val[4 * j] = q.x; val[4 * j + 1] = q.y; val[4 * j + 2] = q.z; val[4 * j + 3] = q.w;
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          const unsigned e0 = d0 + (unsigned)(32 * half + k), e1 = d1 + (unsigned)(32 * half + k);
          if (e0 < (unsigned)kTwBins) val[k] = bins[e0];
          else if (e1 < (unsigned)kTwBins) val[k] = bins[kTwBins + e1];
        }
      }
    };
    unsigned both = 0;
    if (all0 || all1) {
      // the sum does not care about the order: lanes start at different bins of their 64 (a common start is a 64-way bank
      // conflict on every read)
      const unsigned* p = all0 ? bins + d0 : bins + kTwBins + d1;
      for (int k = 0; k < 64; ++k) both += p[(k + lane) & 63];
    } else {
#pragma unroll 1
      for (int half = 0; half < 2; ++half) {
        unsigned val[32];
        counts(half, val);
#pragma unroll
        for (int k = 0; k < 32; ++k) both += val[k];
      }
    }
    TW_STAMP(5);
    Pair mine = {both, 0};
    Pair total;
    const Pair ex = block_exclusive_scan(mine, &total, wave_tot);
    TW_STAMP(6);
    const int bias = (int)flip;                                       // 0x8000 for int16 keys
    auto clamped = [&](int q) {
      long long r = ranks[q];
      if (r < 0) r = 0;
      if ((unsigned long long)r >= total.c) r = (long long)total.c - 1;
      return (unsigned long long)r;
    };
    bool owner = false;
    for (int q = 0; q < nranks; ++q) {
      const unsigned long long r = clamped(q);
      owner = owner || (r >= ex.c && r < ex.c + mine.c);
    }
    if (owner) {                                                      // a handful of threads of the workgroup
      unsigned before = 0;                                            // the counts of this thread's bins in front of the half
#pragma unroll 1
      for (int half = 0; half < 2; ++half) {
        unsigned val[32];
        counts(half, val);
        unsigned here = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k) here += val[k];
        const unsigned long long lo = ex.c + before, hi = lo + here;
        for (int q = 0; q < nranks; ++q) {
          const unsigned long long r = clamped(q);
          if (r >= lo && r < hi) {
            unsigned acc = 0;                                         // < 2^32: bins of one frame
            const unsigned want = (unsigned)(r - lo);
            int at = 31;
            bool found = false;
#pragma unroll
            for (int k = 0; k < 32; ++k) {
              acc += val[k];
              if (!found && want < acc) { at = k; found = true; }
            }
            stats[frame * nranks + q] = b0 + 32 * half + at - bias;
          }
        }
        before += here;
      }
    }
#if PL_TW_TIMING
    TW_STAMP(7);
    if (threadIdx.x == 0)
      for (int k = 0; k < 8; ++k) row[65520 + k] = (uint32_t)(tw_stamp[k] - tw_stamp[0]);   // 10 ns units (100 MHz)
#endif
    return;
  }
  // the windows go out with plain stores: no global atomic ever touched a bin inside a window
  for (int i = threadIdx.x; i < kTwBins; i += kHistThreads) {
    row[w0 + i] = bins[i];
    row[w1 + i] = bins[kTwBins + i];
  }
}

}  // namespace

struct HistEdges { int h, w, window; int32_t *d_min, *d_max; const int64_t* d_ranks; int nranks; int32_t* d_stats; };
static int hist16_impl(const void* in, int dtype, int64_t n, int64_t count, uint32_t* d_hist, uint16_t* d_tile_max,
                       HistEdges edges, void* stream);

extern "C" int pl_hist16(const void* in, int dtype, int64_t n, int64_t count, uint32_t* d_hist, void* stream) {
  return hist16_impl(in, dtype, n, count, d_hist, nullptr, HistEdges{0, 0, 0, nullptr, nullptr, nullptr, 0, nullptr}, stream);
}

extern "C" int pl_hist16_tiles(const void* in, int dtype, int64_t n, int64_t count, uint32_t* d_hist, uint16_t* d_tile_max,
                               void* stream) {
  PL_REQUIRE(d_tile_max, "null pointer");
  return hist16_impl(in, dtype, n, count, d_hist, d_tile_max, HistEdges{0, 0, 0, nullptr, nullptr, nullptr, 0, nullptr}, stream);
}

extern "C" int pl_edge_minmax(const void* in, int dtype, int64_t n, int h, int w, int window, int32_t* d_min, int32_t* d_max,
                              void* stream);

/* pl_hist16_tiles + pl_edge_minmax in the one launch (what the per-image half of WLBaseImage.analyze asks of a frame before
 * its decisions: pylinac/winston_lutz.py:709-712, 775, 1109-1133) */
extern "C" int pl_order_stats_from_hist(const uint32_t* d_hist, int dtype, int64_t n, const int64_t* d_ranks, int nranks,
                                        int32_t* d_out, void* stream);

extern "C" int pl_hist16_wl(const void* in, int dtype, int64_t n, int h, int w, uint32_t* d_hist, uint16_t* d_tile_max,
                            int edge_window, int32_t* d_edge_min, int32_t* d_edge_max, const int64_t* d_ranks, int nranks,
                            int32_t* d_order_stats, void* stream) {
  PL_REQUIRE(d_tile_max && d_edge_min && d_edge_max, "null pointer");
  PL_REQUIRE(h > 0 && w > 0 && edge_window > 0, "bad shape");
  PL_REQUIRE((d_ranks == nullptr) == (d_order_stats == nullptr) && (!d_ranks || nranks > 0), "ranks and their output go together");
  PL_REQUIRE(!d_ranks || (reinterpret_cast<uintptr_t>(d_hist) & 15) == 0, "the table must be 16-byte aligned");
  return hist16_impl(in, dtype, n, (int64_t)h * w, d_hist, d_tile_max,
                     HistEdges{h, w, edge_window, d_edge_min, d_edge_max, d_ranks, nranks, d_order_stats}, stream);
}

static int hist16_impl(const void* in, int dtype, int64_t n, int64_t count, uint32_t* d_hist, uint16_t* d_tile_max,
                       HistEdges edges, void* stream) {
  PL_REQUIRE(in && d_hist, "null pointer");
  PL_REQUIRE(n >= 0 && count > 0, "bad shape");
  PL_REQUIRE(dtype == PL_U16 || dtype == PL_I16, "16-bit integer frames only");
  if (n == 0) return PL_OK;
  PL_REQUIRE(pl_cdiv(n, 8) * 8 * 4 <= 0x7fffffffLL, "batch too large");
  const unsigned flip = dtype == PL_I16 ? 0x8000u : 0u;
  hipStream_t st = (hipStream_t)stream;
  const unsigned short* src = (const unsigned short*)in;
  // (multi-part kernels) measured on MI355X, 256 x 1024^2: 2 parts / one LDS atomic per pixel 0.16 ms; 2 parts / run-merged 0.22;
  // 4 parts / per pixel 0.26; 4 parts / run-merged 0.31 (the kernel is VALU-bound: instructions per pixel
  // times the number of parts that look at it)
  // frames of at least a quarter of a megapixel: ONE read by one workgroup per frame (two LDS windows + global atomics for the
  // pixels in between); smaller frames, or no 152 KiB of LDS: the multi-part kernels
  static std::atomic<int> two_window{0};             // 0 untried, 1 available, -1 refused
  if (two_window == 0) {
    bool ok = hipFuncSetAttribute((const void*)hist16_two_window_kernel<kTwBinsWide, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)tw_lds(kTwBinsWide)) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void*)hist16_two_window_kernel<kTwBinsWl, 8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)tw_lds(kTwBinsWl)) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    two_window = ok ? 1 : -1;
  }
  if (two_window == 1 && count >= 262144) {
    if (edges.d_stats && n > pl_cu_count())            // pl_hist16_wl: Winston-Lutz frames, and more of them than CUs
      hipLaunchKernelGGL((hist16_two_window_kernel<kTwBinsWl, 8>), dim3((unsigned)n), dim3(kHistThreads), tw_lds(kTwBinsWl), st, src,
                         count, flip, d_hist, d_tile_max, edges.h, edges.w, edges.window, edges.d_min, edges.d_max, edges.d_ranks,
                         edges.nranks, edges.d_stats);
    else
      hipLaunchKernelGGL((hist16_two_window_kernel<kTwBinsWide, 4>), dim3((unsigned)n), dim3(kHistThreads), tw_lds(kTwBinsWide), st,
                         src, count, flip, d_hist, d_tile_max, edges.h, edges.w, edges.window, edges.d_min, edges.d_max,
                         edges.d_ranks, edges.nranks, edges.d_stats);
    return pl_check_launch("pl_hist16");
  }
  // (the multi-part kernels know nothing of edges: the stand-alone kernel)
  if (edges.d_min) {
    const int rc0 = pl_edge_minmax(in, dtype, n, edges.h, edges.w, edges.window, edges.d_min, edges.d_max, stream);
    if (rc0 != PL_OK) return rc0;
  }
  // (the multi-part kernels keep no tile maxima: every tile says "look inside")
  if (d_tile_max && hipMemsetAsync(d_tile_max, 0xff, (size_t)n * (size_t)((count + 511) / 512) * sizeof(uint16_t), st) != hipSuccess) {
    pl_set_error("pl_hist16_tiles: memset failed");
    return PL_ERR_HIP;
  }
  int rc = launch_hist16<2, false>(src, n, count, flip, d_hist, st);
  if (rc != 0) rc = launch_hist16<4, false>(src, n, count, flip, d_hist, st);   // 64 KiB LDS needs no opt-in
  PL_REQUIRE(rc == 0, "launch configuration rejected");
  if (edges.d_ranks) return pl_order_stats_from_hist(d_hist, dtype, n, edges.d_ranks, edges.nranks, edges.d_stats, stream);
  return pl_check_launch("pl_hist16");
}

extern "C" int pl_otsu_from_hist(const uint32_t* d_hist, int dtype, int64_t n, int32_t* d_thr,
                                 int32_t* d_min, int32_t* d_max, void* stream) {
  PL_REQUIRE(d_hist && d_thr, "null pointer");
  PL_REQUIRE(dtype == PL_U16 || dtype == PL_I16, "16-bit integer frames only");
  PL_REQUIRE(n >= 0 && n <= 0x7fffffffLL, "bad batch");
  if (n == 0) return PL_OK;
  hipLaunchKernelGGL(otsu_kernel, dim3((unsigned)n), dim3(kHistThreads), 0, (hipStream_t)stream, d_hist,
                     dtype == PL_I16 ? 32768 : 0, d_thr, d_min, d_max, (const int32_t*)nullptr);
  return pl_check_launch("pl_otsu_from_hist");
}

extern "C" int pl_order_stats_from_hist(const uint32_t* d_hist, int dtype, int64_t n,
                                        const int64_t* d_ranks, int nranks, int32_t* d_out,
                                        void* stream) {
  PL_REQUIRE(d_hist && d_ranks && d_out, "null pointer");
  PL_REQUIRE(dtype == PL_U16 || dtype == PL_I16, "16-bit integer frames only");
  PL_REQUIRE(n >= 0 && n <= 0x7fffffffLL && nranks > 0, "bad batch");
  if (n == 0) return PL_OK;
  hipLaunchKernelGGL(order_stats_kernel, dim3((unsigned)n), dim3(kHistThreads), 0, (hipStream_t)stream,
                     d_hist, dtype == PL_I16 ? 32768 : 0, d_ranks, nranks, d_out);
  return pl_check_launch("pl_order_stats_from_hist");
}

/* skimage.filters.threshold_otsu straight from the 16-bit frames (pylinac/ct.py:3323, 3338; acr.py:1409).  d_lo / d_hi:
 * optional per-frame bounds (int32[n], d_lo <= values <= d_hi), both NULL = the kernel places its window from a row
 * sample.  d_thr / d_min / d_max as pl_otsu_from_hist; d_flag int32[n] scratch; d_hist uint32[n][65536] workspace,
 * touched only for frames that do not fit the 38 912-bin LDS window. */
// median.hip: 3x3 median of the frames whose gate flag is non-zero (0 = launched)
int pl_median3_gated(const void* in, void* out, int is_signed, int64_t n, int h, int w, const int32_t* d_gate, hipStream_t st);

namespace {
template <typename T, bool MED3>
int otsu16_launch(const void* in, void* scratch, int dtype, int64_t n, int64_t count, int h, int w, const int32_t* d_lo,
                  const int32_t* d_hi, int32_t* d_thr, int32_t* d_min, int32_t* d_max, int32_t* d_flag, uint32_t* d_hist,
                  hipStream_t st, const char* who) {
  const unsigned flip = dtype == PL_I16 ? 0x8000u : 0u;
  const int bias = dtype == PL_I16 ? 32768 : 0;
  const size_t lds = kOtsuLds;                                  // bins + the spare bin of the branch-free tally + scratch
  static std::atomic<bool> attr{false};                        // one flag per instantiation
  if (!attr) {
    if (hipFuncSetAttribute((const void*)otsu16_window_kernel<T, MED3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess) {
      pl_set_error("%s: LDS attribute: %s", who, hipGetErrorString(hipGetLastError()));
      return PL_ERR_HIP;
    }
    attr = true;
  }
  // small batches: several workgroups per frame (the table d_hist merges them; it and the flags start at zero)
  int parts = 1;
  const int cus = pl_cu_count();
  while (parts < 8 && n * parts * 2 <= cus) parts *= 2;
  if (MED3) { const int row_groups = (h + 31) / 32; while (parts > 1 && parts > row_groups) parts /= 2; }
  if (count < 65536) parts = 1;
  if (parts > 1) {
    hipError_t e = hipMemsetAsync(d_hist, 0, (size_t)n * 65536 * sizeof(uint32_t), st);
    if (e == hipSuccess) e = hipMemsetAsync(d_flag, 0, (size_t)n * sizeof(int32_t), st);
    if (e != hipSuccess) { pl_set_error("%s: memset: %s", who, hipGetErrorString(e)); return PL_ERR_HIP; }
  }
#ifdef PL_OTSU_FULL_ALWAYS                                    // development variant: every frame through the full-range kernel
  (void)hipMemsetAsync(d_flag, 0, (size_t)n * sizeof(int32_t), st);
#else
  hipLaunchKernelGGL((otsu16_window_kernel<T, MED3>), dim3((unsigned)(n * parts)), dim3(kHistThreads), lds, st,
                     (const unsigned short*)in, count, h, w, flip, bias, d_lo, d_hi, d_thr, d_min, d_max, d_flag, parts, d_hist);
#endif
  // frames too wide for the window: the full-range kernel (packed 16-bit counters), every workgroup gated by d_flag; the
  // medians are computed on the fly again for exactly those frames.  (Round 3: gated median plane + two-part histogram +
  // scan, 2.2 x the window kernel's time on a stretched batch.)  Frames beyond 2^26 pixels keep the table path.
  if (count <= (int64_t)kFullEvents * 32768) {
    static std::atomic<bool> attr2{false};
    if (!attr2) {
      if (hipFuncSetAttribute((const void*)otsu16_full_kernel<T, MED3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFullLds) !=
          hipSuccess) {
        pl_set_error("%s: LDS attribute: %s", who, hipGetErrorString(hipGetLastError()));
        return PL_ERR_HIP;
      }
      attr2 = true;
    }
    hipLaunchKernelGGL((otsu16_full_kernel<T, MED3>), dim3((unsigned)n), dim3(kHistThreads), kFullLds, st, (const unsigned short*)in,
                       count, h, w, flip, bias, d_thr, d_min, d_max, d_flag);
    return pl_check_launch(who);
  }
  const unsigned short* plane = (const unsigned short*)in;
  if (MED3) {
    if (pl_median3_gated(in, scratch, dtype == PL_I16, n, h, w, d_flag, st) != 0) {
      pl_set_error("%s: gated median launch rejected", who);
      return PL_ERR_INVALID_ARG;
    }
    plane = (const unsigned short*)scratch;
  }
  int rc = launch_hist16<2, false>(plane, n, count, flip, d_hist, st, d_flag);
  if (rc != 0) rc = launch_hist16<4, false>(plane, n, count, flip, d_hist, st, d_flag);
  if (rc != 0) {
    pl_set_error("%s: launch configuration rejected", who);
    return PL_ERR_INVALID_ARG;
  }
  hipLaunchKernelGGL(otsu_kernel, dim3((unsigned)n), dim3(kHistThreads), 0, st, d_hist, bias, d_thr, d_min, d_max,
                     (const int32_t*)d_flag);
  return pl_check_launch(who);
}
}  // namespace

extern "C" int pl_otsu16(const void* in, int dtype, int64_t n, int64_t count, const int32_t* d_lo, const int32_t* d_hi,
                         int32_t* d_thr, int32_t* d_min, int32_t* d_max, int32_t* d_flag, uint32_t* d_hist,
                         void* stream) {
  PL_REQUIRE(in && d_thr && d_flag && d_hist, "null pointer");
  PL_REQUIRE((d_lo == nullptr) == (d_hi == nullptr), "give both bounds or neither");
  PL_REQUIRE(n >= 0 && n <= 0x7fffffffLL / 8 && count > 0, "bad shape");
  PL_REQUIRE(dtype == PL_U16 || dtype == PL_I16, "16-bit integer frames only");
  if (n == 0) return PL_OK;
  return otsu16_launch<unsigned short, false>(in, nullptr, dtype, n, count, 0, 0, d_lo, d_hi, d_thr, d_min, d_max, d_flag, d_hist,
                                              (hipStream_t)stream, "pl_otsu16");
}

extern "C" int pl_median3_otsu16(const void* in, void* scratch, int dtype, int64_t n, int h, int w, const int32_t* d_lo,
                                 const int32_t* d_hi, int32_t* d_thr, int32_t* d_min, int32_t* d_max, int32_t* d_flag,
                                 uint32_t* d_hist, void* stream) {
  PL_REQUIRE(in && scratch && d_thr && d_flag && d_hist, "null pointer");
  PL_REQUIRE(in != scratch, "scratch must be a distinct buffer");
  PL_REQUIRE((d_lo == nullptr) == (d_hi == nullptr), "give both bounds or neither");
  PL_REQUIRE(n >= 0 && n <= 0x7fffffffLL / 8 && h > 0 && w > 0, "bad shape");
  PL_REQUIRE(dtype == PL_U16 || dtype == PL_I16, "16-bit integer frames only");
  PL_REQUIRE(pl_median3_rows_covers(in, h, w) && (reinterpret_cast<uintptr_t>(scratch) & 15) == 0,
             "needs h > 1, width % 8 == 0 and 16-byte aligned frames (run pl_median2d + pl_otsu16 otherwise)");
  if (n == 0) return PL_OK;
  const int64_t count = (int64_t)h * w;
  return dtype == PL_I16
             ? otsu16_launch<short, true>(in, scratch, dtype, n, count, h, w, d_lo, d_hi, d_thr, d_min, d_max, d_flag, d_hist,
                                          (hipStream_t)stream, "pl_median3_otsu16")
             : otsu16_launch<unsigned short, true>(in, scratch, dtype, n, count, h, w, d_lo, d_hi, d_thr, d_min, d_max, d_flag,
                                                   d_hist, (hipStream_t)stream, "pl_median3_otsu16");
}
