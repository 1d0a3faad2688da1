// Profile extraction: axis reductions of a frame.
//
// Replaces the ad-hoc numpy reductions the analyzers use in place of an Image.profile():
//   np.mean(image, axis)   pylinac/picketfence.py:747-750, pylinac/field_analysis.py:1094-1117
//   np.sum(array, axis)    pylinac/picketfence.py:1513-1514, pylinac/field_analysis.py:488-506
//   np.max(central, axis)  pylinac/starshot.py:216-217
// Integer frames are summed in int64 (exact; numpy's float64/uint64 accumulation of integers is
// exact as well, so any order agrees), float frames are accumulated in the frame's own precision,
// sequentially along axis 0 (numpy's order for a C-contiguous frame) and by a wave tree along
// axis 1 (numpy uses pairwise summation there: equal to ~1 ulp, not bitwise).
//
// pl_threshold_colsum_u16 fuses BaseImage.threshold (pylinac/core/image.py:797-800) with the
// axis-0 column sums of the thresholded frame: one read and one write of the frame, 16-byte
// accesses (4 independent loads in flight per lane), per-lane uint32 partial sums over a 128-row
// band, one uint64 atomic per column/band.
#include "pl_common.h"
#include "median3_rows.h"

namespace {

constexpr int kThreads = 256;

template <typename T> struct Acc { using type = long long; };
template <> struct Acc<float> { using type = float; };
template <> struct Acc<double> { using type = double; };
template <typename A> struct AccIsFloat { static constexpr bool value = false; };
template <> struct AccIsFloat<float> { static constexpr bool value = true; };
template <> struct AccIsFloat<double> { static constexpr bool value = true; };

template <typename T>
__device__ __forceinline__ double finish(typename Acc<T>::type acc, int op, int count) {
  using A = typename Acc<T>::type;
  if (op == PL_MEAN) {
    if constexpr (!AccIsFloat<A>::value) return (double)acc / (double)count;  // float64 mean of integers
    else return (double)(A)(acc / (A)count);                            // mean in the frame's precision
  }
  return (double)acc;
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
reduce_axis0_kernel(const T* __restrict__ in, int h, int w, int col_tiles, int op,
                    double* __restrict__ out) {
  using A = typename Acc<T>::type;
  const int ct = blockIdx.x % col_tiles;
  const size_t frame = blockIdx.x / col_tiles;
  const int c = ct * kThreads + threadIdx.x;
  if (c >= w) return;
  const T* p = in + frame * (size_t)h * w + c;
  if (op == PL_SUM || op == PL_MEAN) {
    A acc = 0;
    for (int r = 0; r < h; ++r) acc += (A)p[(size_t)r * w];
    out[frame * w + c] = finish<T>(acc, op, h);
  } else {
    T best = p[0];
    for (int r = 1; r < h; ++r) {
      T v = p[(size_t)r * w];
      best = (op == PL_MAX) ? (v > best ? v : best) : (v < best ? v : best);
    }
    out[frame * w + c] = (double)best;
  }
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
reduce_axis1_kernel(const T* __restrict__ in, int64_t rows_total, int w, int op,
                    double* __restrict__ out) {
  using A = typename Acc<T>::type;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (kThreads / PL_WAVE) + (threadIdx.x >> 6);
  if (row >= rows_total) return;
  const T* p = in + row * (size_t)w;
  if (op == PL_SUM || op == PL_MEAN) {
    A acc = 0;
    for (int c = lane; c < w; c += PL_WAVE) acc += (A)p[c];
    acc = pl_wave_reduce(acc, [](A a, A b) { return a + b; });
    if (lane == 0) out[row] = finish<T>(acc, op, w);
  } else {
    T best = p[lane < w ? lane : 0];
    for (int c = lane; c < w; c += PL_WAVE) {
      T v = p[c];
      best = (op == PL_MAX) ? (v > best ? v : best) : (v < best ? v : best);
    }
    best = pl_wave_reduce(best, [op](T a, T b) { return (op == PL_MAX) ? (a > b ? a : b) : (a < b ? a : b); });
    if (lane == 0) out[row] = (double)best;
  }
}

// ------------------------------------------------------------- fused threshold + column sums
constexpr int kBandRows = 128;
constexpr int kTcThreads = 128;  // 128 lanes x 8 columns = 1024 columns per sweep
constexpr int kTcUnroll = 4;     // independent 16-byte loads in flight per lane

__global__ void __launch_bounds__(kTcThreads)
threshold_colsum_kernel(const unsigned short* __restrict__ in, unsigned short* __restrict__ out, int h,
                        int w, int bands, const int32_t* __restrict__ thr,
                        unsigned long long* __restrict__ colsum) {
  const unsigned id = pl_xcd_remap(blockIdx.x, gridDim.x);
  const int band = id % bands;
  const size_t frame = id / bands;
  const int t = thr[frame];
  const int r0 = band * kBandRows;
  const int r1 = (r0 + kBandRows < h) ? r0 + kBandRows : h;
  const unsigned short* f = in + frame * (size_t)h * w;
  unsigned short* o = out + frame * (size_t)h * w;
  unsigned long long* cs = colsum + frame * (size_t)w;
  const bool vec = ((w & 7) == 0) && ((reinterpret_cast<uintptr_t>(f) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(o) & 15) == 0);
  if (vec) {
    for (int c = threadIdx.x * 8; c < w; c += kTcThreads * 8) {
      unsigned s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      auto apply = [&](uint4 q) -> uint4 {
        union { uint4 q; unsigned short e[8]; } u;
        u.q = q;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const unsigned short v = ((int)u.e[k] >= t) ? u.e[k] : (unsigned short)0;
          u.e[k] = v;
          s[k] += v;
        }
        return u.q;
      };
      int r = r0;
      for (; r + kTcUnroll <= r1; r += kTcUnroll) {
        uint4 q[kTcUnroll];
#pragma unroll
        for (int k = 0; k < kTcUnroll; ++k) q[k] = *reinterpret_cast<const uint4*>(f + (size_t)(r + k) * w + c);
#pragma unroll
        for (int k = 0; k < kTcUnroll; ++k) *reinterpret_cast<uint4*>(o + (size_t)(r + k) * w + c) = apply(q[k]);
      }
      for (; r < r1; ++r)
        *reinterpret_cast<uint4*>(o + (size_t)r * w + c) = apply(*reinterpret_cast<const uint4*>(f + (size_t)r * w + c));
#pragma unroll
      for (int k = 0; k < 8; ++k) atomicAdd(cs + c + k, (unsigned long long)s[k]);
    }
  } else {
    for (int c = threadIdx.x; c < w; c += kTcThreads) {
      unsigned s = 0;
      for (int r = r0; r < r1; ++r) {
        unsigned short v = f[(size_t)r * w + c];
        v = ((int)v >= t) ? v : (unsigned short)0;
        o[(size_t)r * w + c] = v;
        s += v;
      }
      atomicAdd(cs + c, (unsigned long long)s);
    }
  }
}

// The same with the 3x3 MEDIAN of the frame as the thresholded quantity, computed on the fly (pl_median3_rows): the median
// plane is never written.  One workgroup = 4 waves = one block of 512 columns x one band of 128 rows; wave v walks row group v
// (32 rows) of the band; the four waves' column sums meet in LDS and leave as one 64-bit atomic per column.
constexpr int kMtRows = 32;                        // rows per wave: two halo rows are re-read per wave
constexpr int kMtWaves = kBandRows / kMtRows;      // 4
// PARTS: the band's column sums are STORED as uint32 parts[frame][band][column] (128 rows x 65535 < 2^32) instead of added to
// colsum[frame][column] with 64-bit atomics -- no zeroed table, no atomics; pl_colparts_profile_fwxm adds the bands up.
template <bool PARTS>
__global__ void __launch_bounds__(kMtWaves * PL_WAVE)
median3_threshold_colsum_kernel(const unsigned short* __restrict__ in, unsigned short* __restrict__ out, int h, int w, int bands,
                                int col_groups, const int32_t* __restrict__ thr, unsigned long long* __restrict__ colsum,
                                uint32_t* __restrict__ parts) {
  __shared__ unsigned s_cs[PL_WAVE * 8];
  unsigned id = pl_xcd_remap(blockIdx.x, gridDim.x);
  const int cg = id % col_groups;
  id /= col_groups;
  const int band = id % bands;
  const size_t frame = id / bands;
  const int t = thr[frame];
  const int lane = threadIdx.x & (PL_WAVE - 1), wave = threadIdx.x / PL_WAVE;
  const int c0 = (cg * PL_WAVE + lane) * 8;
  const bool on = c0 < w;
  const int rg = band * kBandRows + kMtRows * wave;     // the wave's first row; rows beyond the frame produce nothing
  const unsigned short* f = in + frame * (size_t)h * w;
  unsigned short* o = out + frame * (size_t)h * w;
  for (int i = threadIdx.x; i < PL_WAVE * 8; i += kMtWaves * PL_WAVE) s_cs[i] = 0;
  __syncthreads();
  unsigned s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rg < h) {                                     // wave-uniform
    pl_median3_rows<unsigned short, kMtRows>(f, h, w, c0, lane, rg, [&](int r, const int (&m)[8]) {
      if (!on) return;
      unsigned v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[k] = m[k] >= t ? (unsigned)m[k] : 0u;
        s[k] += v[k];
      }
      *reinterpret_cast<uint4*>(o + (size_t)r * w + c0) =
          uint4{v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)};
    });
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(&s_cs[lane * 8 + k], s[k]);     // 32 rows x 65535 per wave, 4 waves: < 2^32
  }
  __syncthreads();
  if (PARTS) {
    uint32_t* ps = parts + (frame * (size_t)bands + band) * (size_t)w + (size_t)cg * PL_WAVE * 8;
    for (int i = threadIdx.x; i < PL_WAVE * 8; i += kMtWaves * PL_WAVE)
      if (cg * PL_WAVE * 8 + i < w) ps[i] = s_cs[i];
  } else {
    unsigned long long* cs = colsum + frame * (size_t)w + (size_t)cg * PL_WAVE * 8;
    for (int i = threadIdx.x; i < PL_WAVE * 8; i += kMtWaves * PL_WAVE)
      if (cg * PL_WAVE * 8 + i < w) atomicAdd(cs + i, (unsigned long long)s_cs[i]);
  }
}

__global__ void colsum_to_mean_kernel(const unsigned long long* __restrict__ cs, int64_t total, int h,
                                      double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) out[i] = (double)cs[i] / (double)h;  // np.mean of integers: float64 sum / count
}

}  // namespace

extern "C" int pl_colsum_to_mean(const unsigned long long* d_colsum, int64_t n, int w, int h,
                                 double* d_out, void* stream) {
  PL_REQUIRE(d_colsum && d_out, "null pointer");
  PL_REQUIRE(n >= 0 && w > 0 && h > 0, "bad shape");
  if (n == 0) return PL_OK;
  const int64_t total = n * w;
  hipLaunchKernelGGL(colsum_to_mean_kernel, dim3((unsigned)pl_cdiv(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, d_colsum, total, h, d_out);
  return pl_check_launch("pl_colsum_to_mean");
}

extern "C" int pl_reduce_axis(const void* in, int dtype, int64_t n, int h, int w, int axis, int op,
                              double* d_out, void* stream) {
  PL_REQUIRE(in && d_out, "null pointer");
  PL_REQUIRE(n >= 0 && h > 0 && w > 0, "bad shape");
  PL_REQUIRE(axis == 0 || axis == 1, "axis must be 0 or 1");
  PL_REQUIRE(op >= PL_SUM && op <= PL_MIN, "bad op");
  if (n == 0) return PL_OK;
  hipStream_t st = (hipStream_t)stream;
  if (axis == 0) {
    int col_tiles = (int)pl_cdiv(w, kThreads);
    PL_REQUIRE(n * col_tiles <= 0x7fffffffLL, "batch too large");
    PL_DISPATCH_DTYPE(dtype, T,
                      hipLaunchKernelGGL(reduce_axis0_kernel<T>, dim3((unsigned)(n * col_tiles)),
                                         dim3(kThreads), 0, st, (const T*)in, h, w, col_tiles, op, d_out));
  } else {
    int64_t rows = n * h;
    int64_t blocks = pl_cdiv(rows, kThreads / PL_WAVE);
    PL_REQUIRE(blocks <= 0x7fffffffLL, "batch too large");
    PL_DISPATCH_DTYPE(dtype, T,
                      hipLaunchKernelGGL(reduce_axis1_kernel<T>, dim3((unsigned)blocks), dim3(kThreads), 0,
                                         st, (const T*)in, rows, w, op, d_out));
  }
  return pl_check_launch("pl_reduce_axis");
}

extern "C" int pl_threshold_colsum_u16(const uint16_t* in, uint16_t* out, int64_t n, int h, int w,
                                       const int32_t* d_thr, unsigned long long* d_colsum,
                                       void* stream) {
  PL_REQUIRE(in && out && d_thr && d_colsum, "null pointer");
  PL_REQUIRE(n >= 0 && h > 0 && w > 0, "bad shape");
  if (n == 0) return PL_OK;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(d_colsum, 0, (size_t)n * w * sizeof(unsigned long long), st);
  if (e != hipSuccess) { pl_set_error("pl_threshold_colsum_u16: memset: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
  int bands = (int)pl_cdiv(h, kBandRows);
  PL_REQUIRE(n * bands <= 0x7fffffffLL, "batch too large");
  hipLaunchKernelGGL(threshold_colsum_kernel, dim3((unsigned)(n * bands)), dim3(kTcThreads), 0, st, in, out,
                     h, w, bands, d_thr, d_colsum);
  return pl_check_launch("pl_threshold_colsum_u16");
}

extern "C" int pl_median3_threshold_colsum_u16(const uint16_t* in, uint16_t* out, int64_t n, int h, int w,
                                               const int32_t* d_thr, unsigned long long* d_colsum, void* stream) {
  PL_REQUIRE(in && out && d_thr && d_colsum && in != out, "null or aliased pointers");
  PL_REQUIRE(n >= 0 && h > 0 && w > 0, "bad shape");
  PL_REQUIRE(pl_median3_rows_covers(in, h, w) && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
             "needs h > 1, width % 8 == 0 and 16-byte aligned frames (run pl_median2d + pl_threshold_colsum_u16 otherwise)");
  if (n == 0) return PL_OK;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(d_colsum, 0, (size_t)n * w * sizeof(unsigned long long), st);
  if (e != hipSuccess) { pl_set_error("pl_median3_threshold_colsum_u16: memset: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
  const int bands = (int)pl_cdiv(h, kBandRows), col_groups = (int)pl_cdiv(w / 8, PL_WAVE);
  PL_REQUIRE(n * bands * col_groups <= 0x7fffffffLL, "batch too large");
  hipLaunchKernelGGL(median3_threshold_colsum_kernel<false>, dim3((unsigned)(n * bands * col_groups)), dim3(kMtWaves * PL_WAVE), 0, st,
                     in, out, h, w, bands, col_groups, d_thr, d_colsum, (uint32_t*)nullptr);
  return pl_check_launch("pl_median3_threshold_colsum_u16");
}

// rows per band of pl_median3_threshold_colparts_u16's partial sums: bands = ceil(h / pl_colparts_band_rows())
extern "C" int pl_colparts_band_rows(void) { return kBandRows; }

// The same pass with the column sums left as per-band partial sums d_parts uint32[n][bands][w] (plain stores: nothing to zero,
// no atomics); pl_colparts_profile_fwxm (peaks.hip) turns them into the mean profile and its FWXM record.
extern "C" int pl_median3_threshold_colparts_u16(const uint16_t* in, uint16_t* out, int64_t n, int h, int w,
                                                 const int32_t* d_thr, uint32_t* d_parts, void* stream) {
  PL_REQUIRE(in && out && d_thr && d_parts && in != out, "null or aliased pointers");
  PL_REQUIRE(n >= 0 && h > 0 && w > 0, "bad shape");
  PL_REQUIRE(pl_median3_rows_covers(in, h, w) && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
             "needs h > 1, width % 8 == 0 and 16-byte aligned frames (run pl_median2d + pl_threshold_colsum_u16 otherwise)");
  if (n == 0) return PL_OK;
  const int bands = (int)pl_cdiv(h, kBandRows), col_groups = (int)pl_cdiv(w / 8, PL_WAVE);
  PL_REQUIRE(n * bands * col_groups <= 0x7fffffffLL, "batch too large");
  hipLaunchKernelGGL(median3_threshold_colsum_kernel<true>, dim3((unsigned)(n * bands * col_groups)), dim3(kMtWaves * PL_WAVE), 0,
                     (hipStream_t)stream, in, out, h, w, bands, col_groups, d_thr, (unsigned long long*)nullptr, d_parts);
  return pl_check_launch("pl_median3_threshold_colparts_u16");
}
