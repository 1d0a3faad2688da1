// Varian XIM compressed-pixel decoding (SURVEY.md section 8 "next" row f1).
//
// Replaces: XIM._parse_lookup_table / _get_diffs / _parse_compressed_bytes (pylinac/core/image.py:1180-1296): the
// reference walks the variable-length difference stream run by run and then rebuilds the image one row at a time
// (cumsum per row in a Python loop, ~1 s per 1280 x 1280 image).  Both steps are scans:
//   * byte offset of difference i = exclusive prefix sum of the sizes 1 << code_i (2-bit codes, 4 per lookup byte)
//   * with S_r = row-wise prefix sums of the raw differences, the reference's recurrence is
//       P[r] = P[r-1] + S_r + c_r,   c_1 = -P[0][0],   c_r = c_{r-1} + S_{r-1}[W-1]
//     (derivation and check against the reference decoder: oracle/pylinac_oracle.py xim_decode), i.e. a row scan,
//     a scan of the row totals and a column scan -- all in the array dtype's wrap-around arithmetic (the pixel
//     array is int8/16/32/64 by bytes_per_pixel; 64-bit unsigned accumulators truncated at the store are the same
//     ring arithmetic).
#include "pl_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kItems = 8;                       // differences per lane = two lookup bytes
constexpr int kChunk = kThreads * kItems;       // differences per workgroup

__device__ __forceinline__ unsigned code_of(const unsigned char* __restrict__ lut, int64_t i) {
  return (lut[i >> 2] >> (2 * (i & 3))) & 3u;
}

// pass 1: bytes consumed by each chunk of kChunk differences (+ flag if a code 3 occurs)
__global__ void __launch_bounds__(kThreads)
xim_chunk_bytes_kernel(const unsigned char* __restrict__ lut, int64_t n_diffs, unsigned* __restrict__ chunk_bytes,
                       int* __restrict__ status) {
  __shared__ unsigned s_w[kThreads / PL_WAVE];
  const int64_t base = (int64_t)blockIdx.x * kChunk + (int64_t)threadIdx.x * kItems;
  unsigned sum = 0;
  bool bad = false;
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const int64_t i = base + k;
    if (i < n_diffs) {
      const unsigned c = code_of(lut, i);
      bad |= c == 3u;
      sum += 1u << c;
    }
  }
  if (bad) atomicOr(status, 1);
  sum = pl_wave_reduce(sum, [](unsigned a, unsigned b) { return a + b; });
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0;
    for (int q = 0; q < kThreads / PL_WAVE; ++q) t += s_w[q];
    chunk_bytes[blockIdx.x] = t;
  }
}

// pass 2: exclusive scan of the chunk totals (one workgroup; a few hundred to a few thousand chunks)
__global__ void __launch_bounds__(kThreads)
xim_scan_chunks_kernel(unsigned* __restrict__ chunk_bytes, int n_chunks) {
  __shared__ unsigned s_part[kThreads];
  const int per = (n_chunks + kThreads - 1) / kThreads;
  const int lo = threadIdx.x * per, hi = min(lo + per, n_chunks);
  unsigned sum = 0;
  for (int i = lo; i < hi; ++i) sum += chunk_bytes[i];
  s_part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned run = 0;
    for (int q = 0; q < kThreads; ++q) { const unsigned v = s_part[q]; s_part[q] = run; run += v; }
  }
  __syncthreads();
  unsigned run = s_part[threadIdx.x];
  for (int i = lo; i < hi; ++i) { const unsigned v = chunk_bytes[i]; chunk_bytes[i] = run; run += v; }
}

// pass 3: every difference finds its byte offset and is read, sign-extended and stored in the array dtype;
// the first W + 1 pixels are plain int32
template <typename T>
__global__ void __launch_bounds__(kThreads)
xim_gather_kernel(const unsigned char* __restrict__ lut, const unsigned char* __restrict__ stream,
                  int64_t stream_bytes, int64_t n_diffs, int64_t n_plain, const unsigned* __restrict__ chunk_off,
                  T* __restrict__ a, int* __restrict__ status) {
  __shared__ unsigned s_w[kThreads / PL_WAVE];
  const int64_t base = (int64_t)blockIdx.x * kChunk + (int64_t)threadIdx.x * kItems;
  unsigned sizes[kItems], mine = 0;
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const int64_t i = base + k;
    sizes[k] = (i < n_diffs) ? (1u << (code_of(lut, i) & 3u)) : 0u;
    mine += sizes[k];
  }
  // exclusive scan of `mine` over the workgroup
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned v = __shfl_up(inc, o, 64);
    if (lane >= o) inc += v;
  }
  if (lane == 63) s_w[wv] = inc;
  __syncthreads();
  unsigned off = chunk_off[blockIdx.x] + (inc - mine);
  for (int q = 0; q < wv; ++q) off += s_w[q];
  const int64_t data0 = n_plain * 4;
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const int64_t i = base + k;
    if (i >= n_diffs) break;
    const unsigned sz = sizes[k];
    const int64_t p = data0 + off;
    if (sz > 4u || p + sz > stream_bytes) { atomicOr(status, sz > 4u ? 1 : 2); off += sz; continue; }
    unsigned v = 0;
    for (unsigned q = 0; q < sz; ++q) v |= (unsigned)stream[p + q] << (8 * q);
    int sv = sz == 1 ? (int)(signed char)v : (sz == 2 ? (int)(short)v : (int)v);
    a[n_plain + i] = (T)sv;
    off += sz;
  }
  // the uncompressed head (first row + first pixel of the second row): little-endian int32
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n_plain; i += (int64_t)gridDim.x * kThreads) {
    const unsigned char* q = stream + i * 4;
    const unsigned v = (unsigned)q[0] | ((unsigned)q[1] << 8) | ((unsigned)q[2] << 16) | ((unsigned)q[3] << 24);
    a[i] = (T)(int)v;
  }
}

// row-wise inclusive prefix sums of rows 1 .. H-1 (in place) + the row totals
template <typename T>
__global__ void __launch_bounds__(kThreads)
xim_row_scan_kernel(T* __restrict__ a, int w, unsigned long long* __restrict__ row_total) {
  __shared__ unsigned long long s_w[kThreads / PL_WAVE];
  const int r = blockIdx.x + 1;
  T* row = a + (size_t)r * w;
  const int per = (w + kThreads - 1) / kThreads;
  const int lo = threadIdx.x * per, hi = min(lo + per, w);
  unsigned long long sum = 0;
  for (int c = lo; c < hi; ++c) sum += (unsigned long long)(long long)row[c];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned long long inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long v = __shfl_up(inc, o, 64);
    if (lane >= o) inc += v;
  }
  if (lane == 63) s_w[wv] = inc;
  __syncthreads();
  unsigned long long run = inc - sum;
  for (int q = 0; q < wv; ++q) run += s_w[q];
  for (int c = lo; c < hi; ++c) {
    run += (unsigned long long)(long long)row[c];
    row[c] = (T)(long long)run;
  }
  if (threadIdx.x == kThreads - 1) {
    unsigned long long t = 0;
    for (int q = 0; q < kThreads / PL_WAVE; ++q) t += s_w[q];
    row_total[r] = t;
  }
}

// c_1 = -P[0][0], c_r = c_{r-1} + total_{r-1}
template <typename T>
__global__ void xim_carry_kernel(const T* __restrict__ a, int h, unsigned long long* __restrict__ row_total) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned long long c = 0ull - (unsigned long long)(long long)a[0];
  for (int r = 1; r < h; ++r) {
    const unsigned long long t = row_total[r];
    row_total[r] = c;            // becomes c_r
    c += t;
  }
}

// P[r][c] = P[r-1][c] + S_r[c] + c_r
template <typename T>
__global__ void __launch_bounds__(kThreads)
xim_col_scan_kernel(T* __restrict__ a, int h, int w, const unsigned long long* __restrict__ carry) {
  const int c = blockIdx.x * kThreads + threadIdx.x;
  if (c >= w) return;
  unsigned long long acc = (unsigned long long)(long long)a[c];
  for (int r = 1; r < h; ++r) {
    acc += (unsigned long long)(long long)a[(size_t)r * w + c] + carry[r];
    a[(size_t)r * w + c] = (T)(long long)acc;
  }
}

template <typename T>
int xim_decode_t(const unsigned char* lut, int64_t lut_bytes, const unsigned char* stream, int64_t stream_bytes, int w,
                 int h, T* out, unsigned char* work, hipStream_t st) {
  const int64_t n_plain = (int64_t)w + 1, n_diffs = (int64_t)w * h - w - 1;
  if (stream_bytes < n_plain * 4 || lut_bytes * 4 < n_diffs) { pl_set_error("pl_xim_decode: stream or lookup table too short"); return PL_ERR_INVALID_ARG; }
  const int n_chunks = (int)pl_cdiv(n_diffs > 0 ? n_diffs : 1, kChunk);
  int* status = reinterpret_cast<int*>(work);
  unsigned* chunk = reinterpret_cast<unsigned*>(work + 16);
  unsigned long long* row_total = reinterpret_cast<unsigned long long*>(work + 16 + (((size_t)n_chunks * 4 + 15) & ~(size_t)15));
  if (hipMemsetAsync(status, 0, 16, st) != hipSuccess) { pl_set_error("pl_xim_decode: memset failed"); return PL_ERR_HIP; }
  hipLaunchKernelGGL(xim_chunk_bytes_kernel, dim3(n_chunks), dim3(kThreads), 0, st, lut, n_diffs, chunk, status);
  hipLaunchKernelGGL(xim_scan_chunks_kernel, dim3(1), dim3(kThreads), 0, st, chunk, n_chunks);
  hipLaunchKernelGGL(xim_gather_kernel<T>, dim3(n_chunks), dim3(kThreads), 0, st, lut, stream, stream_bytes, n_diffs,
                     n_plain, chunk, out, status);
  if (h > 1) {
    hipLaunchKernelGGL(xim_row_scan_kernel<T>, dim3(h - 1), dim3(kThreads), 0, st, out, w, row_total);
    hipLaunchKernelGGL(xim_carry_kernel<T>, dim3(1), dim3(64), 0, st, out, h, row_total);
    hipLaunchKernelGGL(xim_col_scan_kernel<T>, dim3((unsigned)pl_cdiv(w, kThreads)), dim3(kThreads), 0, st, out, h, w,
                       row_total);
  }
  return pl_check_launch("pl_xim_decode");
}

}  // namespace

extern "C" int64_t pl_xim_work_bytes(int width, int height) {
  const int64_t n_diffs = (int64_t)width * height - width - 1;
  const int64_t n_chunks = pl_cdiv(n_diffs > 0 ? n_diffs : 1, kChunk);
  return 16 + ((n_chunks * 4 + 15) & ~(int64_t)15) + (int64_t)height * 8;
}

extern "C" int pl_xim_decode(const unsigned char* d_lookup, int64_t lookup_bytes, const unsigned char* d_stream,
                             int64_t stream_bytes, int width, int height, int bytes_per_pixel, void* d_out,
                             unsigned char* d_work, void* stream) {
  PL_REQUIRE(d_lookup && d_stream && d_out && d_work, "null pointer");
  PL_REQUIRE(width > 0 && height > 0, "bad shape");
  hipStream_t st = (hipStream_t)stream;
  switch (bytes_per_pixel) {
    case 1: return xim_decode_t<signed char>(d_lookup, lookup_bytes, d_stream, stream_bytes, width, height, (signed char*)d_out, d_work, st);
    case 2: return xim_decode_t<short>(d_lookup, lookup_bytes, d_stream, stream_bytes, width, height, (short*)d_out, d_work, st);
    case 4: return xim_decode_t<int>(d_lookup, lookup_bytes, d_stream, stream_bytes, width, height, (int*)d_out, d_work, st);
    case 8: return xim_decode_t<long long>(d_lookup, lookup_bytes, d_stream, stream_bytes, width, height, (long long*)d_out, d_work, st);
    default:
      pl_set_error("pl_xim_decode: unsupported bytes per pixel %d", bytes_per_pixel);   // the reference raises ValueError
      return PL_ERR_UNSUPPORTED;
  }
}
