// 2-D / 1-D median filter with scipy semantics.
//
// Replaces: scipy.ndimage.median_filter(array, size=s) as called at
// pylinac/core/array_utils.py:131 (BaseImage.filter(kind="median"), pylinac/core/image.py:695-712;
// the PF noise filter pylinac/picketfence.py:226 uses size 3).
//
// Semantics: s x s window (length-s for a 1-D profile), mode='reflect', origin 0 (window starts
// at i - s/2), rank (s*s)/2, dtype preserved, exact.
//
// 3x3: each lane owns one column and slides down its rows; each new row contributes a SORTED
// horizontal triple (min3/med3/max3 on the VALU), triples are kept in registers for three rows,
// and the median of nine is med3(max3(lows), med3(mids), min3(highs)) -- 7 three-operand VALU ops
// per pixel, no LDS, no cross-lane traffic; the three loads per row are adjacent lanes' bytes and
// coalesce into the same 128-byte lines.
// General s: rank selection by bisection on an order-preserving integer key over an LDS tile.
#include "pl_common.h"
#include "median3_rows.h"

namespace {

constexpr int kThreads = 256;

// order-preserving key types --------------------------------------------------------------------
template <typename T> struct Key;
template <> struct Key<unsigned short> {
  using K = unsigned int; static constexpr int bits = 16;
  __device__ static K enc(unsigned short v) { return v; }
  __device__ static unsigned short dec(K k) { return (unsigned short)k; }
};
template <> struct Key<short> {
  using K = unsigned int; static constexpr int bits = 16;
  __device__ static K enc(short v) { return (unsigned int)((int)v + 32768); }
  __device__ static short dec(K k) { return (short)((int)k - 32768); }
};
template <> struct Key<unsigned char> {
  using K = unsigned int; static constexpr int bits = 8;
  __device__ static K enc(unsigned char v) { return v; }
  __device__ static unsigned char dec(K k) { return (unsigned char)k; }
};
template <> struct Key<int> {
  using K = unsigned int; static constexpr int bits = 32;
  __device__ static K enc(int v) { return (unsigned int)v ^ 0x80000000u; }
  __device__ static int dec(K k) { return (int)(k ^ 0x80000000u); }
};
template <> struct Key<long long> {
  using K = unsigned long long; static constexpr int bits = 64;
  __device__ static K enc(long long v) { return (unsigned long long)v ^ 0x8000000000000000ull; }
  __device__ static long long dec(K k) { return (long long)(k ^ 0x8000000000000000ull); }
};
template <> struct Key<float> {
  using K = unsigned int; static constexpr int bits = 32;
  __device__ static K enc(float v) {
    unsigned int u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  }
  __device__ static float dec(K k) {
    unsigned int u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
  }
};
template <> struct Key<double> {
  using K = unsigned long long; static constexpr int bits = 64;
  __device__ static K enc(double v) {
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
  }
  __device__ static double dec(K k) {
    unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
  }
};

template <typename K> __device__ __forceinline__ K kmin(K a, K b) { return a < b ? a : b; }
template <typename K> __device__ __forceinline__ K kmax(K a, K b) { return a > b ? a : b; }
template <typename K> __device__ __forceinline__ K kmed3(K a, K b, K c) {
  if constexpr (sizeof(K) == 4) return (K)pl_umed3((unsigned)a, (unsigned)b, (unsigned)c);
  else return kmax(kmin(a, b), kmin(kmax(a, b), c));
}

// ------------------------------------------------------------------------------------ 3x3 fast
template <typename T, int ROWS>
__global__ void __launch_bounds__(kThreads)
median3_kernel(const T* __restrict__ in, T* __restrict__ out, int h, int w, int col_tiles,
               int row_groups) {
  using K = typename Key<T>::K;
  unsigned id = pl_xcd_remap(blockIdx.x, gridDim.x);
  const int ct = id % col_tiles;
  id /= col_tiles;
  const int rg = id % row_groups;
  const size_t frame = id / row_groups;
  const int c = ct * kThreads + threadIdx.x;
  if (c >= w) return;
  const int r0 = rg * ROWS;
  const T* f = in + frame * (size_t)h * w;
  T* o = out + frame * (size_t)h * w;
  const int cl = pl_reflect(c - 1, w), cr = pl_reflect(c + 1, w);

  K lo[3], mi[3], hi[3];
  auto load_row = [&](int r, int slot) {
    const T* p = f + (size_t)pl_reflect(r, h) * w;
    K a = Key<T>::enc(p[cl]), b = Key<T>::enc(p[c]), d = Key<T>::enc(p[cr]);
    lo[slot] = kmin(kmin(a, b), d);
    hi[slot] = kmax(kmax(a, b), d);
    mi[slot] = kmed3(a, b, d);
  };
  load_row(r0 - 1, 0);
  load_row(r0, 1);
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const int r = r0 + i;
    if (r >= h) break;
    load_row(r + 1, (i + 2) % 3);
    K a = kmax(kmax(lo[0], lo[1]), lo[2]);
    K b = kmed3(mi[0], mi[1], mi[2]);
    K d = kmin(kmin(hi[0], hi[1]), hi[2]);
    o[(size_t)r * w + c] = Key<T>::dec(kmed3(a, b, d));
  }
}

// -------------------------------------------------------------- 3x3 fast, 16-bit dtypes, packed pairs
// One lane owns a PAIR of adjacent columns (one 32-bit word) and slides down ROWS rows.  Per row it
// loads the words left/centre/right of its pair (row pointers are wave-uniform -> saddr + constant
// lane offset, no per-load address arithmetic), forms the two sorted horizontal triples with
// v_min3 / v_max3 / v_med3 and emits the two medians as one packed 32-bit store: ~10 VALU ops and
// 0.75 vector-memory instructions per pixel (the per-pixel kernel above needs 20 and 4).
template <typename T, int ROWS>
__global__ void __launch_bounds__(kThreads)
median3_pair_kernel(const T* __restrict__ in, T* __restrict__ out, int h, int w, int pair_tiles,
                    int row_groups) {
  static_assert(sizeof(T) == 2, "packed-pair kernel is for 16-bit dtypes");
  unsigned id = pl_xcd_remap(blockIdx.x, gridDim.x);
  const int pt = id % pair_tiles;
  id /= pair_tiles;
  const int rg = id % row_groups;
  const size_t frame = id / row_groups;
  const int npairs = w >> 1;  // w is even (checked by the launcher)
  const int pi = pt * kThreads + threadIdx.x;
  if (pi >= npairs) return;
  const int r0 = rg * ROWS;
  const T* f = in + frame * (size_t)h * w;
  T* o = out + frame * (size_t)h * w;
  const bool has_l = pi > 0, has_r = pi + 1 < npairs;
  const unsigned offc = (unsigned)pi * 4u;
  const unsigned offl = has_l ? offc - 4u : offc, offr = has_r ? offc + 4u : offc;

  int lo0[3], mi0[3], hi0[3], lo1[3], mi1[3], hi1[3];
  auto load_row = [&](int r, int slot) {
    const char* row = reinterpret_cast<const char*>(f + (size_t)pl_reflect(r, h) * w);  // wave-uniform
    const unsigned L = *reinterpret_cast<const unsigned*>(row + offl);
    const unsigned C = *reinterpret_cast<const unsigned*>(row + offc);
    const unsigned R = *reinterpret_cast<const unsigned*>(row + offr);
    const int b = (int)(T)(C & 0xffffu), c = (int)(T)(C >> 16);
    // reflect at the frame's left/right edge: column -1 -> column 0, column w -> column w-1
    const int a = has_l ? (int)(T)(L >> 16) : b;
    const int d = has_r ? (int)(T)(R & 0xffffu) : c;
    lo0[slot] = min(min(a, b), c); hi0[slot] = max(max(a, b), c); mi0[slot] = pl_smed3(a, b, c);
    lo1[slot] = min(min(b, c), d); hi1[slot] = max(max(b, c), d); mi1[slot] = pl_smed3(b, c, d);
  };
  load_row(r0 - 1, 0);
  load_row(r0, 1);
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const int r = r0 + i;
    if (r >= h) break;
    load_row(r + 1, (i + 2) % 3);
    const int m0 = pl_smed3(max(max(lo0[0], lo0[1]), lo0[2]), pl_smed3(mi0[0], mi0[1], mi0[2]),
                            min(min(hi0[0], hi0[1]), hi0[2]));
    const int m1 = pl_smed3(max(max(lo1[0], lo1[1]), lo1[2]), pl_smed3(mi1[0], mi1[1], mi1[2]),
                            min(min(hi1[0], hi1[1]), hi1[2]));
    char* orow = reinterpret_cast<char*>(o + (size_t)r * w);
    *reinterpret_cast<unsigned*>(orow + offc) = ((unsigned)m0 & 0xffffu) | ((unsigned)m1 << 16);
  }
}

// 3x3 median, 16-bit, EIGHT columns per lane (one 16-byte load and one 16-byte store per row): pl_median3_rows
// (median3_rows.h) does the arithmetic; 0.5 vector-memory instructions per pixel.  gate: optional per-frame flags; a frame
// whose flag is 0 is skipped (the fused pipeline materialises the median plane only for frames its one-pass Otsu cannot hold).
// Gated launches (the Otsu fallback of the EPID pipeline) are PERSISTENT: a small fixed grid whose workgroups first look at
// the whole gate array -- nothing flagged (the common case): every workgroup leaves after one 1 KB read, which costs ~2 us
// of launch instead of the ~8 us an early-exiting full grid takes to dispatch -- and otherwise loop over the virtual blocks.
template <typename T, int ROWS>
__global__ void __launch_bounds__(kThreads)
median3_oct_kernel(const T* __restrict__ in, T* __restrict__ out, int h, int w, int col_waves, int row_groups,
                   int64_t n_waves, const int32_t* __restrict__ gate, int64_t n_frames, unsigned virtual_blocks) {
  const int lane = threadIdx.x & (PL_WAVE - 1);
  if (gate) {
    int any = 0;
    for (int64_t i = threadIdx.x; i < n_frames; i += kThreads) any |= gate[i];
    if (!__syncthreads_or(any)) return;
  }
  for (unsigned vb = blockIdx.x; vb < virtual_blocks; vb += gridDim.x) {
    const int64_t gw = (int64_t)pl_xcd_remap(vb, virtual_blocks) * (kThreads / PL_WAVE) + threadIdx.x / PL_WAVE;
    if (gw >= n_waves) continue;
    const int cw = (int)(gw % col_waves);
    const int64_t t = gw / col_waves;
    const int rg = (int)(t % row_groups);
    const size_t frame = (size_t)(t / row_groups);
    if (gate && gate[frame] == 0) continue;          // wave-uniform
    const int c0 = (cw * PL_WAVE + lane) * 8;        // first of the lane's 8 columns
    const bool active = c0 < w;                      // w % 8 == 0: all 8 inside or none
    const T* f = in + frame * (size_t)h * w;
    T* o = out + frame * (size_t)h * w;
    pl_median3_rows<T, ROWS>(f, h, w, c0, lane, rg * ROWS, [&](int r, const int (&m)[8]) {
      if (active)
        *reinterpret_cast<uint4*>(o + (size_t)r * w + c0) =
            uint4{pl_pack16(m[0], m[1]), pl_pack16(m[2], m[3]), pl_pack16(m[4], m[5]), pl_pack16(m[6], m[7])};
    });
  }
}

// --------------------------------------------------------------------- general size (LDS tile)
// 16x16 outputs per block; tile (16+sh-1) x (16+sw-1) keys staged in LDS; each lane bisects the
// key space: smallest key v with  #{window <= v} >= rank+1.
template <typename T>
__global__ void __launch_bounds__(kThreads)
median_general_kernel(const T* __restrict__ in, T* __restrict__ out, int h, int w, int sh, int sw,
                      int tiles_x, int tiles_y, int rank) {
  using KT = Key<T>;
  using K = typename KT::K;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  K* tile = reinterpret_cast<K*>(smem_raw);
  constexpr int TS = 16;
  unsigned id = blockIdx.x;
  const int tx = id % tiles_x;
  id /= tiles_x;
  const int ty = id % tiles_y;
  const size_t frame = id / tiles_y;
  const T* f = in + frame * (size_t)h * w;
  T* o = out + frame * (size_t)h * w;
  const int th = TS + sh - 1, tw = TS + sw - 1;
  const int r_base = ty * TS - sh / 2, c_base = tx * TS - sw / 2;
  for (int e = threadIdx.x; e < th * tw; e += kThreads) {
    int rr = pl_reflect(r_base + e / tw, h), cc = pl_reflect(c_base + e % tw, w);
    tile[e] = KT::enc(f[(size_t)rr * w + cc]);
  }
  __syncthreads();
  const int ly = threadIdx.x / TS, lx = threadIdx.x % TS;
  const int r = ty * TS + ly, c = tx * TS + lx;
  if (r >= h || c >= w) return;
  const K* win = tile + ly * tw + lx;
  K lo = 0, hi = (KT::bits == 64) ? ~(K)0 : (K)(((unsigned long long)1 << KT::bits) - 1);
  const int need = rank + 1;
  while (lo < hi) {
    K mid = lo + (hi - lo) / 2;
    int cnt = 0;
    for (int y = 0; y < sh; ++y)
      for (int x = 0; x < sw; ++x) cnt += (win[y * tw + x] <= mid) ? 1 : 0;
    if (cnt >= need) hi = mid; else lo = mid + 1;
  }
  o[(size_t)r * w + c] = KT::dec(lo);
}

template <typename T>
int median_t(const T* in, T* out, int64_t n, int h, int w, int size, hipStream_t st, const int32_t* gate = nullptr) {
  if (size == 1) {
    hipError_t e = hipMemcpyAsync(out, in, (size_t)n * h * w * sizeof(T), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) { pl_set_error("pl_median2d: copy failed: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
    return PL_OK;
  }
  if constexpr (sizeof(T) == 2) {
    if (size == 3 && h > 1 && w >= 8 && (w & 7) == 0 && ((reinterpret_cast<uintptr_t>(in) & 15) == 0) &&
        ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && (((size_t)h * w) & 7) == 0) {
      constexpr int ROWS = 16;
      const int col_waves = (int)pl_cdiv(w / 8, PL_WAVE), row_groups = (int)pl_cdiv(h, ROWS);
      const int64_t n_waves = n * col_waves * row_groups;
      const int64_t blocks = pl_cdiv(n_waves, kThreads / PL_WAVE);
      if (blocks > 0x7fffffffLL) { pl_set_error("pl_median2d: batch too large"); return PL_ERR_INVALID_ARG; }
      const unsigned grid = gate ? (unsigned)(blocks < 512 ? blocks : 512) : (unsigned)blocks;
      hipLaunchKernelGGL((median3_oct_kernel<T, ROWS>), dim3(grid), dim3(kThreads), 0, st, in, out, h, w,
                         col_waves, row_groups, n_waves, gate, n, (unsigned)blocks);
      return pl_check_launch("pl_median2d");
    }
    if (size == 3 && h > 1 && w >= 4 && (w & 1) == 0 && ((reinterpret_cast<uintptr_t>(in) & 3) == 0) &&
        ((reinterpret_cast<uintptr_t>(out) & 3) == 0)) {
      constexpr int ROWS = 16;
      int pair_tiles = (int)pl_cdiv(w / 2, kThreads), row_groups = (int)pl_cdiv(h, ROWS);
      int64_t blocks = n * pair_tiles * row_groups;
      if (blocks > 0x7fffffffLL) { pl_set_error("pl_median2d: batch too large"); return PL_ERR_INVALID_ARG; }
      hipLaunchKernelGGL((median3_pair_kernel<T, ROWS>), dim3((unsigned)blocks), dim3(kThreads), 0, st, in, out,
                         h, w, pair_tiles, row_groups);
      return pl_check_launch("pl_median2d");
    }
  }
  if (size == 3 && h > 1) {
    constexpr int ROWS = 16;
    int col_tiles = (int)pl_cdiv(w, kThreads), row_groups = (int)pl_cdiv(h, ROWS);
    int64_t blocks = n * col_tiles * row_groups;
    if (blocks > 0x7fffffffLL) { pl_set_error("pl_median2d: batch too large"); return PL_ERR_INVALID_ARG; }
    hipLaunchKernelGGL((median3_kernel<T, ROWS>), dim3((unsigned)blocks), dim3(kThreads), 0, st, in,
                       out, h, w, col_tiles, row_groups);
    return pl_check_launch("pl_median2d");
  }
  const int sh = (h > 1) ? size : 1, sw = size;
  const int rank = (sh * sw) / 2;
  const size_t lds = (size_t)(16 + sh - 1) * (16 + sw - 1) * sizeof(typename Key<T>::K);
  if (lds > 160 * 1024) { pl_set_error("pl_median2d: window %d too large", size); return PL_ERR_UNSUPPORTED; }
  int tiles_x = (int)pl_cdiv(w, 16), tiles_y = (int)pl_cdiv(h, 16);
  int64_t blocks = n * tiles_x * tiles_y;
  if (blocks > 0x7fffffffLL) { pl_set_error("pl_median2d: batch too large"); return PL_ERR_INVALID_ARG; }
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)median_general_kernel<T>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { pl_set_error("pl_median2d: LDS attribute: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
  }
  hipLaunchKernelGGL(median_general_kernel<T>, dim3((unsigned)blocks), dim3(kThreads), lds, st, in, out,
                     h, w, sh, sw, tiles_x, tiles_y, rank);
  return pl_check_launch("pl_median2d");
}

}  // namespace

extern "C" int pl_median2d(const void* in, void* out, int dtype, int64_t n, int h, int w, int size,
                           void* stream) {
  PL_REQUIRE(in && out && in != out, "null or aliased pointers");
  PL_REQUIRE(n >= 0 && h > 0 && w > 0, "bad shape");
  PL_REQUIRE(size >= 1, "size must be >= 1");
  if (n == 0) return PL_OK;
  hipStream_t st = (hipStream_t)stream;
  PL_DISPATCH_DTYPE(dtype, T, return median_t<T>((const T*)in, (T*)out, n, h, w, size, st));
  return PL_OK;
}

// 3x3 median of the frames whose gate flag is non-zero only (16-bit, geometry of pl_median3_rows_covers): the fused
// median + Otsu path's way of materialising the median plane for the frames its LDS window cannot hold.  0 = launched.
int pl_median3_gated(const void* in, void* out, int is_signed, int64_t n, int h, int w, const int32_t* d_gate, hipStream_t st) {
  if (!pl_median3_rows_covers(in, h, w) || (reinterpret_cast<uintptr_t>(out) & 15)) return -1;
  return is_signed ? median_t<short>((const short*)in, (short*)out, n, h, w, 3, st, d_gate)
                   : median_t<unsigned short>((const unsigned short*)in, (unsigned short*)out, n, h, w, 3, st, d_gate);
}
