// Connected components, hole filling and binary centroids.
//
// Replaces:
//   scipy.ndimage.binary_fill_holes(mask)   pylinac/winston_lutz.py:777  (default structure:
//        4-connected background; a hole = background component not connected to the border)
//   scipy.ndimage.center_of_mass(mask)      pylinac/winston_lutz.py:778  (mean of the True
//        coordinates: exact integer sums / count in float64)
//   skimage.measure.label(mask, connectivity) pylinac/metrics/utils.py:131, pylinac/ct.py:3345
//        (labels numbered in raster order of each component's first pixel)
//
// Labelling is a lock-free union-find on the frame's own index space (Playne/Hawick/Komura):
// every foreground pixel starts as its own root, is united with its already-visited neighbours
// (left, up; plus the two upper diagonals for 8-connectivity) with atomicMin links that always
// point to the SMALLER index, then paths are compressed.  The root of a component is therefore
// its first pixel in raster order -- exactly the order skimage/scipy number labels in -- and
// sequential numbering is a prefix count of roots.
//
// Run-based start (fewer atomics on large regions: the background of a Winston-Lutz frame is ONE component of a
// million pixels): the init kernel links every foreground pixel straight to the first pixel of its horizontal run
// inside its 64-lane chunk (ballot + count-leading-zeros, no atomics), so the merge kernel only unites
//   * a chunk's first pixel with its left neighbour (runs that cross a chunk boundary),
//   * the FIRST pixel of each overlap between a run and the run above it (pixel i skips the union with i-w when
//     i-1 and i-w-1 are foreground too: i ~ i-1 ~ i-w-1 ~ i-w already),
//   * 8-connectivity: the upper-left diagonal only when neither i-1 nor i-w is foreground, the upper-right one only
//     when i-w is not (otherwise the row above connects them).
// Every skipped union is implied by ones that other lanes perform in the same launch; roots stay the smallest index.
#include "pl_common.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ int find_root(const int* __restrict__ L, int i) {
  int p = L[i];
  while (p != i) {
    i = p;
    p = L[i];
  }
  return i;
}

__device__ __forceinline__ void unite(int* L, int a, int b) {
  bool done;
  do {
    a = find_root(L, a);
    b = find_root(L, b);
    if (a < b) {
      const int old = atomicMin(&L[b], a);
      done = (old == b);
      b = old;
    } else if (b < a) {
      const int old = atomicMin(&L[a], b);
      done = (old == a);
      a = old;
    } else {
      done = true;
    }
  } while (!done);
}

// Measured on MI355X (profiles/r02a_first_call_summary.txt, config #4: 512 x 1024^2 frames, fill-holes of a field mask
// whose background is one component of a million pixels): per-pixel start (one or two atomic unions per pixel, round 1)
// 133.5 ms, run-based start 20.3 ms.  The per-pixel variant is gone.
// fg(i) = (mask[i] != 0) ^ invert.  L[i] = first pixel of i's run within its 64-lane chunk (lanes = consecutive pixels)
__global__ void __launch_bounds__(kThreads)
ccl_init_kernel(const uint8_t* __restrict__ mask, int invert, int64_t total, int64_t per_frame, int w,
                int* __restrict__ L) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const bool in_range = g < total;
  const int i = in_range ? (int)(g % per_frame) : 0;
  const bool fg = in_range && (((mask[g] != 0) ? 1 : 0) != invert);
  const int lane = threadIdx.x & 63;
  const unsigned long long fgm = __ballot(fg);
  // a run starts where the previous lane is background, or the pixel opens a row (i % w == 0 also covers a new frame)
  const bool starts = fg && (lane == 0 || !((fgm >> (lane - 1)) & 1ull) || (i % w) == 0);
  const unsigned long long sm = __ballot(starts);
  if (!in_range) return;
  if (!fg) { L[g] = -1; return; }
  const unsigned long long upto = sm & (lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull));
  const int start_lane = 63 - __builtin_clzll(upto);      // upto != 0: the run containing this lane has a start
  L[g] = i - (lane - start_lane);
}

__global__ void __launch_bounds__(kThreads)
ccl_merge_kernel(int* __restrict__ Lall, int64_t total, int h, int w, int conn8) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= total) return;
  const int64_t per_frame = (int64_t)h * w;
  int* L = Lall + (g / per_frame) * per_frame;
  const int i = (int)(g % per_frame);
  if (L[i] < 0) return;
  const int r = i / w, c = i % w;
  const bool left = c > 0 && L[i - 1] >= 0;
  // same mapping of lanes to pixels as ccl_init_kernel: lane 0 of a chunk is where a run may continue from the left
  if (left && (threadIdx.x & 63) == 0) unite(L, i, i - 1);
  if (r > 0) {
    const bool up = L[i - w] >= 0;
    const bool upleft = c > 0 && L[i - w - 1] >= 0;
    if (up && !(left && upleft)) unite(L, i, i - w);
    if (conn8) {
      if (upleft && !up && !left) unite(L, i, i - w - 1);
      if (!up && c + 1 < w && L[i - w + 1] >= 0) unite(L, i, i - w + 1);
    }
  }
}

// Flatten the forest.  After the run-based merge a pixel points to the head of its run, run heads point along the row
// and row heads point up a chain whose length depends on the order in which the hardware happened to run the unions
// (up to the number of rows in flight).  Pass 1 (heads_only) lets only the run heads -- row starts, chunk starts and
// pixels whose left neighbour is background: the nodes other pixels point to -- walk that chain; pass 2 then costs
// every pixel at most two hops.  Without pass 1 a million background pixels would each walk the whole chain.
__global__ void __launch_bounds__(kThreads)
ccl_compress_kernel(int* __restrict__ Lall, int64_t total, int64_t per_frame, int w, int heads_only) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= total) return;
  int* L = Lall + (g / per_frame) * per_frame;
  const int i = (int)(g % per_frame);
  if (L[i] < 0) return;
  if (heads_only && !((i % w) == 0 || (threadIdx.x & 63) == 0 || L[i - 1] < 0)) return;
  L[i] = find_root(L, i);
}

// labels := root index + 1 (0 = background); used as the public "raw" labelling
__global__ void ccl_export_kernel(const int* __restrict__ L, int64_t total, int32_t* __restrict__ out) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= total) return;
  out[g] = L[g] < 0 ? 0 : L[g] + 1;
}

// sequential numbering: one 1024-lane workgroup per frame walks the frame in raster order with a
// running count of roots; rank[root] is written in place of the root's own entry (as -(rank) - 2 so
// that it cannot be mistaken for an index), then every pixel reads its root's rank.
__global__ void __launch_bounds__(1024)
ccl_rank_roots_kernel(int* __restrict__ Lall, int64_t per_frame, int32_t* __restrict__ nlabels) {
  __shared__ int wave_tot[16];
  __shared__ int base;
  int* L = Lall + (int64_t)blockIdx.x * per_frame;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int64_t off = 0; off < per_frame; off += 1024) {
    const int64_t i = off + threadIdx.x;
    const int is_root = (i < per_frame && L[i] == (int)i) ? 1 : 0;
    const unsigned long long b = __ballot(is_root);
    const int pre = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wv] = __popcll(b);
    __syncthreads();
    int wbase = 0, tot = 0;
    for (int k = 0; k < 16; ++k) {
      if (k < wv) wbase += wave_tot[k];
      tot += wave_tot[k];
    }
    const int cur = base;
    if (is_root) L[i] = -(cur + wbase + pre + 1) - 1;  // label k (1-based) stored as -(k) - 1 <= -2
    __syncthreads();
    if (threadIdx.x == 0) base = cur + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0 && nlabels) nlabels[blockIdx.x] = base;
}

__global__ void ccl_apply_rank_kernel(const int* __restrict__ Lall, int64_t total, int64_t per_frame,
                                      int32_t* __restrict__ out) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= total) return;
  const int* L = Lall + (g / per_frame) * per_frame;
  int v = L[g % per_frame];
  if (v == -1) { out[g] = 0; return; }
  if (v >= 0) v = L[v];  // non-root pixel: its root holds the encoded rank
  out[g] = -(v + 1);
}

// ---- fill holes: background components that do not touch the frame border become foreground ----
__global__ void border_flag_kernel(const int* __restrict__ Lall, int64_t n, int h, int w,
                                   uint8_t* __restrict__ flags /* [n][h*w], zeroed */) {
  const int64_t per_frame = (int64_t)h * w;
  const int64_t border = 2LL * w + 2LL * h;
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= n * border) return;
  const int64_t frame = g / border;
  int64_t k = g % border;
  int r, c;
  if (k < w) { r = 0; c = (int)k; }
  else if (k < 2LL * w) { r = h - 1; c = (int)(k - w); }
  else if (k < 2LL * w + h) { r = (int)(k - 2LL * w); c = 0; }
  else { r = (int)(k - 2LL * w - h); c = w - 1; }
  const int* L = Lall + frame * per_frame;
  const int root = L[(int64_t)r * w + c];
  if (root >= 0) flags[frame * per_frame + root] = 1;
}

__global__ void fill_apply_kernel(const uint8_t* __restrict__ mask, const int* __restrict__ Lall,
                                  const uint8_t* __restrict__ flags, int64_t total, int64_t per_frame,
                                  uint8_t* __restrict__ out) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= total) return;
  const int64_t fbase = (g / per_frame) * per_frame;
  const int root = Lall[g];  // labelling of the BACKGROUND: -1 where the mask is set
  uint8_t v = mask[g] != 0 ? 1 : 0;
  if (root >= 0 && !flags[fbase + root]) v = 1;
  out[g] = v;
}

// ---- binary centroid: exact integer sums of row / column indices and the pixel count -----------
__global__ void __launch_bounds__(kThreads)
centroid_kernel(const uint8_t* __restrict__ mask, int h, int w, int bpf,
                unsigned long long* __restrict__ sums /* [n][3], zeroed */) {
  const int64_t frame = blockIdx.x / bpf;
  const int chunk = blockIdx.x % bpf;
  const int64_t per_frame = (int64_t)h * w;
  const uint8_t* m = mask + frame * per_frame;
  unsigned long long sr = 0, sc = 0, cnt = 0;
  const int64_t lo = (int64_t)chunk * 65536, hi = (lo + 65536 < per_frame) ? lo + 65536 : per_frame;
  for (int64_t i = lo + threadIdx.x; i < hi; i += kThreads) {
    if (m[i]) { sr += (unsigned long long)(i / w); sc += (unsigned long long)(i % w); ++cnt; }
  }
  auto add = [](unsigned long long a, unsigned long long b) { return a + b; };
  sr = pl_wave_reduce(sr, add); sc = pl_wave_reduce(sc, add); cnt = pl_wave_reduce(cnt, add);
  if ((threadIdx.x & 63) == 0 && cnt) {
    atomicAdd(&sums[frame * 3 + 0], sr);
    atomicAdd(&sums[frame * 3 + 1], sc);
    atomicAdd(&sums[frame * 3 + 2], cnt);
  }
}

__global__ void centroid_finish_kernel(const unsigned long long* __restrict__ sums, int64_t n,
                                       double* __restrict__ out /* [n][3]: row, col, count */) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const double cnt = (double)sums[i * 3 + 2];
  out[i * 3 + 0] = (double)sums[i * 3 + 0] / cnt;  // 0/0 -> NaN, like scipy on an empty mask
  out[i * 3 + 1] = (double)sums[i * 3 + 1] / cnt;
  out[i * 3 + 2] = cnt;
}

// a >= thr after ground+normalize in the reference's float64 arithmetic: ((a - sub) / div) >= thr
template <typename T>
__global__ void scaled_binary_kernel(const T* __restrict__ in, int64_t total, int64_t per_frame,
                                     const double* __restrict__ sub, const double* __restrict__ div,
                                     const double* __restrict__ thr, uint8_t* __restrict__ out) {
  const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (g >= total) return;
  const int64_t f = g / per_frame;
  const double grounded = (double)in[g] - sub[f];  // exact for integer dtypes (array - array.min())
  out[g] = (grounded / div[f] >= thr[f]) ? 1 : 0;
}

int run_ccl(const uint8_t* mask, int invert, int64_t n, int h, int w, int conn, int* L, hipStream_t st) {
  const int64_t per_frame = (int64_t)h * w, total = n * per_frame;
  const unsigned blocks = (unsigned)pl_cdiv(total, kThreads);
  hipLaunchKernelGGL(ccl_init_kernel, dim3(blocks), dim3(kThreads), 0, st, mask, invert, total, per_frame, w, L);
  hipLaunchKernelGGL(ccl_merge_kernel, dim3(blocks), dim3(kThreads), 0, st, L, total, h, w, conn == 8 ? 1 : 0);
  hipLaunchKernelGGL(ccl_compress_kernel, dim3(blocks), dim3(kThreads), 0, st, L, total, per_frame, w, 1);
  hipLaunchKernelGGL(ccl_compress_kernel, dim3(blocks), dim3(kThreads), 0, st, L, total, per_frame, w, 0);
  return pl_check_launch("ccl");
}

}  // namespace

// shared with ct.hip (clear_border): union-find roots of the (optionally inverted) mask
int pl_ccl_roots(const uint8_t* mask, int invert, int64_t n, int h, int w, int conn, int* L, hipStream_t st) {
  return run_ccl(mask, invert, n, h, w, conn, L, st);
}

#define PL_CCL_CHECK_SHAPE()                                                          \
  PL_REQUIRE(n >= 0 && h > 0 && w > 0, "bad shape");                                  \
  PL_REQUIRE((int64_t)h * w <= 0x7fffffffLL, "frame too large for 32-bit labels");    \
  PL_REQUIRE(pl_cdiv(n * (int64_t)h * w, kThreads) <= 0x7fffffffLL, "batch too large"); \
  if (n == 0) return PL_OK;

extern "C" int pl_label(const uint8_t* d_mask, int64_t n, int h, int w, int connectivity,
                        int32_t* d_labels, int32_t* d_work, int32_t* d_nlabels, void* stream) {
  PL_REQUIRE(d_mask && d_labels && d_work, "null pointer");
  PL_REQUIRE(connectivity == 4 || connectivity == 8, "connectivity must be 4 or 8");
  PL_CCL_CHECK_SHAPE();
  hipStream_t st = (hipStream_t)stream;
  const int64_t per_frame = (int64_t)h * w, total = n * per_frame;
  if (int rc = run_ccl(d_mask, 0, n, h, w, connectivity, d_work, st)) return rc;
  PL_REQUIRE(n <= 0x7fffffffLL, "batch too large");
  hipLaunchKernelGGL(ccl_rank_roots_kernel, dim3((unsigned)n), dim3(1024), 0, st, d_work, per_frame, d_nlabels);
  hipLaunchKernelGGL(ccl_apply_rank_kernel, dim3((unsigned)pl_cdiv(total, kThreads)), dim3(kThreads), 0, st,
                     d_work, total, per_frame, d_labels);
  return pl_check_launch("pl_label");
}

extern "C" int pl_fill_holes(const uint8_t* d_mask, uint8_t* d_out, int64_t n, int h, int w,
                             int connectivity_bg, int32_t* d_work, uint8_t* d_flags, void* stream) {
  PL_REQUIRE(d_mask && d_out && d_work && d_flags, "null pointer");
  PL_REQUIRE(connectivity_bg == 4 || connectivity_bg == 8, "connectivity must be 4 or 8");
  PL_CCL_CHECK_SHAPE();
  hipStream_t st = (hipStream_t)stream;
  const int64_t per_frame = (int64_t)h * w, total = n * per_frame;
  if (int rc = run_ccl(d_mask, 1, n, h, w, connectivity_bg, d_work, st)) return rc;
  hipError_t e = hipMemsetAsync(d_flags, 0, (size_t)total, st);
  if (e != hipSuccess) { pl_set_error("pl_fill_holes: memset: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
  const int64_t border = n * (2LL * w + 2LL * h);
  hipLaunchKernelGGL(border_flag_kernel, dim3((unsigned)pl_cdiv(border, kThreads)), dim3(kThreads), 0, st, d_work,
                     n, h, w, d_flags);
  hipLaunchKernelGGL(fill_apply_kernel, dim3((unsigned)pl_cdiv(total, kThreads)), dim3(kThreads), 0, st, d_mask,
                     d_work, d_flags, total, per_frame, d_out);
  return pl_check_launch("pl_fill_holes");
}

extern "C" int pl_binary_centroid(const uint8_t* d_mask, int64_t n, int h, int w,
                                  unsigned long long* d_sums, double* d_out, void* stream) {
  PL_REQUIRE(d_mask && d_sums && d_out, "null pointer");
  PL_CCL_CHECK_SHAPE();
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(d_sums, 0, (size_t)n * 3 * sizeof(unsigned long long), st);
  if (e != hipSuccess) { pl_set_error("pl_binary_centroid: memset: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
  const int bpf = (int)pl_cdiv((int64_t)h * w, 65536);
  PL_REQUIRE(n * bpf <= 0x7fffffffLL, "batch too large");
  hipLaunchKernelGGL(centroid_kernel, dim3((unsigned)(n * bpf)), dim3(kThreads), 0, st, d_mask, h, w, bpf, d_sums);
  hipLaunchKernelGGL(centroid_finish_kernel, dim3((unsigned)pl_cdiv(n, kThreads)), dim3(kThreads), 0, st, d_sums, n,
                     d_out);
  return pl_check_launch("pl_binary_centroid");
}

extern "C" int pl_scaled_binary(const void* in, int dtype, int64_t n, int64_t count, const double* d_sub,
                                const double* d_div, const double* d_thr, uint8_t* d_out, void* stream) {
  PL_REQUIRE(in && d_sub && d_div && d_thr && d_out, "null pointer");
  PL_REQUIRE(n >= 0 && count > 0, "bad shape");
  if (n == 0) return PL_OK;
  const int64_t total = n * count;
  PL_REQUIRE(pl_cdiv(total, kThreads) <= 0x7fffffffLL, "batch too large");
  PL_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL(scaled_binary_kernel<T>, dim3((unsigned)pl_cdiv(total, kThreads)),
                                       dim3(kThreads), 0, (hipStream_t)stream, (const T*)in, total, count, d_sub,
                                       d_div, d_thr, d_out));
  return pl_check_launch("pl_scaled_binary");
}

// ---- fused field CAX: threshold -> binary_fill_holes -> center_of_mass without a mask or a label plane --------------------
// pylinac/winston_lutz.py:775-779 per frame.  Holes (background components that miss the frame border) can only lie
// inside the bounding box of the foreground, and every background pixel outside that box reaches the frame border by
// walking straight outwards.  So: pass 1 streams the frame once and reduces the foreground's count, coordinate sums and
// bounding box; pass 2 (one workgroup per frame) re-thresholds only the box grown by one pixel into LDS, flood-fills the
// background from the window border (4-connected, scipy's default structure) and adds the unreached background pixels
// to the sums.  A Winston-Lutz field is ~60 pixels wide in a 1024^2 frame: the window is 62 x 62 instead of a
// million-pixel labelling problem (config #4: 39 us per frame with the general path).
namespace {

constexpr int kCaxMaxWindow = 147456;   // 384 x 384 bytes of LDS for the window mask

// ((double)a - s) / d >= t is monotone in a (d > 0): for 16-bit frames the smallest integer that passes is found by
// bisection with the SAME float64 expression, and every pixel is then one integer comparison instead of a float64
// division.  Returns 65536 (unsigned) / 32768 (signed) when no value passes.  d <= 0 or NaN: callers keep the float test.
template <typename T>
__device__ __forceinline__ int cax_int_threshold(double s, double d, double t) {
  constexpr int kLo = (T)-1 < (T)0 ? -32768 : 0, kHi = (T)-1 < (T)0 ? 32767 : 65535;
  auto pass = [&](int a) { return ((double)a - s) / d >= t; };
  if (!pass(kHi)) return kHi + 1;
  int lo = kLo, hi = kHi;               // invariant: pass(hi)
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    if (pass(mid)) hi = mid; else lo = mid + 1;
  }
  return hi;
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
cax_reduce_kernel(const T* __restrict__ in, int h, int w, int bpf, const double* __restrict__ sub,
                  const double* __restrict__ div, const double* __restrict__ thr,
                  unsigned long long* __restrict__ acc /* [n][8]: cnt, sum r, sum c, rmin, rmax, cmin, cmax, pad */) {
  const int64_t frame = blockIdx.x / bpf;
  const int chunk = blockIdx.x % bpf;
  const int64_t per_frame = (int64_t)h * w;
  const T* f = in + frame * per_frame;
  const double s = sub[frame], d = div[frame], t = thr[frame];
  unsigned long long cnt = 0, sr = 0, sc = 0;
  unsigned rmin = 0xffffffffu, rmax = 0, cmin = 0xffffffffu, cmax = 0;
  const int64_t lo = (int64_t)chunk * 65536, hi = (lo + 65536 < per_frame) ? lo + 65536 : per_frame;
  const bool use_int = sizeof(T) == 2 && d > 0.0;
  int ithr = 0;
  if constexpr (sizeof(T) == 2) { if (use_int) ithr = cax_int_threshold<T>(s, d, t); }
  if (sizeof(T) == 2 && use_int && (w & 7) == 0 && (reinterpret_cast<uintptr_t>(f) & 15) == 0) {
    // eight pixels of one row per 16-byte load; the row / column of the vector is divided out once, not per pixel
    const uint4* vf = reinterpret_cast<const uint4*>(f);
    for (int64_t v = lo / 8 + threadIdx.x; v < hi / 8; v += kThreads) {
      const uint4 q = vf[v];
      const unsigned wd[4] = {q.x, q.y, q.z, q.w};
      const int64_t i = v * 8;
      const unsigned r = (unsigned)(i / w), c0 = (unsigned)(i % w);
      unsigned m = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int a0 = (int)(T)(wd[k] & 0xffffu), a1 = (int)(T)(wd[k] >> 16);
        m |= (a0 >= ithr ? 1u : 0u) << (2 * k);
        m |= (a1 >= ithr ? 1u : 0u) << (2 * k + 1);
      }
      if (m) {
        const unsigned n = (unsigned)__popc(m);
        unsigned csum = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) csum += ((m >> k) & 1u) * (c0 + (unsigned)k);
        cnt += n; sr += (unsigned long long)r * n; sc += csum;
        rmin = r < rmin ? r : rmin; rmax = r > rmax ? r : rmax;
        const unsigned cl = c0 + (unsigned)__builtin_ctz(m), ch = c0 + 31u - (unsigned)__builtin_clz(m);
        cmin = cl < cmin ? cl : cmin; cmax = ch > cmax ? ch : cmax;
      }
    }
    for (int64_t i = (hi / 8) * 8 + threadIdx.x; i < hi; i += kThreads) {   // tail of a chunk that is not a multiple of 8
      if ((int)f[i] >= ithr) {
        const unsigned r = (unsigned)(i / w), c = (unsigned)(i % w);
        ++cnt; sr += r; sc += c;
        rmin = r < rmin ? r : rmin; rmax = r > rmax ? r : rmax;
        cmin = c < cmin ? c : cmin; cmax = c > cmax ? c : cmax;
      }
    }
  } else
  for (int64_t i = lo + threadIdx.x; i < hi; i += kThreads) {
    bool fg;
    if (use_int) {
      fg = (int)f[i] >= ithr;
    } else {
      const double grounded = (double)f[i] - s;   // exact for integer dtypes (array - array.min())
      fg = grounded / d >= t;
    }
    if (fg) {
      const unsigned r = (unsigned)(i / w), c = (unsigned)(i % w);
      ++cnt; sr += r; sc += c;
      rmin = r < rmin ? r : rmin; rmax = r > rmax ? r : rmax;
      cmin = c < cmin ? c : cmin; cmax = c > cmax ? c : cmax;
    }
  }
  auto add = [](unsigned long long a, unsigned long long b) { return a + b; };
  auto mn = [](unsigned a, unsigned b) { return a < b ? a : b; };
  auto mx = [](unsigned a, unsigned b) { return a > b ? a : b; };
  cnt = pl_wave_reduce(cnt, add); sr = pl_wave_reduce(sr, add); sc = pl_wave_reduce(sc, add);
  rmin = pl_wave_reduce(rmin, mn); rmax = pl_wave_reduce(rmax, mx);
  cmin = pl_wave_reduce(cmin, mn); cmax = pl_wave_reduce(cmax, mx);
  if ((threadIdx.x & 63) == 0 && cnt) {
    unsigned long long* a = acc + frame * 8;
    atomicAdd(&a[0], cnt); atomicAdd(&a[1], sr); atomicAdd(&a[2], sc);
    atomicMin(&a[3], (unsigned long long)rmin); atomicMax(&a[4], (unsigned long long)rmax);
    atomicMin(&a[5], (unsigned long long)cmin); atomicMax(&a[6], (unsigned long long)cmax);
  }
}

// The same reduction driven by the 512-pixel tile maxima pl_hist16_tiles left behind: a tile whose largest key is below the
// integer threshold cannot hold a foreground pixel and is never read -- a 20 x 20 mm field on a 1024^2 panel is a few dozen of
// the frame's 2 048 tiles, so the second full read of the batch (r04z: 0.20 ms per 512 frames) becomes a 4 KiB table scan
// per frame.  16-bit frames of a whole number of tiles with 16-byte aligned rows of 8-pixel vectors; exact integer
// moments like cax_reduce_kernel (the sums are order-independent).  A frame whose divisor is not positive keeps the float64
// test and looks at every tile.
template <typename T>
__global__ void __launch_bounds__(kThreads)
cax_reduce_tiles_kernel(const T* __restrict__ in, int h, int w, const double* __restrict__ sub, const double* __restrict__ div,
                        const double* __restrict__ thr, const unsigned short* __restrict__ tile_max,
                        unsigned long long* __restrict__ acc) {
  // ONE workgroup per frame: (1) every thread looks at its share of the tile maxima and the tiles that may hold foreground
  // are collected in LDS (any order: the sums do not care); (2) the waves take the candidates in turn, FOUR tiles' loads in
  // flight per wave -- the first version walked its candidates one dependent load at a time: 62 us per 512 frames, all of it
  // latency (r05c), against 199 for the full pass
  __shared__ unsigned short s_tiles[4096];
  __shared__ int s_count;
  const int64_t frame = blockIdx.x;
  const int64_t per_frame = (int64_t)h * w;
  const int ntiles = (int)(per_frame >> 9);
  const T* f = in + frame * per_frame;
  const unsigned short* tm = tile_max + frame * ntiles;
  const double s = sub[frame], d = div[frame], t = thr[frame];
  const bool use_int = d > 0.0;
  const int ithr = use_int ? cax_int_threshold<T>(s, d, t) : 0;
  constexpr unsigned flip = (T)-1 < (T)0 ? 0x8000u : 0u;
  // key(v) >= key(ithr) <=> v >= ithr; an ithr beyond the type's range (nothing passes) maps beyond every 16-bit key
  const unsigned kthr = use_int ? (unsigned)(ithr + (int)flip) : 0u;
  unsigned long long cnt = 0, sr = 0, sc = 0;
  unsigned rmin = 0xffffffffu, rmax = 0, cmin = 0xffffffffu, cmax = 0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint4* vf = reinterpret_cast<const uint4*>(f);
  auto take = [&](int tile, const uint4& q) {
    const unsigned wd[4] = {q.x, q.y, q.z, q.w};
    const int64_t i = ((int64_t)tile * 64 + lane) * 8;
    const unsigned r = (unsigned)(i / w), c0 = (unsigned)(i % w);
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int a0 = (int)(T)(wd[k] & 0xffffu), a1 = (int)(T)(wd[k] >> 16);
      const bool p0 = use_int ? a0 >= ithr : ((double)a0 - s) / d >= t;
      const bool p1 = use_int ? a1 >= ithr : ((double)a1 - s) / d >= t;
      m |= (p0 ? 1u : 0u) << (2 * k);
      m |= (p1 ? 1u : 0u) << (2 * k + 1);
    }
    if (m) {
      const unsigned n = (unsigned)__popc(m);
      unsigned csum = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) csum += ((m >> k) & 1u) * (c0 + (unsigned)k);
      cnt += n; sr += (unsigned long long)r * n; sc += csum;
      rmin = r < rmin ? r : rmin; rmax = r > rmax ? r : rmax;
      const unsigned cl = c0 + (unsigned)__builtin_ctz(m), ch = c0 + 31u - (unsigned)__builtin_clz(m);
      cmin = cl < cmin ? cl : cmin; cmax = ch > cmax ? ch : cmax;
    }
  };
  for (int base = 0; base < ntiles; base += 4096) {                   // (a 1024^2 frame has 2 048 tiles: one round)
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    const int top = ntiles - base < 4096 ? ntiles - base : 4096;
    for (int i = threadIdx.x; i < top; i += kThreads)
      if ((unsigned)tm[base + i] >= kthr) s_tiles[atomicAdd(&s_count, 1)] = (unsigned short)i;
    __syncthreads();
    const int nc = s_count;
    constexpr int W = kThreads / 64, U = 4;
    for (int k = wv * U; k < nc; k += W * U) {                        // wave-uniform
      int tl[U];
      uint4 q[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        tl[u] = base + (int)s_tiles[k + u < nc ? k + u : k];
        q[u] = vf[(int64_t)tl[u] * 64 + lane];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (k + u < nc) take(tl[u], q[u]);
    }
    __syncthreads();                                                  // (the list is rebuilt in the next round)
  }
  // the frame's accumulator row is WRITTEN by this workgroup (its only one): no initialisation launch, no global atomics
  auto add = [](unsigned long long a, unsigned long long b) { return a + b; };
  auto mn = [](unsigned a, unsigned b) { return a < b ? a : b; };
  auto mx = [](unsigned a, unsigned b) { return a > b ? a : b; };
  __shared__ unsigned long long s_sum[kThreads / 64][3];
  __shared__ unsigned s_box[kThreads / 64][4];
  cnt = pl_wave_reduce(cnt, add); sr = pl_wave_reduce(sr, add); sc = pl_wave_reduce(sc, add);
  rmin = pl_wave_reduce(rmin, mn); rmax = pl_wave_reduce(rmax, mx);
  cmin = pl_wave_reduce(cmin, mn); cmax = pl_wave_reduce(cmax, mx);
  if (lane == 0) {
    s_sum[wv][0] = cnt; s_sum[wv][1] = sr; s_sum[wv][2] = sc;
    s_box[wv][0] = rmin; s_box[wv][1] = rmax; s_box[wv][2] = cmin; s_box[wv][3] = cmax;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < kThreads / 64; ++k) {
      cnt += s_sum[k][0]; sr += s_sum[k][1]; sc += s_sum[k][2];
      rmin = mn(rmin, s_box[k][0]); rmax = mx(rmax, s_box[k][1]); cmin = mn(cmin, s_box[k][2]); cmax = mx(cmax, s_box[k][3]);
    }
    unsigned long long* a = acc + frame * 8;                          // cax_init_kernel's values where nothing was found
    a[0] = cnt; a[1] = sr; a[2] = sc;
    a[3] = cnt ? (unsigned long long)rmin : ~0ull; a[4] = cnt ? (unsigned long long)rmax : 0ull;
    a[5] = cnt ? (unsigned long long)cmin : ~0ull; a[6] = cnt ? (unsigned long long)cmax : 0ull;
    a[7] = 0;
  }
}

__global__ void cax_init_kernel(unsigned long long* __restrict__ acc, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  unsigned long long* a = acc + i * 8;
  a[0] = 0; a[1] = 0; a[2] = 0; a[3] = ~0ull; a[4] = 0; a[5] = ~0ull; a[6] = 0; a[7] = 0;
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
cax_window_kernel(const T* __restrict__ in, int h, int w, const double* __restrict__ sub, const double* __restrict__ div,
                  const double* __restrict__ thr, const unsigned long long* __restrict__ acc, double* __restrict__ out,
                  int32_t* __restrict__ status) {
  extern __shared__ unsigned char win[];   // 0 background (unreached), 1 foreground, 2 background reached from the border
  __shared__ unsigned long long s_add[3];
  const int64_t frame = blockIdx.x;
  const unsigned long long* a = acc + frame * 8;
  const unsigned long long cnt = a[0];
  if (cnt == 0) {   // empty mask: scipy's center_of_mass divides 0 by 0
    if (threadIdx.x == 0) { out[frame * 3] = 0.0 / 0.0; out[frame * 3 + 1] = 0.0 / 0.0; out[frame * 3 + 2] = 0.0; status[frame] = 0; }
    return;
  }
  const int r0 = (int)a[3] > 0 ? (int)a[3] - 1 : 0, r1 = (int)a[4] + 1 < h ? (int)a[4] + 1 : h - 1;   // inclusive window
  const int c0 = (int)a[5] > 0 ? (int)a[5] - 1 : 0, c1 = (int)a[6] + 1 < w ? (int)a[6] + 1 : w - 1;
  const int wh = r1 - r0 + 1, ww = c1 - c0 + 1;
  if ((int64_t)wh * ww > kCaxMaxWindow) {
    if (threadIdx.x == 0) status[frame] = 1;   // window too large for LDS: the caller takes the general path
    return;
  }
  const int npx = wh * ww;
  const T* f = in + frame * (int64_t)h * w;
  const double s = sub[frame], d = div[frame], t = thr[frame];
  if (threadIdx.x < 3) s_add[threadIdx.x] = 0;
  const bool use_int = sizeof(T) == 2 && d > 0.0;
  int ithr = 0;
  if constexpr (sizeof(T) == 2) { if (use_int) ithr = cax_int_threshold<T>(s, d, t); }
  for (int e = threadIdx.x; e < npx; e += kThreads) {
    const int r = e / ww, c = e % ww;
    const T px = f[(int64_t)(r0 + r) * w + c0 + c];
    const bool fg = use_int ? ((int)px >= ithr) : (((double)px - s) / d >= t);
    const bool edge = r == 0 || c == 0 || r == wh - 1 || c == ww - 1;
    win[e] = fg ? 1 : (edge ? 2 : 0);
  }
  __syncthreads();
  for (;;) {   // 4-connected flood fill of the background from the window border
    int changed = 0;
    for (int e = threadIdx.x; e < npx; e += kThreads) {
      if (win[e]) continue;
      const int r = e / ww, c = e % ww;   // interior pixel: all four neighbours exist
      if (win[e - ww] == 2 || win[e + ww] == 2 || win[e - 1] == 2 || win[e + 1] == 2) { win[e] = 2; changed = 1; }
      (void)r; (void)c;
    }
    if (!__syncthreads_or(changed)) break;
  }
  unsigned long long hc = 0, hr = 0, hcol = 0;
  for (int e = threadIdx.x; e < npx; e += kThreads) {
    if (win[e] == 0) { ++hc; hr += (unsigned long long)(r0 + e / ww); hcol += (unsigned long long)(c0 + e % ww); }
  }
  auto add = [](unsigned long long x, unsigned long long y) { return x + y; };
  hc = pl_wave_reduce(hc, add); hr = pl_wave_reduce(hr, add); hcol = pl_wave_reduce(hcol, add);
  if ((threadIdx.x & 63) == 0 && hc) { atomicAdd(&s_add[0], hc); atomicAdd(&s_add[1], hr); atomicAdd(&s_add[2], hcol); }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double total = (double)(cnt + s_add[0]);
    out[frame * 3 + 0] = (double)(a[1] + s_add[1]) / total;   // exact integer sums / count in float64, like center_of_mass
    out[frame * 3 + 1] = (double)(a[2] + s_add[2]) / total;
    out[frame * 3 + 2] = total;
    status[frame] = 0;
  }
}

}  // namespace

/* ndimage.center_of_mass(ndimage.binary_fill_holes(((a - sub) / div) >= thr)) per frame (pylinac/winston_lutz.py:775-779)
 * without materialising the mask: d_out float64[n][3] = (row, col, filled pixel count); d_acc uint64[n][8] scratch;
 * d_status int32[n]: 0 done, 1 = the foreground's bounding box exceeds the 384 x 384 LDS window (use pl_scaled_binary ->
 * pl_fill_holes -> pl_binary_centroid for that frame). */
static int field_cax_impl(const void* in, int dtype, int64_t n, int h, int w, const double* d_sub, const double* d_div,
                          const double* d_thr, const uint16_t* d_tile_max, unsigned long long* d_acc, double* d_out,
                          int32_t* d_status, void* stream);

extern "C" int pl_field_cax(const void* in, int dtype, int64_t n, int h, int w, const double* d_sub,
                            const double* d_div, const double* d_thr, unsigned long long* d_acc, double* d_out,
                            int32_t* d_status, void* stream) {
  return field_cax_impl(in, dtype, n, h, w, d_sub, d_div, d_thr, nullptr, d_acc, d_out, d_status, stream);
}

/* pl_field_cax with the streaming pass driven by pl_hist16_tiles' tile maxima (see cax_reduce_tiles_kernel).  16-bit frames
 * whose pixel count is a multiple of 512, width a multiple of 8 and base 16-byte aligned; anything else takes the full pass. */
extern "C" int pl_field_cax_tiles(const void* in, int dtype, int64_t n, int h, int w, const double* d_sub, const double* d_div,
                                  const double* d_thr, const uint16_t* d_tile_max, unsigned long long* d_acc, double* d_out,
                                  int32_t* d_status, void* stream) {
  PL_REQUIRE(d_tile_max, "null pointer");
  return field_cax_impl(in, dtype, n, h, w, d_sub, d_div, d_thr, d_tile_max, d_acc, d_out, d_status, stream);
}

static int field_cax_impl(const void* in, int dtype, int64_t n, int h, int w, const double* d_sub, const double* d_div,
                          const double* d_thr, const uint16_t* d_tile_max, unsigned long long* d_acc, double* d_out,
                          int32_t* d_status, void* stream) {
  PL_REQUIRE(in && d_sub && d_div && d_thr && d_acc && d_out && d_status, "null pointer");
  PL_CCL_CHECK_SHAPE();
  hipStream_t st = (hipStream_t)stream;
  const int bpf = (int)pl_cdiv((int64_t)h * w, 65536);
  PL_REQUIRE(n * bpf <= 0x7fffffffLL, "batch too large");
  static std::atomic<bool> attr{false};
  const bool tiles_ok = d_tile_max && (dtype == PL_U16 || dtype == PL_I16) && (((int64_t)h * w) & 511) == 0 && (w & 7) == 0 &&
                        (reinterpret_cast<uintptr_t>(in) & 15) == 0;
  if (!tiles_ok) hipLaunchKernelGGL(cax_init_kernel, dim3((unsigned)pl_cdiv(n, kThreads)), dim3(kThreads), 0, st, d_acc, n);
  PL_DISPATCH_DTYPE(dtype, T, {
    if (!attr) {
      // every instantiation that can be launched gets the opt-in (cheap; done once per process)
      (void)hipFuncSetAttribute((const void*)cax_window_kernel<unsigned short>, hipFuncAttributeMaxDynamicSharedMemorySize, kCaxMaxWindow);
      (void)hipFuncSetAttribute((const void*)cax_window_kernel<short>, hipFuncAttributeMaxDynamicSharedMemorySize, kCaxMaxWindow);
      (void)hipFuncSetAttribute((const void*)cax_window_kernel<unsigned char>, hipFuncAttributeMaxDynamicSharedMemorySize, kCaxMaxWindow);
      (void)hipFuncSetAttribute((const void*)cax_window_kernel<int>, hipFuncAttributeMaxDynamicSharedMemorySize, kCaxMaxWindow);
      (void)hipFuncSetAttribute((const void*)cax_window_kernel<long long>, hipFuncAttributeMaxDynamicSharedMemorySize, kCaxMaxWindow);
      (void)hipFuncSetAttribute((const void*)cax_window_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, kCaxMaxWindow);
      (void)hipFuncSetAttribute((const void*)cax_window_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, kCaxMaxWindow);
      attr = true;
    }
    bool tiled = false;
    if constexpr (sizeof(T) == 2) {
      if (tiles_ok) {
        hipLaunchKernelGGL(cax_reduce_tiles_kernel<T>, dim3((unsigned)n), dim3(kThreads), 0, st, (const T*)in, h, w, d_sub, d_div,
                           d_thr, d_tile_max, d_acc);
        tiled = true;
      }
    }
    if (!tiled)
      hipLaunchKernelGGL(cax_reduce_kernel<T>, dim3((unsigned)(n * bpf)), dim3(kThreads), 0, st, (const T*)in, h, w, bpf,
                         d_sub, d_div, d_thr, d_acc);
    hipLaunchKernelGGL(cax_window_kernel<T>, dim3((unsigned)n), dim3(kThreads), kCaxMaxWindow, st, (const T*)in, h, w,
                       d_sub, d_div, d_thr, d_acc, d_out, d_status);
  });
  return pl_check_launch("pl_field_cax");
}

// ---- WLBaseImage._clean_edges' edge test (pylinac/winston_lutz.py:1109-1133): min / max over the four window_size-wide
// edge strips of every frame.  One workgroup per frame.
namespace {
template <typename T>
__global__ void __launch_bounds__(kThreads)
edge_minmax_kernel(const T* __restrict__ in, int h, int w, int ws, int32_t* __restrict__ emin, int32_t* __restrict__ emax) {
  __shared__ int s_mn[kThreads / 64], s_mx[kThreads / 64];
  const T* f = in + (int64_t)blockIdx.x * h * w;
  int mn = 0x7fffffff, mx = -0x7fffffff - 1;
  auto see = [&](int v) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; };
  const int band = ws < h ? ws : h;
  for (int e = threadIdx.x; e < band * w; e += kThreads) {          // top and bottom strips
    see((int)f[e]);
    see((int)f[(int64_t)(h - band) * w + e]);
  }
  const int cb = ws < w ? ws : w;
  for (int e = threadIdx.x; e < h * cb; e += kThreads) {            // left and right strips
    const int r = e / cb, c = e % cb;
    see((int)f[(int64_t)r * w + c]);
    see((int)f[(int64_t)r * w + (w - cb) + c]);
  }
  mn = pl_wave_reduce(mn, [](int a, int b) { return a < b ? a : b; });
  mx = pl_wave_reduce(mx, [](int a, int b) { return a > b ? a : b; });
  if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < kThreads / 64; ++k) { mn = s_mn[k] < mn ? s_mn[k] : mn; mx = s_mx[k] > mx ? s_mx[k] : mx; }
    emin[blockIdx.x] = mn;
    emax[blockIdx.x] = mx;
  }
}
}  // namespace

/* min / max over the four `window`-pixel-wide edge strips of every 16-bit frame (int32[n] each): the edge test of
 * WLBaseImage._clean_edges (pylinac/winston_lutz.py:1109-1133). */
extern "C" int pl_edge_minmax(const void* in, int dtype, int64_t n, int h, int w, int window, int32_t* d_min,
                              int32_t* d_max, void* stream) {
  PL_REQUIRE(in && d_min && d_max, "null pointer");
  PL_REQUIRE(dtype == PL_U16 || dtype == PL_I16, "16-bit integer frames only");
  PL_REQUIRE(window > 0, "window must be positive");
  PL_CCL_CHECK_SHAPE();
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PL_U16)
    hipLaunchKernelGGL(edge_minmax_kernel<unsigned short>, dim3((unsigned)n), dim3(kThreads), 0, st,
                       (const unsigned short*)in, h, w, window, d_min, d_max);
  else
    hipLaunchKernelGGL(edge_minmax_kernel<short>, dim3((unsigned)n), dim3(kThreads), 0, st, (const short*)in, h, w,
                       window, d_min, d_max);
  return pl_check_launch("pl_edge_minmax");
}


// ---- the per-frame scalar decisions of WLBaseImage.analyze on the device -----------------------------------------------
namespace {
struct WlFrac { double t[7]; };

// numpy's _lerp (np.percentile, method "linear"): a + (b - a) t, and b - (b - a)(1 - t) where t >= 0.5
__device__ __forceinline__ double wl_lerp(double a, double b, double t) {
  const double d = b - a;
  return t >= 0.5 ? b - d * (1.0 - t) : a + d * t;
}

__global__ void wl_decisions_kernel(const int32_t* __restrict__ stats, const int32_t* __restrict__ emin, const int32_t* __restrict__ emax,
                                    int64_t n, WlFrac f, int32_t* __restrict__ inverted, int32_t* __restrict__ noisy,
                                    double* __restrict__ vmin, double* __restrict__ vmax, double* __restrict__ gmax, double* __restrict__ thr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t* s = stats + i * 16;               // min, max, lower neighbours of the 7 percentiles, upper neighbours
  const double mn = (double)s[0], mx = (double)s[1];
  auto pct = [&](int q) { return wl_lerp((double)s[2 + q], (double)s[9 + q], f.t[q]); };
  // image.py:899-926 check_inversion_by_histogram((0.01, 50, 99.99)): invert when the median sits nearer the top
  const double p_lo = pct(2), p_mid = pct(3), p_hi = pct(4);
  inverted[i] = __builtin_fabs(p_mid - p_lo) > __builtin_fabs(p_mid - p_hi) ? 1 : 0;
  // winston_lutz.py:1109-1133 _clean_edges: an edge strip more than 10 % of the (p5 .. p99.5) range outside it
  const double e0 = pct(5), e1 = pct(6), rng = e1 - e0;
  noisy[i] = ((double)emin[i] < e0 - rng / 10.0) || ((double)emax[i] > e1 + rng / 10.0) ? 1 : 0;
  // winston_lutz.py:711-712, 775-776: ground() / normalize(), then (p99.9 - p5) / 2 + p5 of the float64 frame: the order
  // statistics pushed through the same float64 operations
  const double g = mx - mn;
  const double a0 = ((double)s[2] - mn) / g, b0 = ((double)s[9] - mn) / g;
  const double a1 = ((double)s[3] - mn) / g, b1 = ((double)s[10] - mn) / g;
  const double p0 = wl_lerp(a0, b0, f.t[0]), p1 = wl_lerp(a1, b1, f.t[1]);
  vmin[i] = mn;
  vmax[i] = mx;
  gmax[i] = g;
  thr[i] = (p1 - p0) / 2.0 + p0;
}
}  // namespace

extern "C" int pl_wl_decisions(const int32_t* d_stats, const int32_t* d_edge_min, const int32_t* d_edge_max, int64_t n,
                               const double* h_frac, int32_t* d_inverted, int32_t* d_noisy, double* d_vmin, double* d_vmax,
                               double* d_gmax, double* d_thr, void* stream) {
  PL_REQUIRE(d_stats && d_edge_min && d_edge_max && h_frac && d_inverted && d_noisy && d_vmin && d_vmax && d_gmax && d_thr,
             "null pointer");
  PL_REQUIRE(n >= 0, "bad shape");
  if (n == 0) return PL_OK;
  WlFrac f;
  for (int k = 0; k < 7; ++k) f.t[k] = h_frac[k];
  hipLaunchKernelGGL(wl_decisions_kernel, dim3((unsigned)pl_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, d_stats,
                     d_edge_min, d_edge_max, n, f, d_inverted, d_noisy, d_vmin, d_vmax, d_gmax, d_thr);
  return pl_check_launch("pl_wl_decisions");
}
