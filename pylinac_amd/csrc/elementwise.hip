// Per-frame min/max and the elementwise mutators of BaseImage.
//
// Replaces (numpy expressions in the reference):
//   pl_minmax      array.min()/array.max()                  pylinac/core/array_utils.py:66,77,102
//   pl_ground      array - array.min() + value              pylinac/core/array_utils.py:92-102
//   pl_normalize   array / val   (-> float64)               pylinac/core/array_utils.py:63-71
//   pl_invert      -array + array.max() + array.min()       pylinac/core/array_utils.py:74-77
//   pl_scale       array * scalar  (stretch)                pylinac/core/array_utils.py:168
//   pl_cast_wrap   np.array(float64, dtype=T) incl. the wrap of negative values (convert_to_dtype :198)
//   pl_threshold   np.where(a >= t, a, 0) / (a <= t)        pylinac/core/image.py:785-800
//   pl_as_binary   np.where(a >= t, 1, 0)                   pylinac/core/image.py:802-815
//
// All are HBM-bound streaming kernels: 16-byte loads/stores per lane, one contiguous chunk of one
// frame per block (so the per-frame scalar is block-uniform and read through the scalar cache).
#include "pl_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kChunkBytes = 64 * 1024;  // bytes of input one block streams (read + write kernels reach 5.3-5.6 TB/s with it on
                                        // 256 x 1024 x 1024 uint16; 256 KiB or 1 MiB blocks measured the same or slower; the pure
                                        // reduction pl_minmax is the exception, see kMinmaxChunkBytes)

template <typename T> struct Vec16 { static constexpr int N = 16 / sizeof(T); };
template <typename T> struct is_floating { static constexpr bool value = false; };
template <> struct is_floating<float> { static constexpr bool value = true; };
template <> struct is_floating<double> { static constexpr bool value = true; };

struct Plan { int64_t chunk; int bpf; };  // elements per block, blocks per frame

template <typename T>
static Plan make_plan(int64_t count) {
  Plan p;
  p.chunk = kChunkBytes / (int64_t)sizeof(T);
  p.bpf = (int)pl_cdiv(count, p.chunk);
  return p;
}

// Apply f(elem, frame) over one block's chunk with 16-byte vector access when aligned.
template <typename TI, typename TO, typename F>
__device__ __forceinline__ void stream_chunk(const TI* __restrict__ in, TO* __restrict__ out,
                                             int64_t count, int64_t chunk, int bpf, F f) {
  const int64_t frame = blockIdx.x / bpf;
  const int64_t off = (int64_t)(blockIdx.x % bpf) * chunk;
  const int64_t len = (count - off) < chunk ? (count - off) : chunk;
  const TI* src = in + frame * count + off;
  TO* dst = out + frame * count + off;
  constexpr int N = Vec16<TI>::N;
  constexpr int OB = N * (int)sizeof(TO);          // output bytes per lane-iteration
  constexpr int OA = OB >= 16 ? 16 : OB;           // widest naturally aligned store we can use
  struct alignas(OA) OutPack { TO e[N]; };
  const bool aligned = ((reinterpret_cast<uintptr_t>(src) & 15) == 0) &&
                       ((reinterpret_cast<uintptr_t>(dst) & (OA - 1)) == 0);
  if (aligned) {
    const int64_t nvec = len / N;
    auto one = [&](int64_t v, const uint4& q) {
      union { uint4 q; TI e[N]; } u;
      u.q = q;
      OutPack r;
#pragma unroll
      for (int k = 0; k < N; ++k) r.e[k] = f(u.e[k], frame);
      reinterpret_cast<OutPack*>(dst)[v] = r;
    };
    constexpr int U = 4;                            // loads in flight per lane
    int64_t v = threadIdx.x;
    for (; v + (U - 1) * kThreads < nvec; v += U * kThreads) {
      uint4 q[U];
#pragma unroll
      for (int u = 0; u < U; ++u) q[u] = reinterpret_cast<const uint4*>(src)[v + u * kThreads];
#pragma unroll
      for (int u = 0; u < U; ++u) one(v + u * kThreads, q[u]);
    }
    for (; v < nvec; v += kThreads) one(v, reinterpret_cast<const uint4*>(src)[v]);
    for (int64_t i = nvec * N + threadIdx.x; i < len; i += kThreads) dst[i] = f(src[i], frame);
  } else {
    for (int64_t i = threadIdx.x; i < len; i += kThreads) dst[i] = f(src[i], frame);
  }
}

// ------------------------------------------------------------------------------------ min / max
__device__ __forceinline__ void atomic_min_f64(double* addr, double v) {
  unsigned long long* a = reinterpret_cast<unsigned long long*>(addr);
  unsigned long long old = *a;
  while (v < __longlong_as_double((long long)old)) {
    unsigned long long assumed = old;
    old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(v));
    if (old == assumed) break;
  }
}
__device__ __forceinline__ void atomic_max_f64(double* addr, double v) {
  unsigned long long* a = reinterpret_cast<unsigned long long*>(addr);
  unsigned long long old = *a;
  while (v > __longlong_as_double((long long)old)) {
    unsigned long long assumed = old;
    old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(v));
    if (old == assumed) break;
  }
}

__global__ void minmax_init(double* mn, double* mx, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i < n) { mn[i] = __longlong_as_double(0x7ff0000000000000LL); mx[i] = __longlong_as_double((long long)0xfff0000000000000ULL); }
}

#ifndef PL_MINMAX_LOADS
#define PL_MINMAX_LOADS 8
#endif
#ifndef PL_MINMAX_CHUNK_KB
#define PL_MINMAX_CHUNK_KB 256
#endif
constexpr int kMinmaxLoads = PL_MINMAX_LOADS;                // 16-byte loads a lane keeps in flight
constexpr int64_t kMinmaxChunkBytes = PL_MINMAX_CHUNK_KB * 1024;   // per workgroup (the elementwise kernels' 64 KiB blocks live too
                                                                   // short for a pure reduction: 3.7 TB/s on 400 MB of frames)
template <typename T>
__global__ void __launch_bounds__(kThreads)
minmax_kernel(const T* __restrict__ in, int64_t count, int64_t chunk, int bpf, double* mn, double* mx) {
  const int64_t frame = blockIdx.x / bpf;
  const int64_t off = (int64_t)(blockIdx.x % bpf) * chunk;
  const int64_t len = (count - off) < chunk ? (count - off) : chunk;
  const T* src = in + frame * count + off;
  constexpr int N = Vec16<T>::N;
  T lo = src[0], hi = src[0];
  if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    const int64_t nvec = len / N;
    const uint4* vsrc = reinterpret_cast<const uint4*>(src);
    // kMinmaxLoads 16-byte loads per lane in flight (a streaming kernel's rate is its bytes in flight over the loaded latency:
    // one load per lane and iteration reached 3.4 TB/s on 256 x 768 x 1024 uint16); 16-bit integers through the packed min / max
    if constexpr (sizeof(T) == 2 && !is_floating<T>::value) {
      typedef T T2 __attribute__((ext_vector_type(2)));
      T2 lo2 = {lo, lo}, hi2 = {hi, hi};
      auto see = [&](const uint4& q) {
        const unsigned wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const T2 x = __builtin_bit_cast(T2, wds[k]);
          lo2 = __builtin_elementwise_min(lo2, x);
          hi2 = __builtin_elementwise_max(hi2, x);
        }
      };
      int64_t v = threadIdx.x;
      for (; v + (kMinmaxLoads - 1) * kThreads < nvec; v += kMinmaxLoads * kThreads) {
        uint4 q[kMinmaxLoads];
#pragma unroll
        for (int u = 0; u < kMinmaxLoads; ++u) q[u] = vsrc[v + u * kThreads];
#pragma unroll
        for (int u = 0; u < kMinmaxLoads; ++u) see(q[u]);
      }
      for (; v < nvec; v += kThreads) see(vsrc[v]);
      lo = lo2.x < lo2.y ? lo2.x : lo2.y;
      hi = hi2.x > hi2.y ? hi2.x : hi2.y;
    } else {
      auto see = [&](const uint4& q) {
        union { uint4 q; T e[N]; } u;
        u.q = q;
#pragma unroll
        for (int k = 0; k < N; ++k) { lo = u.e[k] < lo ? u.e[k] : lo; hi = u.e[k] > hi ? u.e[k] : hi; }
      };
      int64_t v = threadIdx.x;
      for (; v + (kMinmaxLoads - 1) * kThreads < nvec; v += kMinmaxLoads * kThreads) {
        uint4 q[kMinmaxLoads];
#pragma unroll
        for (int u = 0; u < kMinmaxLoads; ++u) q[u] = vsrc[v + u * kThreads];
#pragma unroll
        for (int u = 0; u < kMinmaxLoads; ++u) see(q[u]);
      }
      for (; v < nvec; v += kThreads) see(vsrc[v]);
    }
    for (int64_t i = nvec * N + threadIdx.x; i < len; i += kThreads) { T e = src[i]; lo = e < lo ? e : lo; hi = e > hi ? e : hi; }
  } else {
    for (int64_t i = threadIdx.x; i < len; i += kThreads) { T e = src[i]; lo = e < lo ? e : lo; hi = e > hi ? e : hi; }
  }
  double dlo = (double)lo, dhi = (double)hi;
  dlo = pl_wave_reduce(dlo, [](double a, double b) { return a < b ? a : b; });
  dhi = pl_wave_reduce(dhi, [](double a, double b) { return a > b ? a : b; });
  __shared__ double slo[kThreads / PL_WAVE], shi[kThreads / PL_WAVE];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { slo[wv] = dlo; shi[wv] = dhi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < kThreads / PL_WAVE; ++k) { dlo = slo[k] < dlo ? slo[k] : dlo; dhi = shi[k] > dhi ? shi[k] : dhi; }
    atomic_min_f64(mn + frame, dlo);
    atomic_max_f64(mx + frame, dhi);
  }
}

// min/max of the pixels selected by a (frame-shared) uint8 mask: edges[rr, cc].min()/max() of
// get_regions' disk selection (pylinac/ct.py:3334-3338)
__global__ void __launch_bounds__(kThreads)
minmax_masked_kernel(const double* __restrict__ in, const unsigned char* __restrict__ mask, int64_t count,
                     int bpf, double* mn, double* mx) {
  const int64_t frame = blockIdx.x / bpf;
  const int64_t chunk = 65536;
  const int64_t lo_i = (int64_t)(blockIdx.x % bpf) * chunk;
  const int64_t hi_i = (lo_i + chunk < count) ? lo_i + chunk : count;
  const double* src = in + frame * count;
  double lo = __longlong_as_double(0x7ff0000000000000LL), hi = __longlong_as_double((long long)0xfff0000000000000ULL);
  for (int64_t i = lo_i + threadIdx.x; i < hi_i; i += kThreads)
    if (mask[i]) { const double v = src[i]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
  lo = pl_wave_reduce(lo, [](double a, double b) { return a < b ? a : b; });
  hi = pl_wave_reduce(hi, [](double a, double b) { return a > b ? a : b; });
  if ((threadIdx.x & 63) == 0) { atomic_min_f64(mn + frame, lo); atomic_max_f64(mx + frame, hi); }
}

// ----------------------------------------------------------------------------- elementwise ops
template <typename T>
__global__ void __launch_bounds__(kThreads)
ground_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t count, int64_t chunk, int bpf,
              const double* __restrict__ mn, double value) {
  stream_chunk<T, T>(in, out, count, chunk, bpf, [=](T a, int64_t fr) -> T {
    if constexpr (!is_floating<T>::value) {
      // numpy: integer array - same-dtype scalar (+ python int): modular arithmetic in the dtype
      long long v = (long long)a - (long long)mn[fr] + (long long)value;
      return (T)v;
    } else {
      return (T)(a - (T)mn[fr] + (T)value);
    }
  });
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
normalize_kernel(const T* __restrict__ in, double* __restrict__ out, int64_t count, int64_t chunk,
                 int bpf, const double* __restrict__ val) {
  stream_chunk<T, double>(in, out, count, chunk, bpf,
                          [=](T a, int64_t fr) -> double { return (double)a / val[fr]; });
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
invert_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t count, int64_t chunk, int bpf,
              const double* __restrict__ mn, const double* __restrict__ mx) {
  stream_chunk<T, T>(in, out, count, chunk, bpf, [=](T a, int64_t fr) -> T {
    if constexpr (!is_floating<T>::value) {
      long long v = -(long long)a + (long long)mx[fr] + (long long)mn[fr];  // modular, like numpy
      return (T)v;
    } else {
      return (T)((-a + (T)mx[fr]) + (T)mn[fr]);
    }
  });
}

// np.invert on an integer array: the bitwise complement in the array's own type (array_utils.py:80-89).  No float64
// scalars are involved, so 64-bit values keep every bit (round 1 routed this through -a + max + min with the type's
// extrema as doubles: 2^63 is not an int64).
template <typename T>
__global__ void __launch_bounds__(kThreads)
bit_invert_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i < total) out[i] = (T)~in[i];
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
scale_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t count, int64_t chunk, int bpf,
             double factor) {
  stream_chunk<T, T>(in, out, count, chunk, bpf, [=](T a, int64_t) -> T {
    if constexpr (is_floating<T>::value) return (T)(a * (T)factor);
    else return (T)((long long)a * (long long)factor);
  });
}

// frame - frame.min() as uint16, where that is EXACT: every value an integer, no difference beyond `max_range`.  Lets the
// uint16 analyzers take what the reference's loader may hand over instead -- int16 frames of a signed panel, float64 frames
// that hold integers (`dtype=float`, or rescale tags with an integer slope and intercept) -- with identical results:
// ground() / normalize() only ever see a - min (picketfence.py:322-323, winston_lutz.py:711-712).  flag[frame] = 1 marks a
// frame that does not qualify (its output is unspecified).
template <typename T>
__global__ void __launch_bounds__(kThreads)
to_u16_exact_kernel(const T* __restrict__ in, unsigned short* __restrict__ out, int64_t count, int64_t chunk, int bpf,
                    const double* __restrict__ mn, double max_range, int32_t* __restrict__ flag) {
  bool bad = false;
  stream_chunk<T, unsigned short>(in, out, count, chunk, bpf, [&](T a, int64_t fr) -> unsigned short {
    const double d = (double)a - mn[fr];                   // exact for integers below 2^53
    bad |= !(d >= 0.0 && d <= max_range) || d != floor(d);
    return (unsigned short)(unsigned)(d >= 0.0 && d <= 65535.0 ? d : 0.0);
  });
  if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(&flag[blockIdx.x / bpf], 1);
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
threshold_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t count, int64_t chunk, int bpf,
                 const double* __restrict__ thr, int thr_stride, int kind) {
  stream_chunk<T, T>(in, out, count, chunk, bpf, [=](T a, int64_t fr) -> T {
    const double t = thr[fr * thr_stride];
    const bool keep = kind == 0 ? ((double)a >= t) : ((double)a <= t);
    return keep ? a : (T)0;
  });
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
binary_kernel(const T* __restrict__ in, unsigned char* __restrict__ out, int64_t count, int64_t chunk,
              int bpf, const double* __restrict__ thr, int thr_stride) {
  stream_chunk<T, unsigned char>(in, out, count, chunk, bpf, [=](T a, int64_t fr) -> unsigned char {
    return ((double)a >= thr[fr * thr_stride]) ? 1 : 0;
  });
}

static int check_grid(int64_t n, int bpf, const char* who) {
  if (n * bpf > 0x7fffffffLL) { pl_set_error("%s: batch too large for one launch", who); return PL_ERR_INVALID_ARG; }
  return PL_OK;
}

}  // namespace

#define PL_EW_PROLOGUE(who)                                        \
  PL_REQUIRE(in && out, "null pointer");                           \
  PL_REQUIRE(n >= 0 && count > 0, "bad shape");                    \
  if (n == 0) return PL_OK;                                        \
  hipStream_t st = (hipStream_t)stream;

extern "C" int pl_minmax(const void* in, int dtype, int64_t n, int64_t count, double* d_min,
                         double* d_max, void* stream) {
  PL_REQUIRE(in && d_min && d_max, "null pointer");
  PL_REQUIRE(n >= 0 && count > 0, "bad shape");
  if (n == 0) return PL_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(minmax_init, dim3((unsigned)pl_cdiv(n, kThreads)), dim3(kThreads), 0, st, d_min, d_max, n);
  PL_DISPATCH_DTYPE(dtype, T, {
    Plan p;
    p.chunk = kMinmaxChunkBytes / (int64_t)sizeof(T);
    p.bpf = (int)pl_cdiv(count, p.chunk);
    if (n * p.bpf < 512) {                                  // few frames: more, smaller blocks
      p.chunk = kChunkBytes / (int64_t)sizeof(T);
      p.bpf = (int)pl_cdiv(count, p.chunk);
    }
    if (int rc = check_grid(n, p.bpf, "pl_minmax")) return rc;
    hipLaunchKernelGGL(minmax_kernel<T>, dim3((unsigned)(n * p.bpf)), dim3(kThreads), 0, st, (const T*)in,
                       count, p.chunk, p.bpf, d_min, d_max);
  });
  return pl_check_launch("pl_minmax");
}

extern "C" int pl_minmax_masked(const double* in, const uint8_t* d_mask, int64_t n, int64_t count, double* d_min,
                                double* d_max, void* stream) {
  PL_REQUIRE(in && d_mask && d_min && d_max, "null pointer");
  PL_REQUIRE(n >= 0 && count > 0, "bad shape");
  if (n == 0) return PL_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(minmax_init, dim3((unsigned)pl_cdiv(n, kThreads)), dim3(kThreads), 0, st, d_min, d_max, n);
  const int bpf = (int)pl_cdiv(count, 65536);
  if (int rc = check_grid(n, bpf, "pl_minmax_masked")) return rc;
  hipLaunchKernelGGL(minmax_masked_kernel, dim3((unsigned)(n * bpf)), dim3(kThreads), 0, st, in, d_mask, count, bpf,
                     d_min, d_max);
  return pl_check_launch("pl_minmax_masked");
}

extern "C" int pl_ground(const void* in, void* out, int dtype, int64_t n, int64_t count,
                         const double* d_min, double value, void* stream) {
  PL_EW_PROLOGUE("pl_ground");
  PL_REQUIRE(d_min, "null d_min");
  PL_DISPATCH_DTYPE(dtype, T, {
    Plan p = make_plan<T>(count);
    if (int rc = check_grid(n, p.bpf, "pl_ground")) return rc;
    hipLaunchKernelGGL(ground_kernel<T>, dim3((unsigned)(n * p.bpf)), dim3(kThreads), 0, st, (const T*)in,
                       (T*)out, count, p.chunk, p.bpf, d_min, value);
  });
  return pl_check_launch("pl_ground");
}

extern "C" int pl_normalize(const void* in, double* out, int dtype, int64_t n, int64_t count,
                            const double* d_val, void* stream) {
  PL_EW_PROLOGUE("pl_normalize");
  PL_REQUIRE(d_val, "null d_val");
  PL_DISPATCH_DTYPE(dtype, T, {
    Plan p = make_plan<T>(count);
    if (int rc = check_grid(n, p.bpf, "pl_normalize")) return rc;
    hipLaunchKernelGGL(normalize_kernel<T>, dim3((unsigned)(n * p.bpf)), dim3(kThreads), 0, st,
                       (const T*)in, out, count, p.chunk, p.bpf, d_val);
  });
  return pl_check_launch("pl_normalize");
}

extern "C" int pl_invert(const void* in, void* out, int dtype, int64_t n, int64_t count,
                         const double* d_min, const double* d_max, void* stream) {
  PL_EW_PROLOGUE("pl_invert");
  PL_REQUIRE(d_min && d_max, "null min/max");
  PL_DISPATCH_DTYPE(dtype, T, {
    Plan p = make_plan<T>(count);
    if (int rc = check_grid(n, p.bpf, "pl_invert")) return rc;
    hipLaunchKernelGGL(invert_kernel<T>, dim3((unsigned)(n * p.bpf)), dim3(kThreads), 0, st, (const T*)in,
                       (T*)out, count, p.chunk, p.bpf, d_min, d_max);
  });
  return pl_check_launch("pl_invert");
}

// np.array(float64_values, dtype=T) as numpy does it on x86-64 (C conversion through a 64-bit integer, then narrowing:
// negative values wrap into unsigned types).  convert_to_dtype (array_utils.py:171-198) relies on exactly that wrap:
// `relative * range - max - 1` is negative for unsigned targets and lands on the right value modulo 2^bits.
template <typename T>
__global__ void cast_wrap_kernel(const double* __restrict__ in, T* __restrict__ out, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  if (is_floating<T>::value) out[i] = (T)in[i];
  else out[i] = (T)(long long)in[i];
}

extern "C" int pl_cast_wrap(const double* in, void* out, int dtype, int64_t count, void* stream) {
  PL_REQUIRE(in && out, "null pointer");
  PL_REQUIRE(count >= 0, "bad count");
  if (count == 0) return PL_OK;
  PL_REQUIRE(pl_cdiv(count, kThreads) <= 0x7fffffffLL, "batch too large for one launch");
  PL_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL(cast_wrap_kernel<T>, dim3((unsigned)pl_cdiv(count, kThreads)), dim3(kThreads), 0,
                                       (hipStream_t)stream, in, (T*)out, count));
  return pl_check_launch("pl_cast_wrap");
}

// Per-unit result records: out[i][j] = (double)column_j[i * stride_j + offset_j] + add_j for up to 16 columns of float64 or
// int32 device arrays -- the ONE table an analyzer's batch sends to the host (the Winston-Lutz record: field centre, BB
// centroid + window offset, counts, status words, decision flags), assembled in one launch instead of a dozen framework
// kernels of a few microseconds each (r04z: 0.2 of the 1.07 ms of a 512-frame Winston-Lutz pass).
struct PackCols {
  const void* ptr[16];
  int64_t stride[16], offset[16];
  double add[16];
  int is_i32[16];
};

__global__ void pack_columns_kernel(PackCols c, int k, int64_t n, double* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (t >= n * k) return;
  const int64_t i = t / k;
  const int j = (int)(t - i * k);
  const int64_t at = i * c.stride[j] + c.offset[j];
  const double v = c.is_i32[j] ? (double)static_cast<const int32_t*>(c.ptr[j])[at] : static_cast<const double*>(c.ptr[j])[at];
  out[t] = v + c.add[j];
}

extern "C" int pl_pack_columns(const void* const* d_cols, const int* is_int32, const int64_t* strides, const int64_t* offsets,
                               const double* adds, int k, int64_t n, double* d_out, void* stream) {
  PL_REQUIRE(d_cols && is_int32 && strides && offsets && adds && d_out, "null pointer");
  PL_REQUIRE(k >= 1 && k <= 16 && n >= 0, "1..16 columns");
  if (n == 0) return PL_OK;
  PackCols c{};
  for (int j = 0; j < k; ++j) {
    PL_REQUIRE(d_cols[j] != nullptr, "null column");
    c.ptr[j] = d_cols[j]; c.stride[j] = strides[j]; c.offset[j] = offsets[j]; c.add[j] = adds[j]; c.is_i32[j] = is_int32[j];
  }
  PL_REQUIRE(pl_cdiv(n * k, kThreads) <= 0x7fffffffLL, "batch too large");
  hipLaunchKernelGGL(pack_columns_kernel, dim3((unsigned)pl_cdiv(n * k, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, c, k, n,
                     d_out);
  return pl_check_launch("pl_pack_columns");
}

extern "C" int pl_to_u16_exact(const void* in, int dtype, int64_t n, int64_t count, const double* d_min, double max_range,
                               uint16_t* out, int32_t* d_flag, void* stream) {
  PL_EW_PROLOGUE("pl_to_u16_exact");
  PL_REQUIRE(d_min && d_flag, "null pointer");
  PL_REQUIRE(dtype == PL_I16 || dtype == PL_F64 || dtype == PL_I32, "int16, int32 or float64 frames");
  PL_REQUIRE(max_range >= 0.0 && max_range <= 65535.0, "max_range in [0, 65535]");
  hipError_t e = hipMemsetAsync(d_flag, 0, (size_t)n * sizeof(int32_t), st);
  if (e != hipSuccess) { pl_set_error("pl_to_u16_exact: memset: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
  PL_DISPATCH_DTYPE(dtype, T, {
    Plan p = make_plan<T>(count);
    if (int rc = check_grid(n, p.bpf, "pl_to_u16_exact")) return rc;
    hipLaunchKernelGGL(to_u16_exact_kernel<T>, dim3((unsigned)(n * p.bpf)), dim3(kThreads), 0, st, (const T*)in, out, count,
                       p.chunk, p.bpf, d_min, max_range, d_flag);
  });
  return pl_check_launch("pl_to_u16_exact");
}

extern "C" int pl_scale(const void* in, void* out, int dtype, int64_t n, int64_t count, double factor,
                        void* stream) {
  PL_EW_PROLOGUE("pl_scale");
  PL_DISPATCH_DTYPE(dtype, T, {
    Plan p = make_plan<T>(count);
    if (int rc = check_grid(n, p.bpf, "pl_scale")) return rc;
    hipLaunchKernelGGL(scale_kernel<T>, dim3((unsigned)(n * p.bpf)), dim3(kThreads), 0, st, (const T*)in,
                       (T*)out, count, p.chunk, p.bpf, factor);
  });
  return pl_check_launch("pl_scale");
}

extern "C" int pl_threshold(const void* in, void* out, int dtype, int64_t n, int64_t count,
                            const double* d_thr, int thr_stride, int kind, void* stream) {
  PL_EW_PROLOGUE("pl_threshold");
  PL_REQUIRE(d_thr && (thr_stride == 0 || thr_stride == 1), "bad threshold array");
  PL_REQUIRE(kind == 0 || kind == 1, "kind must be 0 (high) or 1 (low)");
  PL_DISPATCH_DTYPE(dtype, T, {
    Plan p = make_plan<T>(count);
    if (int rc = check_grid(n, p.bpf, "pl_threshold")) return rc;
    hipLaunchKernelGGL(threshold_kernel<T>, dim3((unsigned)(n * p.bpf)), dim3(kThreads), 0, st,
                       (const T*)in, (T*)out, count, p.chunk, p.bpf, d_thr, thr_stride, kind);
  });
  return pl_check_launch("pl_threshold");
}

extern "C" int pl_as_binary(const void* in, uint8_t* out, int dtype, int64_t n, int64_t count,
                            const double* d_thr, int thr_stride, void* stream) {
  PL_EW_PROLOGUE("pl_as_binary");
  PL_REQUIRE(d_thr && (thr_stride == 0 || thr_stride == 1), "bad threshold array");
  PL_DISPATCH_DTYPE(dtype, T, {
    Plan p = make_plan<T>(count);
    if (int rc = check_grid(n, p.bpf, "pl_as_binary")) return rc;
    hipLaunchKernelGGL(binary_kernel<T>, dim3((unsigned)(n * p.bpf)), dim3(kThreads), 0, st, (const T*)in,
                       out, count, p.chunk, p.bpf, d_thr, thr_stride);
  });
  return pl_check_launch("pl_as_binary");
}

/* skimage.transform.rotate(image, angle, mode="edge") as BaseImage.rotate calls it (pylinac/core/image.py:780-783):
 * order-1 (order-0 for bool images, skimage's default there) warp of every output pixel (tfr, tfc) through the 2x3 inverse map the host built (c = M00*tfc + M01*tfr + M02,
 * r = M10*tfc + M11*tfr + M12, products and sums rounded one by one), bilinear over floor/ceil neighbours with indices
 * clamped to the frame ('edge'), then clipped to the frame's own [min, max] (clip=True).  Arithmetic in the frame's type. */
template <typename T, int ORDER>
__global__ void warp_affine_kernel(const T* __restrict__ in, T* __restrict__ out, int h, int w, T m00, T m01, T m02,
                                       T m10, T m11, T m12, const double* __restrict__ d_min,
                                       const double* __restrict__ d_max) {
  const int64_t frame = blockIdx.z;
  const int tfc = blockIdx.x * blockDim.x + threadIdx.x;
  const int tfr = blockIdx.y * blockDim.y + threadIdx.y;
  if (tfc >= w || tfr >= h) return;
  const T* img = in + frame * (int64_t)h * w;
  const T x = (T)tfc, y = (T)tfr;
  const T c = m00 * x + m01 * y + m02;
  const T r = m10 * x + m11 * y + m12;
  auto clampi = [](T v, int hi) { return v < (T)0 ? 0 : (v > (T)hi ? hi : (int)v); };
  if (ORDER == 0) {   // nearest neighbour: C round() (halves away from zero), no clipping
    out[frame * (int64_t)h * w + (int64_t)tfr * w + tfc] = img[(int64_t)clampi(round(r), h - 1) * w + clampi(round(c), w - 1)];
    return;
  }
  const T fr = floor(r), fc = floor(c);
  const T dr = r - fr, dc = c - fc;
  const int r0 = clampi(fr, h - 1), r1 = clampi(ceil(r), h - 1);
  const int c0 = clampi(fc, w - 1), c1 = clampi(ceil(c), w - 1);
  const T tl = img[(int64_t)r0 * w + c0], tr = img[(int64_t)r0 * w + c1];
  const T bl = img[(int64_t)r1 * w + c0], br = img[(int64_t)r1 * w + c1];
  // The compiled interpolation keeps `top` / `bottom` in double and writes `1 - dc` with a double constant, so for float32
  // frames only the products dc * top_right and dc * bottom_right are rounded to float32 (measured on scikit-image 0.18.3's
  // _warp_fast: 3000 / 3000 random 2x2 probes); for float64 frames every step is double.
  const double top = (1.0 - (double)dc) * (double)tl + (double)(T)(dc * tr);
  const double bottom = (1.0 - (double)dc) * (double)bl + (double)(T)(dc * br);
  T v = (T)((1.0 - (double)dr) * top + (double)dr * bottom);
  const T lo = (T)d_min[frame], hi = (T)d_max[frame];
  v = v < lo ? lo : (v > hi ? hi : v);
  out[frame * (int64_t)h * w + (int64_t)tfr * w + tfc] = v;
}

extern "C" int pl_warp_affine(const void* in, void* out, int dtype, int64_t n, int64_t h, int64_t w, int order,
                              const double* h_matrix, const double* d_min, const double* d_max, void* stream) {
  PL_REQUIRE(in && out && h_matrix && d_min && d_max, "null pointer");
  PL_REQUIRE(order == 0 || order == 1, "order must be 0 (nearest) or 1 (bilinear)");
  PL_REQUIRE(in != out, "in-place rotation is not supported");
  PL_REQUIRE(n >= 0 && n <= 65535 && h > 0 && w > 0 && h <= 0x3fffffff && w <= 0x3fffffff, "bad shape");
  if (n == 0) return PL_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 block(64, 4), grid((unsigned)pl_cdiv(w, 64), (unsigned)pl_cdiv(h, 4), (unsigned)n);
  const double* m = h_matrix;
  switch (dtype) {
    case PL_F64:
      if (order == 0)
        hipLaunchKernelGGL((warp_affine_kernel<double, 0>), grid, block, 0, st, (const double*)in, (double*)out, (int)h, (int)w,
                           m[0], m[1], m[2], m[3], m[4], m[5], d_min, d_max);
      else
        hipLaunchKernelGGL((warp_affine_kernel<double, 1>), grid, block, 0, st, (const double*)in, (double*)out, (int)h, (int)w,
                           m[0], m[1], m[2], m[3], m[4], m[5], d_min, d_max);
      break;
    case PL_F32:   /* warp() casts the matrix to the image's float type (skimage/transform/_warps.py, matrix.astype) */
      if (order == 0)
        hipLaunchKernelGGL((warp_affine_kernel<float, 0>), grid, block, 0, st, (const float*)in, (float*)out, (int)h, (int)w,
                           (float)m[0], (float)m[1], (float)m[2], (float)m[3], (float)m[4], (float)m[5], d_min, d_max);
      else
        hipLaunchKernelGGL((warp_affine_kernel<float, 1>), grid, block, 0, st, (const float*)in, (float*)out, (int)h, (int)w,
                           (float)m[0], (float)m[1], (float)m[2], (float)m[3], (float)m[4], (float)m[5], d_min, d_max);
      break;
    default: pl_set_error("pl_warp_affine: float32 / float64 frames only (integers are converted by the host)");
             return PL_ERR_UNSUPPORTED;
  }
  return pl_check_launch("pl_warp_affine");
}

/* np.invert(array) for integer frames (pylinac/core/array_utils.py:80-89): out = ~in in the array's own type. */
extern "C" int pl_bit_invert(const void* in, void* out, int dtype, int64_t total, void* stream) {
  PL_REQUIRE(in && out, "null pointer");
  PL_REQUIRE(total >= 0 && pl_cdiv(total, kThreads) <= 0x7fffffffLL, "bad size");
  if (total == 0) return PL_OK;
  const unsigned blocks = (unsigned)pl_cdiv(total, kThreads);
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case PL_U8: hipLaunchKernelGGL(bit_invert_kernel<unsigned char>, dim3(blocks), dim3(kThreads), 0, st, (const unsigned char*)in, (unsigned char*)out, total); break;
    case PL_U16: hipLaunchKernelGGL(bit_invert_kernel<unsigned short>, dim3(blocks), dim3(kThreads), 0, st, (const unsigned short*)in, (unsigned short*)out, total); break;
    case PL_I16: hipLaunchKernelGGL(bit_invert_kernel<short>, dim3(blocks), dim3(kThreads), 0, st, (const short*)in, (short*)out, total); break;
    case PL_I32: hipLaunchKernelGGL(bit_invert_kernel<int>, dim3(blocks), dim3(kThreads), 0, st, (const int*)in, (int*)out, total); break;
    case PL_I64: hipLaunchKernelGGL(bit_invert_kernel<long long>, dim3(blocks), dim3(kThreads), 0, st, (const long long*)in, (long long*)out, total); break;
    default: pl_set_error("pl_bit_invert: integer dtypes only"); return PL_ERR_UNSUPPORTED;
  }
  return pl_check_launch("pl_bit_invert");
}
