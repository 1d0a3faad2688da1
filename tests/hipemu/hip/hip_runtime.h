// TEST INFRASTRUCTURE ONLY -- a tiny wave64 execution model for running the HIP kernels of pylinac_amd/csrc on
// the CPU, so that kernel LOGIC can be checked against the oracle in the `-m "not gpu"` suite (this container has
// no GPU and a round has 90 GPU-minutes).  Nothing under pylinac_amd/ includes, loads or links this: the product
// has no CPU path (tests/test_cabi.py checks that).  tests/hipemu/build.py compiles selected csrc/*.hip files with
// g++ against THIS header (it shadows <hip/hip_runtime.h>) into tests/hipemu/_build/libpylinac_emu.so.
//
// Execution model: one workgroup at a time, every work-item a ucontext fiber on ONE OS thread.
//   __syncthreads()              fiber parks until every live fiber of the workgroup is parked at a block barrier
//   __shfl* / __ballot / ...     the 64 lanes of a wave exchange through a slot table: park, the scheduler snapshots
//                                the slots of the lanes that arrived (= the active mask) and releases them together
//   __shared__                   `static` (one workgroup runs at a time); dynamic LDS: build.py rewrites the
//                                `extern __shared__ T name[];` declaration into a pointer to hipemu::dyn_lds()
//   atomics                      plain read-modify-write (single OS thread)
// What it does NOT model: timing, memory coalescing, cross-workgroup spinning, code that relies on implicit wave
// lock-step without a barrier or wave intrinsic.  Floating point is IEEE double / float on both sides (g++ builds
// with -ffp-contract=off like hipcc does for this library); libm functions may differ from ROCm's in the last ulp.
#pragma once
#include <cstdlib>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <type_traits>

#define PL_HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
#define HIPEMU_VEC(T, name)                                   \
  struct name##2 { T x, y; };                                  \
  struct name##3 { T x, y, z; };                               \
  struct alignas(4 * sizeof(T) > 16 ? 16 : 4 * sizeof(T)) name##4 { T x, y, z, w; };
HIPEMU_VEC(unsigned, uint)
HIPEMU_VEC(int, int)
HIPEMU_VEC(float, float)
HIPEMU_VEC(double, double)
HIPEMU_VEC(unsigned short, ushort)
HIPEMU_VEC(short, short)
HIPEMU_VEC(unsigned char, uchar)
HIPEMU_VEC(unsigned long long, ulonglong)
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
inline double2 make_double2(double x, double y) { return double2{x, y}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }
typedef struct ihipStream_t* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorLaunchFailure = 719 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

namespace hipemu {
struct Idx { unsigned x, y, z; };
extern Idx g_threadIdx, g_blockIdx;
extern dim3 g_blockDim, g_gridDim;
void launch(dim3 grid, dim3 block, size_t dyn_lds_bytes, const std::function<void()>& body);
void* dyn_lds();
void block_barrier();
int block_barrier_or(int pred);
// park this lane with `mine`; returns the values of the whole wave and the mask of the lanes that took part
void wave_exchange(uint64_t mine, uint64_t out[64], uint64_t* active);
hipError_t take_error();
inline int lane_id() { return (int)((g_threadIdx.x + g_blockDim.x * (g_threadIdx.y + g_blockDim.y * g_threadIdx.z)) & 63u); }

template <typename T>
inline uint64_t to_bits(T v) {
  static_assert(sizeof(T) <= 8, "wave exchange moves at most 8 bytes");
  uint64_t b = 0;
  memcpy(&b, &v, sizeof(T));
  return b;
}
template <typename T>
inline T from_bits(uint64_t b) {
  T v;
  memcpy(&v, &b, sizeof(T));
  return v;
}
template <typename T>
inline T shfl_from(T v, int src) {
  uint64_t all[64], act;
  wave_exchange(to_bits(v), all, &act);
  if (src < 0 || src > 63 || !((act >> src) & 1ull)) return v;
  return from_bits<T>(all[src]);
}
}  // namespace hipemu

#define threadIdx (hipemu::g_threadIdx)
#define blockIdx (hipemu::g_blockIdx)
#define blockDim (hipemu::g_blockDim)
#define gridDim (hipemu::g_gridDim)

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
  hipemu::launch((grid), (block), (size_t)(lds), [&]() { (kernel)(__VA_ARGS__); })

inline const char* hipGetErrorString(hipError_t) { return "hipemu: kernel deadlocked (divergent barrier?)"; }
inline hipError_t hipGetLastError() { return hipemu::take_error(); }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
template <typename F>
inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
struct hipFuncAttributes { size_t sharedSizeBytes; };
inline hipError_t hipFuncGetAttributes(hipFuncAttributes* a, const void*) { a->sharedSizeBytes = 2048; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) {   // HIPEMU_CU_COUNT: a smaller chip (launch policies that ask for it)
  const char* e = getenv("HIPEMU_CU_COUNT");
  *v = e && atoi(e) > 0 ? atoi(e) : 256;
  return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }

// ---- barriers and wave intrinsics ------------------------------------------------------------------------
inline void __syncthreads() { hipemu::block_barrier(); }
inline void __threadfence() {}   // one OS thread: every store is visible at once
inline int __syncthreads_or(int p) { return hipemu::block_barrier_or(p); }
#define __builtin_amdgcn_wave_barrier() ((void)hipemu::shfl_from(0, 0))
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_fract(x) ((x) - floor(x))
#define __builtin_amdgcn_rsq(x) (1.0 / sqrt((double)(x)))   /* v_rsq_f64: a seed; the kernels refine it */
#define __builtin_amdgcn_fractf(x) ((x) - floorf(x))
#define __builtin_amdgcn_sqrtf(x) sqrtf(x)                   /* v_sqrt_f32 (1 ulp on the device; correctly rounded here) */
// v_mfma_i32_16x16x64_i8: byte s of lane (m, g) of A meets byte s of lane (n, g) of B (m, n = lane & 15, g = lane >> 4);
// D[m = 4 * (lane >> 4) + reg][n = lane & 15] (the layout scripts/ubench/mfma_i8.hip checks on the device).  All 64 lanes
// must take part (the kernels call it wave-uniformly).
typedef int hipemu_v4i __attribute__((ext_vector_type(4)));
inline hipemu_v4i hipemu_mfma_i32_16x16x64_i8(hipemu_v4i a, hipemu_v4i b, hipemu_v4i c) {
  uint64_t all[4][64], act;
  uint64_t mine[4];
  memcpy(&mine[0], &a, 16);
  memcpy(&mine[2], &b, 16);
  for (int q = 0; q < 4; ++q) hipemu::wave_exchange(mine[q], all[q], &act);
  const int lane = hipemu::lane_id();
  const int n = lane & 15;
  hipemu_v4i d = c;
  for (int r = 0; r < 4; ++r) {
    const int m = 4 * (lane >> 4) + r;
    int acc = 0;
    for (int g = 0; g < 4; ++g) {
      signed char ab[16], bb[16];
      uint64_t t[2] = {all[0][m + 16 * g], all[1][m + 16 * g]};
      memcpy(ab, t, 16);
      uint64_t u[2] = {all[2][n + 16 * g], all[3][n + 16 * g]};
      memcpy(bb, u, 16);
      for (int s = 0; s < 16; ++s) acc += (int)ab[s] * (int)bb[s];
    }
    d[r] = c[r] + acc;
  }
  return d;
}
#define __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, x, y, z) hipemu_mfma_i32_16x16x64_i8((a), (b), (c))
// MUBUF raw buffer access: resource = base pointer + num_records; offsets are plain byte offsets; an access whose lane
// offset reaches num_records is dropped (loads return 0), as the hardware's range check does for raw buffers
struct hipemu_rsrc { char* base; unsigned num; };
#define __amdgpu_buffer_rsrc_t hipemu_rsrc
inline hipemu_rsrc __builtin_amdgcn_make_buffer_rsrc(void* p, short, int num, int) { return hipemu_rsrc{(char*)p, (unsigned)num}; }
inline bool hipemu_in_range(hipemu_rsrc r, int voff, unsigned size) { return (unsigned long long)(unsigned)voff + size <= (unsigned long long)r.num; }
inline unsigned __builtin_amdgcn_raw_buffer_load_b32(hipemu_rsrc r, int voff, int soff, int) {
  unsigned v = 0;
  if (hipemu_in_range(r, voff, 4)) memcpy(&v, r.base + (size_t)(unsigned)voff + (size_t)(unsigned)soff, 4);
  return v;
}
inline void __builtin_amdgcn_raw_buffer_store_b32(unsigned v, hipemu_rsrc r, int voff, int soff, int) {
  if (hipemu_in_range(r, voff, 4)) memcpy(r.base + (size_t)(unsigned)voff + (size_t)(unsigned)soff, &v, 4);
}
typedef unsigned hipemu_v2u __attribute__((ext_vector_type(2)));
inline hipemu_v2u __builtin_amdgcn_raw_buffer_load_b64(hipemu_rsrc r, int voff, int soff, int) {
  hipemu_v2u v = {0u, 0u};
  if (hipemu_in_range(r, voff, 8)) memcpy(&v, r.base + (size_t)(unsigned)voff + (size_t)(unsigned)soff, 8);
  return v;
}
inline void __builtin_amdgcn_raw_buffer_store_b64(hipemu_v2u v, hipemu_rsrc r, int voff, int soff, int) {
  if (hipemu_in_range(r, voff, 8)) memcpy(r.base + (size_t)(unsigned)voff + (size_t)(unsigned)soff, &v, 8);
}
typedef unsigned hipemu_v4u __attribute__((ext_vector_type(4)));
inline void __builtin_amdgcn_raw_buffer_store_b128(hipemu_v4u v, hipemu_rsrc r, int voff, int soff, int) {
  if (hipemu_in_range(r, voff, 16)) memcpy(r.base + (size_t)(unsigned)voff + (size_t)(unsigned)soff, &v, 16);
}
// v_permlane16_swap_b32 (gfx950): rows of 16 lanes; row 1 of the first operand <-> row 0 of the second, row 3 <-> row 2.
// Returns {new first, new second}.  Every lane of the wave must take part.
inline hipemu_v2u hipemu_permlane16_swap(unsigned a, unsigned b) {
  uint64_t all[64], act;
  hipemu::wave_exchange(((uint64_t)b << 32) | a, all, &act);
  const int lane = hipemu::lane_id();
  const int row = lane >> 4;
  hipemu_v2u r = {a, b};
  if (row & 1) r[0] = (unsigned)(all[lane - 16] >> 32);        // first.row(odd) <- second.row(even)
  else r[1] = (unsigned)(all[lane + 16] & 0xffffffffull);       // second.row(even) <- first.row(odd)
  return r;
}
#define __builtin_amdgcn_permlane16_swap(a, b, fi, bc) hipemu_permlane16_swap((a), (b))
// v_alignbit_b32: low 32 bits of ({hi, lo} >> (s & 31))
#define __builtin_amdgcn_alignbit(hi, lo, s) \
  ((unsigned)(((((unsigned long long)(unsigned)(hi)) << 32) | (unsigned)(lo)) >> ((s) & 31)))
// v_perm_b32: result byte i = byte sel[i] of the 8-byte pool {src1 (bytes 0-3), src0 (bytes 4-7)}; 0x0c -> 0x00
inline unsigned hipemu_perm(unsigned src0, unsigned src1, unsigned sel) {
  const unsigned long long pool = ((unsigned long long)src0 << 32) | src1;
  unsigned r = 0;
  for (int i = 0; i < 4; ++i) {
    const unsigned s = (sel >> (8 * i)) & 0xffu;
    unsigned b = 0;
    if (s <= 7) b = (unsigned)((pool >> (8 * s)) & 0xffu);
    else if (s == 0x0c) b = 0;
    else if (s >= 0x0d) b = 0xffu;
    else b = ((pool >> (16 * (s - 8) + 15)) & 1u) ? 0xffu : 0u;   // 8..11: sign of a 16-bit half
    r |= b << (8 * i);
  }
  return r;
}
#define __builtin_amdgcn_perm(a, b, s) hipemu_perm((a), (b), (s))
#define __builtin_amdgcn_readfirstlane(v) (hipemu::readfirstlane(v))
namespace hipemu {
template <typename T>
inline T readfirstlane(T v) {
  uint64_t all[64], act;
  wave_exchange(to_bits(v), all, &act);
  return from_bits<T>(all[__builtin_ctzll(act)]);
}
}  // namespace hipemu
// DPP controls the kernels use: wave_shr:1 (0x138: lane i <- lane i - 1) / wave_shl:1 (0x130: lane i <- lane i + 1),
// quad_perm (0x00 - 0xff: two selector bits per lane of a quad), row_half_mirror (0x141) / row_mirror (0x140),
// row_shr:n (0x110 + n: lane i <- lane i - n inside its row of 16), row_bcast:15 (0x142: lane 15 of each row to the next
// row), row_bcast:31 (0x143: lane 31 to rows 2 and 3).  A lane without a source, or whose row / bank the masks disable,
// keeps `old` (bound_ctrl: 0 instead, for a missing source).
inline int hipemu_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  uint64_t all[64], act;
  hipemu::wave_exchange(hipemu::to_bits(src), all, &act);
  const int lane = hipemu::lane_id();
  const int row = lane >> 4, in_row = lane & 15;
  if (!((row_mask >> row) & 1) || !((bank_mask >> (in_row >> 2)) & 1)) return old;
  int from = -1;
  if (ctrl >= 0 && ctrl <= 0xff) from = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);   // quad_perm
  else if (ctrl == 0x141) from = (lane & ~7) | (7 - (lane & 7));                            // row_half_mirror
  else if (ctrl == 0x140) from = (lane & ~15) | (15 - (lane & 15));                         // row_mirror
  else if (ctrl == 0x138) from = lane - 1;
  else if (ctrl == 0x130) from = lane + 1 < 64 ? lane + 1 : -1;
  else if (ctrl > 0x110 && ctrl <= 0x11f) from = in_row >= (ctrl & 15) ? lane - (ctrl & 15) : -1;
  else if (ctrl == 0x142) from = row >= 1 ? 16 * row - 1 : -1;
  else if (ctrl == 0x143) from = row >= 2 ? 31 : -1;
  if (from < 0 || from > 63 || !((act >> from) & 1ull)) return bound_ctrl ? 0 : old;
  return hipemu::from_bits<int>(all[from]);
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) hipemu_update_dpp((old), (src), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_readlane(v, l) (hipemu::shfl_from((v), (l)))
template <typename T>
inline T __shfl(T v, int src, int width = 64) { (void)width; return hipemu::shfl_from(v, src); }
template <typename T>
inline T __shfl_xor(T v, int mask, int width = 64) { (void)width; return hipemu::shfl_from(v, hipemu::lane_id() ^ mask); }
template <typename T>
inline T __shfl_up(T v, unsigned d, int width = 64) { (void)width; return hipemu::shfl_from(v, hipemu::lane_id() - (int)d); }
template <typename T>
inline T __shfl_down(T v, unsigned d, int width = 64) { (void)width; return hipemu::shfl_from(v, hipemu::lane_id() + (int)d); }
inline unsigned long long __ballot(int pred) {
  uint64_t all[64], act;
  hipemu::wave_exchange(pred ? 1u : 0u, all, &act);
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l)
    if (((act >> l) & 1ull) && all[l]) m |= 1ull << l;
  return m;
}

// ---- device math that is global-namespace in HIP ---------------------------------------------------------------
using std::max;
using std::min;
inline void sincospi(double x, double* s, double* c) {
  // exact at the multiples of 1/2 like ROCm's sincospi (the kernels reduce their twiddle indices first)
  double r = fmod(x, 2.0);
  if (r < 0) r += 2.0;
  if (r == 0.0) { *s = 0.0; *c = 1.0; }
  else if (r == 0.5) { *s = 1.0; *c = 0.0; }
  else if (r == 1.0) { *s = 0.0; *c = -1.0; }
  else if (r == 1.5) { *s = -1.0; *c = 0.0; }
  else { *s = sin(M_PI * r); *c = cos(M_PI * r); }
}

// ---- bit casts / integer intrinsics ----------------------------------------------------------------------
inline double __longlong_as_double(long long v) { return hipemu::from_bits<double>((uint64_t)v); }
inline long long __double_as_longlong(double v) { return (long long)hipemu::to_bits(v); }
inline float __uint_as_float(unsigned v) { return hipemu::from_bits<float>(v); }
inline unsigned __float_as_uint(float v) { return (unsigned)hipemu::to_bits(v); }
inline int __float_as_int(float v) { return (int)hipemu::to_bits(v); }
inline float __int_as_float(int v) { return hipemu::from_bits<float>((unsigned)v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }

// ---- atomics (one OS thread: plain read-modify-write, returning the old value) -------------------------------
template <typename T, typename U>
inline T atomicAdd(T* p, U v) { T o = *p; *p = (T)(o + (T)v); return o; }
#define __hip_atomic_load(p, order, scope) (*(p))           /* one OS thread: every store is visible at once */
#define __hip_atomic_store(p, v, order, scope) ((void)(*(p) = (v)))
#define __hip_atomic_fetch_add(p, v, order, scope) atomicAdd((p), (v))
template <typename T, typename U>
inline T atomicSub(T* p, U v) { T o = *p; *p = (T)(o - (T)v); return o; }
template <typename T, typename U>
inline T atomicOr(T* p, U v) { T o = *p; *p = (T)(o | (T)v); return o; }
template <typename T, typename U>
inline T atomicAnd(T* p, U v) { T o = *p; *p = (T)(o & (T)v); return o; }
template <typename T, typename U>
inline T atomicMin(T* p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <typename T, typename U>
inline T atomicMax(T* p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <typename T, typename U>
inline T atomicExch(T* p, U v) { T o = *p; *p = (T)v; return o; }
template <typename T, typename U, typename V>
inline T atomicCAS(T* p, U expected, V desired) { T o = *p; if (o == (T)expected) *p = (T)desired; return o; }
