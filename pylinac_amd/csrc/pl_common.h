// Shared device/host helpers for libpylinac_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/pylinac_hip.h"

#define PL_WAVE 64

// ---- host side -------------------------------------------------------------------------------
void pl_set_error(const char* fmt, ...);
int pl_check_launch(const char* what);

#define PL_REQUIRE(cond, msg)                    \
  do {                                           \
    if (!(cond)) {                               \
      pl_set_error("%s: %s", __func__, msg);     \
      return PL_ERR_INVALID_ARG;                 \
    }                                            \
  } while (0)

static inline int64_t pl_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// compute units of the current device (queried once per device; 256 on an MI355X): grid-shaping heuristics only
int pl_cu_count();
// The "opt in to more than 64 KB of dynamic LDS" flags next to the launches are std::atomic: hipFuncSetAttribute is
// idempotent, so two host threads racing through a first call both set it and both store `true` -- no data race.

// ---- device side -----------------------------------------------------------------------------
// scipy 'reflect' (half-sample symmetric:  d c b a | a b c d | d c b a), valid for any distance.
__device__ __forceinline__ int pl_reflect(int i, int n) {
  if ((unsigned)i < (unsigned)n) return i;
  // single reflection (the only case for halos narrower than the frame): no integer division
  const int m1 = (i < 0) ? (-i - 1) : (2 * n - 1 - i);
  if ((unsigned)m1 < (unsigned)n) return m1;
  const int p = 2 * n;
  int m = i % p;
  if (m < 0) m += p;
  return m >= n ? p - 1 - m : m;
}

// three-operand VALU median/min/max on 32-bit keys (the compiler only forms min3/max3 on its own)
__device__ __forceinline__ unsigned pl_umed3(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ int pl_smed3(int a, int b, int c) {
  int r;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// Blocks b and b+8 share an XCD (observed dispatch, speed only).  Map the hardware block id onto
// a logical id such that CONSECUTIVE logical ids run on the SAME XCD, so that neighbouring tiles
// of one frame (which share halo rows) hit the same 4 MiB L2.  Bijective for any grid size
// (cdna_hip_programming.md, "XCD swizzle must be bijective").
__device__ __forceinline__ unsigned pl_xcd_remap(unsigned b, unsigned nwg) {
  unsigned q = nwg >> 3, r = nwg & 7u, xcd = b & 7u;
  unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (b >> 3);
}

template <typename T>
__device__ __forceinline__ T pl_from_double(double v);
// C cast double -> integer == truncation toward zero (scipy ni_support.c CASE_COPY_LINE_TO_DATA)
template <>
__device__ __forceinline__ unsigned short pl_from_double<unsigned short>(double v) {
  return (unsigned short)(unsigned int)v;
}
template <>
__device__ __forceinline__ short pl_from_double<short>(double v) {
  return (short)(int)v;
}
template <>
__device__ __forceinline__ float pl_from_double<float>(double v) {
  return (float)v;
}
template <>
__device__ __forceinline__ double pl_from_double<double>(double v) {
  return v;
}
template <>
__device__ __forceinline__ int pl_from_double<int>(double v) {
  return (int)v;
}
template <>
__device__ __forceinline__ long long pl_from_double<long long>(double v) {
  return (long long)v;
}
template <>
__device__ __forceinline__ unsigned char pl_from_double<unsigned char>(double v) {
  return (unsigned char)(unsigned int)v;
}

// Buffer addressing (MUBUF): base in a 4-SGPR resource, a wave-uniform byte offset in an SGPR, the lane's byte offset in
// ONE VGPR -- strided row walks cost no per-access VALU address arithmetic.  Bounds checking is not relied upon
// (num_records covers the whole 32-bit offset range); callers keep every offset inside their allocation.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pl_make_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}
__device__ __forceinline__ unsigned pl_buffer_load_u32(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned uniform_off) {
  return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, (int)lane_off, (int)uniform_off, 0);
}
__device__ __forceinline__ void pl_buffer_store_u32(unsigned v, __amdgpu_buffer_rsrc_t r, unsigned lane_off,
                                                    unsigned uniform_off) {
  __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)lane_off, (int)uniform_off, 0);
}

// 8-byte forms, and a BOUNDED resource: an access whose lane offset lies at or beyond `bytes` is dropped by the hardware's
// range check (stores vanish, loads return 0) -- predication without a branch
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pl_make_rsrc_bounded(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint2 pl_buffer_load_u64(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned uniform_off) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)lane_off, (int)uniform_off, 0);
  return uint2{(unsigned)v[0], (unsigned)v[1]};
}
__device__ __forceinline__ void pl_buffer_store_u64(uint2 v, __amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned uniform_off) {
  typedef unsigned pl_v2u __attribute__((ext_vector_type(2)));
  __builtin_amdgcn_raw_buffer_store_b64(pl_v2u{v.x, v.y}, r, (int)lane_off, (int)uniform_off, 0);
}

__device__ __forceinline__ void pl_buffer_store_u128(uint4 v, __amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned uniform_off) {
  typedef unsigned pl_v4u __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(pl_v4u{v.x, v.y, v.z, v.w}, r, (int)lane_off, (int)uniform_off, 0);
}

// neighbour exchange inside a wave on the VALU's data-parallel path (DPP wave_shr:1 / wave_shl:1), NOT through the LDS
// crossbar (__shfl_up / __shfl_down compile to ds_bpermute_b32, which queues behind a kernel's own LDS atomics).
// pl_wave_from_prev: lane i receives lane i - 1's value (lane 0 keeps `v`); pl_wave_from_next: lane i receives lane i + 1's.
__device__ __forceinline__ int pl_wave_from_prev(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int pl_wave_from_next(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false); }

// q = (a - s) / d, the float64 quotient every picket-fence kernel evaluates per pixel (ground() / normalize() folded in:
// s = the frame's minimum, d = max - min).  A float64 division is ~12 instructions, several of them quarter rate; when s and
// d are integers of the uint16 range (d >= 1) the correctly rounded quotient is three full-rate operations:
//   q0 = x * r;  rem = fma(-q0, d, x);  q = fma(rem, r, q0)      with r = RN(1 / d), x = a - s (exact)
// (Markstein's correction step).  That this equals RN(x / d) for EVERY integer |x| <= 65535 and d in [1, 65535] is checked
// exhaustively (4.3e9 pairs, tests/fma_quotient_check.c, run by the CPU suite).  Anything else divides.
struct PlQuot {
  double s, d, r;
  bool fast;
};
__device__ __forceinline__ PlQuot pl_quot_make(double s, double d) {
  PlQuot k;
  k.s = s;
  k.d = d;
  k.fast = d >= 1.0 && d <= 65535.0 && d == floor(d) && s >= 0.0 && s <= 65535.0 && s == floor(s);
  k.r = k.fast ? 1.0 / d : 0.0;
  return k;
}
// `a` must be an integer of the uint16 range when k.fast (the callers pass pixels of uint16 frames)
__device__ __forceinline__ double pl_quot(const PlQuot& k, double a) {
  const double x = a - k.s;
  if (k.fast) {
    const double q0 = x * k.r;
    const double rem = fma(-q0, k.d, x);
    return fma(rem, k.r, q0);
  }
  return x / k.d;
}

// One LDS atomic add at an ABSOLUTE LDS byte address (a kernel whose whole LDS is its dynamic block has that block at
// address 0: a table's byte offset is its address, and the per-element "base + offset" add disappears -- the dynamic block's
// base is a link-time symbol the compiler cannot fold).  pl_lds_base() is what a kernel checks before relying on it.
typedef __attribute__((address_space(3))) unsigned pl_lds_u32;
__device__ __forceinline__ void pl_lds_add_abs(unsigned byte_addr, unsigned v) {
  __hip_atomic_fetch_add((pl_lds_u32*)(uintptr_t)byte_addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ unsigned pl_lds_base(const void* lds_ptr) { return (unsigned)reinterpret_cast<uintptr_t>(lds_ptr); }

// A pointer into memory the kernel never writes, read at wave-uniform addresses: through the CONSTANT address space the
// compiler uses scalar loads (s_load: no vector memory instruction, no place in the in-order vmcnt queue).  A plain
// `const T* __restrict__` read inside a loop that also stores came out as a vector load followed by s_waitcnt vmcnt(0).
#define PL_CONSTANT_AS __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ const PL_CONSTANT_AS T* pl_constant_ptr(const T* p) {
  return (const PL_CONSTANT_AS T*)p;
}

// LDS written by some lanes of a wave and read by others: the hardware runs a wave's LDS operations in order, the fences keep
// the COMPILER from reordering them (wavefront scope: no wait instruction is generated)
__device__ __forceinline__ void pl_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the float64 of lane `l` (wave-uniform) in every lane, on the scalar path (two v_readlane_b32; __shfl would go through the
// LDS crossbar)
__device__ __forceinline__ double pl_readlane_f64(double v, int l) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}

// wave-level reductions (64 lanes, xor butterflies -> every lane holds the result)
template <typename T, typename F>
__device__ __forceinline__ T pl_wave_reduce(T v, F f) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = f(v, __shfl_xor(v, o, 64));
  return v;
}

// Wave-wide reductions of IDEMPOTENT operations (min, max: f(v, v) = v) on the VALU's data-parallel path: four row_shr steps
// leave each row's result in its lane 15, row_bcast:15 / row_bcast:31 carry it across the rows, v_readlane hands lane 63's
// total to every lane -- about twenty vector instructions.  The xor butterflies of pl_wave_reduce compile to ds_bpermute_b32
// (two per step for a float64), an LDS round trip each: ~1 400 cycles per float64 reduction, and the per-window / per-peak
// kernels are chains of those.  Not for sums: the order of a floating-point sum is part of the results.
template <int CTRL, int RM>
__device__ __forceinline__ int pl_dpp_keep(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, RM, 0xf, false); }

template <typename F>
__device__ __forceinline__ int pl_wave_reduce_idem(int v, F f) {
  v = f(v, pl_dpp_keep<0x111, 0xf>(v));
  v = f(v, pl_dpp_keep<0x112, 0xf>(v));
  v = f(v, pl_dpp_keep<0x114, 0xf>(v));
  v = f(v, pl_dpp_keep<0x118, 0xf>(v));
  v = f(v, pl_dpp_keep<0x142, 0xa>(v));
  v = f(v, pl_dpp_keep<0x143, 0xc>(v));
  return __builtin_amdgcn_readlane(v, 63);
}
template <typename F>
__device__ __forceinline__ double pl_wave_reduce_idem(double v, F f) {
  auto step = [&](auto tag_lo, auto tag_hi) {
    const long long b = __double_as_longlong(v);
    const int lo = tag_lo((int)b), hi = tag_hi((int)(b >> 32));
    v = f(v, __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo)));
  };
  step([](int x) { return pl_dpp_keep<0x111, 0xf>(x); }, [](int x) { return pl_dpp_keep<0x111, 0xf>(x); });
  step([](int x) { return pl_dpp_keep<0x112, 0xf>(x); }, [](int x) { return pl_dpp_keep<0x112, 0xf>(x); });
  step([](int x) { return pl_dpp_keep<0x114, 0xf>(x); }, [](int x) { return pl_dpp_keep<0x114, 0xf>(x); });
  step([](int x) { return pl_dpp_keep<0x118, 0xf>(x); }, [](int x) { return pl_dpp_keep<0x118, 0xf>(x); });
  step([](int x) { return pl_dpp_keep<0x142, 0xa>(x); }, [](int x) { return pl_dpp_keep<0x142, 0xa>(x); });
  step([](int x) { return pl_dpp_keep<0x143, 0xc>(x); }, [](int x) { return pl_dpp_keep<0x143, 0xc>(x); });
  return pl_readlane_f64(v, 63);
}

// Sums over the EIGHT lanes of a lane's aligned group (lanes 8k .. 8k + 7), every lane gets the total: two quad permutes and a
// half-row mirror on the DPP path (one vector move per 32-bit word and step) where __shfl_xor is an LDS round trip each.
// Integer sums only (any order is exact).
__device__ __forceinline__ unsigned pl_group8_sum(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);     // quad_perm [1, 0, 3, 2]
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);     // quad_perm [2, 3, 0, 1]
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);    // row_half_mirror: lane i <- lane 7 - i
  return v;
}
__device__ __forceinline__ unsigned long long pl_group8_sum(unsigned long long v) {
  auto from = [](unsigned long long x, auto mv) {
    return ((unsigned long long)(unsigned)mv((int)(x >> 32)) << 32) | (unsigned long long)(unsigned)mv((int)x);
  };
  v += from(v, [](int x) { return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true); });
  v += from(v, [](int x) { return __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true); });
  v += from(v, [](int x) { return __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true); });
  return v;
}

// dispatch a dtype enum onto a template parameter
#define PL_DISPATCH_DTYPE(dtype, T, ...)                         \
  switch (dtype) {                                               \
    case PL_U16: { using T = unsigned short; __VA_ARGS__; break; } \
    case PL_I16: { using T = short; __VA_ARGS__; break; }        \
    case PL_F32: { using T = float; __VA_ARGS__; break; }        \
    case PL_F64: { using T = double; __VA_ARGS__; break; }       \
    case PL_U8: { using T = unsigned char; __VA_ARGS__; break; } \
    case PL_I32: { using T = int; __VA_ARGS__; break; }          \
    case PL_I64: { using T = long long; __VA_ARGS__; break; }    \
    default:                                                     \
      pl_set_error("%s: unsupported dtype %d", __func__, (int)(dtype)); \
      return PL_ERR_UNSUPPORTED;                                 \
  }
