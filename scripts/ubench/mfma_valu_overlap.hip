// Do v_mfma_i32_16x16x64_i8 and plain integer VALU overlap on one gfx950 SIMD?  Four loops, one workgroup per CU:
//   M  MFMA only (8 independent accumulators per iteration)          V  VALU only (32 v_lshl_add_u32 per iteration)
//   I  both, interleaved 1 MFMA : 4 VALU inside ONE wave             P  both, on PARTNER waves of the same SIMD (512-thread
//                                                                       workgroup: waves 0-3 run M's loop, waves 4-7 run V's)
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_valu_overlap.hip -o scripts/ubench/mfma_valu_overlap
#include <hip/hip_runtime.h>

#include <cstdio>

typedef int v4i __attribute__((ext_vector_type(4)));

#define MFMA(acc) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define VALU4(x0, x1, x2, x3)                                   \
  asm volatile("v_lshl_add_u32 %0, %0, 1, %4\n\t"               \
               "v_lshl_add_u32 %1, %1, 1, %4\n\t"               \
               "v_lshl_add_u32 %2, %2, 1, %4\n\t"               \
               "v_lshl_add_u32 %3, %3, 1, %4"                   \
               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(k))
#define VALU4F(x0, x1, x2, x3)                                  \
  asm volatile("v_and_b32 %0, %0, %4\n\t"                       \
               "v_xor_b32 %1, %1, %4\n\t"                       \
               "v_add_u32 %2, %2, %4\n\t"                       \
               "v_sub_u32 %3, %3, %4"                           \
               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(k))

template <int MODE, bool FAST>   // 0 = M, 1 = V, 2 = I, 3 = P
__global__ void __launch_bounds__(512) loop(const v4i* src, int* dst, int iters) {
  const v4i a = src[threadIdx.x & 63], b = src[64 + (threadIdx.x & 63)];
  v4i c[8];
  for (int q = 0; q < 8; ++q) c[q] = v4i{q, 0, 0, 0};
  int x[8];
  for (int q = 0; q < 8; ++q) x[q] = threadIdx.x + q;
  const int k = dst[0];
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && threadIdx.x < 256);
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && threadIdx.x >= 256);
  if (MODE == 2) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        MFMA(c[q]);
        if (FAST) VALU4F(x[0], x[1], x[2], x[3]); else VALU4(x[(q & 1) * 4], x[(q & 1) * 4 + 1], x[(q & 1) * 4 + 2], x[(q & 1) * 4 + 3]);
      }
    }
  } else if (do_m) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 8; ++q) MFMA(c[q]);
    }
  } else if (do_v) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (FAST) VALU4F(x[0], x[1], x[2], x[3]); else VALU4(x[(q & 1) * 4], x[(q & 1) * 4 + 1], x[(q & 1) * 4 + 2], x[(q & 1) * 4 + 3]);
      }
    }
  }
  v4i s = c[0];
  for (int q = 1; q < 8; ++q) s += c[q];
  int t = s[0] + s[1] + s[2] + s[3];
  for (int q = 0; q < 8; ++q) t += x[q];
  dst[1 + blockIdx.x * blockDim.x + threadIdx.x] = t;
}

int main() {
  v4i* src;
  int* dst;
  hipMalloc(&src, 128 * sizeof(v4i));
  hipMalloc(&dst, (1 + 256 * 512) * sizeof(int) * 4);
  hipMemset(src, 1, 128 * sizeof(v4i));
  hipMemset(dst, 0, (1 + 256 * 512) * sizeof(int));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  auto run = [&](auto kern, int threads, const char* name) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, src, dst, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, src, dst, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %8.3f ms  = %6.2f ns per iteration (8 MFMA and/or 32 VALU per wave)\n", name, ms, ms * 1e6 / iters);
  };
  run(loop<0, false>, 256, "M  MFMA only, 1 wave/SIMD");
  run(loop<1, false>, 256, "V  VALU only (v_lshl_add_u32), 1 wave/SIMD");
  run(loop<1, true>, 256, "V' VALU only (and/xor/add/sub), 1 wave/SIMD");
  run(loop<2, false>, 256, "I  interleaved in one wave (lshl_add)");
  run(loop<2, true>, 256, "I' interleaved in one wave (and/xor/add/sub)");
  run(loop<3, false>, 512, "P  partner waves: M on waves 0-3, V on waves 4-7");
  run(loop<3, true>, 512, "P' partner waves: M on waves 0-3, V' on waves 4-7");
  run(loop<0, false>, 512, "MM MFMA only, 2 waves/SIMD");
  run(loop<1, false>, 512, "VV VALU only, 2 waves/SIMD");
  run(loop<2, false>, 512, "II interleaved, 2 waves/SIMD");
  return 0;
}
