// gfx950: 9 x v_mfma_i32_16x16x64_i8 per "tile" against ~60 integer VALU in ONE instruction stream, for two VALU mixes:
//   FULL  only full-rate opcodes (v_ashrrev_i32, v_add_u32, v_and_b32, v_or_b32, v_sub_u32)
//   HALF  the half-rate kind the compiler picks for the same arithmetic (v_lshl_add_u32, v_add3_u32, v_perm_b32, v_lshlrev_b32)
// each with 1, 2 and 3 waves per SIMD.  Prints ns per tile per SIMD for MFMA only, VALU only, and both.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_valu_mix.hip -o scripts/ubench/mfma_valu_mix
#include <hip/hip_runtime.h>

#include <cstdio>

typedef int v4i __attribute__((ext_vector_type(4)));

#define MFMA(acc) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FULL6(x0, x1, x2)                                                                            \
  asm volatile("v_ashrrev_i32 %0, 8, %0\n\tv_add_u32 %1, %1, %3\n\tv_and_b32 %2, %2, %3\n\t"        \
               "v_or_b32 %0, %0, %3\n\tv_sub_u32 %1, %1, %3\n\tv_ashrrev_i32 %2, 1, %2"             \
               : "+v"(x0), "+v"(x1), "+v"(x2) : "v"(k))
#define HALF6(x0, x1, x2)                                                                            \
  asm volatile("v_lshl_add_u32 %0, %0, 1, %3\n\tv_add3_u32 %1, %1, %3, %3\n\tv_perm_b32 %2, %2, %3, %3\n\t" \
               "v_lshlrev_b32 %0, 1, %0\n\tv_lshl_add_u32 %1, %1, 1, %3\n\tv_add3_u32 %2, %2, %3, %3" \
               : "+v"(x0), "+v"(x1), "+v"(x2) : "v"(k))

template <int MODE, bool FULL>   // bit 0: MFMA, bit 1: VALU
__global__ void __launch_bounds__(1024) loop(const v4i* src, int* dst, int iters) {
  const v4i a = src[threadIdx.x & 63], b = src[64 + (threadIdx.x & 63)];
  v4i c[5];
  for (int q = 0; q < 5; ++q) c[q] = v4i{q, 0, 0, 0};
  int x[6];
  for (int q = 0; q < 6; ++q) x[q] = threadIdx.x + q;
  const int k = dst[0];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 10; ++q) {                 // ten slots: MFMA in nine of them, six VALU in each
      if ((MODE & 1) && q < 9) MFMA(c[q % 5]);
      if (MODE & 2) {
        if (FULL) FULL6(x[(q & 1) * 3], x[(q & 1) * 3 + 1], x[(q & 1) * 3 + 2]);
        else HALF6(x[(q & 1) * 3], x[(q & 1) * 3 + 1], x[(q & 1) * 3 + 2]);
      }
    }
  }
  v4i s = c[0];
  for (int q = 1; q < 5; ++q) s += c[q];
  int t = s[0] + s[1] + s[2] + s[3];
  for (int q = 0; q < 6; ++q) t += x[q];
  dst[1 + blockIdx.x * blockDim.x + threadIdx.x] = t;
}

int main() {
  v4i* src;
  int* dst;
  (void)hipMalloc(&src, 128 * sizeof(v4i));
  (void)hipMalloc(&dst, (1 + 256 * 1024) * sizeof(int));
  (void)hipMemset(src, 1, 128 * sizeof(v4i));
  (void)hipMemset(dst, 0, (1 + 256 * 1024) * sizeof(int));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 10000;
  auto run = [&](auto kern, int waves, const char* name) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(256 * waves), 0, 0, src, dst, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256 * waves), 0, 0, src, dst, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %d wave(s)/SIMD: %7.2f ns per tile per SIMD\n", name, waves, ms * 1e6 / iters / waves);
  };
  for (int wv = 1; wv <= 3; ++wv) {
    run(loop<1, true>, wv, "9 MFMA");
    run(loop<2, true>, wv, "60 VALU full-rate");
    run(loop<2, false>, wv, "60 VALU half-rate");
    run(loop<3, true>, wv, "9 MFMA + 60 VALU full-rate");
    run(loop<3, false>, wv, "9 MFMA + 60 VALU half-rate");
  }
  return 0;
}
