// Development micro-benchmark #2: VOP2 (4-byte) encodings vs VOP3 (8-byte), to see which forms issue in 2 cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 256
#define OPS(X) \
  X(0, "v_fmac_f32 %0, %1, %2", 1) X(1, "v_fmac_f32 %0, %3, %2", 1) X(2, "v_dot2c_i32_i16 %0, %1, %2", 1) \
  X(3, "v_dot2c_i32_i16 %0, %3, %2", 1) X(4, "v_dot4c_i32_i8 %0, %1, %2", 1) X(5, "v_add_u32_e64 %0, %1, %0", 1) \
  X(6, "v_pk_add_u16 %0, %1, %0", 1) X(7, "v_add_f32 %0, %1, %0", 1) X(8, "v_mul_f32 %0, %1, %0", 1) \
  X(9, "v_xor_b32 %0, %1, %0", 1) X(10, "v_lshlrev_b32 %0, 3, %0", 1) X(11, "v_mul_u32_u24 %0, %1, %0", 1) \
  X(12, "v_add3_u32 %0, %1, %0, %2", 1) X(13, "v_alignbit_b32 %0, %1, %0, 16", 1) X(14, "v_and_b32 %0, %1, %0", 1) \
  X(15, "v_mov_b32 %0, %1", 1) X(16, "v_cvt_f32_ubyte0 %0, %0", 1) X(17, "v_min_u32 %0, %1, %0", 1) \
  X(18, "v_pk_mad_u16 %0, %1, %2, %0", 1) X(19, "v_dot2c_f32_f16 %0, %1, %2", 1) X(20, "v_sub_u32 %0, %0, %1", 1) \
  X(21, "v_add_u32 %0, %3, %0", 1) X(22, "v_mad_u32_u16 %0, %1, %2, %0", 1) X(23, "v_add_u16 %0, %1, %0", 1) \
  X(24, "v_fma_f32 %0, %1, %2, %0", 1) X(25, "v_perm_b32 %0, %1, %0, %3", 1) X(26, "v_lshl_or_b32 %0, %1, 16, %0", 1) \
  X(27, "v_cvt_f32_u32 %0, %0", 1) X(28, "v_bfe_u32 %0, %0, 3, 9", 1) X(29, "v_lshrrev_b32 %0, 16, %0", 1)

template <int OP>
__global__ void __launch_bounds__(256) k(unsigned* out, unsigned seed, unsigned wsg) {
  unsigned a[8], x = threadIdx.x * 2654435761u + seed, y = x ^ 0x1234567u;
  for (int i = 0; i < 8; ++i) a[i] = x + i;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#define X(N, S, C) if (OP == N) asm volatile(S : "+v"(a[i]) : "v"(x), "v"(y), "s"(wsg));
        OPS(X)
#undef X
      }
    }
  }
  unsigned s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 0x12345u) out[0] = s;
}

template <int OP>
void run(const char* name) {
  unsigned* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8;
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u, 0x00010001u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u, 0x00010001u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double winst = (double)blocks * 4 * ITER * 64;
  const double rate = winst / 1024.0 / (ms * 1e-3);
  printf("%-36s %7.3f ms  -> %5.2f cycles/instr @2.4GHz\n", name, ms, 2.4e9 / rate);
  hipFree(out);
}

int main() {
#define X(N, S, C) run<N>(S);
  OPS(X)
#undef X
  return 0;
}
