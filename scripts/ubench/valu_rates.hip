// Development micro-benchmark (not part of the product): issue rate of the VALU instructions the Gaussian
// decision arithmetic could be built from, on gfx950.   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP 64
#define ITER 256

template <int OP>
__global__ void __launch_bounds__(256) k(unsigned* out, unsigned seed, unsigned wsg) {
  unsigned a[8], x = threadIdx.x * 2654435761u + seed, w = wsg;  // w: wave-uniform (SGPR)
  for (int i = 0; i < 8; ++i) a[i] = x + i;
  double d[4] = {1.0 + x, 2.0 + x, 3.0, 4.0};
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "s"(w));
        if (OP == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "s"(w));
        if (OP == 2) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "s"(w));
        if (OP == 3) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "s"(w));
        if (OP == 4) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(x), "s"(w));
        if (OP == 5) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
        if (OP == 9) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "s"(w));
        if (OP == 10) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
        if (OP == 11) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
        if (OP == 12) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(unsigned long long*)&d[i & 3]) : "v"(x), "v"(a[i]) : "vcc");
      }
      if (OP == 6) {
#pragma unroll
        for (int i = 0; i < 4; ++i)   // 4 packed ops = 8 lanes-ops, counted as 4 instructions
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(unsigned long long*)&a[2 * i]) : "v"(*(unsigned long long*)&d[0]), "v"(*(unsigned long long*)&d[1]));
      }
      if (OP == 7) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 3]), "v"(d[(i + 2) & 3]));
      }
      if (OP == 13) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(*(unsigned long long*)&a[2 * i]) : "v"(*(unsigned long long*)&d[0]));
      }
    }
  }
  unsigned s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  s += (unsigned)(d[0] + d[1] + d[2] + d[3]);
  if (s == 0x12345u) out[0] = s;
}

template <int OP>
void run(const char* name, int per_rep) {
  unsigned* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u, 0x00010001u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u, 0x00010001u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double winst = (double)blocks * 4 * ITER * (REP / 8) * per_rep;   // wave-instructions
  const double per_simd_per_s = winst / 1024.0 / (ms * 1e-3);
  printf("%-28s %8.3f ms  %6.2f G wave-instr/s/SIMD  -> %5.2f cycles/instr @2.4GHz  (%5.2f @2.1GHz)\n", name, ms,
         per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, 2.1e9 / per_simd_per_s);
  hipFree(out);
}

int main() {
  run<0>("v_dot2_u32_u16 (v,s)", 8);
  run<9>("v_dot2_i32_i16", 8);
  run<1>("v_fma_f32", 8);
  run<6>("v_pk_fma_f32 (per instr)", 4);
  run<13>("v_pk_add_f32 (per instr)", 4);
  run<7>("v_fma_f64", 4);
  run<2>("v_mad_u32_u24", 8);
  run<3>("v_dot4_u32_u8", 8);
  run<4>("v_perm_b32", 8);
  run<5>("v_add_u32", 8);
  run<10>("v_cvt_f32_u32", 8);
  run<11>("v_mul_lo_u32", 8);
  run<12>("v_mad_u64_u32", 8);
  return 0;
}
