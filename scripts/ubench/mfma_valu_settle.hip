// Settles the question the two round-2 microbenchmarks left open: do v_mfma_i32_16x16x64_i8 and full-rate
// integer VALU overlap on one gfx950 SIMD for gauss2d_mm's REAL tile mix (9 MFMA + 44 full-rate VALU per 16 x 16 tile)?
// Wall time alone cannot tell "the pipes add" from "the clock drops under the denser stream" (DVFS), so every row reports
//   ns per tile per SIMD (HIP events)        cycles per tile per SIMD (s_memtime delta of one wave / tiles / waves-per-SIMD)
//   effective shader clock = cycles / ns
// Rows: M (MFMA only), V (VALU only), I (1 MFMA : ~5 VALU interleaved in ONE wave, independent registers), B (blocked: 9 MFMA
// then 44 VALU), P (partner waves of one SIMD: waves 0-3 of a 512-thread workgroup run M, waves 4-7 run V; only with an even
// number of waves per SIMD), C (the kernel's carry chain: three tiles in lock step, VALU shifts consume the MFMA results --
// compiler-scheduled intrinsics); each at 1 - 4 waves per SIMD, with constant and with random operand bytes.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_valu_settle.hip -o scripts/ubench/mfma_valu_settle
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

#define MFMA(acc) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
// five / four full-rate opcodes on independent registers (no chain shorter than 4 instructions)
#define V5(x0, x1, x2, x3, x4)                                                                                       \
  asm volatile("v_ashrrev_i32 %0, 8, %0\n\tv_add_u32 %1, %1, %5\n\tv_and_b32 %2, %2, %5\n\tv_or_b32 %3, %3, %5\n\t" \
               "v_xor_b32 %4, %4, %5"                                                                                \
               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4) : "v"(k))
#define V4(x0, x1, x2, x3)                                                                                       \
  asm volatile("v_ashrrev_i32 %0, 8, %0\n\tv_add_u32 %1, %1, %4\n\tv_and_b32 %2, %2, %4\n\tv_sub_u32 %3, %3, %4" \
               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(k))

enum { kM = 0, kV = 1, kI = 2, kB = 3, kP = 4, kC = 5 };

template <int MODE>
__global__ void __launch_bounds__(1024) loop(const v4i* src, int* dst, long long* clk, int iters) {
  const v4i a = src[threadIdx.x & 63], b = src[64 + (threadIdx.x & 63)];
  v4i c[9];
  for (int q = 0; q < 9; ++q) c[q] = v4i{q, 0, 0, 0};
  int x[10];
  for (int q = 0; q < 10; ++q) x[q] = threadIdx.x + q;
  const int k = dst[0];
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  bool do_m = MODE == kM, do_v = MODE == kV;
  if (MODE == kP) { do_m = wave < nw / 2; do_v = !do_m; }
  const long long t0 = clock64();
  const long long w0 = wall_clock64();
  if (MODE == kI) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 9; ++q) {                               // 9 x (1 MFMA + 5 VALU) - 1 = 9 MFMA + 44 VALU
        MFMA(c[q]);
        if (q < 8) V5(x[(q & 1) * 5], x[(q & 1) * 5 + 1], x[(q & 1) * 5 + 2], x[(q & 1) * 5 + 3], x[(q & 1) * 5 + 4]);
        else V4(x[0], x[1], x[2], x[3]);
      }
    }
  } else if (MODE == kB) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 9; ++q) MFMA(c[q]);
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        if (q < 8) V5(x[(q & 1) * 5], x[(q & 1) * 5 + 1], x[(q & 1) * 5 + 2], x[(q & 1) * 5 + 3], x[(q & 1) * 5 + 4]);
        else V4(x[0], x[1], x[2], x[3]);
      }
    }
  } else if (MODE == kC) {
    // the kernel's chain (gaussian_mm.hip mm_tiles): three tiles in lock step, five levels, shifts between the levels, the
    // decision byte and a packed result per output; 9 MFMA + ~44 VALU per tile, scheduled by the compiler
    v4i lo[3], hi[3];
    for (int i = 0; i < 3; ++i) { lo[i] = a + v4i{i, i, i, i}; hi[i] = b - v4i{i, i, i, i}; }
    v4i keep = v4i{0, 0, 0, 0};
    for (int it = 0; it < iters; it += 3) {
      v4i t[3], t3[3], t4[3], r[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) t[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(lo[i], b, c[0], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 3; ++i) t[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(hi[i], a, t[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 3; ++i) t[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(lo[i], a, t[i] >> 8, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 3; ++i) t[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(hi[i], b, t[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 3; ++i) t3[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(lo[i], b, t[i] >> 8, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 3; ++i) t3[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(hi[i], a, t3[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 3; ++i) t4[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(lo[i], a, (t3[i] >> 8) + c[1], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 3; ++i) t4[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(hi[i], b, t4[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        r[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(hi[i], a, t4[i] >> 8, 0, 0, 0);
        const v4i z = (t3[i] | t4[i]) & 255;
        unsigned zm = (unsigned)z[0] < (unsigned)z[1] ? z[0] : z[1];
        zm = zm < (unsigned)z[2] ? zm : z[2];
        zm = zm < (unsigned)z[3] ? zm : z[3];
        const unsigned p0 = __builtin_amdgcn_perm(r[i][1], r[i][0], 0x05040100u), p1 = __builtin_amdgcn_perm(r[i][3], r[i][2], 0x05040100u);
        keep += v4i{(int)p0, (int)p1, (int)(zm == 0u), 0};
        // next iteration's samples depend on this one's results: nothing hoists
        lo[i] = lo[i] ^ v4i{(int)p0, 0, 0, 0};
      }
    }
    c[2] += keep;
  } else if (do_m) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 9; ++q) MFMA(c[q]);
    }
  } else if (do_v) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        if (q < 8) V5(x[(q & 1) * 5], x[(q & 1) * 5 + 1], x[(q & 1) * 5 + 2], x[(q & 1) * 5 + 3], x[(q & 1) * 5 + 4]);
        else V4(x[0], x[1], x[2], x[3]);
      }
    }
  }
  v4i s = c[0];
  for (int q = 1; q < 9; ++q) s += c[q];
  int t = s[0] + s[1] + s[2] + s[3];
  for (int q = 0; q < 10; ++q) t += x[q];
  const long long t1 = clock64();
  const long long w1 = wall_clock64();
  dst[1 + blockIdx.x * blockDim.x + threadIdx.x] = t;
  if ((threadIdx.x & 63) == 0) {
    clk[2 * (blockIdx.x * nw + wave)] = t1 - t0;
    clk[2 * (blockIdx.x * nw + wave) + 1] = w1 - w0;
  }
}

int main(int argc, char** argv) {
  const int only_waves = argc > 1 ? atoi(argv[1]) : 0;     // run only this many waves per SIMD
  const int mult = argc > 2 ? atoi(argv[2]) : 1;           // iteration multiplier (sustained-power runs)
  v4i* src;
  int* dst;
  long long* clk;
  (void)hipMalloc(&src, 128 * sizeof(v4i));
  (void)hipMalloc(&dst, (1 + 256 * 1024) * sizeof(int));
  (void)hipMalloc(&clk, 2 * 256 * 16 * sizeof(long long));
  (void)hipMemset(dst, 0, (1 + 256 * 1024) * sizeof(int));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 9999 * mult;
  std::vector<long long> h(2 * 256 * 16);
  for (int rnd = 0; rnd < 2; ++rnd) {
    std::vector<unsigned char> bytes(128 * 16);
    for (auto& v : bytes) v = rnd ? (unsigned char)(rand() & 255) : 1;
    (void)hipMemcpy(src, bytes.data(), bytes.size(), hipMemcpyHostToDevice);
    printf("== operand bytes: %s\n", rnd ? "random" : "constant 1");
    auto run = [&](auto kern, int waves, const char* name, int tiles_per_iter_x2 /* tiles per SIMD-wave, doubled */) {
      const int threads = 256 * waves, nw = threads / 64;
      hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, src, dst, clk, 99);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, src, dst, clk, iters);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      (void)hipMemcpy(h.data(), clk, 2 * 256 * nw * sizeof(long long), hipMemcpyDeviceToHost);
      double cyc = 0, wall = 0;
      for (int i = 0; i < 256 * nw; ++i) { cyc += (double)h[2 * i]; wall += (double)h[2 * i + 1]; }
      cyc /= 256 * nw;
      wall /= 256 * nw;
      // tiles a SIMD completes in the kernel: waves-per-SIMD x iters (P: half the waves run each stream -> waves / 2 tiles of
      // MFMA work and waves / 2 tiles of VALU work per iteration, i.e. waves / 2 whole tiles)
      const double tiles = (double)iters * waves * tiles_per_iter_x2 / 2.0;
      printf("%-44s %d w/SIMD: %7.2f ns/tile/SIMD  %7.1f memtime-ticks/tile/SIMD  %7.1f realtime-ticks(100MHz?)/kernel-us %6.2f  -> ticks/ns %.3f\n",
             name, waves, ms * 1e6 / tiles, cyc / tiles, wall, ms * 1e3, cyc / (ms * 1e6));
    };
    for (int wv = 1; wv <= 4; ++wv) {
      if (only_waves && wv != only_waves) continue;
      run(loop<kM>, wv, "M  9 MFMA", 2);
      run(loop<kV>, wv, "V  44 VALU full-rate", 2);
      run(loop<kI>, wv, "I  interleaved in one wave", 2);
      run(loop<kB>, wv, "B  blocked in one wave (9 M then 44 V)", 2);
      if (wv % 2 == 0) run(loop<kP>, wv, "P  partner waves (half M, half V)", 1);
      run(loop<kC>, wv, "C  kernel's carry chain, 3 tiles in lock step", 2);
    }
  }
  return 0;
}
