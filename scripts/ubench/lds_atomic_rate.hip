// ds_add_u32 (no return) cost per 64-lane instruction on gfx950 for address patterns a histogram produces:
//   distinct consecutive bins | 16 distinct (4 lanes each) | one bin | random in 64 bins | random in 8192 bins
// 16 waves per workgroup, one workgroup per CU (the Otsu kernel's shape), 4096 instructions per wave.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/lds_atomic_rate.hip -o scripts/ubench/lds_atomic_rate
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void __launch_bounds__(1024) k(const unsigned* __restrict__ idx, unsigned* out, int iters) {
  extern __shared__ unsigned bins[];
  for (int i = threadIdx.x; i < 38912; i += 1024) bins[i] = 0;
  __syncthreads();
  unsigned a[8];
  for (int q = 0; q < 8; ++q) a[q] = idx[q * 1024 + threadIdx.x];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 8; ++q) atomicAdd(&bins[a[q]], 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = bins[0] + bins[100];
}

int main() {
  const int iters = 512;
  unsigned *didx, *dout;
  (void)hipMalloc(&didx, 8 * 1024 * 4);
  (void)hipMalloc(&dout, 256 * 4);
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 38912 * 4);
  unsigned h[8 * 1024];
  const char* names[] = {"64 distinct consecutive bins per wave", "16 distinct bins per wave (4 lanes each)", "one bin per wave",
                         "random in a 64-bin window", "random in a 8192-bin window", "random in a 64-bin window, per-wave windows apart"};
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int pat = 0; pat < 6; ++pat) {
    unsigned s = 12345;
    for (int q = 0; q < 8; ++q)
      for (int t = 0; t < 1024; ++t) {
        s = s * 1664525u + 1013904223u;
        const int lane = t & 63, wave = t >> 6;
        unsigned v;
        switch (pat) {
          case 0: v = 1000 + lane + 64 * q; break;
          case 1: v = 1000 + (lane >> 2) + 16 * q; break;
          case 2: v = 1000 + q; break;
          case 3: v = 1000 + (s >> 26); break;
          case 4: v = 1000 + (s >> 19); break;
          default: v = 1000 + 2000 * wave + (s >> 26); break;
        }
        h[q * 1024 + t] = v;
      }
    (void)hipMemcpy(didx, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(256), dim3(1024), 38912 * 4, 0, didx, dout, 8);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(1024), 38912 * 4, 0, didx, dout, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // per CU: 16 waves x iters x 8 instructions
    printf("%-52s %7.3f ms  = %6.1f ns per ds_add_u32 wave-instruction per CU\n", names[pat], ms, ms * 1e6 / (16.0 * iters * 8));
  }
  return 0;
}
