// v_mfma_i32_16x16x64_i8 on gfx950: (1) operand / result lane layout check against a CPU product with random int8
// operands (A slot = lane (m = l & 15, group g = l >> 4), byte s  <->  B slot = lane (n = l & 15, group g), byte s;
// D[m = 4 * (l >> 4) + reg][n = l & 15]); (2) issue rate with independent and dependent accumulators.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_i8.hip -o scripts/ubench/mfma_i8 && scripts/ubench/mfma_i8
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void one(const v4i* a, const v4i* b, v4i* d) {
  v4i acc = {1, 2, 3, 4};
  acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
  d[threadIdx.x] = acc;
}

template <int NACC>
__global__ void rate(const v4i* a, const v4i* b, v4i* d, int iters) {
  v4i x = a[threadIdx.x & 63], y = b[threadIdx.x & 63];
  v4i acc[NACC];
  for (int q = 0; q < NACC; ++q) acc[q] = v4i{q, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, y, acc[q], 0, 0, 0);
  }
  v4i s = acc[0];
  for (int q = 1; q < NACC; ++q) s += acc[q];
  d[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  std::vector<int8_t> A(64 * 16), B(64 * 16);
  srand(3);
  for (auto& v : A) v = (int8_t)(rand() % 256 - 128);
  for (auto& v : B) v = (int8_t)(rand() % 256 - 128);
  v4i *da, *db, *dd;
  hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 1 << 24);
  hipMemcpy(da, A.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(db, B.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, da, db, dd);
  std::vector<int> D(256);
  hipMemcpy(D.data(), dd, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int m = 4 * (l >> 4) + r, n = l & 15;
      long want = r + 1;
      for (int g = 0; g < 4; ++g)
        for (int s = 0; s < 16; ++s) want += (long)A[(m + 16 * g) * 16 + s] * (long)B[(n + 16 * g) * 16 + s];
      if (want != D[l * 4 + r]) { if (bad < 5) printf("mismatch lane %d reg %d: got %d want %ld\n", l, r, D[l * 4 + r], want); ++bad; }
    }
  printf("layout check: %s (%d mismatches of 256)\n", bad ? "FAILED" : "ok", bad);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](auto kern, int nacc, const char* name) {
    const int iters = 2000, blocks = 256 * 8, threads = 256;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, da, db, dd, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, da, db, dd, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * (threads / 64) * iters * nacc;
    printf("%s: %.3f ms, %.1f TOPS, %.2f ns per MFMA per SIMD-equivalent (1024 SIMDs)\n", name, ms,
           n * 32768.0 / (ms * 1e-3) / 1e12, ms * 1e6 / (n / 1024.0));
  };
  run(rate<1>, 1, "dependent chain (1 acc)");
  run(rate<4>, 4, "4 independent accs");
  run(rate<8>, 8, "8 independent accs");
  return bad ? 1 : 0;
}
