// Timing attribution for gauss2d_mm (pylinac_amd/csrc/gaussian_mm.hip): the kernel source is compiled here with
// -DPL_G2D_VARIANT=<bits> (see the switch list next to PL_G2D_VARIANT in that file) and timed on 256 x 1024 x 1024 uint16
// frames, sigma 5.  Variants other than 0 compute garbage: this is a stopwatch, not a test.
//   scripts/build_g2d_variants.sh 0 3 7 ...   ->  scripts/ubench/g2d_v<bits>
#include "../../pylinac_amd/csrc/gaussian_mm.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

void pl_set_error(const char*, ...) {}
int pl_check_launch(const char*) { return 0; }
int pl_cu_count() { return 256; }

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 256, h = 1024, w = 1024, R = 20;
  const int epid = argc > 2 ? atoi(argv[2]) : 0;   // 1: EPID-like frames (background 2000, 595-pixel plateau 40000, +-400 noise)
  const size_t px = (size_t)n * h * w;
  std::vector<unsigned short> host(px);
  unsigned s = 12345u;
  for (size_t i = 0; i < px; ++i) {
    s = s * 1664525u + 1013904223u;
    const size_t y = (i / w) % h, x = i % w;
    const bool in_field = y > 214 && y < 810 && x > 214 && x < 810;
    host[i] = epid ? (unsigned short)((in_field ? 39600 : 1600) + (s >> 22) * 800 / 1024)
                   : (unsigned short)(20000 + y * 10 + (s >> 22));             // ramp + noise
  }
  double wts[2 * 20 + 1], sum = 0;
  for (int k = -R; k <= R; ++k) sum += (wts[k + R] = std::exp(-0.5 / 25.0 * k * k));
  for (auto& v : wts) v /= sum;
  unsigned short *din, *dout;
  hipMalloc(&din, px * 2);
  hipMalloc(&dout, px * 2);
  hipMemcpy(din, host.data(), px * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) pl_gauss_mm2d_launch(din, dout, 0, n, h, w, wts, R, 0);
  hipDeviceSynchronize();
  const int iters = 10;
  hipEventRecord(e0);
  for (int it = 0; it < iters; ++it) pl_gauss_mm2d_launch(din, dout, 0, n, h, w, wts, R, 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("variant %3d: %.4f ms per launch of %d frames, %s data (%s)\n", PL_G2D_VARIANT, ms / iters, n, epid ? "EPID-like" : "ramp+noise",
         hipGetErrorString(hipGetLastError()));
  return 0;
}
