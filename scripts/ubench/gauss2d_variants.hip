// Timing attribution for gauss2d_mm (pylinac_amd/csrc/gaussian_mm.hip): the kernel source is compiled here with
// -DPL_G2D_VARIANT=<bits> (see the switch list next to PL_G2D_VARIANT in that file) and timed on 256 x 1024 x 1024 uint16
// frames, sigma 5.  Variants other than 0 compute garbage: this is a stopwatch, not a test.
//   scripts/build_g2d_variants.sh 0 3 7 ...   ->  scripts/ubench/g2d_v<bits>
#include "../../pylinac_amd/csrc/gaussian_mm.hip"

#include <time.h>

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
    host[i] = epid == 2 ? (unsigned short)(1000 + ((s >> 22) & 3))          // low-toggle data: the power-cap probe
            : epid ? (unsigned short)((in_field ? 39600 : 1600) + (s >> 22) * 800 / 1024)
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
  const int iters = argc > 3 ? atoi(argv[3]) : 10;
  hipEventRecord(e0);
  for (int it = 0; it < iters; ++it) pl_gauss_mm2d_launch(din, dout, 0, n, h, w, wts, R, 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  if (argc > 4) {   // DVFS ramp: after `idle_ms` of idleness, time consecutive groups of 5 launches
    const int idle_ms = atoi(argv[4]);
    hipDeviceSynchronize();
    struct timespec ts = {idle_ms / 1000, (idle_ms % 1000) * 1000000L};
    nanosleep(&ts, nullptr);
    std::vector<hipEvent_t> ev(61);
    for (auto& e : ev) hipEventCreate(&e);
    hipEventRecord(ev[0]);
    for (int g = 0; g < 60; ++g) {
      for (int it = 0; it < 5; ++it) pl_gauss_mm2d_launch(din, dout, 0, n, h, w, wts, R, 0);
      hipEventRecord(ev[g + 1]);
    }
    hipEventSynchronize(ev[60]);
    printf("ramp after %d ms idle (ms per launch, groups of 5):", idle_ms);
    for (int g = 0; g < 60; ++g) {
      float t;
      hipEventElapsedTime(&t, ev[g], ev[g + 1]);
      printf(" %.3f", t / 5);
    }
    printf("\n");
  }
#if PL_G2D_TIMING
  {
    std::vector<unsigned long long> dbg(4096 * 8 * 4);
    hipMemcpyFromSymbol(dbg.data(), HIP_SYMBOL(g2d_dbg), dbg.size() * 8);
    const int wgs = n * 4 < 4096 ? n * 4 : 4096;
    const char* names[4] = {"W+load-issue", "axis0", "barrier", "axis1"};
    for (int wv = 0; wv < 8; ++wv) {
      printf("  wave %d:", wv);
      double tot = 0;
      for (int k = 0; k < 4; ++k) {
        double a = 0;
        for (int b = 0; b < wgs; ++b) a += (double)dbg[(b * 8 + wv) * 4 + k];
        a /= wgs * 64.0;                                       // per step (64 steps per workgroup at 1024 rows)
        tot += a;
        printf("  %s %7.0f", names[k], a);
      }
      printf("   = %7.0f memtime ticks per step\n", tot);
    }
  }
#endif
  printf("variant %3d: %.4f ms per launch of %d frames, %s data (%s)\n", PL_G2D_VARIANT, ms / iters, n, epid == 2 ? "low-toggle (1000 + 2 random bits)" : epid ? "EPID-like" : "ramp+noise",
         hipGetErrorString(hipGetLastError()));
  return 0;
}
