// TEST INFRASTRUCTURE: exhaustive proof of the quotient identity pl_quot (csrc/pl_common.h) relies on.
// For every integer x in [0, 65535] and integer d in [1, 65535]:  RN(x / d) == fma(fma(-q0, d, x), r, q0) with q0 = RN(x * r),
// r = RN(1 / d).  (Odd symmetry of round-to-nearest extends it to negative x.)  Exit status 0 = no mismatch.
#include <math.h>
#include <omp.h>
#include <stdio.h>
int main() {
  long bad = 0;
#pragma omp parallel for reduction(+:bad) schedule(dynamic, 64)
  for (int di = 1; di <= 65535; ++di) {
    const double d = (double)di, r = 1.0 / d;
    for (int xi = 0; xi <= 65535; ++xi) {
      const double x = (double)xi;
      const double q0 = x * r;
      const double rem = fma(-q0, d, x);
      const double q = fma(rem, r, q0);
      if (q != x / d) ++bad;
    }
  }
  printf("mismatches: %ld\n", bad);
  return bad != 0;
}
