// Batched four-parameter Hill fits (the Levenberg-Marquardt remainder).
//
// Replaces: pylinac.core.hill.Hill.fit (pylinac/core/hill.py:18-30) as SingleProfile.inflection_data calls it for both
// penumbrae of a profile (pylinac/core/profile.py:1676-1708):
//     curve_fit(hill_func, x_data, y_data, p0=(min(y), max(y), median(x), 0))
// scipy.optimize.curve_fit without bounds or a Jacobian is scipy.optimize.leastsq = MINPACK's lmdif with ftol = xtol =
// 1.49012e-8, gtol = 0, maxfev = 200 * (n + 1), epsfcn = machine epsilon, factor = 100, mode 1 (variables scaled by the
// column norms of the Jacobian).  MINPACK is a third-party dependency of the reference (scipy >= 1.11, pyproject.toml:30-47),
// absent from /root/reference; this file restates the PUBLISHED algorithm (More, "The Levenberg-Marquardt algorithm:
// implementation and theory", 1978; the minpack routines lmdif, fdjac2, qrfac, lmpar, qrsolv, enorm): forward-difference
// Jacobian, Householder QR with column pivoting, the trust-region parameter by More's iteration, the same acceptance and
// termination tests -- so that the fit stops where scipy's stops.  Parity is anchored on the reference's own fitted
// parameters and inflection points (tests/golden/hill.npz, produced by its SingleProfile / Hill through real scipy) at the
// 1e-5 that row already uses downstream of a fit (the reference reproduces its OWN parameters only to ~1e-7 run to run:
// numpy's vectorised pow is not bit-reproducible).
//
// Two kernels, identical results (n = 4 parameters, m <= 1024 samples): hill_fit_group_kernel -- EIGHT lanes per fit, the
// fit's vectors in LDS, for the batches this mostly serves (thousands of penumbra windows of a few dozen samples) -- and
// hill_fit_kernel -- one LANE per fit, the Jacobian (4 x m), two m-vectors and the samples in a caller-provided workspace in
// global memory, transposed so that the fits of a wave sit side by side -- for longer windows and for batches that fill the chip
// with one lane per fit.
#include <math.h>

#include "pl_common.h"

namespace {

constexpr int kHillN = 4;
constexpr int kHillThreads = 64;

// hill_func, pylinac/core/hill.py:67-78.  NOT inlined: pow() is ~1 500 instructions and the fit kernels call this from several
// places; with everything inlined they were 75 / 120 KB of code against a 64 KB instruction cache.
__device__ __attribute__((noinline)) double hill_value4(double x, double a, double b, double c, double d) {
  return a + (b - a) / (1.0 + pow(c / x, d));
}
__device__ __forceinline__ double hill_value(double x, const double* p) { return hill_value4(x, p[0], p[1], p[2], p[3]); }

// minpack enorm: the Euclidean norm with separate accumulators for small, intermediate and large components
template <typename P>
__device__ __forceinline__ double hill_enorm_core(int n, P x, int64_t stride) {
  const double rdwarf = 3.834e-20, rgiant = 1.304e19;
  double s1 = 0.0, s2 = 0.0, s3 = 0.0, x1max = 0.0, x3max = 0.0;
  const double agiant = rgiant / (double)n;
  for (int i = 0; i < n; ++i) {
    const double xabs = fabs(x[(size_t)i * stride]);
    if (xabs > rdwarf && xabs < agiant) {
      s2 += xabs * xabs;
    } else if (xabs <= rdwarf) {
      if (xabs > x3max) {
        const double t = x3max / xabs;
        s3 = 1.0 + s3 * (t * t);
        x3max = xabs;
      } else if (xabs != 0.0) {
        const double t = xabs / x3max;
        s3 += t * t;
      }
    } else {
      if (xabs > x1max) {
        const double t = x1max / xabs;
        s1 = 1.0 + s1 * (t * t);
        x1max = xabs;
      } else {
        const double t = xabs / x1max;
        s1 += t * t;
      }
    }
  }
  if (s1 != 0.0) return x1max * sqrt(s1 + (s2 / x1max) / x1max);
  if (s2 != 0.0) {
    if (s2 >= x3max) return sqrt(s2 * (1.0 + (x3max / s2) * (x3max * s3)));
    return sqrt(x3max * ((s2 / x3max) + (x3max * s3)));
  }
  return x3max * sqrt(s3);
}
// the norm of one of the four-vectors in registers (inlined: a call would force the array into scratch memory) ...
__device__ __forceinline__ double hill_enorm(int n, const double* x, int stride) { return hill_enorm_core(n, x, (int64_t)stride); }
// ... and of an m-vector in the workspace (one copy of the code)
__device__ __attribute__((noinline)) double hill_enorm_vec(int n, const double* x, int64_t stride) { return hill_enorm_core(n, x, stride); }

// minpack qrsolv for n = 4: given the pivoted R (upper triangle of r, column-major r[j * 4 + i]; the strict lower triangle
// is overwritten with the transposed strict upper triangle of S), solve for x with D x = 0 appended in the least squares
// sense.  sdiag receives the diagonal of S.
__device__ void hill_qrsolv(double* r, const int* ipvt, const double* diag, const double* qtb, double* x, double* sdiag,
                            double* wa) {
  constexpr int n = kHillN;
  for (int j = 0; j < n; ++j) {
    for (int i = j; i < n; ++i) r[j * n + i] = r[i * n + j];
    x[j] = r[j * n + j];
    wa[j] = qtb[j];
  }
  for (int j = 0; j < n; ++j) {
    const int l = ipvt[j];
    if (diag[l] != 0.0) {
      for (int k = j; k < n; ++k) sdiag[k] = 0.0;
      sdiag[j] = diag[l];
      double qtbpj = 0.0;
      for (int k = j; k < n; ++k) {
        if (sdiag[k] == 0.0) continue;
        double cs, sn;
        if (fabs(r[k * n + k]) < fabs(sdiag[k])) {
          const double cotan = r[k * n + k] / sdiag[k];
          sn = 0.5 / sqrt(0.25 + 0.25 * (cotan * cotan));
          cs = sn * cotan;
        } else {
          const double tn = sdiag[k] / r[k * n + k];
          cs = 0.5 / sqrt(0.25 + 0.25 * (tn * tn));
          sn = cs * tn;
        }
        r[k * n + k] = cs * r[k * n + k] + sn * sdiag[k];
        const double temp = cs * wa[k] + sn * qtbpj;
        qtbpj = -sn * wa[k] + cs * qtbpj;
        wa[k] = temp;
        for (int i = k + 1; i < n; ++i) {
          const double t2 = cs * r[k * n + i] + sn * sdiag[i];
          sdiag[i] = -sn * r[k * n + i] + cs * sdiag[i];
          r[k * n + i] = t2;
        }
      }
    }
    sdiag[j] = r[j * n + j];
    r[j * n + j] = x[j];
  }
  int nsing = n;
  for (int j = 0; j < n; ++j) {
    if (sdiag[j] == 0.0 && nsing == n) nsing = j;
    if (nsing < n) wa[j] = 0.0;
  }
  for (int k = 0; k < nsing; ++k) {
    const int j = nsing - 1 - k;
    double sum = 0.0;
    for (int i = j + 1; i < nsing; ++i) sum += r[j * n + i] * wa[i];
    wa[j] = (wa[j] - sum) / sdiag[j];
  }
  for (int j = 0; j < n; ++j) x[ipvt[j]] = wa[j];
}

// minpack lmpar for n = 4: the Levenberg-Marquardt parameter par such that || D x || is within 10 % of delta
__device__ void hill_lmpar(double* r, const int* ipvt, const double* diag, const double* qtb, double delta, double* par,
                           double* x, double* sdiag, double* wa1, double* wa2) {
  constexpr int n = kHillN;
  const double dwarf = 2.2250738585072014e-308;
  int nsing = n;
  for (int j = 0; j < n; ++j) {
    wa1[j] = qtb[j];
    if (r[j * n + j] == 0.0 && nsing == n) nsing = j;
    if (nsing < n) wa1[j] = 0.0;
  }
  for (int k = 0; k < nsing; ++k) {
    const int j = nsing - 1 - k;
    wa1[j] /= r[j * n + j];
    const double temp = wa1[j];
    for (int i = 0; i < j; ++i) wa1[i] -= r[j * n + i] * temp;
  }
  for (int j = 0; j < n; ++j) x[ipvt[j]] = wa1[j];
  int iter = 0;
  for (int j = 0; j < n; ++j) wa2[j] = diag[j] * x[j];
  double dxnorm = hill_enorm(n, wa2, 1);
  double fp = dxnorm - delta;
  if (fp <= 0.1 * delta) {
    if (iter == 0) *par = 0.0;
    return;
  }
  double parl = 0.0;
  if (nsing >= n) {
    for (int j = 0; j < n; ++j) {
      const int l = ipvt[j];
      wa1[j] = diag[l] * (wa2[l] / dxnorm);
    }
    for (int j = 0; j < n; ++j) {
      double sum = 0.0;
      for (int i = 0; i < j; ++i) sum += r[j * n + i] * wa1[i];
      wa1[j] = (wa1[j] - sum) / r[j * n + j];
    }
    const double temp = hill_enorm(n, wa1, 1);
    parl = ((fp / delta) / temp) / temp;
  }
  for (int j = 0; j < n; ++j) {
    double sum = 0.0;
    for (int i = 0; i <= j; ++i) sum += r[j * n + i] * qtb[i];
    wa1[j] = sum / diag[ipvt[j]];
  }
  const double gnorm = hill_enorm(n, wa1, 1);
  double paru = gnorm / delta;
  if (paru == 0.0) paru = dwarf / fmin(delta, 0.1);
  *par = fmax(*par, parl);
  *par = fmin(*par, paru);
  if (*par == 0.0) *par = gnorm / dxnorm;
  for (;;) {
    ++iter;
    if (*par == 0.0) *par = fmax(dwarf, 0.001 * paru);
    const double temp = sqrt(*par);
    for (int j = 0; j < n; ++j) wa1[j] = temp * diag[j];
    hill_qrsolv(r, ipvt, wa1, qtb, x, sdiag, wa2);
    for (int j = 0; j < n; ++j) wa2[j] = diag[j] * x[j];
    dxnorm = hill_enorm(n, wa2, 1);
    const double fp_old = fp;
    fp = dxnorm - delta;
    if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= fp_old && fp_old < 0.0) || iter == 10) return;
    for (int j = 0; j < n; ++j) {
      const int l = ipvt[j];
      wa1[j] = diag[l] * (wa2[l] / dxnorm);
    }
    for (int j = 0; j < n; ++j) {
      wa1[j] /= sdiag[j];
      const double t2 = wa1[j];
      for (int i = j + 1; i < n; ++i) wa1[i] -= r[j * n + i] * t2;
    }
    const double t3 = hill_enorm(n, wa1, 1);
    const double parc = ((fp / delta) / t3) / t3;
    if (fp > 0.0) parl = fmax(parl, *par);
    if (fp < 0.0) paru = fmin(paru, *par);
    *par = fmax(parl, *par + parc);
  }
}

__global__ void __launch_bounds__(kHillThreads)
hill_fit_kernel(const double* __restrict__ xs, const double* __restrict__ ys, const int32_t* __restrict__ lens, int64_t nfits,
                int mmax, int64_t stride, double* __restrict__ work /* [8 * mmax][nfits] */, double* __restrict__ params,
                int32_t* __restrict__ info_out, int32_t* __restrict__ nfev_out, double* __restrict__ step_out) {
  constexpr int n = kHillN;
  const int64_t fit = (int64_t)blockIdx.x * kHillThreads + threadIdx.x;
  if (fit >= nfits) return;
  const int m = lens ? lens[fit] : mmax;
  const double* xd = xs + fit * stride;
  const double* yd = ys + fit * stride;
  // The workspace is TRANSPOSED: element k of fit f at work[k * nfits + f], so that the 64 fits of a wave touch 64 consecutive
  // doubles whenever they are at the same place in the algorithm (a per-fit slab made every access 64 cache lines: the
  // kernel was bound by that latency, 2.3 ms for 8 192 ten-sample fits).  S = distance between a fit's consecutive elements.
  const size_t S = (size_t)nfits;
  double* fjac = work + fit;                                // column-major: fjac[(j * mmax + i) * S]
  double* fvec = fjac + 4 * (size_t)mmax * S;
  double* wa4 = fvec + (size_t)mmax * S;
  double* xt = wa4 + (size_t)mmax * S;                      // the fit's own copy of its samples, same layout
  double* yt = xt + (size_t)mmax * S;
  double* out = params + fit * n;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  if (m < n || m > mmax) {                                  // curve_fit raises for fewer samples than parameters
    for (int j = 0; j < n; ++j) out[j] = nan;
    info_out[fit] = -1;
    if (nfev_out) nfev_out[fit] = 0;
    if (step_out) step_out[fit] = nan;
    return;
  }
  // p0 = (min(y), max(y), np.median(x), 0): the median of the (sorted, as np.arange makes them) x values by selection
  double x[n];
  {
    bool finite = true;                                     // curve_fit(check_finite=True) raises ValueError for NaN / inf
    for (int i = 0; i < m; ++i) {
      xt[i * S] = xd[i];
      yt[i * S] = yd[i];
      finite = finite && fabs(xd[i]) <= 1.7976931348623157e308 && fabs(yd[i]) <= 1.7976931348623157e308;
    }
    if (!finite) {
      for (int j = 0; j < n; ++j) out[j] = nan;
      info_out[fit] = -4;
      if (nfev_out) nfev_out[fit] = 0;
      if (step_out) step_out[fit] = nan;
      return;
    }
    double mn = yd[0], mx = yd[0];
    for (int i = 1; i < m; ++i) { mn = yt[i * S] < mn ? yt[i * S] : mn; mx = yt[i * S] > mx ? yt[i * S] : mx; }
    // order statistics k_lo, k_hi of x by counting (m is a few dozen)
    const int k_hi = m / 2, k_lo = (m & 1) ? k_hi : k_hi - 1;
    double v_lo = 0.0, v_hi = 0.0;
    for (int a = 0; a < m; ++a) {
      const double va = xt[a * S];
      int rank = 0;
      for (int b = 0; b < m; ++b) rank += (xt[b * S] < va || (xt[b * S] == va && b < a)) ? 1 : 0;
      if (rank == k_lo) v_lo = va;
      if (rank == k_hi) v_hi = va;
    }
    x[0] = mn; x[1] = mx; x[2] = (m & 1) ? v_hi : (v_lo + v_hi) / 2.0; x[3] = 0.0;
  }
  const double epsmch = 2.220446049250313e-16;
  const double ftol = 1.49012e-8, xtol = 1.49012e-8, gtol = 0.0, factor = 100.0;
  const int maxfev = 200 * (n + 1);
  auto residuals = [&](const double* p, double* f) {        // curve_fit minimises func(x, *p) - y
    for (int i = 0; i < m; ++i) f[i * S] = hill_value(xt[i * S], p) - yt[i * S];
  };
  double diag[n], qtf[n], wa1[n], wa2[n], wa3[n], r[n * n], sdiag[n];
  int ipvt[n];
  int info = 0, nfev = 1, iter = 1;
  residuals(x, fvec);
  double fnorm = hill_enorm_vec(m, fvec, (int64_t)S);
  double par = 0.0, delta = 0.0, xnorm = 0.0;
  double last_step = 0.0;                                   // |last accepted step| / |x| in the scaled variables (step_out)
  bool done = false;
  while (!done) {
    // ---- fdjac2: forward differences
    {
      const double eps = sqrt(epsmch);                      // epsfcn = machine epsilon
      for (int j = 0; j < n; ++j) {
        const double temp = x[j];
        double hstep = eps * fabs(temp);
        if (hstep == 0.0) hstep = eps;
        x[j] = temp + hstep;
        residuals(x, wa4);
        x[j] = temp;
        for (int i = 0; i < m; ++i) fjac[((size_t)j * mmax + i) * S] = (wa4[i * S] - fvec[i * S]) / hstep;
      }
      nfev += n;
    }
    // ---- qrfac with column pivoting (rdiag -> wa1, acnorm -> wa2, work -> wa3)
    {
      for (int j = 0; j < n; ++j) {
        wa2[j] = hill_enorm_vec(m, fjac + (size_t)j * mmax * S, (int64_t)S);
        wa1[j] = wa2[j];
        wa3[j] = wa1[j];
        ipvt[j] = j;
      }
      for (int j = 0; j < n; ++j) {
        int kmax = j;
        for (int k = j; k < n; ++k)
          if (wa1[k] > wa1[kmax]) kmax = k;
        if (kmax != j) {
          for (int i = 0; i < m; ++i) {
            const double t = fjac[((size_t)j * mmax + i) * S];
            fjac[((size_t)j * mmax + i) * S] = fjac[((size_t)kmax * mmax + i) * S];
            fjac[((size_t)kmax * mmax + i) * S] = t;
          }
          wa1[kmax] = wa1[j];
          wa3[kmax] = wa3[j];
          const int k = ipvt[j]; ipvt[j] = ipvt[kmax]; ipvt[kmax] = k;
        }
        double ajnorm = hill_enorm_vec(m - j, fjac + ((size_t)j * mmax + j) * S, (int64_t)S);
        if (ajnorm != 0.0) {
          if (fjac[((size_t)j * mmax + j) * S] < 0.0) ajnorm = -ajnorm;
          for (int i = j; i < m; ++i) fjac[((size_t)j * mmax + i) * S] /= ajnorm;
          fjac[((size_t)j * mmax + j) * S] += 1.0;
          for (int k = j + 1; k < n; ++k) {
            double sum = 0.0;
            for (int i = j; i < m; ++i) sum += fjac[((size_t)j * mmax + i) * S] * fjac[((size_t)k * mmax + i) * S];
            const double temp = sum / fjac[((size_t)j * mmax + j) * S];
            for (int i = j; i < m; ++i) fjac[((size_t)k * mmax + i) * S] -= temp * fjac[((size_t)j * mmax + i) * S];
            if (wa1[k] != 0.0) {
              double t = fjac[((size_t)k * mmax + j) * S] / wa1[k];
              t = 1.0 - t * t;
              wa1[k] *= sqrt(t > 0.0 ? t : 0.0);
              const double q = wa1[k] / wa3[k];
              if (0.05 * (q * q) <= epsmch) {
                wa1[k] = hill_enorm_vec(m - j - 1, fjac + ((size_t)k * mmax + j + 1) * S, (int64_t)S);
                wa3[k] = wa1[k];
              }
            }
          }
        }
        wa1[j] = -ajnorm;
      }
    }
    if (iter == 1) {
      for (int j = 0; j < n; ++j) {
        diag[j] = wa2[j];
        if (wa2[j] == 0.0) diag[j] = 1.0;
      }
      for (int j = 0; j < n; ++j) wa3[j] = diag[j] * x[j];
      xnorm = hill_enorm(n, wa3, 1);
      delta = factor * xnorm;
      if (delta == 0.0) delta = factor;
    }
    // ---- (q transpose) * fvec, first n components in qtf
    for (int i = 0; i < m; ++i) wa4[i * S] = fvec[i * S];
    for (int j = 0; j < n; ++j) {
      if (fjac[((size_t)j * mmax + j) * S] != 0.0) {
        double sum = 0.0;
        for (int i = j; i < m; ++i) sum += fjac[((size_t)j * mmax + i) * S] * wa4[i * S];
        const double temp = -sum / fjac[((size_t)j * mmax + j) * S];
        for (int i = j; i < m; ++i) wa4[i * S] += fjac[((size_t)j * mmax + i) * S] * temp;
      }
      fjac[((size_t)j * mmax + j) * S] = wa1[j];
      qtf[j] = wa4[j * S];
    }
    // the n x n upper triangle R (column j, rows 0 .. j) in a register-sized copy: r[j * n + i]
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) r[j * n + i] = i <= j ? fjac[((size_t)j * mmax + i) * S] : 0.0;
    // ---- norm of the scaled gradient
    double gnorm = 0.0;
    if (fnorm != 0.0)
      for (int j = 0; j < n; ++j) {
        const int l = ipvt[j];
        if (wa2[l] == 0.0) continue;
        double sum = 0.0;
        for (int i = 0; i <= j; ++i) sum += r[j * n + i] * (qtf[i] / fnorm);
        gnorm = fmax(gnorm, fabs(sum / wa2[l]));
      }
    if (gnorm <= gtol) { info = 4; break; }
    for (int j = 0; j < n; ++j) diag[j] = fmax(diag[j], wa2[j]);
    // ---- inner loop: steps until one is accepted
    for (;;) {
      double rr[n * n];
      for (int k = 0; k < n * n; ++k) rr[k] = r[k];       // lmpar / qrsolv scribble on the lower triangle
      hill_lmpar(rr, ipvt, diag, qtf, delta, &par, wa1, sdiag, wa2, wa3);
      double xnew[n];
      for (int j = 0; j < n; ++j) {
        wa1[j] = -wa1[j];
        xnew[j] = x[j] + wa1[j];
        wa3[j] = diag[j] * wa1[j];
      }
      const double pnorm = hill_enorm(n, wa3, 1);
      if (iter == 1) delta = fmin(delta, pnorm);
      residuals(xnew, wa4);
      ++nfev;
      const double fnorm1 = hill_enorm_vec(m, wa4, (int64_t)S);
      double actred = -1.0;
      if (0.1 * fnorm1 < fnorm) { const double t = fnorm1 / fnorm; actred = 1.0 - t * t; }
      for (int j = 0; j < n; ++j) wa3[j] = 0.0;
      for (int j = 0; j < n; ++j) {
        const double temp = wa1[ipvt[j]];
        for (int i = 0; i <= j; ++i) wa3[i] += r[j * n + i] * temp;
      }
      const double temp1 = hill_enorm(n, wa3, 1) / fnorm;
      const double temp2 = (sqrt(par) * pnorm) / fnorm;
      const double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
      const double dirder = -(temp1 * temp1 + temp2 * temp2);
      const double ratio = prered != 0.0 ? actred / prered : 0.0;
      if (ratio <= 0.25) {
        double temp = actred >= 0.0 ? 0.5 : 0.5 * dirder / (dirder + 0.5 * actred);
        if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
        delta = temp * fmin(delta, pnorm / 0.1);
        par /= temp;
      } else if (par == 0.0 || ratio >= 0.75) {
        delta = pnorm / 0.5;
        par *= 0.5;
      }
      if (ratio >= 1.0e-4) {                                // successful iteration
        for (int j = 0; j < n; ++j) {
          x[j] = xnew[j];
          wa2[j] = diag[j] * x[j];
        }
        for (int i = 0; i < m; ++i) fvec[i * S] = wa4[i * S];
        xnorm = hill_enorm(n, wa2, 1);
        fnorm = fnorm1;
        last_step = xnorm > 0.0 ? pnorm / xnorm : (pnorm == 0.0 ? 0.0 : __longlong_as_double(0x7ff0000000000000LL));   // all scaled parameters zero: never NaN
        ++iter;
      }
      const bool small_red = fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0;
      if (small_red) info = 1;
      if (delta <= xtol * xnorm) info = 2;
      if (small_red && info == 2) info = 3;
      if (info != 0) { done = true; break; }
      if (nfev >= maxfev) info = 5;
      if (fabs(actred) <= epsmch && prered <= epsmch && 0.5 * ratio <= 1.0) info = 6;
      if (delta <= epsmch * xnorm) info = 7;
      if (gnorm <= epsmch) info = 8;
      if (info != 0) { done = true; break; }
      if (ratio >= 1.0e-4) break;                           // next outer iteration: a new Jacobian
    }
  }
  for (int j = 0; j < n; ++j) out[j] = x[j];
  info_out[fit] = info;
  if (nfev_out) nfev_out[fit] = nfev;
  if (step_out) step_out[fit] = last_step;
}

// ---- the same fit by a GROUP of eight lanes -----------------------------------------------------------------------------
// One lane per fit leaves a wave with the serial work of its slowest lane: 8 192 ten-sample windows took 2.3 ms on 128 of the
// chip's 1 024 SIMDs, three quarters of it in pow().  Here eight lanes share a fit: the LEADER runs MINPACK's algorithm
// exactly as hill_fit_kernel does (same operations in the same order: identical parameters, info and nfev -- the tests compare
// the two kernels bit for bit), the whole group evaluates the model -- the 4 m forward differences of a Jacobian or the m
// residuals of a trial step, one pow() each, a pure function of (sample, parameters) whoever computes it.  The wave alternates
// between an evaluation pass (all lanes) and a serial step (leaders) until every group is done; all of a fit's vectors live in
// LDS.  The control flow between the two is wave-uniform (groups in different phases of different fits are predicated), so the
// only cross-lane traffic is LDS ordered by pl_wave_sync().
constexpr int kHillGroup = 8;
constexpr int kHillComm = 16;                               // doubles per group: x[4], h[4], xnew[4], phase
enum { kHillInit = 0, kHillJac = 1, kHillInner = 2, kHillDone = 3 };

__host__ __device__ constexpr size_t hill_group_doubles(int mmax) { return 8 * (size_t)mmax + kHillComm; }

__global__ void __launch_bounds__(PL_WAVE)
hill_fit_group_kernel(const double* __restrict__ xs, const double* __restrict__ ys, const int32_t* __restrict__ lens, int64_t nfits,
                      int mmax, int64_t stride, double* __restrict__ params, int32_t* __restrict__ info_out,
                      int32_t* __restrict__ nfev_out, double* __restrict__ step_out) {
  constexpr int n = kHillN, G = kHillGroup;
  extern __shared__ __attribute__((aligned(16))) unsigned char hill_lds[];
  const int lane = threadIdx.x, grp = lane / G, sub = lane % G;
  const int64_t fit = (int64_t)blockIdx.x * (PL_WAVE / G) + grp;
  const bool leader = sub == 0;
  double* fjac = reinterpret_cast<double*>(hill_lds) + grp * hill_group_doubles(mmax);   // column-major: fjac[j * mmax + i]
  double* fvec = fjac + 4 * (size_t)mmax;
  double* wa4 = fvec + mmax;
  double* xt = wa4 + mmax;
  double* yt = xt + mmax;
  double* comm = yt + mmax;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  const double epsmch = 2.220446049250313e-16;
  const double ftol = 1.49012e-8, xtol = 1.49012e-8, gtol = 0.0, factor = 100.0;
  const int maxfev = 200 * (n + 1);
  const int m = fit < nfits ? (lens ? lens[fit] : mmax) : 0;
  int phase = kHillInit;
  if (fit >= nfits) {
    phase = kHillDone;
  } else if (m < n || m > mmax) {                            // curve_fit raises for fewer samples than parameters
    if (leader) {
      for (int j = 0; j < n; ++j) params[fit * n + j] = nan;
      info_out[fit] = -1;
      if (nfev_out) nfev_out[fit] = 0;
    }
    phase = kHillDone;
  } else {
    const double* xd = xs + fit * stride;
    const double* yd = ys + fit * stride;
    if (leader) comm[12] = 0.0;
    pl_wave_sync();
    bool finite = true;                                      // curve_fit(check_finite=True) raises ValueError for NaN / inf
    for (int i = sub; i < m; i += G) {
      xt[i] = xd[i];
      yt[i] = yd[i];
      finite = finite && fabs(xd[i]) <= 1.7976931348623157e308 && fabs(yd[i]) <= 1.7976931348623157e308;
    }
    if (!finite) comm[12] = 1.0;                             // (any lane of the group: the same value)
  }
  pl_wave_sync();
  if (phase != kHillDone && comm[12] != 0.0) {
    if (leader) {
      for (int j = 0; j < n; ++j) params[fit * n + j] = nan;
      info_out[fit] = -4;
      if (nfev_out) nfev_out[fit] = 0;
    }
    phase = kHillDone;
  }
  pl_wave_sync();
  // p0 = (min(y), max(y), np.median(x), 0): the order statistics of x by counting, the group's lanes sharing the candidates
  if (phase != kHillDone) {
    const int k_hi = m / 2, k_lo = (m & 1) ? k_hi : k_hi - 1;
    for (int a = sub; a < m; a += G) {
      const double va = xt[a];
      int rank = 0;
      for (int b = 0; b < m; ++b) rank += (xt[b] < va || (xt[b] == va && b < a)) ? 1 : 0;
      if (rank == k_lo) comm[4] = va;
      if (rank == k_hi) comm[5] = va;
    }
  }
  pl_wave_sync();
  // the leader's state (hill_fit_kernel's locals)
  double x[n], xnew[n], diag[n], qtf[n], wa1[n], wa2[n], wa3[n], r[n * n], sdiag[n];
  int ipvt[n];
  int info = 0, nfev = 0, iter = 1;
  double fnorm = 0.0, par = 0.0, delta = 0.0, xnorm = 0.0, gnorm = 0.0, pnorm = 0.0, last_step = 0.0;
  if (leader && phase != kHillDone) {
    double mn = yt[0], mx = yt[0];
    for (int i = 1; i < m; ++i) { mn = yt[i] < mn ? yt[i] : mn; mx = yt[i] > mx ? yt[i] : mx; }
    x[0] = mn; x[1] = mx; x[2] = (m & 1) ? comm[5] : (comm[4] + comm[5]) / 2.0; x[3] = 0.0;
    for (int j = 0; j < n; ++j) comm[8 + j] = x[j];          // the first evaluation: the residuals at p0
  }
  pl_wave_sync();

  // the leader's step towards a trial point: lmpar, xnew = x + p, pnorm (hill_fit_kernel's inner loop up to its evaluation)
  auto trial_step = [&]() {
    double rr[n * n];
    for (int k = 0; k < n * n; ++k) rr[k] = r[k];           // lmpar / qrsolv scribble on the lower triangle
    hill_lmpar(rr, ipvt, diag, qtf, delta, &par, wa1, sdiag, wa2, wa3);
    for (int j = 0; j < n; ++j) {
      wa1[j] = -wa1[j];
      xnew[j] = x[j] + wa1[j];
      wa3[j] = diag[j] * wa1[j];
    }
    pnorm = hill_enorm(n, wa3, 1);
    if (iter == 1) delta = fmin(delta, pnorm);
    for (int j = 0; j < n; ++j) comm[8 + j] = xnew[j];
  };
  auto ask_jacobian = [&]() {                                // fdjac2's steps: epsfcn = machine epsilon
    const double eps = sqrt(epsmch);
    for (int j = 0; j < n; ++j) {
      double hstep = eps * fabs(x[j]);
      if (hstep == 0.0) hstep = eps;
      comm[j] = x[j];
      comm[4 + j] = hstep;
    }
  };
  auto finish = [&]() {
    for (int j = 0; j < n; ++j) params[fit * n + j] = x[j];
    info_out[fit] = info;
    if (nfev_out) nfev_out[fit] = nfev;
    if (step_out) step_out[fit] = last_step;
  };

  while (__ballot(phase != kHillDone) != 0ull) {
    // ---- evaluation pass: every lane of the group
    if (phase == kHillJac) {
      for (int e = sub; e < n * m; e += G) {
        const int j = e / m, i = e - j * m;
        const double hstep = comm[4 + j];
        double p[n];
        for (int k = 0; k < n; ++k) p[k] = k == j ? comm[k] + hstep : comm[k];   // x[j] = temp + hstep (a select: j is a lane's own)
        const double v = hill_value(xt[i], p) - yt[i];
        fjac[(size_t)j * mmax + i] = (v - fvec[i]) / hstep;
      }
    } else if (phase != kHillDone) {
      double p[n];
      for (int k = 0; k < n; ++k) p[k] = comm[8 + k];
      for (int i = sub; i < m; i += G) wa4[i] = hill_value(xt[i], p) - yt[i];
    }
    pl_wave_sync();
    // ---- serial step: the leaders
    if (leader && phase != kHillDone) {
      int next = phase;
      bool step = false;                                     // a trial step is due (one call site: lmpar is large)
      if (phase == kHillInit) {
        for (int i = 0; i < m; ++i) fvec[i] = wa4[i];
        nfev = 1;
        fnorm = hill_enorm_vec(m, fvec, 1);
        ask_jacobian();
        next = kHillJac;
      } else if (phase == kHillJac) {
        nfev += n;
        // ---- qrfac with column pivoting (rdiag -> wa1, acnorm -> wa2, work -> wa3)
        for (int j = 0; j < n; ++j) {
          wa2[j] = hill_enorm_vec(m, fjac + (size_t)j * mmax, 1);
          wa1[j] = wa2[j];
          wa3[j] = wa1[j];
          ipvt[j] = j;
        }
        for (int j = 0; j < n; ++j) {
          int kmax = j;
          for (int k = j; k < n; ++k)
            if (wa1[k] > wa1[kmax]) kmax = k;
          if (kmax != j) {
            for (int i = 0; i < m; ++i) {
              const double t = fjac[(size_t)j * mmax + i];
              fjac[(size_t)j * mmax + i] = fjac[(size_t)kmax * mmax + i];
              fjac[(size_t)kmax * mmax + i] = t;
            }
            wa1[kmax] = wa1[j];
            wa3[kmax] = wa3[j];
            const int k = ipvt[j]; ipvt[j] = ipvt[kmax]; ipvt[kmax] = k;
          }
          double ajnorm = hill_enorm_vec(m - j, fjac + (size_t)j * mmax + j, 1);
          if (ajnorm != 0.0) {
            if (fjac[(size_t)j * mmax + j] < 0.0) ajnorm = -ajnorm;
            for (int i = j; i < m; ++i) fjac[(size_t)j * mmax + i] /= ajnorm;
            fjac[(size_t)j * mmax + j] += 1.0;
            for (int k = j + 1; k < n; ++k) {
              double sum = 0.0;
              for (int i = j; i < m; ++i) sum += fjac[(size_t)j * mmax + i] * fjac[(size_t)k * mmax + i];
              const double temp = sum / fjac[(size_t)j * mmax + j];
              for (int i = j; i < m; ++i) fjac[(size_t)k * mmax + i] -= temp * fjac[(size_t)j * mmax + i];
              if (wa1[k] != 0.0) {
                double t = fjac[(size_t)k * mmax + j] / wa1[k];
                t = 1.0 - t * t;
                wa1[k] *= sqrt(t > 0.0 ? t : 0.0);
                const double q = wa1[k] / wa3[k];
                if (0.05 * (q * q) <= epsmch) {
                  wa1[k] = hill_enorm_vec(m - j - 1, fjac + (size_t)k * mmax + j + 1, 1);
                  wa3[k] = wa1[k];
                }
              }
            }
          }
          wa1[j] = -ajnorm;
        }
        if (iter == 1) {
          for (int j = 0; j < n; ++j) {
            diag[j] = wa2[j];
            if (wa2[j] == 0.0) diag[j] = 1.0;
          }
          for (int j = 0; j < n; ++j) wa3[j] = diag[j] * x[j];
          xnorm = hill_enorm(n, wa3, 1);
          delta = factor * xnorm;
          if (delta == 0.0) delta = factor;
        }
        // ---- (q transpose) * fvec, first n components in qtf
        for (int i = 0; i < m; ++i) wa4[i] = fvec[i];
        for (int j = 0; j < n; ++j) {
          if (fjac[(size_t)j * mmax + j] != 0.0) {
            double sum = 0.0;
            for (int i = j; i < m; ++i) sum += fjac[(size_t)j * mmax + i] * wa4[i];
            const double temp = -sum / fjac[(size_t)j * mmax + j];
            for (int i = j; i < m; ++i) wa4[i] += fjac[(size_t)j * mmax + i] * temp;
          }
          fjac[(size_t)j * mmax + j] = wa1[j];
          qtf[j] = wa4[j];
        }
        for (int j = 0; j < n; ++j)
          for (int i = 0; i < n; ++i) r[j * n + i] = i <= j ? fjac[(size_t)j * mmax + i] : 0.0;
        // ---- norm of the scaled gradient
        gnorm = 0.0;
        if (fnorm != 0.0)
          for (int j = 0; j < n; ++j) {
            const int l = ipvt[j];
            if (wa2[l] == 0.0) continue;
            double sum = 0.0;
            for (int i = 0; i <= j; ++i) sum += r[j * n + i] * (qtf[i] / fnorm);
            gnorm = fmax(gnorm, fabs(sum / wa2[l]));
          }
        if (gnorm <= gtol) {
          info = 4;
          finish();
          next = kHillDone;
        } else {
          for (int j = 0; j < n; ++j) diag[j] = fmax(diag[j], wa2[j]);
          step = true;
          next = kHillInner;
        }
      } else {                                               // kHillInner: the trial point's residuals are in wa4
        ++nfev;
        const double fnorm1 = hill_enorm_vec(m, wa4, 1);
        double actred = -1.0;
        if (0.1 * fnorm1 < fnorm) { const double t = fnorm1 / fnorm; actred = 1.0 - t * t; }
        for (int j = 0; j < n; ++j) wa3[j] = 0.0;
        for (int j = 0; j < n; ++j) {
          const double temp = wa1[ipvt[j]];
          for (int i = 0; i <= j; ++i) wa3[i] += r[j * n + i] * temp;
        }
        const double temp1 = hill_enorm(n, wa3, 1) / fnorm;
        const double temp2 = (sqrt(par) * pnorm) / fnorm;
        const double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
        const double dirder = -(temp1 * temp1 + temp2 * temp2);
        const double ratio = prered != 0.0 ? actred / prered : 0.0;
        if (ratio <= 0.25) {
          double temp = actred >= 0.0 ? 0.5 : 0.5 * dirder / (dirder + 0.5 * actred);
          if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
          delta = temp * fmin(delta, pnorm / 0.1);
          par /= temp;
        } else if (par == 0.0 || ratio >= 0.75) {
          delta = pnorm / 0.5;
          par *= 0.5;
        }
        if (ratio >= 1.0e-4) {                               // successful iteration
          for (int j = 0; j < n; ++j) {
            x[j] = xnew[j];
            wa2[j] = diag[j] * x[j];
          }
          for (int i = 0; i < m; ++i) fvec[i] = wa4[i];
          xnorm = hill_enorm(n, wa2, 1);
          fnorm = fnorm1;
          last_step = xnorm > 0.0 ? pnorm / xnorm : (pnorm == 0.0 ? 0.0 : __longlong_as_double(0x7ff0000000000000LL));   // all scaled parameters zero: never NaN
          ++iter;
        }
        const bool small_red = fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0;
        if (small_red) info = 1;
        if (delta <= xtol * xnorm) info = 2;
        if (small_red && info == 2) info = 3;
        if (info == 0) {
          if (nfev >= maxfev) info = 5;
          if (fabs(actred) <= epsmch && prered <= epsmch && 0.5 * ratio <= 1.0) info = 6;
          if (delta <= epsmch * xnorm) info = 7;
          if (gnorm <= epsmch) info = 8;
        }
        if (info != 0) {
          finish();
          next = kHillDone;
        } else if (ratio >= 1.0e-4) {                        // next outer iteration: a new Jacobian
          ask_jacobian();
          next = kHillJac;
        } else {
          step = true;
        }
      }
      if (step) trial_step();
      comm[12] = (double)next;
    }
    pl_wave_sync();
    if (phase != kHillDone) phase = (int)comm[12];
    pl_wave_sync();
  }
}

// first index i in [0, n) with x[i] >= v (np.searchsorted side="left"), n if none
__device__ __forceinline__ int hill_lower_bound(const double* __restrict__ x, int n, double v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (x[mid] < v) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// scipy's linear interp1d with extrapolation (SingleProfile._y_original_to_interp, profile.py:1227-1235)
__device__ __forceinline__ double hill_lookup(const double* __restrict__ xi, const double* __restrict__ v, int S, double q) {
  int hi = hill_lower_bound(xi, S, q);
  hi = hi < 1 ? 1 : (hi > S - 1 ? S - 1 : hi);
  const int lo = hi - 1;
  const double slope = (v[hi] - v[lo]) / (xi[hi] - xi[lo]);
  return slope * (q - xi[lo]) + v[lo];
}

// The two penumbra windows of SingleProfile.inflection_data (profile.py:1676-1700), one workgroup per profile:
//   left_idx  = _x_interp_to_original(first peak of the derivative), right_idx = ...(last valley)
//   half      = int(round(hill_window_ratio * abs(right_idx - left_idx) / 2))
//   x_left    = [x for x in np.arange(left_idx - half, left_idx + half) if x >= 0]
//   x_right   = [x for x in np.arange(right_idx - half, right_idx + half) if x < len(d1)]
//   y         = _y_original_to_interp(x)
// np.arange on floats: length = ceil(stop - start), element 0 = start, element 1 = start + 1.0, element i = start + i * delta
// with delta = element 1 - element 0 (numpy's DOUBLE_fill); the edges come from np.interp's grid-point rule (to_original below).
__global__ void __launch_bounds__(kHillThreads)
hill_windows_kernel(const double* __restrict__ xi, const double* __restrict__ values, int S, const int32_t* __restrict__ pk_count,
                    const int32_t* __restrict__ pk_idx, int cap_p, const int32_t* __restrict__ vl_count,
                    const int32_t* __restrict__ vl_idx, int cap_v, double ratio, int mmax, double* __restrict__ xw,
                    double* __restrict__ yw, int32_t* __restrict__ lens, double* __restrict__ edges) {
  const int64_t p = blockIdx.x;
  const double* v = values + p * (int64_t)S;
  const int np_ = pk_count[p], nv = vl_count[p];
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  if (np_ <= 0 || nv <= 0) {                                // the reference indexes an empty list here (IndexError)
    if (threadIdx.x == 0) {
      lens[2 * p] = lens[2 * p + 1] = 0;
      edges[2 * p] = edges[2 * p + 1] = nan;
    }
    return;
  }
  // _x_interp_to_original: interp1d(arange(S), x_indices) does not extrapolate, so scipy hands it to np.interp, which returns
  // x_indices[k] itself on a grid point (no slope arithmetic: no last-bit difference on resampled grids)
  auto to_original = [&](int k) { return xi[k < 0 ? 0 : (k > S - 1 ? S - 1 : k)]; };
  const double left = to_original(pk_idx[p * cap_p]);
  const double right = to_original(vl_idx[p * cap_v + (nv < cap_v ? nv : cap_v) - 1]);
  const double half = rint(ratio * fabs(right - left) / 2.0);          // python's round(): half to even
  if (threadIdx.x == 0) { edges[2 * p] = left; edges[2 * p + 1] = right; }
  for (int side = 0; side < 2; ++side) {
    const double mid = side ? right : left;
    const double start = mid - half, stop = mid + half;
    const double span = ceil(stop - start);
    int len = span > 0.0 ? (span > 1.0e9 ? 1000000000 : (int)span) : 0;
    const double next = start + 1.0;
    const double delta = next - start;
    auto at = [&](int i) { return i == 0 ? start : (i == 1 ? next : start + (double)i * delta); };
    // the filters keep a suffix (x >= 0) or a prefix (x < S) of the increasing sequence
    int first = 0;
    if (side == 0) {
      while (first < len && !(at(first) >= 0.0)) ++first;
    } else {
      while (len > 0 && !(at(len - 1) < (double)S)) --len;
    }
    int m = len - first;
    if (m > mmax) m = -m;                                   // cannot happen for the mmax the host sizes; reported, not truncated
    const int64_t f = 2 * p + side;
    if (threadIdx.x == 0) lens[f] = m;
    for (int j = threadIdx.x; j < m; j += kHillThreads) {
      const double x = at(first + j);
      xw[f * mmax + j] = x;
      yw[f * mmax + j] = hill_lookup(xi, v, S, x);
    }
  }
}

// Hill.inflection_idx and Hill.y at it (pylinac/core/hill.py:32-36, 56-65)
__global__ void hill_inflection_kernel(const double* __restrict__ params, int64_t nfits, double* __restrict__ out) {
  const int64_t f = (int64_t)blockIdx.x * kHillThreads + threadIdx.x;
  if (f >= nfits) return;
  const double* p = params + f * 4;
  const double idx = p[2] * pow((p[3] - 1.0) / (p[3] + 1.0), 1.0 / p[3]);
  out[2 * f] = idx;
  out[2 * f + 1] = hill_value(idx, p);
}

// SingleProfile.penumbra for the Hill edge method (pylinac/core/profile.py:1852-1898): the positions where the fitted curve
// takes lower / 50 and upper / 50 of its value at the inflection point (Hill.x, hill.py:48-54), their distance, and the
// curve's gradient at the inflection point (Hill.gradient_at, hill.py:38-46).  out[f] = lower index, lower value, upper
// index, upper value, |upper index - lower index|, gradient.
__global__ void hill_penumbra_kernel(const double* __restrict__ params, const double* __restrict__ infl, int64_t nfits,
                                     double lower, double upper, double* __restrict__ out) {
  const int64_t f = (int64_t)blockIdx.x * kHillThreads + threadIdx.x;
  if (f >= nfits) return;
  const double a = params[f * 4], b = params[f * 4 + 1], c = params[f * 4 + 2], d = params[f * 4 + 3];
  const double x0 = infl[2 * f], y0 = infl[2 * f + 1];
  const double lo_v = y0 * lower / 50.0, hi_v = y0 * upper / 50.0;
  auto x_at = [&](double y) { return c * pow((y - a) / (b - y), 1.0 / d); };
  const double lo_i = x_at(lo_v), hi_i = x_at(hi_v);
  const double cxd = pow(c / x0, d);
  double* o = out + f * 6;
  o[0] = lo_i;
  o[1] = lo_v;
  o[2] = hi_i;
  o[3] = hi_v;
  o[4] = fabs(hi_i - lo_i);
  o[5] = (b - a) * d * cxd / (((cxd + 1.0) * (cxd + 1.0)) * x0);
}

// per-profile look-ups values_i(q_ij) through scipy's linear interp1d (SingleProfile._y_original_to_interp)
__global__ void profile_lookup_kernel(const double* __restrict__ xi, const double* __restrict__ values, int S,
                                      const double* __restrict__ q, int nq, int64_t total, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kHillThreads + threadIdx.x;
  if (i >= total) return;
  out[i] = hill_lookup(xi, values + (i / nq) * (int64_t)S, S, q[i]);
}

// SingleProfile._x_interp_to_original for fractional positions: np.interp(q, arange(S), x_indices) (numpy's compiled interp:
// j = the interval with j <= q < j + 1, the grid value itself when q == j, else slope * (q - j) + x_indices[j] with
// slope = (x_indices[j + 1] - x_indices[j]) / ((j + 1) - j); positions outside [0, S - 1] take the end values)
__global__ void index_to_original_kernel(const double* __restrict__ xi, int S, const double* __restrict__ q, int64_t total,
                                         double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kHillThreads + threadIdx.x;
  if (i >= total) return;
  const double x = q[i];
  double r;
  if (x != x) r = x;
  else if (x <= 0.0) r = xi[0];
  else if (x >= (double)(S - 1)) r = xi[S - 1];
  else {
    const int j = (int)floor(x);
    if (x == (double)j) r = xi[j];
    else {
      const double slope = (xi[j + 1] - xi[j]) / ((double)(j + 1) - (double)j);
      r = slope * (x - (double)j) + xi[j];
    }
  }
  out[i] = r;
}

}  // namespace

extern "C" int pl_index_to_original(const double* d_x_indices, int s, const double* d_q, int64_t total, double* d_out,
                                    void* stream) {
  PL_REQUIRE(d_x_indices && d_q && d_out, "null pointer");
  PL_REQUIRE(s >= 2 && total >= 0, "bad shape");
  if (total == 0) return PL_OK;
  PL_REQUIRE(pl_cdiv(total, kHillThreads) <= 0x7fffffffLL, "batch too large");
  hipLaunchKernelGGL(index_to_original_kernel, dim3((unsigned)pl_cdiv(total, kHillThreads)), dim3(kHillThreads), 0,
                     (hipStream_t)stream, d_x_indices, s, d_q, total, d_out);
  return pl_check_launch("pl_index_to_original");
}

static int hill_fit_impl(const double* d_x, const double* d_y, const int32_t* d_lens, int64_t n, int mmax, int64_t stride,
                         double* d_work, double* d_params, int32_t* d_info, int32_t* d_nfev, double* d_last_step, void* stream) {
  PL_REQUIRE(d_x && d_y && d_work && d_params && d_info, "null pointer");
  PL_REQUIRE(n >= 0 && mmax >= 4 && mmax <= 1024 && stride >= mmax, "bad shape (4 .. 1024 samples per fit)");
  if (n == 0) return PL_OK;
  // windows whose vectors fit the LDS eight groups to a wave: the group kernel; longer ones: one lane per fit, global workspace
  // (the group kernel spends eight lanes' issue slots on every serial step: it wins while the batch leaves SIMDs idle -- 1.45
  // against 2.3 ms for 8 192 fits -- and draws level at 32 768 fits, where one lane per fit already fills half the chip)
  const size_t lds = hill_group_doubles(mmax) * sizeof(double) * (PL_WAVE / kHillGroup);
  if (lds <= 64 * 1024 && n <= 32768) {
    const int64_t blocks = pl_cdiv(n, PL_WAVE / kHillGroup);
    PL_REQUIRE(blocks <= 0x7fffffffLL, "batch too large");
    hipLaunchKernelGGL(hill_fit_group_kernel, dim3((unsigned)blocks), dim3(PL_WAVE), lds, (hipStream_t)stream, d_x, d_y, d_lens, n,
                       mmax, stride, d_params, d_info, d_nfev, d_last_step);
    return pl_check_launch("pl_hill_fit");
  }
  PL_REQUIRE(pl_cdiv(n, kHillThreads) <= 0x7fffffffLL, "batch too large");
  hipLaunchKernelGGL(hill_fit_kernel, dim3((unsigned)pl_cdiv(n, kHillThreads)), dim3(kHillThreads), 0, (hipStream_t)stream, d_x, d_y,
                     d_lens, n, mmax, stride, d_work, d_params, d_info, d_nfev, d_last_step);
  return pl_check_launch("pl_hill_fit");
}

extern "C" int pl_hill_fit(const double* d_x, const double* d_y, const int32_t* d_lens, int64_t n, int mmax, int64_t stride,
                           double* d_work, double* d_params, int32_t* d_info, int32_t* d_nfev, void* stream) {
  return hill_fit_impl(d_x, d_y, d_lens, n, mmax, stride, d_work, d_params, d_info, d_nfev, nullptr, stream);
}

/* pl_hill_fit + d_last_step float64 [n]: the length of the LAST ACCEPTED Levenberg-Marquardt step relative to the parameter
 * vector (both in MINPACK's scaled variables).  MINPACK stops on the REDUCTION of the sum of squares (ftol); in a flat valley
 * that happens while the parameters are still moving, and where exactly it happens depends on the last bit of pow(): a fit
 * whose last step is above ~1e-6 reports a point that scipy reproduces to 1e-3, not 1e-5 (NaN: no fit was attempted). */
extern "C" int pl_hill_fit_ex(const double* d_x, const double* d_y, const int32_t* d_lens, int64_t n, int mmax, int64_t stride,
                              double* d_work, double* d_params, int32_t* d_info, int32_t* d_nfev, double* d_last_step, void* stream) {
  PL_REQUIRE(d_last_step, "null pointer");
  return hill_fit_impl(d_x, d_y, d_lens, n, mmax, stride, d_work, d_params, d_info, d_nfev, d_last_step, stream);
}

extern "C" int pl_hill_windows(const double* d_x_indices, const double* d_values, int64_t n, int s, const int32_t* d_peak_count,
                               const int32_t* d_peak_idx, int cap_peaks, const int32_t* d_valley_count,
                               const int32_t* d_valley_idx, int cap_valleys, double window_ratio, int mmax, double* d_xw,
                               double* d_yw, int32_t* d_lens, double* d_edges, void* stream) {
  PL_REQUIRE(d_x_indices && d_values && d_peak_count && d_peak_idx && d_valley_count && d_valley_idx && d_xw && d_yw && d_lens &&
                 d_edges, "null pointer");
  PL_REQUIRE(n >= 0 && s >= 2 && cap_peaks >= 1 && cap_valleys >= 1 && mmax >= 1, "bad shape");
  if (n == 0) return PL_OK;
  PL_REQUIRE(n <= 0x7fffffffLL, "batch too large");
  hipLaunchKernelGGL(hill_windows_kernel, dim3((unsigned)n), dim3(kHillThreads), 0, (hipStream_t)stream, d_x_indices, d_values, s,
                     d_peak_count, d_peak_idx, cap_peaks, d_valley_count, d_valley_idx, cap_valleys, window_ratio, mmax, d_xw, d_yw,
                     d_lens, d_edges);
  return pl_check_launch("pl_hill_windows");
}

extern "C" int pl_hill_inflection(const double* d_params, int64_t n, double* d_out, void* stream) {
  PL_REQUIRE(d_params && d_out, "null pointer");
  PL_REQUIRE(n >= 0, "bad shape");
  if (n == 0) return PL_OK;
  PL_REQUIRE(pl_cdiv(n, kHillThreads) <= 0x7fffffffLL, "batch too large");
  hipLaunchKernelGGL(hill_inflection_kernel, dim3((unsigned)pl_cdiv(n, kHillThreads)), dim3(kHillThreads), 0, (hipStream_t)stream,
                     d_params, n, d_out);
  return pl_check_launch("pl_hill_inflection");
}

extern "C" int pl_profile_lookup(const double* d_x_indices, const double* d_values, int64_t n, int s, const double* d_q, int nq,
                                 double* d_out, void* stream) {
  PL_REQUIRE(d_x_indices && d_values && d_q && d_out, "null pointer");
  PL_REQUIRE(n >= 0 && s >= 2 && nq >= 1, "bad shape");
  if (n == 0) return PL_OK;
  const int64_t total = n * nq;
  PL_REQUIRE(pl_cdiv(total, kHillThreads) <= 0x7fffffffLL, "batch too large");
  hipLaunchKernelGGL(profile_lookup_kernel, dim3((unsigned)pl_cdiv(total, kHillThreads)), dim3(kHillThreads), 0,
                     (hipStream_t)stream, d_x_indices, d_values, s, d_q, nq, total, d_out);
  return pl_check_launch("pl_profile_lookup");
}

extern "C" int pl_hill_penumbra(const double* d_params, const double* d_inflection, int64_t n, double lower, double upper,
                                double* d_out, void* stream) {
  PL_REQUIRE(d_params && d_inflection && d_out, "null pointer");
  PL_REQUIRE(n >= 0 && lower <= upper, "bad arguments (lower <= upper)");
  if (n == 0) return PL_OK;
  PL_REQUIRE(pl_cdiv(n, kHillThreads) <= 0x7fffffffLL, "batch too large");
  hipLaunchKernelGGL(hill_penumbra_kernel, dim3((unsigned)pl_cdiv(n, kHillThreads)), dim3(kHillThreads), 0, (hipStream_t)stream,
                     d_params, d_inflection, n, lower, upper, d_out);
  return pl_check_launch("pl_hill_penumbra");
}
