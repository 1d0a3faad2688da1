// CatPhanBase.find_phantom_axis (pylinac/ct.py:2398-2446) on the device, so that config #5's pass has no mid-pass host
// synchronisation: the per-slice phantom ROI table (pl_edge_regions) -> per volume the centre line that places the CTP528
// circle profiles (pl_circle_profile_combined_ex).
//
// One workgroup per volume: the slices that show the phantom (status 0), np.median of their centres, the
// np.isclose(median, c, atol=3, rtol=0.01) screen on both axes, and a first-order least-squares fit of each centre
// coordinate against the slice number over the slices that pass both -- the closed form about the means, where the
// reference calls np.polyfit (LAPACK gelsd: an SVD whose rounding no device code can reproduce bit for bit).  The fit is
// therefore NOT the reported result: it only places the profiles of the first pass.  ct.ctp528_batch recomputes the exact
// np.polyfit on the host from the same ROI table WHILE the device goes on, and accepts a profile only if the two centres
// differ by less than that profile's decision margin (the distance of any tap from a rounding / bounds decision); otherwise
// that slice is repeated about the exact centre.  The median and the screen ARE exact (order statistics, the same float64
// expression as numpy's).  flag: 0 ok, 1 no slice shows the phantom (the reference raises ValueError), 2 fewer than two
// slices pass the screen.
#include "pl_common.h"

namespace {

__global__ void __launch_bounds__(256)
phantom_axis_fit_kernel(const double* __restrict__ roi, int spv, double x_adj, double y_adj, double* __restrict__ fit,
                        double* __restrict__ centers, int32_t* __restrict__ flag) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ax_smem[];
  double* cx = reinterpret_cast<double*>(ax_smem);
  double* cy = cx + spv;
  int* seen = reinterpret_cast<int*>(cy + spv);
  __shared__ double s_med[4];
  __shared__ int s_m;
  __shared__ double s_fit[4];
  const int tid = threadIdx.x;
  const int64_t v = blockIdx.x;
  const double* r = roi + v * (int64_t)spv * 8;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  if (tid == 0) s_m = 0;
  __syncthreads();
  int mine = 0;
  for (int z = tid; z < spv; z += 256) {
    const int sn = r[z * 8] == 0.0;
    seen[z] = sn;
    cx[z] = r[z * 8 + 4] + x_adj;
    cy[z] = r[z * 8 + 3] + y_adj;
    mine += sn;
  }
  if (mine) atomicAdd(&s_m, mine);
  __syncthreads();
  const int m = s_m;
  if (m == 0) {
    if (tid == 0) {
      flag[v] = 1;
      for (int k = 0; k < 4; ++k) fit[v * 4 + k] = nan;
    }
    for (int z = tid; z < spv; z += 256) {
      centers[(v * spv + z) * 2] = nan;
      centers[(v * spv + z) * 2 + 1] = nan;
    }
    return;
  }
  // np.median: the element(s) of rank (m - 1) / 2 and m / 2 among the seen centres (ties ranked by position)
  const int ra = (m - 1) / 2, rb = m / 2;
  for (int i = tid; i < spv; i += 256) {
    if (!seen[i]) continue;
    const double xi = cx[i], yi = cy[i];
    int rx = 0, ry = 0;
    for (int j = 0; j < spv; ++j) {
      if (!seen[j]) continue;
      rx += (cx[j] < xi) | ((cx[j] == xi) & (j < i));
      ry += (cy[j] < yi) | ((cy[j] == yi) & (j < i));
    }
    if (rx == ra) s_med[0] = xi;
    if (rx == rb) s_med[1] = xi;
    if (ry == ra) s_med[2] = yi;
    if (ry == rb) s_med[3] = yi;
  }
  __syncthreads();
  // np.mean of the one or two middle elements: (a + a) / 2 == a exactly
  const double medx = (s_med[0] + s_med[1]) / 2.0, medy = (s_med[2] + s_med[3]) / 2.0;
  // the np.isclose screen (exact, per slice) and the sums of the fit.  The sums are formed in whatever order the block
  // reduction takes: this line only PLACES the profiles (its centres are verified against np.polyfit through the margins),
  // so its last bits are free -- a serial loop on one lane was 40 us of the pass
  __shared__ double s_red[4][4];
  auto block_sum4 = [&](double (&v)[4]) {                    // -> the four sums in every thread
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = pl_wave_reduce(v[q], [](double a, double b) { return a + b; });
    __syncthreads();
    if ((tid & 63) == 0)
      for (int q = 0; q < 4; ++q) s_red[tid >> 6][q] = v[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (s_red[0][q] + s_red[1][q]) + (s_red[2][q] + s_red[3][q]);
  };
  double a1[4] = {0.0, 0.0, 0.0, 0.0};                       // count, sum z, sum x, sum y
  for (int z = tid; z < spv; z += 256) {
    // np.isclose(a = median, b = c): |a - b| <= atol + rtol * |b|
    const bool ok = seen[z] && fabs(medx - cx[z]) <= 3.0 + 0.01 * fabs(cx[z]) && fabs(medy - cy[z]) <= 3.0 + 0.01 * fabs(cy[z]);
    seen[z] = ok ? 1 : 0;                                    // (each slice is read and written by its own thread only)
    if (ok) { a1[0] += 1.0; a1[1] += (double)z; a1[2] += cx[z]; a1[3] += cy[z]; }
  }
  block_sum4(a1);
  const int k = (int)a1[0];
  const int code = k < 2 ? 2 : 0;
  double f[4] = {nan, nan, nan, nan};
  if (code == 0) {                                           // (block-uniform)
    const double zb = a1[1] / k, xb = a1[2] / k, yb = a1[3] / k;
    double a2[4] = {0.0, 0.0, 0.0, 0.0};                     // szz, szx, szy
    for (int z = tid; z < spv; z += 256) {
      if (!seen[z]) continue;
      const double dz = (double)z - zb;
      a2[0] += dz * dz;
      a2[1] += dz * (cx[z] - xb);
      a2[2] += dz * (cy[z] - yb);
    }
    block_sum4(a2);
    f[0] = a2[1] / a2[0];
    f[1] = xb - f[0] * zb;
    f[2] = a2[2] / a2[0];
    f[3] = yb - f[2] * zb;
  }
  if (tid == 0) {
    for (int q = 0; q < 4; ++q) {
      s_fit[q] = f[q];
      fit[v * 4 + q] = f[q];
    }
    flag[v] = code;
  }
  __syncthreads();
  for (int z = tid; z < spv; z += 256) {                     // np.poly1d(fit)(z) = fit[0] * z + fit[1]
    centers[(v * spv + z) * 2] = s_fit[0] * (double)z + s_fit[1];
    centers[(v * spv + z) * 2 + 1] = s_fit[2] * (double)z + s_fit[3];
  }
}

}  // namespace

extern "C" int pl_phantom_axis_fit(const double* d_roi, int64_t n_volumes, int slices_per_volume, double x_adjustment,
                                   double y_adjustment, double* d_fit, double* d_centers, int32_t* d_flag, void* stream) {
  PL_REQUIRE(d_roi && d_fit && d_centers && d_flag, "null pointer");
  PL_REQUIRE(n_volumes >= 0 && n_volumes <= 0x7fffffffLL && slices_per_volume > 0 && slices_per_volume <= 4096, "bad shape");
  if (n_volumes == 0) return PL_OK;
  const size_t lds = (size_t)slices_per_volume * 20;
  hipLaunchKernelGGL(phantom_axis_fit_kernel, dim3((unsigned)n_volumes), dim3(256), lds, (hipStream_t)stream, d_roi,
                     slices_per_volume, x_adjustment, y_adjustment, d_fit, d_centers, d_flag);
  return pl_check_launch("pl_phantom_axis_fit");
}
