// Batched 1-D peak finding with scipy.signal.find_peaks semantics, plus pylinac's own pre/post
// steps.
//
// Replaces: pylinac.core.profile.find_peaks (pylinac/core/profile.py:2545-2623) and
// _parse_peak_args (:2626-2649), i.e. the call
//     scipy.signal.find_peaks(trimmed, rel_height=1-fwxm_height, width=min_width,
//                             height=threshold, distance=peak_separation, prominence=required)
// followed by "keep the max_number largest by peak_props[peak_sort], re-sorted left to right".
//
// One 256-lane workgroup per profile.  Stages (the order of scipy's filters is preserved):
//   A  min/max of the FULL profile (ratio threshold: height = min + thr*(max-min))
//   B  local maxima (strict rise, plateau -> midpoint (l+r)/2, strict fall; ends never peaks)
//      + height filter, compacted IN ORDER with ballot/popcount scans
//   C  distance filter: priority = height, highest first, processed sequentially by one lane
//      exactly like _select_by_peak_distance (ties: stable order -- scipy's own tie order comes
//      from np.argsort's default introsort and is implementation-defined)
//   D  prominences + bases: one lane per peak walks outwards in LDS
//   E  prominence filter   F  widths at rel_height (+ width filter)
//   G  top-max_number by key (stable, reversed)   H  ordered output compaction
// Float64 throughout, operations in scipy's order, so the float results are bit-identical.
#include <math.h>

#include "pl_common.h"
#include "peaks_device.h"

namespace {

constexpr int kThreads = kPkThreads;

// STAGE = true: the (trimmed) profile lives in LDS and every walk is a ds_read; keeping the two
// cases in separate instantiations lets the compiler know the address space (a runtime select
// between an LDS and a global pointer degrades every access to a slow FLAT load).
template <bool STAGE, int NT>
__global__ void __launch_bounds__(kThreads)
find_peaks_kernel(const double* __restrict__ x, int64_t nprof, int slot_bytes, int len_all, const int32_t* __restrict__ lens,
                  const int32_t* __restrict__ regions, int64_t stride,
                  pl_peak_params prm, int cap, int maxc, int32_t* __restrict__ d_count, int32_t* __restrict__ d_idx,
                  int32_t* __restrict__ d_lb, int32_t* __restrict__ d_rb, double* __restrict__ d_props,
                  int32_t* __restrict__ d_status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  __shared__ Scan scan_a[kThreads / NT];
  __shared__ double s_red_a[kThreads / NT][2 * (kThreads / PL_WAVE)];
  __shared__ int s_cnt_a[kThreads / NT];
  const int slot = threadIdx.x / NT, tid = threadIdx.x % NT;
  const PeakLds L{smem_all + (size_t)slot * slot_bytes, &scan_a[slot], s_red_a[slot], &s_cnt_a[slot]};
  const int64_t prof = (int64_t)blockIdx.x * (kThreads / NT) + slot;
  if (prof >= nprof) return;                     // NT == 64: a whole wave leaves; NT == 256: never taken
  const int len = lens ? lens[prof] : len_all;   // ragged batches: per-profile length
  // search region: the batch's, or this profile's own [lo, hi) (python slice semantics resolved by the caller)
  const int rlo = regions ? regions[2 * prof] : prm.region_lo, rhi = regions ? regions[2 * prof + 1] : prm.region_hi;
  find_peaks_profile<STAGE, NT>(x + prof * stride, len, rlo, rhi, prm, cap, maxc, L, tid, d_count + prof, d_idx + prof * cap,
                                d_lb + prof * cap, d_rb + prof * cap, d_props + prof * 6 * (int64_t)cap, d_status + prof);
}

__global__ void fwxm_record_kernel(const int32_t* __restrict__ count, const int32_t* __restrict__ idx,
                                   const double* __restrict__ props, int cap, int64_t n,
                                   double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fwxm_record_one(count[i], idx + i * cap, props + i * 6 * (int64_t)cap, cap, out + i * 8);
}

// The tail of the EPID pipeline in ONE launch, one workgroup per frame: per-band column sums of the thresholded frame ->
// np.mean(frame, 0) (pylinac/picketfence.py:747-750: float64 sum of integers / count) -> find_peaks on that profile ->
// the FWXM record.  Three launches (pl_colsum_to_mean, pl_find_peaks, pl_fwxm_record: 4 + 25 + 4 us of kernels and the gaps
// between them) and the column-sum memset + 64-bit atomics in front of them become one.
template <bool STAGE>
__global__ void __launch_bounds__(kThreads)
colparts_profile_fwxm_kernel(const uint32_t* __restrict__ parts, int bands, int w, int h, pl_peak_params prm, int cap, int maxc,
                             double* __restrict__ profile, int32_t* __restrict__ d_count, int32_t* __restrict__ d_idx,
                             int32_t* __restrict__ d_lb, int32_t* __restrict__ d_rb, double* __restrict__ d_props,
                             int32_t* __restrict__ d_status, double* __restrict__ fwxm) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  __shared__ Scan scan;
  __shared__ double s_red[2 * (kThreads / PL_WAVE)];
  __shared__ int s_cnt;
  const int64_t frame = blockIdx.x;
  const int tid = threadIdx.x;
  const uint32_t* pf = parts + frame * (int64_t)bands * w;
  double* prof = profile + frame * (int64_t)w;
  for (int i = tid; i < w; i += kThreads) {
    unsigned long long cs = 0;
    for (int b = 0; b < bands; ++b) cs += pf[(size_t)b * w + i];
    prof[i] = (double)cs / (double)h;               // np.mean of integers: float64 sum / count
  }
  __syncthreads();                                  // the profile row (global memory) is this workgroup's own
  const PeakLds L{smem_all, &scan, s_red, &s_cnt};
  find_peaks_profile<STAGE, kThreads>(prof, w, prm.region_lo, prm.region_hi, prm, cap, maxc, L, tid, d_count + frame,
                                      d_idx + frame * cap, d_lb + frame * cap, d_rb + frame * cap,
                                      d_props + frame * 6 * (int64_t)cap, d_status + frame);
  __syncthreads();
  if (tid == 0) fwxm_record_one(d_count[frame], d_idx + frame * cap, d_props + frame * 6 * (int64_t)cap, cap, fwxm + frame * 8);
}

// CTP528CP504.mtf's searches (pylinac/ct.py:1511-1544) for every (profile, line-pair region) pair in ONE launch, a wave per
// pair: the `max_number` most prominent peaks inside the region, and -- when exactly that many were found -- the valleys
// (peaks of the negated profile) between the outermost two of them.  Rounds 1-3 issued sixteen pl_find_peaks_regions
// launches plus the torch glue between them per batch.
constexpr int kPvMaxRegions = 16, kPvCap = 8;
struct PvRegions {
  pl_peak_params pk[kPvMaxRegions], vl[kPvMaxRegions];
  int n;
};

__global__ void __launch_bounds__(kThreads)
peak_valley_kernel(const double* __restrict__ x, int64_t nprof, int64_t stride, int len, PvRegions R, int slot_bytes, int maxc,
                   int cap_p, int cap_v, int32_t* __restrict__ d_pk_count, double* __restrict__ d_pk_height,
                   int32_t* __restrict__ d_vl_count, double* __restrict__ d_vl_value, double* __restrict__ d_means) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  constexpr int NW = kThreads / PL_WAVE;
  __shared__ Scan scan_a[NW];
  __shared__ double s_red_a[NW][2 * NW];
  __shared__ int s_cnt_a[NW];
  __shared__ int32_t o_cnt[NW], o_st[NW], o_idx[NW][kPvCap], o_lb[NW][kPvCap], o_rb[NW][kPvCap];
  __shared__ double o_p[NW][6 * kPvCap];
  const int slot = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / PL_WAVE)), tid = threadIdx.x % PL_WAVE;
  const int64_t unit = (int64_t)blockIdx.x * NW + slot;
  if (unit >= nprof * R.n) return;                     // a whole wave leaves
  const int64_t prof = unit / R.n;
  const int k = (int)(unit - prof * R.n);
  const PeakLds L{smem_all + (size_t)slot * slot_bytes, &scan_a[slot], s_red_a[slot], &s_cnt_a[slot]};
  const double* xp = x + prof * stride;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  const pl_peak_params pk = R.pk[k];
  find_peaks_profile<true, PL_WAVE>(xp, len, pk.region_lo, pk.region_hi, pk, cap_p, maxc, L, tid, &o_cnt[slot], o_idx[slot],
                                    o_lb[slot], o_rb[slot], o_p[slot], &o_st[slot]);
  group_sync<PL_WAVE>();
  const int cnt = o_cnt[slot];
  const int lo = cnt > 0 ? o_idx[slot][0] : 0, hi = cnt > 0 ? o_idx[slot][cnt - 1] : 0;   // peaks leave in index order
  if (tid == 0) d_pk_count[unit] = cnt;
  const double hv = tid < cnt && tid < cap_p ? o_p[slot][tid] : nan;      // the peak heights stay in registers for the mean
  if (tid < cap_p) d_pk_height[unit * cap_p + tid] = hv;
  group_sync<PL_WAVE>();
  int vcnt = 0;
  if (cnt == pk.max_number && cnt > 0) {               // wave-uniform
    const pl_peak_params vl = R.vl[k];
    find_peaks_profile<true, PL_WAVE>(xp, len, lo, hi, vl, cap_v, maxc, L, tid, &o_cnt[slot], o_idx[slot], o_lb[slot], o_rb[slot],
                                      o_p[slot], &o_st[slot], -1.0);
    group_sync<PL_WAVE>();
    vcnt = o_cnt[slot];
  }
  if (tid == 0) d_vl_count[unit] = vcnt;
  if (tid < cap_v) d_vl_value[unit * cap_v + tid] = tid < vcnt ? xp[o_idx[slot][tid]] : nan;   // values[valley_idxs]
  if (d_means) {                                          // wave-uniform
    // max_values.mean() / min_values.mean() (pylinac/ct.py:1530, 1536): np.mean of fewer than eight values is the plain
    // left-to-right float64 sum divided by the count; an empty selection gives NaN.  The peak mean is only meaningful when
    // the region held exactly max_number peaks -- the caller's `break` test -- and is NaN otherwise.
    double pm = nan, vm = nan;
    if (cnt == pk.max_number && cnt > 0) {
      double sum = __shfl(hv, 0, PL_WAVE);
      for (int j = 1; j < cnt; ++j) sum = sum + __shfl(hv, j, PL_WAVE);
      pm = sum / (double)cnt;
      if (vcnt > 0) {
        double vs = xp[o_idx[slot][0]];
        for (int j = 1; j < vcnt; ++j) vs = vs + xp[o_idx[slot][j]];
        vm = vs / (double)vcnt;
      }
    }
    if (tid == 0) {
      d_means[unit * 2] = pm;
      d_means[unit * 2 + 1] = vm;
    }
  }
}

}  // namespace

extern "C" int pl_fwxm_record(const int32_t* d_count, const int32_t* d_idx, const double* d_props,
                              int cap, int64_t n, double* d_out, void* stream) {
  PL_REQUIRE(d_count && d_idx && d_props && d_out, "null pointer");
  PL_REQUIRE(n >= 0 && cap > 0, "bad shape");
  if (n == 0) return PL_OK;
  hipLaunchKernelGGL(fwxm_record_kernel, dim3((unsigned)pl_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     d_count, d_idx, d_props, cap, n, d_out);
  return pl_check_launch("pl_fwxm_record");
}

extern "C" int pl_find_peaks_var(const double* d_x, int64_t n, int len, const int32_t* d_lens, int64_t stride,
                                 const pl_peak_params* params, int cap, int32_t* d_count, int32_t* d_idx,
                                 int32_t* d_left_base, int32_t* d_right_base, double* d_props,
                                 int32_t* d_status, void* stream);
extern "C" int pl_find_peaks_regions(const double* d_x, int64_t n, int len, const int32_t* d_lens, int64_t stride,
                                     const pl_peak_params* params, const int32_t* d_regions, int cap, int32_t* d_count,
                                     int32_t* d_idx, int32_t* d_left_base, int32_t* d_right_base, double* d_props,
                                     int32_t* d_status, void* stream);

extern "C" int pl_find_peaks(const double* d_x, int64_t n, int len, int64_t stride,
                             const pl_peak_params* params, int cap, int32_t* d_count, int32_t* d_idx,
                             int32_t* d_left_base, int32_t* d_right_base, double* d_props,
                             int32_t* d_status, void* stream) {
  return pl_find_peaks_var(d_x, n, len, nullptr, stride, params, cap, d_count, d_idx, d_left_base, d_right_base,
                           d_props, d_status, stream);
}

// ragged form: profile i has d_lens[i] <= len samples (row stride `stride`); the search region of
// `params` is clipped to each profile's own length
extern "C" int pl_find_peaks_var(const double* d_x, int64_t n, int len, const int32_t* d_lens, int64_t stride,
                                 const pl_peak_params* params, int cap, int32_t* d_count, int32_t* d_idx,
                                 int32_t* d_left_base, int32_t* d_right_base, double* d_props,
                                 int32_t* d_status, void* stream) {
  return pl_find_peaks_regions(d_x, n, len, d_lens, stride, params, nullptr, cap, d_count, d_idx, d_left_base,
                               d_right_base, d_props, d_status, stream);
}

// per-profile search regions: d_regions int32[n][2] = [lo, hi) of profile i (NULL: the region of `params`).  The
// reference's per-image loops search each profile where ITS peaks were (CTP528CP504.mtf: valleys between the outermost
// peaks of a line-pair region, pylinac/ct.py:1526-1533).
extern "C" int pl_find_peaks_regions(const double* d_x, int64_t n, int len, const int32_t* d_lens, int64_t stride,
                                     const pl_peak_params* params, const int32_t* d_regions, int cap, int32_t* d_count,
                                     int32_t* d_idx, int32_t* d_left_base, int32_t* d_right_base, double* d_props,
                                     int32_t* d_status, void* stream) {
  PL_REQUIRE(d_x && params && d_count && d_idx && d_left_base && d_right_base && d_props && d_status,
             "null pointer");
  PL_REQUIRE(n >= 0 && n <= 0x7fffffffLL && len > 0 && stride >= len && cap > 0, "bad shape");
  PL_REQUIRE(params->distance >= 1, "distance must be >= 1");
  if (n == 0) return PL_OK;
  int lo = params->region_lo < 0 ? 0 : params->region_lo;
  int hi = params->region_hi > len ? len : params->region_hi;
  int m = hi > lo ? hi - lo : 0;
  if (d_regions) m = len;                      // per-profile regions: size the tables for the longest possible one
  int maxc = m / 2 + 1;
  if (maxc > kMaxCand) maxc = kMaxCand;
  const int stage_x = (m <= kStageMax) ? 1 : 0;
  size_t lds = (size_t)maxc * (8 + 8 + 4 * 4) + 8 + (stage_x ? (size_t)m * 8 : 0);
  lds = (lds + 15) & ~(size_t)15;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)find_peaks_kernel<true, kThreads>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)find_peaks_kernel<false, kThreads>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              150 * 1024);
    if (e != hipSuccess) { pl_set_error("pl_find_peaks: LDS attribute: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
    attr_set = true;
  }
  if (stage_x && m <= kShortMax)
    // short profiles (picket-fence windows, 38 samples each, half a million per batch): one WAVE per profile, four
    // profiles per workgroup, no workgroup barrier anywhere
    hipLaunchKernelGGL((find_peaks_kernel<true, PL_WAVE>), dim3((unsigned)pl_cdiv(n, kThreads / PL_WAVE)), dim3(kThreads),
                       lds * (kThreads / PL_WAVE), (hipStream_t)stream, d_x, n, (int)lds, len, d_lens, d_regions, stride,
                       *params, cap, maxc, d_count, d_idx, d_left_base, d_right_base, d_props, d_status);
  else if (stage_x)
    hipLaunchKernelGGL((find_peaks_kernel<true, kThreads>), dim3((unsigned)n), dim3(kThreads), lds, (hipStream_t)stream, d_x,
                       n, (int)lds, len, d_lens, d_regions, stride, *params, cap, maxc, d_count, d_idx, d_left_base,
                       d_right_base, d_props, d_status);
  else
    hipLaunchKernelGGL((find_peaks_kernel<false, kThreads>), dim3((unsigned)n), dim3(kThreads), lds, (hipStream_t)stream, d_x,
                       n, (int)lds, len, d_lens, d_regions, stride, *params, cap, maxc, d_count, d_idx, d_left_base,
                       d_right_base, d_props, d_status);
  return pl_check_launch("pl_find_peaks");
}

// per-band column sums (pl_median3_threshold_colparts_u16) -> mean profile, its peaks, the FWXM record: see the kernel
extern "C" int pl_colparts_profile_fwxm(const uint32_t* d_parts, int64_t n, int bands, int w, int h, const pl_peak_params* params,
                                        int cap, double* d_profile, int32_t* d_count, int32_t* d_idx, int32_t* d_left_base,
                                        int32_t* d_right_base, double* d_props, int32_t* d_status, double* d_fwxm, void* stream) {
  PL_REQUIRE(d_parts && params && d_profile && d_count && d_idx && d_left_base && d_right_base && d_props && d_status && d_fwxm,
             "null pointer");
  PL_REQUIRE(n >= 0 && n <= 0x7fffffffLL && bands > 0 && w > 0 && h > 0 && cap > 0, "bad shape");
  PL_REQUIRE(params->distance >= 1, "distance must be >= 1");
  if (n == 0) return PL_OK;
  const int lo = params->region_lo < 0 ? 0 : params->region_lo;
  const int hi = params->region_hi > w ? w : params->region_hi;
  const int m = hi > lo ? hi - lo : 0;
  int maxc = m / 2 + 1;
  if (maxc > kMaxCand) maxc = kMaxCand;
  const bool stage_x = m <= kStageMax;
  size_t lds = (size_t)maxc * (8 + 8 + 4 * 4) + 8 + (stage_x ? (size_t)m * 8 : 0);
  lds = (lds + 15) & ~(size_t)15;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)colparts_profile_fwxm_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       150 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)colparts_profile_fwxm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) { pl_set_error("pl_colparts_profile_fwxm: LDS attribute: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
    attr_set = true;
  }
  if (stage_x)
    hipLaunchKernelGGL(colparts_profile_fwxm_kernel<true>, dim3((unsigned)n), dim3(kThreads), lds, (hipStream_t)stream, d_parts, bands,
                       w, h, *params, cap, maxc, d_profile, d_count, d_idx, d_left_base, d_right_base, d_props, d_status, d_fwxm);
  else
    hipLaunchKernelGGL(colparts_profile_fwxm_kernel<false>, dim3((unsigned)n), dim3(kThreads), lds, (hipStream_t)stream, d_parts, bands,
                       w, h, *params, cap, maxc, d_profile, d_count, d_idx, d_left_base, d_right_base, d_props, d_status, d_fwxm);
  return pl_check_launch("pl_colparts_profile_fwxm");
}

/* CTP528CP504.mtf's peak and valley searches for every (profile, region) pair in one launch: see peak_valley_kernel */
extern "C" int pl_peak_valley_regions(const double* d_x, int64_t n, int len, int64_t stride, const pl_peak_params* peak_params,
                                      const pl_peak_params* valley_params, int nregions, int cap_p, int cap_v,
                                      int32_t* d_pk_count, double* d_pk_height, int32_t* d_vl_count, double* d_vl_value,
                                      double* d_means, void* stream) {
  PL_REQUIRE(d_x && peak_params && valley_params && d_pk_count && d_pk_height && d_vl_count && d_vl_value, "null pointer");
  PL_REQUIRE(n >= 0 && len > 0 && stride >= len, "bad shape");
  PL_REQUIRE(nregions >= 1 && nregions <= kPvMaxRegions && cap_p >= 1 && cap_p <= kPvCap && cap_v >= 1 && cap_v <= kPvCap,
             "1..16 regions, capacities 1..8");
  PL_REQUIRE(n * nregions <= 0x7fffffffLL, "batch too large");
  if (n == 0) return PL_OK;
  PvRegions R;
  R.n = nregions;
  int m = 0;
  for (int k = 0; k < nregions; ++k) {
    R.pk[k] = peak_params[k];
    R.vl[k] = valley_params[k];
    PL_REQUIRE(R.pk[k].distance >= 1 && R.vl[k].distance >= 1, "distance must be >= 1");
    PL_REQUIRE(R.pk[k].max_number >= 1 && R.pk[k].max_number <= cap_p, "a peak count of 1..cap_p per region");
    const int lo = R.pk[k].region_lo < 0 ? 0 : R.pk[k].region_lo, hi = R.pk[k].region_hi > len ? len : R.pk[k].region_hi;
    if (hi - lo > m) m = hi - lo;
  }
  PL_REQUIRE(m <= 1024, "search regions of at most 1024 samples (a wave per region)");
  const int maxc = m / 2 + 1;
  size_t lds = (size_t)maxc * (8 + 8 + 4 * 4) + 8 + (size_t)m * 8;
  lds = (lds + 15) & ~(size_t)15;
  const int64_t units = n * nregions;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)peak_valley_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) { pl_set_error("pl_peak_valley_regions: LDS attribute: %s", hipGetErrorString(e)); return PL_ERR_HIP; }
    attr_set = true;
  }
  hipLaunchKernelGGL(peak_valley_kernel, dim3((unsigned)pl_cdiv(units, kThreads / PL_WAVE)), dim3(kThreads),
                     lds * (kThreads / PL_WAVE), (hipStream_t)stream, d_x, n, stride, len, R, (int)lds, maxc, cap_p, cap_v,
                     d_pk_count, d_pk_height, d_vl_count, d_vl_value, d_means);
  return pl_check_launch("pl_peak_valley_regions");
}
