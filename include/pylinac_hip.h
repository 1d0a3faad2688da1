/*
 * pylinac_hip.h -- C ABI of libpylinac_hip.so, the MI355X (gfx950) image-QA compute core.
 *
 * The reference (jrkerns/pylinac v3.46.0) has NO native/FFI boundary: its hot path is numpy glue
 * over scipy.ndimage / scipy.signal / skimage calls.  This header therefore
 * declares the entry points a binding for that path would need, one per third-party call the
 * reference makes; each cites the reference call site it replaces.
 *
 * Conventions
 *   - Batched, stateless, stream-ordered.  All pointers are DEVICE pointers unless the name says
 *     "host".  Frames are [n][h][w] row-major and densely packed.  Nothing is allocated inside;
 *     the caller owns every buffer (PyTorch tensors in the Python host layer).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - Return value: pl_status (0 = PL_OK).  pl_last_error() gives a thread-local message.
 *   - Floating-point device code is compiled with -ffp-contract=off: every kernel reproduces the
 *     reference's float64 operation ORDER, so integer outputs are bit-exact and float outputs are
 *     bit-exact wherever the reference's own order is defined.
 */
#ifndef PYLINAC_HIP_H
#define PYLINAC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PL_ABI_VERSION 3

typedef enum pl_status {
  PL_OK = 0,
  PL_ERR_INVALID_ARG = 1,
  PL_ERR_UNSUPPORTED = 2,
  PL_ERR_HIP = 3
} pl_status;

typedef enum pl_dtype {
  PL_U16 = 0,
  PL_I16 = 1,
  PL_F32 = 2,
  PL_F64 = 3,
  PL_U8 = 4,
  PL_I32 = 5,
  PL_I64 = 6
} pl_dtype;

typedef enum pl_reduce_op { PL_SUM = 0, PL_MEAN = 1, PL_MAX = 2, PL_MIN = 3 } pl_reduce_op;

/* peak_props keys pylinac sorts by: pylinac/core/profile.py:2616-2618 */
typedef enum pl_peak_sort { PL_SORT_PROMINENCES = 0, PL_SORT_PEAK_HEIGHTS = 1, PL_SORT_WIDTHS = 2 } pl_peak_sort;

int pl_abi_version(void);
const char* pl_status_string(int status);
const char* pl_last_error(void);
/* 1 when a HIP device is usable from this process */
int pl_device_available(void);

/* ---- a2: ndimage.gaussian_filter (pylinac/core/array_utils.py:133) ------------------------------
 * One correlate1d pass along `axis` (0 = rows/vertical, 1 = columns/horizontal) of every frame,
 * mode='reflect', scipy's symmetric summation order in float64, result cast into `dtype`
 * (truncation toward zero for integer dtypes).  The 2*radius+1 float64 taps (scipy's _gaussian_kernel1d; the
 * host layer computes them) are passed TWICE: d_weights on the device (read by the float64 kernels) and
 * h_weights in host memory (the packed kernels for 16-bit frames receive them as kernel arguments).
 * h_weights may be NULL: the calls then run the float64 kernels on d_weights (same frames, slower; no
 * device-to-host fetch, no stream synchronisation, legal under stream capture) -- pass both on a hot
 * path.  The caller guarantees that the two copies hold the same taps.  in != out. */
int pl_gaussian1d(const void* in, void* out, int dtype, int64_t n, int h, int w, int axis,
                  const double* d_weights, const double* h_weights, int radius, void* stream);
/* axis 0 into tmp, then axis 1 into out: ndimage.gaussian_filter on a 2-D frame. */
int pl_gaussian2d(const void* in, void* out, void* tmp, int dtype, int64_t n, int h, int w,
                  const double* d_weights, const double* h_weights, int radius, void* stream);

/* ---- a1: ndimage.median_filter(size=s) (pylinac/core/array_utils.py:131) ------------------------
 * s x s window (h > 1) or length-s window (h == 1), mode='reflect', origin 0, rank (s*s)/2. */
int pl_median2d(const void* in, void* out, int dtype, int64_t n, int h, int w, int size,
                void* stream);

/* ---- a5: min/max reductions + ground/normalize/invert (pylinac/core/array_utils.py:63-102) ----- */
int pl_minmax(const void* in, int dtype, int64_t n, int64_t count, double* d_min, double* d_max,
              void* stream);
/* out = a - d_min[i] + value           (same dtype)   array_utils.py:102 */
int pl_ground(const void* in, void* out, int dtype, int64_t n, int64_t count, const double* d_min,
              double value, void* stream);
/* out(f64) = a / d_val[i]                              array_utils.py:70 */
int pl_normalize(const void* in, double* out, int dtype, int64_t n, int64_t count,
                 const double* d_val, void* stream);
/* out = -a + d_max[i] + d_min[i]       (same dtype)   array_utils.py:77 */
int pl_invert(const void* in, void* out, int dtype, int64_t n, int64_t count, const double* d_min,
              const double* d_max, void* stream);
/* np.invert(array) for integer frames (pylinac/core/array_utils.py:80-89): out = ~in in the array's own type
 * (PL_U8 / PL_U16 / PL_I16 / PL_I32 / PL_I64; wider unsigned types travel as their same-width signed bits). */
int pl_bit_invert(const void* in, void* out, int dtype, int64_t total, void* stream);

/* BaseImage.rotate (pylinac/core/image.py:780-783) = skimage.transform.rotate(array, angle, mode="edge"): order-1 warp
 * (order 0 = nearest neighbour with C round(), skimage's default for bool images, no clipping) with the row-major 2x3 inverse map h_matrix (host memory, 6 doubles: c = m0*col + m1*row + m2, r = m3*col + m4*row + m5),
 * edge-clamped neighbours, result clipped to [d_min[i], d_max[i]].  PL_F32 / PL_F64 frames (the host converts integers the
 * way skimage.img_as_float does); in != out. */
int pl_warp_affine(const void* in, void* out, int dtype, int64_t n, int64_t h, int64_t w, int order, const double* h_matrix,
                   const double* d_min, const double* d_max, void* stream);

/* The per-unit record table an analyzer's batch returns (e.g. WLBaseImage.analyze's field centre / BB centroid / status per
 * image, pylinac/winston_lutz.py:709-762): d_out float64 [n][k], out[i][j] = (double)col_j[i * strides[j] + offsets[j]] +
 * adds[j]; d_cols = k device pointers (HOST array of pointers) to float64 (is_int32[j] = 0) or int32 (1) arrays; strides /
 * offsets in elements; 1 <= k <= 16. */
int pl_pack_columns(const void* const* d_cols, const int* is_int32, const int64_t* strides, const int64_t* offsets,
                    const double* adds, int k, int64_t n, double* d_out, void* stream);

/* frame - frame.min() as uint16 where that is exact (every value an integer, no difference beyond max_range <= 65535):
 * the bridge from what the reference's loader may produce -- int16 frames, float64 frames holding integers -- to the
 * uint16 analyzers, whose ground() / normalize() (pylinac/picketfence.py:322-323, pylinac/winston_lutz.py:711-712) only ever
 * see a - min.  dtype PL_I16 / PL_I32 / PL_F64; d_min [n] = the frames' minima (pl_minmax); d_flag int32 [n]: 1 = the frame
 * does not qualify (non-integer value or range beyond max_range; its output is unspecified).  max_range 32767 for int16
 * reproduces where the reference's own int16 ground() stops being exact (array_utils.py:102). */
int pl_to_u16_exact(const void* in, int dtype, int64_t n, int64_t count, const double* d_min, double max_range,
                    uint16_t* out, int32_t* d_flag, void* stream);

/* out = a * factor   (same dtype; the multiply inside stretch(), array_utils.py:168) */
int pl_scale(const void* in, void* out, int dtype, int64_t n, int64_t count, double factor,
             void* stream);
/* out = np.array(in, dtype=T) for float64 in: C conversion through a 64-bit integer then narrowing (negative values wrap
 * into unsigned types), which convert_to_dtype's `relative * range - max - 1` depends on (array_utils.py:171-198). */
int pl_cast_wrap(const double* in, void* out, int dtype, int64_t count, void* stream);

/* ---- a3: BaseImage.threshold / as_binary (pylinac/core/image.py:785-815) ------------------------
 * kind 0 ('high'): out = a >= t ? a : 0 ; kind 1 ('low'): out = a <= t ? a : 0.
 * d_thr holds one float64 threshold per frame (thr_stride 1) or one for all (thr_stride 0). */
int pl_threshold(const void* in, void* out, int dtype, int64_t n, int64_t count,
                 const double* d_thr, int thr_stride, int kind, void* stream);
/* out(u8) = a >= t */
int pl_as_binary(const void* in, uint8_t* out, int dtype, int64_t n, int64_t count,
                 const double* d_thr, int thr_stride, void* stream);

/* ---- a4/a6: per-frame integer histogram, Otsu, order statistics ---------------------------------
 * hist: uint32 [n][65536]; bin b counts value b (PL_U16) or value b-32768 (PL_I16).  Every bin
 * is written (no zero-fill needed). */
int pl_hist16(const void* in, int dtype, int64_t n, int64_t count, uint32_t* d_hist, void* stream);
/* The same pass, which also leaves the largest KEY (value for PL_U16, value + 32768 for PL_I16) of every 512-pixel tile
 * [512 t, 512 t + 512) of every frame in d_tile_max uint16 [n][ceil(count / 512)] -- 0xffff for a tile the kernel did not
 * look at as one unit (frames below 2^18 pixels, unaligned frames, a ragged tail): "look inside".  pl_field_cax_tiles uses
 * it to visit only the tiles that can hold a pixel above the field threshold (pylinac/winston_lutz.py:775-779). */
int pl_hist16_tiles(const void* in, int dtype, int64_t n, int64_t count, uint32_t* d_hist, uint16_t* d_tile_max, void* stream);
/* pl_hist16_tiles and pl_edge_minmax (min / max over the four edge_window-wide edge strips, int32 [n] each) in one launch:
 * everything the per-image half of WLBaseImage.analyze asks of a frame before its scalar decisions
 * (pylinac/winston_lutz.py:709-712, 775, 1109-1133).  d_ranks / d_order_stats (both or neither; as pl_order_stats_from_hist):
 * the order statistics of the np.percentile calls are selected in the same launch from the histogram while it is still in
 * LDS -- d_hist is then SCRATCH (it only receives the bins outside the kernel's LDS windows) and must be 16-byte aligned. */
int pl_hist16_wl(const void* in, int dtype, int64_t n, int h, int w, uint32_t* d_hist, uint16_t* d_tile_max, int edge_window,
                 int32_t* d_edge_min, int32_t* d_edge_max, const int64_t* d_ranks, int nranks, int32_t* d_order_stats, void* stream);
/* skimage.filters.threshold_otsu on an integer image (pylinac/ct.py:3323,3338; acr.py:1409):
 * one bin per integer in [min,max], float64 class statistics, first argmax.  Outputs int32[n]. */
int pl_otsu_from_hist(const uint32_t* d_hist, int dtype, int64_t n, int32_t* d_thr,
                      int32_t* d_min, int32_t* d_max, void* stream);
/* The same threshold straight from the 16-bit frames.  Frames whose values fit a window of 38 912 consecutive bins
 * are read ONCE (+ a 1/16 row sample that places the window): histogram in 152 KiB of LDS, Otsu scan on the LDS bins,
 * no table in HBM.  d_lo / d_hi: optional per-frame bounds (int32[n], d_lo <= values <= d_hi) that place the window
 * instead of the sample; both NULL otherwise.  Frames that do not fit go through pl_hist16 + pl_otsu_from_hist inside
 * the same call (d_hist uint32[n][65536] workspace is touched only for those; d_flag int32[n] scratch says which). */
int pl_otsu16(const void* in, int dtype, int64_t n, int64_t count, const int32_t* d_lo, const int32_t* d_hi,
              int32_t* d_thr, int32_t* d_min, int32_t* d_max, int32_t* d_flag, uint32_t* d_hist, void* stream);
/* pl_otsu16 of the 3x3 MEDIAN of each h x w frame (Image.filter(3, "median") followed by the Otsu threshold,
 * pylinac/core/image.py:695-712 -> array_utils.py:131, then skimage.filters.threshold_otsu as at pylinac/ct.py:3323)
 * WITHOUT writing the median plane: the one-pass kernel computes the medians on the fly.  Needs h > 1, w % 8 == 0, 16-byte
 * aligned frames.  scratch (n*h*w elements of dtype) receives the median plane of ONLY those frames that do not fit the
 * one-pass window (d_flag[i] = 1) -- they go through pl_median2d's kernel, pl_hist16 and pl_otsu_from_hist inside this call. */
int pl_median3_otsu16(const void* in, void* scratch, int dtype, int64_t n, int h, int w, const int32_t* d_lo,
                      const int32_t* d_hi, int32_t* d_thr, int32_t* d_min, int32_t* d_max, int32_t* d_flag,
                      uint32_t* d_hist, void* stream);
/* exact order statistics: out[i][k] = value with 0-based rank d_ranks[k] in frame i
 * (np.percentile call sites: pylinac/core/image.py:899-926, picketfence.py:229-238). */
int pl_order_stats_from_hist(const uint32_t* d_hist, int dtype, int64_t n, const int64_t* d_ranks,
                             int nranks, int32_t* d_out, void* stream);

/* ---- a7: profile extraction, np.mean/np.sum/np.max/np.min(image, axis) --------------------------
 * (pylinac/picketfence.py:747-750, 1513-1514; starshot.py:216-217).  axis 0 -> out[n][w],
 * axis 1 -> out[n][h]; float64 output; integer inputs are summed exactly. */
int pl_reduce_axis(const void* in, int dtype, int64_t n, int h, int w, int axis, int op,
                   double* d_out, void* stream);

/* threshold + axis-0 integer column sums in one pass over a u16 frame (pipeline fusion of
 * image.py:797-800 and picketfence.py:747-750).  d_colsum: uint64 [n][w], zeroed inside. */
int pl_threshold_colsum_u16(const uint16_t* in, uint16_t* out, int64_t n, int h, int w,
                            const int32_t* d_thr, unsigned long long* d_colsum, void* stream);
/* the same with out = threshold(median3(in)): Image.filter(3, "median") (image.py:695-712), Image.threshold (:785-800)
 * and the axis-0 column sums (picketfence.py:747-750) in ONE pass over the unfiltered frame; the median plane is never
 * written.  Needs h > 1, w % 8 == 0, 16-byte aligned frames. */
int pl_median3_threshold_colsum_u16(const uint16_t* in, uint16_t* out, int64_t n, int h, int w,
                                    const int32_t* d_thr, unsigned long long* d_colsum, void* stream);

/* pl_median3_threshold_colsum_u16 with the column sums left as per-band partial sums d_parts uint32[n][bands][w], bands =
 * ceil(h / pl_colparts_band_rows()): plain stores -- no table to zero, no atomics.  pl_colparts_profile_fwxm (below, with
 * pl_fwxm_record) adds the bands up.  The EPID pipeline's third stage. */
int pl_median3_threshold_colparts_u16(const uint16_t* in, uint16_t* out, int64_t n, int h, int w,
                                      const int32_t* d_thr, uint32_t* d_parts, void* stream);
int pl_colparts_band_rows(void);

/* d_out[n][w] = d_colsum[n][w] / h in float64: the np.mean(axis=0) profile of an integer frame. */
int pl_colsum_to_mean(const unsigned long long* d_colsum, int64_t n, int w, int h, double* d_out,
                      void* stream);

/* ---- a12: CircleProfile / CollapsedCircleProfile (pylinac/core/profile.py:2279-2283, 2473-2483) --
 * out[i][s] = ( sum_k img_i[ nearest(sin[s]*r_ik + cy_i), nearest(cos[s]*r_ik + cx_i) ] ) / divisor
 * with ndimage.map_coordinates(order=0, mode='constant', cval=0) sampling (coordinates outside
 * [0, n-1] give 0).  d_cos/d_sin: float64[nsamp] host-computed tables; d_radii: float64[n][nr];
 * d_cx/d_cy: float64[n]; d_out: float64[n][nsamp]. */
int pl_circle_profile(const void* img, int dtype, int64_t n, int h, int w, const double* d_cos,
                      const double* d_sin, int nsamp, const double* d_radii, int nr,
                      const double* d_cx, const double* d_cy, double divisor, double* d_out,
                      void* stream);

/* The same profile of combine_surrounding_slices(z +- plusminus, "max") (pylinac/ct.py:3351-3386, sampled at ct.py:1561-1580)
 * without building the combined slices: d_stack [n_stack][h][w] holds whole volumes of slices_per_volume slices, profile i
 * belongs to slice d_slice_index[i] (int64 [m]; NULL = slice i, m == n_stack), every tap is the maximum over the slice's
 * neighbours inside its own volume with pl_combine_slices' indexing (negative indices wrap, indices past the end reuse
 * the last slice).  Integer slices; d_radii [m][nr], d_cx / d_cy [m], d_out [m][nsamp]. */
int pl_circle_profile_combined(const void* stack, int dtype, int64_t n_stack, int h, int w, const int64_t* d_slice_index,
                               int64_t m, int64_t slices_per_volume, int plusminus, const double* d_cos,
                               const double* d_sin, int nsamp, const double* d_radii, int nr, const double* d_cx,
                               const double* d_cy, double divisor, double* d_out, void* stream);

/* pl_circle_profile_combined + d_margin float64 [m] (may be NULL; the caller presets it to +inf): per profile the smallest
 * distance of any tap's coordinate + 0.5 to an integer (the nearest-pixel choice of map_coordinates(order=0)) or of a
 * coordinate to 0 / n - 1 (its bounds test) -- how far the centre may move before any sample of the profile changes.  Lets
 * ct.ctp528_batch place the profiles about a centre fitted on the device (pl_phantom_axis_fit) and prove afterwards that
 * the reference's np.polyfit centre (pylinac/ct.py:2440-2445) selects the same pixels. */
int pl_circle_profile_combined_ex(const void* stack, int dtype, int64_t n_stack, int h, int w,
                                  const int64_t* d_slice_index, int64_t m, int64_t slices_per_volume, int plusminus,
                                  const double* d_cos, const double* d_sin, int nsamp, const double* d_radii, int nr,
                                  const double* d_cx, const double* d_cy, double divisor, double* d_out,
                                  double* d_margin, void* stream);

/* pl_circle_profile_combined_ex for a THIN ring (CTP528CP504.circle_profile, pylinac/ct.py:1559-1580: 20 radii within +-4 % of
 * the line-pair radius): the caller promises r_lo <= |radius| <= r_hi for every radius of every profile.  One workgroup per
 * profile copies the annulus of the ring's bounding box -- per row the one or two chords, the maximum over the
 * 2 * plusminus + 1 slices formed on the way -- into LDS with row-contiguous loads and takes the taps from there: the 172 000
 * scattered gathers of a CTP528 profile become 45 000 contiguous loads; consecutive profiles run on the same XCD, whose L2
 * then serves the 2 * plusminus slices neighbours share.  Same samples, same margins.  A tap whose pixel is not inside a staged
 * chord is fetched from the slices themselves (a wrong promise costs time, never a sample); a ring whose annulus exceeds 64 KB
 * of LDS, or plusminus > 3, takes pl_circle_profile_combined_ex. */
int pl_circle_profile_ring(const void* stack, int dtype, int64_t n_stack, int h, int w, const int64_t* d_slice_index,
                           int64_t m, int64_t slices_per_volume, int plusminus, const double* d_cos, const double* d_sin,
                           int nsamp, const double* d_radii, int nr, const double* d_cx, const double* d_cy, double divisor,
                           double r_lo, double r_hi, double* d_out, double* d_margin, void* stream);

/* CatPhanBase.find_phantom_axis (pylinac/ct.py:2398-2446) for the phantom-ROI tables d_roi [n_volumes * spv][8] of
 * pl_edge_regions, one volume after the other: slices with status 0, np.median of their centres, the np.isclose(median, c,
 * atol=3, rtol=0.01) screen on both axes (both exact), then a closed-form first-order least-squares fit of centre against
 * slice number -- a PLACEMENT aid, not the reported fit (the host's np.polyfit is; see pl_circle_profile_combined_ex).
 * d_fit [n_volumes][4] = zx slope, zx intercept, zy slope, zy intercept; d_centers [n_volumes * spv][2] = (x, y) of the fitted
 * line at every slice; d_flag int32 [n_volumes]: 0 ok, 1 no slice shows the phantom, 2 fewer than two slices pass. */
int pl_phantom_axis_fit(const double* d_roi, int64_t n_volumes, int slices_per_volume, double x_adjustment,
                        double y_adjustment, double* d_fit, double* d_centers, int32_t* d_flag, void* stream);

/* ---- a15: ndimage.sobel(image, axis) (pylinac/core/image.py:1006-1007): [-1,0,1] along `axis`
 * then [1,2,1] along the other axis, mode='reflect', each pass cast into the image dtype. */
int pl_sobel(const void* in, void* out, int dtype, int64_t n, int h, int w, int axis, void* stream);

/* ---- a13/a14: connected components, hole filling, binary centroid ------------------------------
 * Masks are uint8 [n][h][w] (non-zero = foreground).  d_work: int32 [n][h][w] scratch owned by the
 * caller (union-find forest); d_flags: uint8 [n][h][w] scratch.
 * pl_label: skimage.measure.label(mask, connectivity) / scipy.ndimage.label numbering (raster order
 *   of each component's first pixel; pylinac/metrics/utils.py:131, pylinac/ct.py:3345);
 *   connectivity 4 or 8; d_nlabels int32[n] (may be NULL).
 * pl_fill_holes: scipy.ndimage.binary_fill_holes (pylinac/winston_lutz.py:777); connectivity_bg is the
 *   BACKGROUND connectivity: 4 for scipy's default structure, 8 for skimage's filled_area
 *   (structure ones((3,3))).
 * pl_binary_centroid: scipy.ndimage.center_of_mass of a mask (winston_lutz.py:778):
 *   d_out float64 [n][3] = row, col, count;  d_sums: uint64 [n][3] scratch.
 * pl_scaled_binary: the threshold_img of find_field_centroids on a ground()+normalize()d frame
 *   without materialising the float64 frame: out = ((a - sub_i) / div_i) >= thr_i  (float64). */
int pl_label(const uint8_t* d_mask, int64_t n, int h, int w, int connectivity, int32_t* d_labels,
             int32_t* d_work, int32_t* d_nlabels, void* stream);
int pl_fill_holes(const uint8_t* d_mask, uint8_t* d_out, int64_t n, int h, int w, int connectivity_bg,
                  int32_t* d_work, uint8_t* d_flags, void* stream);
int pl_binary_centroid(const uint8_t* d_mask, int64_t n, int h, int w, unsigned long long* d_sums,
                       double* d_out, void* stream);
int pl_scaled_binary(const void* in, int dtype, int64_t n, int64_t count, const double* d_sub,
                     const double* d_div, const double* d_thr, uint8_t* d_out, void* stream);
/* ndimage.center_of_mass(ndimage.binary_fill_holes(((a - sub) / div) >= thr)) per frame (pylinac/winston_lutz.py:775-779)
 * WITHOUT materialising the mask or a label plane: one streaming pass reduces the foreground's count, coordinate sums
 * and bounding box, then one workgroup per frame flood-fills the background of the box (grown by one pixel) in LDS and
 * adds the holes.  d_out float64[n][3] = (row, col, filled pixel count); d_acc uint64[n][8] scratch; d_status int32[n]:
 * 0 done, 1 = the bounding box exceeds the 384 x 384 LDS window: use pl_scaled_binary -> pl_fill_holes ->
 * pl_binary_centroid for that frame. */
int pl_field_cax(const void* in, int dtype, int64_t n, int h, int w, const double* d_sub, const double* d_div,
                 const double* d_thr, unsigned long long* d_acc, double* d_out, int32_t* d_status, void* stream);
/* The same with the streaming pass driven by pl_hist16_tiles' d_tile_max (uint16 [n][h * w / 512]): only tiles whose largest
 * value can pass the threshold are read.  Same results; frames that are not 16-bit, not a whole number of 512-pixel tiles,
 * not 8-pixel aligned, take pl_field_cax's full pass. */
int pl_field_cax_tiles(const void* in, int dtype, int64_t n, int h, int w, const double* d_sub, const double* d_div,
                       const double* d_thr, const uint16_t* d_tile_max, unsigned long long* d_acc, double* d_out,
                       int32_t* d_status, void* stream);
/* min / max over the four `window`-pixel-wide edge strips of every 16-bit frame (int32[n] each): the edge test of
 * WLBaseImage._clean_edges (pylinac/winston_lutz.py:1109-1133). */
int pl_edge_minmax(const void* in, int dtype, int64_t n, int h, int w, int window, int32_t* d_min, int32_t* d_max,
                   void* stream);
/* The scalar decisions WLBaseImage.analyze takes per image, from the frame's exact order statistics, WITHOUT a host round
 * trip: check_inversion_by_histogram((0.01, 50, 99.99)) (pylinac/core/image.py:899-926), the edge test of _clean_edges
 * (pylinac/winston_lutz.py:1109-1133: percentiles 5 / 99.5 against the edge strips' extrema), and ground() / normalize() +
 * the field threshold (p99.9 - p5) / 2 + p5 of find_field_centroids (pylinac/winston_lutz.py:711-712, 775-776), all in numpy's
 * float64 operation order (np.percentile's _lerp).  d_stats int32[n][16] = min, max, the LOWER order-statistic neighbour of
 * the percentiles (5, 99.9, 0.01, 50, 99.99, 5, 99.5), then their UPPER neighbours; h_frac[7] (host) = the interpolation
 * weights of those percentiles (the fractional part of q / 100 * (count - 1)).  Outputs (device): d_inverted / d_noisy
 * int32[n], d_vmin / d_vmax / d_gmax (= max - min) / d_thr float64[n]. */
int pl_wl_decisions(const int32_t* d_stats, const int32_t* d_edge_min, const int32_t* d_edge_max, int64_t n,
                    const double* h_frac, int32_t* d_inverted, int32_t* d_noisy, double* d_vmin, double* d_vmax,
                    double* d_gmax, double* d_thr, void* stream);

/* ---- a16: CatPhan slice localisation (pylinac/ct.py:381-425, 3315-3348) --------------------------
 * pl_scharr: skimage.filters.scharr(float image) -> float64 edge magnitude.
 * pl_gaussian2d_mode: ndimage.gaussian_filter with border mode 0 'reflect' / 1 'nearest' / 2 'constant' (cval 0),
 *   (skimage.filters.gaussian uses 'nearest').
 * pl_clip: np.clip.   pl_compare: op 0 >=, 1 >, 2 <=, 3 < against per-frame / broadcast thresholds.
 * pl_hist_uniform: np.histogram(values[mask], bins=nbins) with caller-supplied float64 edges
 *   [n][nbins+1] (np.linspace(min, max, nbins+1)); d_mask uint8[count] shared by all frames or NULL;
 *   d_counts uint32[n][nbins].
 * pl_clear_border: skimage.segmentation.clear_border(bw, buffer_size) (8-connected labelling).
 * pl_region_stats: regionprops raw sums, float64 [n][max_labels][10] = area, bbox(r0,c0,r1,c1),
 *   sum r, sum c, sum w, sum w*r, sum w*c;  d_isum uint64[n][max_labels][7], d_wsum float64
 *   [n][max_labels][3] scratch; d_overflow int32[n] set when a label exceeds max_labels. */
int pl_scharr(const void* in, double* out, int dtype, int64_t n, int h, int w, void* stream);
int pl_gaussian2d_mode(const void* in, void* out, void* tmp, int dtype, int64_t n, int h, int w,
                       const double* d_weights, int radius, int mode, void* stream);
/* min/max over the pixels selected by a frame-shared uint8 mask (edges[rr, cc] of the disk) */
int pl_minmax_masked(const double* in, const uint8_t* d_mask, int64_t n, int64_t count, double* d_min,
                     double* d_max, void* stream);
int pl_clip(const void* in, void* out, int dtype, int64_t n, int64_t count, double lo, double hi,
            void* stream);
int pl_hist_uniform(const double* in, const uint8_t* d_mask, int64_t n, int64_t count,
                    const double* d_edges, int nbins, uint32_t* d_counts, void* stream);
int pl_compare(const void* in, int dtype, int64_t n, int64_t count, const double* d_thr, int thr_stride,
               int op, uint8_t* d_out, void* stream);
int pl_clear_border(const uint8_t* d_mask, uint8_t* d_out, int64_t n, int h, int w, int buffer_size,
                    int32_t* d_work, uint8_t* d_flags, void* stream);
int pl_region_stats(const int32_t* d_labels, const double* d_intensity, int64_t n, int h, int w,
                    int max_labels, unsigned long long* d_isum, double* d_wsum, double* d_stats,
                    int32_t* d_overflow, void* stream);
/* pl_region_moments: exact raw moments per label, uint64 [n][max_labels][6] = m00, m10 (sum r), m01 (sum c),
 *   m20, m02, m11 in image coordinates: what skimage 0.18.3's regionprops.moments_central / inertia_tensor /
 *   orientation / eccentricity are formed from (measure/_regionprops.py:318-322, 394-420; called at
 *   pylinac/planar_imaging.py:2348, pylinac/ct.py:2522-2563).  The host forms the central moments in exact
 *   integer arithmetic, so symmetric regions decide `a - c == 0` exactly. */
int pl_region_moments(const int32_t* d_labels, int64_t n, int h, int w, int max_labels,
                      unsigned long long* d_mom, int32_t* d_overflow, void* stream);
/* np.linspace(lo_i, hi_i, nbins + 1) per frame -> d_edges float64 [n][nbins + 1]: the bin edges np.histogram builds for
 * `bins = nbins` over the range of the selected pixels (feeds pl_hist_uniform without a host round trip). */
int pl_linspace_edges(const double* d_lo, const double* d_hi, int nbins, int64_t n, double* d_edges, void* stream);
/* skimage 0.18.3 threshold_otsu from a 256-bin float histogram (counts uint32 [n][256], edges float64 [n][257]):
 * d_thr[i] = otsu_i * scale (pylinac/ct.py:3338-3340 uses 0.8), d_raw[i] = otsu_i (optional, may be NULL). */
int pl_otsu_from_counts(const uint32_t* d_counts, const double* d_edges, int nbins, int64_t n, double scale,
                        double* d_thr, double* d_raw, void* stream);
/* The edge-image half of the slice localisation as one streaming pass (csrc/edge_stream.hip; pylinac/ct.py:391-392,
 * 3327-3338): ndimage.gaussian_filter(skimage.filters.scharr(frame.astype(float)), mode='nearest') with the taps d_weights
 * (device float64 [2 * radius + 1], radius 1..8), every float64 value bit-identical to pl_scharr + pl_gaussian2d_mode(mode 1).
 *   d_out (may be NULL: extrema only) = the plane as out_dtype PL_F64 [n][h][w], or PL_F32 = RN of the float64 value (1 MiB
 *   per 512 x 512 slice instead of 2; pl_edge_otsu / pl_edge_regions decide on it and fall back to the exact value where the
 *   rounding could matter);  d_rawmax[i] = max of the Scharr magnitude itself (the "no edges" test np.max(edges) < 0.1);
 *   d_min / d_max[i] = exact float64 extrema of the smoothed plane over the selected pixels: d_row_spans (int32 [h][2]:
 *   columns [c0, c1) of each row, shared by all frames -- a disk) or d_mask (uint8 [h][w], shared), or every pixel when both
 *   are NULL (+inf / -inf when nothing is selected).  int16 / uint16 frames. */
int pl_edge_plane(const void* in, int dtype, int64_t n, int h, int w, const double* d_weights, int radius,
                  const int32_t* d_row_spans, const uint8_t* d_mask, void* d_out, int out_dtype, double* d_rawmax,
                  double* d_min, double* d_max, void* stream);
/* skimage.filters.threshold_otsu(edges[selection]) for the plane pl_edge_plane wrote (pylinac/ct.py:3334-3340), one launch:
 * np.histogram's 256 bins over [d_min[i], d_max[i]] (np.linspace edges, numpy's edge-corrected binning) and skimage 0.18.3's
 * class statistics -> d_thr[i] = otsu_i * scale (0.8 in the reference), d_raw_otsu[i] = otsu_i (optional).  d_min / d_max
 * must be the EXACT extrema of the selection (pl_edge_plane's).  On a PL_F32 plane a pixel whose float32 neighbours fall into
 * different bins is recomputed exactly from the slices in_raw (dtype PL_I16 / PL_U16) with the taps d_weights; a PL_F64
 * plane needs neither (in_raw may be NULL).  d_work: uint32 [n][258] scratch (zeroed here; afterwards [i][0..255] = the
 * histogram, [i][257] = how many pixels were recomputed exactly).  Empty selection -> NaN, constant selection -> that value. */
int pl_edge_otsu(const void* d_plane, int plane_dtype, const void* in_raw, int dtype, int64_t n, int h, int w,
                 const double* d_weights, int radius, const int32_t* d_row_spans, const uint8_t* d_mask, const double* d_min,
                 const double* d_max, double scale, uint32_t* d_work, double* d_thr, double* d_raw_otsu, void* stream);
/* round 3's form of the same pass: float64 plane, byte mask (= pl_edge_plane(..., NULL, d_mask, d_out, PL_F64, ...)) */
int pl_scharr_gaussian(const void* in, int dtype, int64_t n, int h, int w, const double* d_weights, int radius,
                       const uint8_t* d_mask, double* d_out, double* d_rawmax, double* d_min, double* d_max, void* stream);
/* The labelling half of get_regions (pylinac/ct.py:3340-3347) for every frame in ONE launch, a workgroup per frame with the
 * mask as a bit plane in LDS and components as sets of row runs (csrc/slice_regions.hip):
 *   bw = in > d_thr[i]  (dtype PL_F64; NaN compares false)   or   bw = in != 0  (dtype PL_U8, d_thr NULL);
 *   clear_border_ext > 0: skimage.segmentation.clear_border(bw, buffer_size = clear_border_ext - 1) (8-connected);
 *   fill_holes != 0: scipy.ndimage.binary_fill_holes(bw) (default structure: 4-connected background);
 *   measure.label(bw) (8-connected, scikit-image's raster-order numbering) and per label k < max_labels
 *   d_table float64 [n][max_labels][7] = area, bbox r0, c0, r1, c1 (half-open), sum of rows, sum of columns (exact
 *   integers); d_count int32[n] = number of labels (may exceed max_labels: the table then holds the first max_labels);
 *   d_status int32[n]: 0, or 1 = more row runs than the LDS list holds -- repeat that frame with pl_compare /
 *   pl_clear_border / pl_fill_holes / pl_label / pl_region_stats.  d_out_mask (optional uint8 [n][h][w]) receives the
 *   final mask.  pl_mask_regions_fits(h, w, max_labels) != 0 says whether the shape fits the LDS form at all. */
int pl_mask_regions_fits(int h, int w, int max_labels);
int pl_mask_regions(const void* in, int dtype, const double* d_thr, int64_t n, int h, int w, int clear_border_ext,
                    int fill_holes, int max_labels, double* d_table, int32_t* d_count, int32_t* d_status,
                    uint8_t* d_out_mask, void* stream);
/* pl_mask_regions on the float32 plane of pl_edge_plane (bw = plane > d_thr[i], decided on the float32 value where its two
 * float32 neighbours agree and recomputed exactly from the slices in_raw / taps d_weights otherwise), and -- when d_roi is
 * given -- Slice.phantom_roi's choice (pylinac/ct.py:381-425) in the same launch: d_roi float64 [n][8] = status, label,
 * filled_area, centroid row, centroid col, bbox r0, c0, r1 of the region whose area is closest to catphan_size (first on
 * ties) -- status 0 ok, 1 no edges (d_rawmax[i] < 0.1), 2 no region, 3 not within a factor 1.3 of catphan_size, 4 more than
 * max_labels regions, 5 run list overflow (d_status[i] = 1: repeat the slice on the general path); columns 1-7 are NaN unless
 * status is 0.  d_table may be NULL. */
int pl_edge_regions(const float* d_plane, const void* in_raw, int dtype, const double* d_weights, int radius,
                    const double* d_thr, int64_t n, int h, int w, int clear_border_ext, int fill_holes, int max_labels,
                    double* d_table, int32_t* d_count, int32_t* d_status, uint8_t* d_out_mask, double catphan_size,
                    const double* d_rawmax, double* d_roi, void* stream);

/* pl_edge_plane in PACKED FLOAT32 (round 6; csrc/edge_stream32.hip): the same smoothed Scharr plane (pylinac/ct.py:391,
 * 3327-3328) at less than half the vector instructions.  d_out [n][h][w] float32 lies within pl_edge_plane32_bracket() bit
 * patterns of the exact float64 value (a stored 0 is exact) -- for consumers that decide from it and recompute exactly what
 * they cannot decide: pl_edge_otsu_ex / pl_edge_regions_ex with that bracket.  d_min / d_max [n]: the EXACT float64 extrema over
 * the row spans (candidates within two brackets of the float32 extrema, recomputed from the slices in scipy's float64
 * sequence); d_rawmax [n]: max of the raw Scharr magnitude, exact below 32 (it only feeds the "no edges" test < 0.1).
 * d_status int32 [n]: 1 = the extrema of this slice could not be certified (two candidates in one lane, or a full list):
 * the caller repeats the slice with pl_edge_plane.  Even widths; d_work: pl_edge_plane32_work_bytes(n) bytes, 16-byte aligned. */
int pl_edge_plane32_bracket(void);
int64_t pl_edge_plane32_work_bytes(int64_t n);
int pl_edge_plane32(const void* in, int dtype, int64_t n, int h, int w, const double* d_weights, int radius,
                    const int32_t* d_row_spans, float* d_out, unsigned char* d_work, double* d_rawmax, double* d_min,
                    double* d_max, int32_t* d_status, void* stream);

/* pl_edge_otsu / pl_edge_regions on a float32 plane that lies within `bracket` bit patterns of the exact value instead of being
 * its correctly rounded image (bracket 1): the plane of pl_edge_plane32.  Every decision -- histogram bin, threshold -- is taken
 * from the plane where the bracket allows and recomputed exactly from the slices otherwise, so the results are those of the
 * exact plane. */
int pl_edge_otsu_ex(const void* d_plane, int plane_dtype, const void* in_raw, int dtype, int64_t n, int h, int w,
                    const double* d_weights, int radius, const int32_t* d_row_spans, const uint8_t* d_mask,
                    const double* d_min, const double* d_max, double scale, uint32_t* d_work, double* d_thr,
                    double* d_raw_otsu, int bracket, void* stream);
int pl_edge_regions_ex(const float* d_plane, const void* in_raw, int dtype, const double* d_weights, int radius,
                       const double* d_thr, int64_t n, int h, int w, int clear_border_ext, int fill_holes, int max_labels,
                       double* d_table, int32_t* d_count, int32_t* d_status, uint8_t* d_out_mask, double catphan_size,
                       const double* d_rawmax, double* d_roi, int bracket, void* stream);
/* combine_surrounding_slices (pylinac/ct.py:3351-3386) for EVERY slice of a stack [n][count] made of whole volumes of
 * slices_per_volume slices: mode 0 = np.max (d_out has the input dtype), mode 1 = np.mean (d_out float64).  The window
 * z-k .. z+k indexes the slice's own volume the way the reference indexes its Python list: a negative index wraps around
 * to the end of the volume; an index beyond the last slice raises IndexError in the reference -- such slices reuse the last
 * slice here and the caller reports them as invalid. */
int pl_combine_slices(const void* in, void* d_out, int dtype, int64_t n, int64_t count, int plusminus, int mode,
                      int64_t slices_per_volume, void* stream);

/* ---- a13: one threshold level of find_features (pylinac/metrics/utils.py:128-180 + features.py) ---
 * Inputs per window i: d_sample float64 [n][h][w] (the stretched sample), the 4-connected label image of
 * `sample > cutoff` (pl_label), its label count and its region table (pl_region_stats, max_labels rows).
 * Every region that does not touch the frame and passes is_right_size_bb / is_round /
 * is_right_circumference / is_symmetric / is_solid appends its weighted centroid (x, y) to
 * d_xy float64 [n][8][2] unless it lies within min_sep_px of a point accepted before it (earlier levels
 * AND earlier regions of this level, in label order: metrics/utils.py:28-36 iterates the list it appends to);
 * d_count int32[n] (zeroed by the caller before level 0),
 * d_level int32[n] (initialised to -1: first level with a hit), d_done int32[n] (zeroed; set once
 * d_count >= max_number, later calls skip the window), d_status int32[n] (0 ok; 1 label table
 * overflow, 2 > 32 candidate regions, 3 region bbox > 160 px, 4 > 8 features). */
int pl_features_level(const double* d_sample, const int32_t* d_labels, const int32_t* d_nlabels,
                      const double* d_stats, int max_labels, int64_t n, int h, int w, double dpmm,
                      double radius_mm, double tol_mm, double min_sep_px, int max_number, int level,
                      int32_t* d_done, int32_t* d_count, double* d_xy, int32_t* d_level, int32_t* d_status,
                      void* stream);
/* The WHOLE threshold sweep of find_features (BB mode; pylinac/metrics/utils.py:120-181) for n windows of h x w float64
 * samples already stretched to [0, 1], one workgroup per window with the window resident in LDS (level map, row runs,
 * union-find over run ids, per-candidate crop analysis).  h_cutoffs: HOST memory, nlevels <= 64 increasing values
 * (imin + step, imin + 2 step, ... accumulated like the reference).  The sweep stops at the first level that completes
 * max_number features.  Outputs as pl_features_level; d_status: 0 ok, 2 / 4 as there, 3 = a candidate's bbox exceeds
 * 64 pixels, 5 = a level has more than 4096 row runs -- for 3 and 5 the caller uses the level-by-level entry points.
 * Windows up to 160 x 160. */
int pl_features_sweep(const double* d_sample, int64_t n, int h, int w, double dpmm, double radius_mm, double tol_mm,
                      double min_sep_px, int max_number, const double* h_cutoffs, int nlevels, int32_t* d_count,
                      double* d_xy, int32_t* d_level, int32_t* d_status, void* stream);
/* The same sweep straight from uint16 FRAMES: the window [top, top + h) x [left, left + w) of every frame, whose sample
 * SizedDiskRegion.calculate builds as stretch(invert((a - frame min) / (frame max - frame min))) (pylinac/winston_lutz.py:
 * 711-712, 788-806; pylinac/metrics/image.py:564-612; pylinac/metrics/utils.py:112-118).  Every map of that chain is monotone,
 * so the workgroup evaluates it per pixel from the window's integer extrema with the float64 operations of pl_ground /
 * pl_normalize / pl_invert / pl_scale in their order: bit-identical samples without nine float64 passes over the windows.
 * d_vmin / d_vmax float64[n]: the FRAME's min / max (ground() / normalize() act on the whole image); invert = 0 for
 * low-density BBs. */
int pl_features_sweep_u16(const uint16_t* d_frames, int64_t n, int frame_h, int frame_w, int top, int left, int h, int w,
                          const double* d_vmin, const double* d_vmax, int invert, double dpmm, double radius_mm, double tol_mm,
                          double min_sep_px, int max_number, const double* h_cutoffs, int nlevels, int32_t* d_count,
                          double* d_xy, int32_t* d_level, int32_t* d_status, void* stream);

/* ---- a13 (fields): one threshold level of GlobalSizedFieldLocator.calculate (pylinac/metrics/image.py:817-897)
 * Inputs per frame: the 8-connected label image of `sample > cutoff` (pl_label), its label count and region
 * table (pl_region_stats).  Regions whose bbox keeps clear of the (buffer_size + 1)-pixel border band
 * (segmentation.clear_border before labelling) and that pass is_right_square_perimeter and
 * is_right_area_square (pylinac/metrics/features.py:69-101) append their UNWEIGHTED centroid (x, y) to
 * d_xy [n][8][2] unless within max(equivalent_diameter of this level's hits) / dpmm of an accepted point.
 * d_done / d_count / d_level / d_status as for pl_features_level. */
int pl_fields_level(const int32_t* d_labels, const int32_t* d_nlabels, const double* d_stats, int max_labels,
                    int64_t n, int h, int w, double dpmm, double field_width_mm, double field_height_mm,
                    double field_tol_mm, int buffer_size, int max_number, int level, int32_t* d_done,
                    int32_t* d_count, double* d_xy, int32_t* d_level, int32_t* d_status, void* stream);

/* ---- f1 ("next" row): Varian XIM compressed pixels (pylinac/core/image.py:1180-1296) -----------------------
 * d_lookup: the file's lookup table (2-bit size codes, 4 per byte); d_stream: the compressed pixel buffer that
 * follows its 4-byte length in the file ((width + 1) int32 values, then one 1/2/4-byte little-endian difference per
 * remaining pixel).  d_out: int8/16/32/64 [height][width] by bytes_per_pixel (1/2/4/8), wrap-around arithmetic like
 * the reference's numpy arrays.  d_work: pl_xim_work_bytes() bytes; its first int32 is a status the caller may read
 * after the stream completes (bit 0: size code 3 in the lookup table -- the reference raises KeyError --,
 * bit 1: stream shorter than the lookup table implies). */
int64_t pl_xim_work_bytes(int width, int height);
int pl_xim_decode(const unsigned char* d_lookup, int64_t lookup_bytes, const unsigned char* d_stream,
                  int64_t stream_bytes, int width, int height, int bytes_per_pixel, void* d_out,
                  unsigned char* d_work, void* stream);

/* ---- f1 ("next" row), the DICOM half: native (uncompressed) Pixel Data -> typed frames -------------------------
 * Replaces `self.metadata.pixel_array` [+ `.astype(dtype)`] [+ `pixels.apply_rescale`] of DicomImage.__init__
 * (pylinac/core/image.py:1431-1444, 363-389) and the per-file loop of the DICOM stacks (image.py:2155-2160, 2234-2243)
 * for a batch: pydicom's numpy handler (pydicom>=2.0,<3, pyproject.toml:40; pixel_data_handlers/numpy_handler.py
 * get_pixeldata + util.pixel_dtype) = np.frombuffer(PixelData, '<' or '>', 'u' or 'i', BitsAllocated / 8 bytes).
 *   d_bytes [nbytes]: a device copy of the file(s) or of a Pixel Data value, 4-byte-aligned start; d_offsets int64 [n]:
 *   the byte at which each frame's first sample lies (any alignment: (7FE0,0010) values start wherever the header ends);
 *   rows x cols samples per frame, SamplesPerPixel 1; bits_allocated 8 / 16 / 32; pixel_representation 0 / 1;
 *   big_endian: Explicit VR Big Endian (1.2.840.10008.1.2.2);
 *   unused_bits 0 = the container value as stored (pydicom 2.x, the reference's pin), 1 = low bits_stored bits kept /
 *   sign-extended from bit bits_stored - 1 (pydicom >= 3 `correct_unused_bits`);
 *   d_out [n][rows][cols], out_dtype = the container dtype (PL_U8 for 8-bit, PL_U16 / PL_I16, PL_I32 for 32-bit: wider
 *   unsigned and int8 travel as same-width bits), PL_F32 (`astype(float32)`) or PL_F64 (`astype(float64)`, and with
 *   rescale != 0: `* slope` then `+ intercept`, two roundings = apply_modality_lut);
 *   d_status int32 [n]: 1 = the frame does not lie inside the buffer (pydicom raises ValueError), frame left untouched. */
int pl_dicom_decode(const unsigned char* d_bytes, int64_t nbytes, const int64_t* d_offsets, int64_t n, int rows, int cols,
                    int bits_allocated, int bits_stored, int pixel_representation, int big_endian, int unused_bits,
                    void* d_out, int out_dtype, int rescale, double slope, double intercept, int32_t* d_status,
                    void* stream);

/* ---- f2 ("next" row, first half): skimage.feature.canny as called at pylinac/planar_imaging.py:574-588 -------
 * float64 images, mask=None.  The caller composes: G = pl_gaussian2d_mode(mode 2) of the image and of an all-ones
 * frame; pl_canny_normalise: smoothed = G(image) / (G(ones) + eps); pl_sobel on axis 1 (jsobel) and axis 0 (isobel);
 * pl_canny_nms: magnitude = hypot and the interpolated non-maximum suppression (uint8 local maxima);
 * pl_order_stats_f64: exact order statistics of the magnitude image for np.percentile thresholds;
 * pl_canny_hysteresis phase 0: low / high masks from d_thresholds [n][2]; the caller labels d_low with pl_label
 * (8-connected); phase 1: keep the labelled segments that contain a high pixel (d_good: n*h*w int32 scratch). */
int pl_canny_normalise(const double* d_g_img, const double* d_g_ones, int64_t n, int64_t per_frame, double* d_out,
                       void* stream);
int pl_canny_nms(const double* d_isobel, const double* d_jsobel, int64_t n, int h, int w, double* d_magnitude,
                 unsigned char* d_local_max, void* stream);
/* canny(mask=...) (skimage/feature/_canny.py, smooth_with_function_and_mask + the eroded mask): pl_canny_mask_prepare writes
 * the image with 0 outside the mask and the mask as float64 (both are then smoothed; pl_canny_normalise divides them per frame);
 * pl_canny_nms_masked is pl_canny_nms restricted to binary_erosion(mask, 3 x 3 ones, border_value=0).  d_mask: uint8, non-zero =
 * inside; mask_per_frame = 1: [n][h][w], 0: one [h][w] plane for the batch. */
int pl_canny_mask_prepare(const double* d_img, const unsigned char* d_mask, int mask_per_frame, int64_t n, int64_t per_frame,
                          double* d_masked, double* d_mask_f, void* stream);
int pl_canny_nms_masked(const double* d_isobel, const double* d_jsobel, int64_t n, int h, int w, const unsigned char* d_mask,
                        int mask_per_frame, double* d_magnitude, unsigned char* d_local_max, void* stream);
int pl_order_stats_f64(const double* d_values, int64_t n, int64_t count, const int64_t* d_ranks, int n_ranks,
                       double* d_out, void* stream);
/* skimage.transform.hough_line(image, theta) accumulator (pylinac/planar_imaging.py:3158): d_image uint8 [h][w]
 * (non-zero = edge), d_cos / d_sin float64 [n_theta] (np.cos / np.sin of the angles, from the host), d_accum uint64
 * [2 * ceil(sqrt(h^2 + w^2))][n_theta] (zeroed here; scikit-image 0.18.3 layout).  The distance bins are
 * np.linspace(-offset, offset, rows). */
int pl_hough_line(const unsigned char* d_image, int h, int w, const double* d_cos, const double* d_sin, int n_theta,
                  unsigned long long* d_accum, void* stream);
/* f2, second half: the dense part of skimage.transform.hough_line_peaks -> skimage.feature.peak._prominent_peaks
 * (scikit-image 0.18.3) as called at pylinac/planar_imaging.py:3160-3166.
 * pl_max_filter1d: ndimage.maximum_filter1d(img, size = 2 * half + 1, axis, mode="constant", cval=0) per frame of
 *   [n][h][w] (any dtype of the enum; the Hough accumulator is passed as PL_I64: counts < 2^63).  Not in place.
 * pl_peak_candidates: mask = (img == img_max) & (double(img) > threshold), uint8 [count]. */
int pl_max_filter1d(const void* in, void* out, int dtype, int64_t n, int h, int w, int axis, int half, void* stream);
int pl_peak_candidates(const void* img, const void* img_max, int dtype, int64_t count, double threshold,
                       unsigned char* mask, void* stream);
int pl_canny_hysteresis(const unsigned char* d_local_max, const double* d_magnitude, const double* d_thresholds,
                        int64_t n, int h, int w, unsigned char* d_low, unsigned char* d_high, const int32_t* d_labels,
                        int32_t* d_good, unsigned char* d_out, int phase, void* stream);

/* ---- f3 ("next" row): ROI statistics after phantom localisation ---------------------------------------
 * DiskROI.circle_mask + pixel_value/mean/std/min/max (pylinac/core/roi.py:104-140) and axis-aligned
 * RectangleROI.pixel_array statistics (:664-704).  d_rois float64 [..][rois_per_frame][4] (roi_frame_stride
 * doubles between frames, 0 = the same ROIs for every frame): kind 0 disk = (cx, cy, radius, unused) with
 * skimage.draw.disk membership; kind 1 rectangle = (r0, r1, c0, c1) half-open.  d_out float64
 * [n][rois_per_frame][6] = count, mean, std (population), min, max, median (np.median).  d_status int32
 * [n][rois_per_frame]: 0 ok, 1 a selected pixel lies outside the frame (rectangles: the window leaves it), 2 the ROI box
 * exceeds 2^28 pixels, 3 empty.  ROIs of up to 16384 pixels are reduced from an LDS copy, larger ones by streaming passes. */
int pl_roi_stats(const void* d_frames, int dtype, int64_t n, int h, int w, const double* d_rois,
                 int rois_per_frame, int64_t roi_frame_stride, int kind, double* d_out, int32_t* d_status,
                 void* stream);

/* RectangleROI.pixels_flat statistics (pylinac/core/roi.py:644-704) for rotated rectangles, as general polygons:
 * d_vertices float64 [..][rois_per_frame][n_vertices][2] = (row, col) (roi_frame_stride doubles between frames, 0 =
 * the same polygons for every frame); membership = skimage.draw.polygon(r, c, shape=(h, w)) of scikit-image 0.18.3
 * (edge and vertex points included).  Output and status as pl_roi_stats (status 3: the polygon misses the frame). */
int pl_polygon_roi_stats(const void* d_frames, int dtype, int64_t n, int h, int w, const double* d_vertices,
                         int n_vertices, int rois_per_frame, int64_t roi_frame_stride, double* d_out,
                         int32_t* d_status, void* stream);

/* ---- a15: BaseImage.gamma, the Bakai gamma map (pylinac/core/image.py:994-1016) ---------------------------------
 * pl_bakai_mask: ref[ref < d_frame_cut[frame]] = NaN (float64) and its float32 copy (the Sobel input: pl_sobel on
 * axis 1 / 0 gives d_grad_x / d_grad_y).  pl_bakai_gamma: |comp - ref| / sqrt(dose_term + dist_term * hypot(gx, gy)^2)
 * with the float32 / float64 mix numpy uses (dose_term = float32((doseTA/100)^2), dist_term = float32(distTA_px^2)). */
int pl_bakai_mask(const double* d_ref, const double* d_frame_cut, int64_t n, int64_t per_frame, double* d_ref_masked,
                  float* d_ref32, void* stream);
int pl_bakai_gamma(const double* d_ref_masked, const double* d_comp, const float* d_grad_x, const float* d_grad_y,
                   float dose_term, float dist_term, int64_t total, double* d_out, void* stream);

/* pylinac/core/gamma.py:105-227 gamma_geometric (the simplex-distance 1-D gamma PhysicalProfileMixin.gamma calls,
 * pylinac/core/profile.py:822-874): d_ref / d_ref_x float64 [n_ref], d_eval / d_eval_x float64 [n_eval] (n_eval >= 2);
 * dose_denominator = max(reference) * dose_to_agreement, threshold_normalized = dose_threshold / dose_to_agreement,
 * eval_x_decreasing = the evaluation coordinates run high -> low.  d_gamma float64 [n_ref] (fill_value below the threshold). */
int pl_gamma_geometric(const double* d_ref, const double* d_ref_x, int n_ref, const double* d_eval, const double* d_eval_x,
                       int n_eval, double dose_denominator, double distance_to_agreement, double threshold_normalized,
                       int eval_x_decreasing, double gamma_cap, double fill_value, double* d_gamma, void* stream);

/* ---- f4 ("next" row, gamma part): pylinac.core.gamma.gamma_2d (pylinac/core/gamma.py:229-330) --------------
 * d_reference / d_evaluation float64 [n][h][w]; dose_fraction = dose_to_agreement / 100; global_dose != 0:
 * dose_ta = dose_fraction * d_ref_max[frame] (reference.max()), else dose_fraction * reference (elementwise).
 * d_dr / d_dc / d_dist2 [n_offsets]: the skimage.draw.disk((0, 0), DTA + 1) offsets and
 * (dr / DTA)^2 + (dc / DTA)^2, from the host.  threshold_normalized = dose_threshold / 100.  d_work:
 * 2 * n * h * w doubles.  d_out float64 [n][h][w]: gamma, gamma_cap where Gamma^2 >= cap^2, fill_value where the
 * normalised reference is NaN or below the threshold.  Bit-identical to the reference. */
int pl_gamma2d(const double* d_reference, const double* d_evaluation, int64_t n, int h, int w, double dose_fraction,
               int global_dose, const double* d_ref_max, const int32_t* d_dr, const int32_t* d_dc,
               const double* d_dist2, int n_offsets, double threshold_normalized, double gamma_cap,
               double fill_value, double* d_work, double* d_out, void* stream);

/* pylinac.core.gamma.gamma_1d (pylinac/core/gamma.py:333-455).  d_ref / d_ref_x [n_ref], d_eval / d_eval_x [n_eval]
 * (abscissae ascending), n_samples = int(DTA * resolution_factor * 2 + 1) search samples per reference point,
 * threshold = reference.max() / 100 * dose_threshold, dose_ta_global = dose_to_agreement / 100 * reference.max(),
 * dose_fraction = dose_to_agreement / 100 (local dose), dta_squared = distance_to_agreement ** 2.
 * d_gamma [n_ref]; d_eval_vals / d_eval_xs [n_ref][n_samples] (rows of skipped points undefined), d_computed int32
 * [n_ref] (0: below the threshold -> fill_value). */
int pl_gamma1d(const double* d_ref, const double* d_ref_x, int n_ref, const double* d_eval, const double* d_eval_x,
               int n_eval, double distance_to_agreement, double dta_squared, int n_samples, double threshold,
               double dose_ta_global, double dose_fraction, int global_dose, double gamma_cap, double fill_value,
               double* d_gamma, double* d_eval_vals, double* d_eval_xs, int32_t* d_computed, void* stream);

/* ---- a11: profile resampling -----------------------------------------------------------------------
 * scipy.interpolate.interp1d(x, y, kind, bounds_error=False, fill_value="extrapolate")(xq) as called by
 * SingleProfile._interpolate (pylinac/core/profile.py:1349-1358).  d_x float64 abscissae (x_stride elements
 * between profiles, 0 = shared by all), d_y float64 [n_profiles][length], d_xq float64 [n_query] (shared),
 * kind 0 = "linear" (scipy's slope form, bit-identical), 1 = "cubic" (the not-a-knot interpolating cubic
 * spline, ~1e-13 relative to scipy's B-spline evaluation; length >= 4).  d_work: 3*n_profiles*length doubles
 * for kind 1 (may be NULL for kind 0); after a kind-1 call its first n_profiles*length doubles hold the spline's
 * second derivatives M[profile][i] at the knots (the piecewise cubic on [x_i, x_i+1], h = x_i+1 - x_i, is
 * (M_i (x_i+1 - x)^3 + M_i+1 (x - x_i)^3) / (6 h) + (y_i / h - M_i h / 6)(x_i+1 - x) + (y_i+1 / h - M_i+1 h / 6)(x - x_i)),
 * which host-side optimisers evaluate point by point (InflectionDerivativeProfile.field_edge_idx, profile.py:656-670).
 * d_out float64 [n_profiles][n_query]. */
int pl_interp1d(const double* d_x, int64_t x_stride, const double* d_y, int64_t n_profiles, int length,
                const double* d_xq, int n_query, int kind, double* d_work, double* d_out, void* stream);
/* scipy.ndimage.zoom(values, zoom, order=3, mode="nearest", grid_mode) of 1-D float64 profiles, as
 * ProfileBase.as_resampled (pylinac/core/profile.py:353-390, grid_mode 0) and PhysicalProfileMixin.as_resampled
 * (:950-1011, grid_mode 1 by default) call it: 12-sample edge padding, cubic B-spline prefilter (mirror initialisation),
 * four-tap evaluation at i * (length - 1) / (out_length - 1), or at (i + 1/2) * length / out_length - 1/2 in grid mode.
 * out_length = round(length * zoom) is the caller's.  d_work: n_profiles * (length + 24) doubles.  ~1e-14 relative to
 * scipy (tests state 1e-12). */
int pl_zoom1d_cubic(const double* d_y, int64_t n_profiles, int length, int out_length, int grid_mode, double* d_work,
                    double* d_out, void* stream);
/* np.gradient(y) (unit spacing, edge_order 1) per profile, as SingleProfile.inflection_data takes it of the
 * smoothed profile (pylinac/core/profile.py:1643-1647).  float64 [n_profiles][length] -> same shape. */
int pl_gradient1d(const double* d_y, int64_t n_profiles, int length, double* d_out, void* stream);

/* ---- a18: noise power spectrum, radial average, ESF-FFT MTF ---------------------------------------
 * pl_nps2d: pylinac/core/nps.py:35-79 noise_power_spectrum_2d.  d_rois float64, n_rois ROIs of which the
 * top-left length x length block is used (roi_stride / row_stride in elements, so ROIs of different shapes
 * can sit zero-padded in one buffer); per ROI: subtract the block mean, |fft2|^2, fftshift; mean over ROIs;
 * times pixel_size^2 / length^2 -> d_out float64 [length][length].  d_work: pl_nps2d_work_doubles() doubles. */
int pl_nps2d(const double* d_rois, int64_t n_rois, int length, int64_t roi_stride, int row_stride,
             double pixel_size, double* d_work, double* d_out, void* stream);
int64_t pl_nps2d_work_doubles(int64_t n_rois, int length);
/* pylinac/core/nps.py:12-32 radial_average about floor(shape/2): bin = int(sqrt(dx^2+dy^2)), mean per bin
 * (sums in numpy's raster order -> bit-identical), d_out float64 [nbins], nbins = 1 + the largest bin. */
int pl_radial_average(const double* d_arr, int h, int w, int nbins, double* d_out, void* stream);
/* pylinac/core/mtf.py:448-456 _compute_esf_mtf for n_esf edge spread functions (float64 [n_esf][lmax], valid
 * lengths d_lens >= 2, d_window float64 [n_esf][lmax] = the window samples of each ESF's own length):
 * |fft(gradient(esf) * window, num_samples)| / DC for k < num_samples/2 -> d_mtf_each [n_esf][num_samples/2];
 * d_mtf_mean [num_samples/2] = mean over the ESFs (mtf.py:375).  d_work: n_esf * (num_samples/2) doubles. */
int pl_esf_mtf(const double* d_esf, const int32_t* d_lens, const double* d_window, int n_esf, int lmax,
               int num_samples, double* d_work, double* d_mtf_each, double* d_mtf_mean, void* stream);

/* ---- a8-a10: pylinac.core.profile.find_peaks over scipy.signal.find_peaks -----------------------
 * (pylinac/core/profile.py:2545-2649).  One 1-D float64 profile per batch item. */
typedef struct pl_peak_params {
  double threshold;       /* `threshold` argument                                             */
  int threshold_is_ratio; /* 1: height = min + threshold*(max-min) over the FULL profile       */
  int distance;           /* scipy `distance` (>= 1), already ceil'ed                          */
  int has_prominence;     /* `required_prominence is not None`                                 */
  double prominence_min;
  double width_min;       /* `min_width` (pylinac passes 0)                                    */
  double rel_height;      /* 1 - fwxm_height                                                   */
  int region_lo;          /* search region [lo, hi) in samples; idx are reported un-trimmed     */
  int region_hi;
  int max_number;         /* <= 0: keep all                                                    */
  int sort_key;           /* pl_peak_sort                                                      */
} pl_peak_params;

/* d_x: float64 [n][len] (row stride = stride elements).  Outputs (capacity `cap` per profile):
 *   d_count int32[n]; d_idx,d_left_base,d_right_base int32[n][cap];
 *   d_props float64[n][6][cap] = heights, prominences, widths, width_heights, left_ips, right_ips
 *   d_status int32[n]: 0 ok, 1 = more than `cap` peaks (truncated).
 * left_ips/right_ips are relative to the trimmed region, exactly like the reference
 * (pylinac/core/profile.py:2613 shifts only the indices). */
int pl_find_peaks(const double* d_x, int64_t n, int len, int64_t stride, const pl_peak_params* params,
                  int cap, int32_t* d_count, int32_t* d_idx, int32_t* d_left_base,
                  int32_t* d_right_base, double* d_props, int32_t* d_status, void* stream);

/* ragged form of pl_find_peaks: profile i has d_lens[i] <= len samples */
int pl_find_peaks_var(const double* d_x, int64_t n, int len, const int32_t* d_lens, int64_t stride,
                      const pl_peak_params* params, int cap, int32_t* d_count, int32_t* d_idx,
                      int32_t* d_left_base, int32_t* d_right_base, double* d_props, int32_t* d_status,
                      void* stream);
/* pl_find_peaks_var with per-profile search regions: d_regions int32[n][2] = [lo, hi) of profile i (NULL = the region of
 * `params`): CTP528CP504.mtf searches the valleys between the outermost peaks of each line-pair region
 * (pylinac/ct.py:1526-1533). */
int pl_find_peaks_regions(const double* d_x, int64_t n, int len, const int32_t* d_lens, int64_t stride,
                          const pl_peak_params* params, const int32_t* d_regions, int cap, int32_t* d_count,
                          int32_t* d_idx, int32_t* d_left_base, int32_t* d_right_base, double* d_props,
                          int32_t* d_status, void* stream);

/* CTP528CP504.mtf's searches (pylinac/ct.py:1511-1544) for every (profile, line-pair region) pair in one launch: per region
 * k < nregions (<= 16) find_peaks with peak_params[k] (host array; its search region and max_number = the expected number
 * of peaks, 1..cap_p) and, when exactly max_number peaks were found, find_valleys = find_peaks of the negated profile with
 * valley_params[k] inside [first peak index, last peak index) (valley_params' own region is ignored).
 *   d_pk_count int32 [n][nregions], d_pk_height float64 [n][nregions][cap_p] (peak_heights in index order, NaN beyond count),
 *   d_vl_count int32 [n][nregions] (0 when the peak count was wrong), d_vl_value float64 [n][nregions][cap_v] (the profile's
 *   values at the valley indices, NaN beyond count); d_means (optional) float64 [n][nregions][2] = np.mean of the peak
 *   heights (NaN unless exactly max_number peaks were found) and of the valley values (NaN when there are none): the
 *   max_values.mean() / min_values.mean() of ct.py:1530, 1536.  Capacities 1..8, regions of at most 1024 samples. */
int pl_peak_valley_regions(const double* d_x, int64_t n, int len, int64_t stride, const pl_peak_params* peak_params,
                           const pl_peak_params* valley_params, int nregions, int cap_p, int cap_v, int32_t* d_pk_count,
                           double* d_pk_height, int32_t* d_vl_count, double* d_vl_value, double* d_means, void* stream);

/* ---- BASELINE config #3: PicketFence.analyze per-image measurement, UP_DOWN pickets ------------
 * (pylinac/picketfence.py:745-803, 847-912, 1605-1628) on uint16 frames whose float64 image would be
 * q = (a - sub_i) / div_i (the constructor's ground()/normalize(), picketfence.py:322-323).
 * pl_scaled_colmean: np.mean(q, 0) -> float64 [n][w] (row-sequential float64 sum, like numpy).
 * pl_pf_pickets: from a pl_find_peaks result on the max-normalised leaf profile: FWXM picket centres
 *   int(round(l + (r - l)/2)), their profile values, and np.median(np.diff(np.sort(idx))) per frame.
 * pl_pf_windows: one window per (frame, leaf, picket slot): status (0 ok, 1 no such picket, 2 rejected by
 *   _is_mlc_peak_in_window, 3 empty/too large), np.median(window, axis=0) grounded and max-normalised
 *   into d_prof [n*nleaves*cap][lmax], its length, and max(approx_idx - spacing/2, 0).
 *   d_leaf_top/d_leaf_bottom: int32 [nleaves] window rows (host geometry, _get_mlc_window).
 * pl_pf_positions: position = FWXM centre (pl_fwxm_record) + offset, NaN where status != 0. */
int pl_scaled_colmean(const uint16_t* in, int64_t n, int h, int w, const double* d_sub, const double* d_div,
                      double* d_out, void* stream);
int pl_pf_pickets(const int32_t* d_count, const double* d_props, int cap, const double* d_prof, int w, int64_t n,
                  int32_t* d_pk_idx, double* d_pk_val, double* d_spacing, void* stream);
int pl_pf_windows(const uint16_t* in, int64_t n, int h, int w, const double* d_sub, const double* d_div,
                  const int32_t* d_pk_count, const int32_t* d_pk_idx, const double* d_pk_val, int cap,
                  const double* d_spacing, const int32_t* d_leaf_top, const int32_t* d_leaf_bottom, int nleaves,
                  double height_threshold, double edge_threshold, int lmax, double* d_prof, int32_t* d_len,
                  double* d_offset, int32_t* d_status, void* stream);
/* pl_pf_windows with the caller's bound on the window height (max_rows = the tallest bottom - top of the leaf table, 1..48):
 * the kernel's LDS per wave follows it, which is what decides how many waves hide each other's latency. */
int pl_pf_windows_rows(const uint16_t* in, int64_t n, int h, int w, const double* d_sub, const double* d_div,
                       const int32_t* d_pk_count, const int32_t* d_pk_idx, const double* d_pk_val, int cap,
                       const double* d_spacing, const int32_t* d_leaf_top, const int32_t* d_leaf_bottom, int nleaves,
                       int max_rows, double height_threshold, double edge_threshold, int lmax, double* d_prof, int32_t* d_len,
                       double* d_offset, int32_t* d_status, void* stream);
int pl_pf_positions(const int32_t* d_status, const double* d_fwxm, const double* d_offset, int64_t m,
                    double* d_pos, void* stream);

/* The window kernel with the FWXM search fused in, for either orientation (pylinac/picketfence.py:746-749, 847-886,
 * 1605-1628): per (frame, leaf, picket slot) d_rec float64 [n*nleaves*cap][3] = centre, left edge, right edge of the window
 * profile's FWXM peak, each + max(approx_idx - spacing/2, 0) -- `position` for separate_leaves False (centre) and True (left,
 * right); NaN where d_status != 0 (codes as pl_pf_windows) or the profile has no peak.  orientation 0 = UP_DOWN (d_leaf_lo /
 * d_leaf_hi are window ROWS, pickets run along the columns), 1 = LEFT_RIGHT (they are window COLUMNS, pickets run along the
 * rows; np.std over axis 0 and np.median over axis 1 of the window, i.e. the transposed computation with numpy's summation
 * order for a non-contiguous axis).  fwxm_params: pl_find_peaks parameters of FWXMProfile.field_edge_idx (fwxm_height,
 * max_number = 1).  d_prof (optional, float64 [n*nleaves*cap][lmax >= 128]) receives the window profiles.  max_rows: the
 * widest leaf in pixels, 1..48 (10 mm leaves on the finest supported EPID at isocentre scale are 45 pixels).
 * exact_deviation: 0 = the edge test max(std) < edge_threshold * median(std) (:855) is decided from exact integer row moments
 * whenever it holds or fails by more than numpy's rounding could move it (1e-7 relative), and numpy's float64 sequence only
 * runs for a window inside that margin; 1 = always evaluate numpy's sequence.  Same results either way. */
int pl_pf_measure(const uint16_t* in, int64_t n, int h, int w, int orientation, const double* d_sub, const double* d_div,
                  const int32_t* d_pk_count, const int32_t* d_pk_idx, const double* d_pk_val, int cap, const double* d_spacing,
                  const int32_t* d_leaf_lo, const int32_t* d_leaf_hi, int nleaves, int max_rows, double height_threshold,
                  double edge_threshold, int exact_deviation, const pl_peak_params* fwxm_params, double* d_rec, int32_t* d_status,
                  double* d_prof, int lmax, void* stream);
/* np.mean(q, 1) -> d_out float64 [n][h] (the leaf profile of LEFT_RIGHT pickets, picketfence.py:749) in numpy's PAIRWISE
 * summation order for the contiguous axis.  The summation tree of a row of w values is laid out by the caller
 * (ops.pairwise_plan): d_leaf_start / d_leaf_len int32 [nleaves] = the leaf blocks (<= 128 values each), d_program int32
 * [2 * nleaves - 1] = postfix order of the recursion (k >= 0: leaf k's sum, -1: add the two sums on top). */
int pl_scaled_rowmean(const uint16_t* in, int64_t n, int h, int w, const double* d_sub, const double* d_div,
                      const int32_t* d_leaf_start, const int32_t* d_leaf_len, int nleaves, const int32_t* d_program, int nprog,
                      double* d_out, void* stream);

/* Hill.fit (pylinac/core/hill.py:18-30: scipy.optimize.curve_fit(hill_func, x, y, p0 = (min(y), max(y), median(x), 0)) =
 * MINPACK lmdif with scipy's defaults) for a batch of penumbra windows, one lane per fit (csrc/hill.hip restates the published
 * Levenberg-Marquardt algorithm; SingleProfile.inflection_data fits two windows per profile, profile.py:1676-1708).
 *   d_x / d_y float64 [n][stride] (fit i uses its first d_lens[i] samples; d_lens NULL = mmax for all), 4 <= samples <= mmax
 *   <= 1024; d_work float64 [8 * mmax][n] scratch (transposed: the fits of a wave side by side); d_params float64 [n][4] = a, b, c, d (NaN when the fit has fewer than four
 *   samples); d_info int32 [n] = MINPACK's info (1-4 converged -- curve_fit accepts exactly these --, 5 maxfev, 6-8 tolerances
 *   too small, -1 too few samples, -4 a NaN or infinity among the samples: curve_fit raises ValueError); d_nfev (optional) int32 [n] function evaluations. */
int pl_hill_fit(const double* d_x, const double* d_y, const int32_t* d_lens, int64_t n, int mmax, int64_t stride,
                double* d_work, double* d_params, int32_t* d_info, int32_t* d_nfev, void* stream);
/* The same fit, which also reports d_last_step float64 [n]: the length of the LAST ACCEPTED Levenberg-Marquardt step relative
 * to the parameter vector (MINPACK's scaled variables).  MINPACK stops on the reduction of the sum of squares; in a flat valley
 * that happens while the parameters still move, and where it happens depends on the last bit of pow(): such a fit (last step
 * above ~1e-6) is reproduced by scipy to 1e-3, not to 1e-5.  NaN where no fit was attempted. */
int pl_hill_fit_ex(const double* d_x, const double* d_y, const int32_t* d_lens, int64_t n, int mmax, int64_t stride,
                   double* d_work, double* d_params, int32_t* d_info, int32_t* d_nfev, double* d_last_step, void* stream);

/* The two penumbra windows SingleProfile.inflection_data fits (pylinac/core/profile.py:1676-1700) for a batch of processed
 * profiles sharing x_indices: left_idx / right_idx = _x_interp_to_original(first derivative peak / last valley),
 * half = int(round(window_ratio * |right - left| / 2)), x = np.arange(idx - half, idx + half) filtered to x >= 0 (left) or
 * x < s (right), y = _y_original_to_interp(x) (scipy's linear interp1d, extrapolating).
 *   d_x_indices float64 [s]; d_values float64 [n][s]; peaks / valleys as pl_find_peaks_regions returns them (index order);
 *   d_xw / d_yw float64 [2n][mmax] (fit 2i = left, 2i + 1 = right window of profile i); d_lens int32 [2n] (0 and NaN edges when
 *   the derivative has no peak or valley -- the reference raises IndexError; negative = longer than mmax, nothing written);
 *   d_edges float64 [n][2] = left_idx, right_idx (the INFLECTION_DERIVATIVE edges). */
int pl_hill_windows(const double* d_x_indices, const double* d_values, int64_t n, int s, const int32_t* d_peak_count,
                    const int32_t* d_peak_idx, int cap_peaks, const int32_t* d_valley_count, const int32_t* d_valley_idx,
                    int cap_valleys, double window_ratio, int mmax, double* d_xw, double* d_yw, int32_t* d_lens, double* d_edges,
                    void* stream);

/* Hill.inflection_idx and Hill.y there (pylinac/core/hill.py:32-36, 56-65): d_params float64 [n][4] -> d_out float64 [n][2] =
 * c * ((d - 1) / (d + 1)) ** (1 / d), and the curve's value at it. */
int pl_hill_inflection(const double* d_params, int64_t n, double* d_out, void* stream);

/* SingleProfile.penumbra for the Hill edge method (pylinac/core/profile.py:1852-1898; Hill.x / Hill.gradient_at,
 * pylinac/core/hill.py:38-54): d_params float64 [n][4], d_inflection float64 [n][2] (pl_hill_inflection's output) ->
 * d_out float64 [n][6] = index where the curve takes lower / 50 of its inflection value, that value, the same for upper, the
 * distance of the two indices, the curve's gradient at the inflection point. */
int pl_hill_penumbra(const double* d_params, const double* d_inflection, int64_t n, double lower, double upper, double* d_out,
                     void* stream);

/* SingleProfile._y_original_to_interp (pylinac/core/profile.py:1227-1235) per profile: d_out[i][j] = scipy's linear interp1d
 * (x_indices, values_i, extrapolating) at d_q[i][j].  d_x_indices float64 [s], d_values float64 [n][s], d_q / d_out [n][nq]. */
int pl_profile_lookup(const double* d_x_indices, const double* d_values, int64_t n, int s, const double* d_q, int nq,
                      double* d_out, void* stream);

/* SingleProfile._x_interp_to_original (pylinac/core/profile.py:1204, 1217-1226): interp1d(range(s), x_indices) without
 * extrapolation = numpy's compiled np.interp (the grid value itself on a grid point).  d_q / d_out float64 [total]. */
int pl_index_to_original(const double* d_x_indices, int s, const double* d_q, int64_t total, double* d_out, void* stream);

/* FWXMProfile.field_edge_idx/center_idx/field_width_px (pylinac/core/profile.py:602-611, 322-344)
 * from a pl_find_peaks result obtained with max_number = 1:
 * d_out float64 [n][8] = n_peaks, peak_idx, height, prominence, left, right, centre, width
 * (NaN-filled when a profile has no peak; the reference raises IndexError there). */
int pl_fwxm_record(const int32_t* d_count, const int32_t* d_idx, const double* d_props, int cap,
                   int64_t n, double* d_out, void* stream);

/* The tail of the EPID pipeline in one launch (one workgroup per frame): d_profile[n][w] = (sum over bands of
 * d_parts[n][bands][w]) / h  = np.mean(frame, 0) (pylinac/picketfence.py:747-750), pl_find_peaks on it with `params`
 * (outputs as pl_find_peaks), then pl_fwxm_record's row d_fwxm[n][8] (pylinac/core/profile.py:602-611, 322-344). */
int pl_colparts_profile_fwxm(const uint32_t* d_parts, int64_t n, int bands, int w, int h, const pl_peak_params* params,
                             int cap, double* d_profile, int32_t* d_count, int32_t* d_idx, int32_t* d_left_base,
                             int32_t* d_right_base, double* d_props, int32_t* d_status, double* d_fwxm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PYLINAC_HIP_H */
