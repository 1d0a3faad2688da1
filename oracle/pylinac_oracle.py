"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the pylinac image-QA hot path.

This file is *not* part of the product.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it, and only as the checker.

It restates, in numpy, the glue that pylinac wraps around scipy / scikit-image for the hot
path of SURVEY.md section 8, each function citing the reference file:line it follows.
The arithmetic itself lives in third-party native code that is not under /root/reference:

  * scipy  (pin in the reference: ``scipy>=1.11.0``, pyproject.toml:30-47; container 1.15.3)
    ``ndimage.gaussian_filter / median_filter`` and ``signal.find_peaks``,
  * scikit-image (``>=0.18``; container: 0.18.3 under /opt/conda/bin/python3.9 only)
    ``filters.threshold_otsu``.

For those two levels this file holds

  (1) the *glue* functions (``filter``, ``threshold``, ``find_peaks`` ...) that call the real
      scipy exactly as the reference does (scipy is present on the GPU box), and
  (2) independent restatements of the published third-party algorithms
      (``gaussian_filter_restated``, ``median_filter_restated``, ``scipy_find_peaks_restated``,
      ``threshold_otsu``) that spell out the exact summation order / tie rules the HIP kernels
      implement.  (2) is pinned against (1) and against golden vectors produced by the
      reference's own code (tests/golden/make_golden.py, tests/test_oracle_golden.py).

Parity status: pinned (reference KATs of SURVEY.md section 8c + golden vectors generated
by importing the reference through oracle/ref_loader.py, and by scikit-image 0.18.3 itself for the
skimage-level restatements).  ``rescale_dicom_values`` restates a THIRD-PARTY algorithm, pydicom's
``pixels.apply_rescale`` (``pydicom>=2.0,<3``, the reference's pyproject.toml:40; not installed in the
build container, so nothing here could run it): it is pinned on the known answers of the reference's
own tests for ``_rescale_dicom_values`` (tests_basic/core/test_image.py:131-229), not on vectors
produced by pydicom.
"""
from __future__ import annotations

import math

import numpy as np
from scipy import ndimage, signal

# --------------------------------------------------------------------------------------
# a1/a2  filtering  (pylinac/core/array_utils.py:105-138, pylinac/core/image.py:695-712)
# --------------------------------------------------------------------------------------


def resolve_filter_size(array: np.ndarray, size):
    """Float size in (0,1) -> int(round(len(array)*size)) clamped to >= 1.
    ``len(array)`` is the ROW count for a 2-D array.  pylinac/core/array_utils.py:124-129."""
    if isinstance(size, float):
        if 0 < size < 1:
            size = int(round(len(array) * size))
            size = max(size, 1)
        else:
            raise ValueError("Float was passed but was not between 0 and 1")
    return size


def filter(array: np.ndarray, size=0.05, kind: str = "median") -> np.ndarray:
    """pylinac/core/array_utils.py:105-138 (the body behind ``BaseImage.filter``)."""
    if not array.size:
        raise ValueError("Array must not be empty")  # array_utils.py:23-26
    size = resolve_filter_size(array, size)
    if kind == "median":
        return ndimage.median_filter(array, size=size)  # array_utils.py:131
    elif kind == "gaussian":
        return ndimage.gaussian_filter(array, sigma=size)  # array_utils.py:133
    raise ValueError(f"Filter type {kind} unsupported. Use one of 'median', 'gaussian'")


def gaussian_kernel1d(sigma: float, truncate: float = 4.0):
    """scipy/ndimage/_filters.py ``_gaussian_kernel1d`` + ``gaussian_filter1d`` (order 0):
    radius ``int(truncate*sigma+0.5)``, ``exp(-0.5/sigma^2 * x^2)`` normalised by its sum.
    Returns (weights[2*radius+1], radius).  The kernel is exactly symmetric."""
    sd = float(sigma)
    lw = int(truncate * sd + 0.5)
    sigma2 = sd * sd
    x = np.arange(-lw, lw + 1)
    phi = np.exp(-0.5 / sigma2 * x**2)
    phi = phi / phi.sum()
    return phi[::-1].copy(), lw


def reflect_index(i, n: int):
    """scipy 'reflect' (half-sample symmetric, ``d c b a | a b c d | d c b a``), any distance."""
    i = np.asarray(i)
    p = 2 * n
    m = np.mod(i, p)
    return np.where(m >= n, p - 1 - m, m)


def _correlate1d_symmetric(a: np.ndarray, w: np.ndarray, lw: int, axis: int) -> np.ndarray:
    """scipy ``NI_Correlate1D`` symmetric branch (ni_filters.c), float64 accumulator:
         acc  = x[0]*w[0]
         acc += (x[-j] + x[+j]) * w[-j]     for j = radius .. 1   (outermost tap first)
    no FMA contraction.  Returns float64 (caller casts into the output dtype)."""
    n = a.shape[axis]
    x = np.moveaxis(a.astype(np.float64), axis, -1)
    idx = reflect_index(np.arange(-lw, n + lw), n)
    x = x[..., idx]
    c = lw
    acc = x[..., c : c + n] * w[lw]
    for j in range(lw, 0, -1):
        acc = acc + (x[..., c - j : c - j + n] + x[..., c + j : c + j + n]) * w[lw - j]
    return np.moveaxis(acc, -1, axis)


def cast_like_scipy(acc: np.ndarray, dtype) -> np.ndarray:
    """scipy copies the double line buffer into the output array with a C cast
    (ni_support.c CASE_COPY_LINE_TO_DATA): truncation toward zero for integer dtypes,
    round-to-nearest for float32."""
    dtype = np.dtype(dtype)
    if dtype.kind in "ui":
        return np.trunc(acc).astype(dtype)
    return acc.astype(dtype)


def gaussian_filter_restated(a: np.ndarray, sigma: float) -> np.ndarray:
    """``ndimage.gaussian_filter(a, sigma)`` as pylinac calls it (array_utils.py:133):
    axis 0 first, then axis 1 ...; the *output of each pass has the input dtype* and feeds the
    next pass."""
    w, lw = gaussian_kernel1d(sigma)
    out = a
    for axis in range(a.ndim):
        out = cast_like_scipy(_correlate1d_symmetric(out, w, lw, axis), a.dtype)
    return out


def median_filter_restated(a: np.ndarray, size: int) -> np.ndarray:
    """``ndimage.median_filter(a, size=s)``: s^ndim window, mode='reflect', origin 0 (window
    starts at ``i - s//2``), rank ``s^ndim // 2``."""
    s = int(size)
    lo = s // 2
    if a.ndim == 1:
        n = a.shape[0]
        idx = reflect_index(np.arange(n)[:, None] - lo + np.arange(s)[None, :], n)
        win = a[idx]
        return np.sort(win, axis=-1)[..., (s) // 2]
    h, w = a.shape
    ri = reflect_index(np.arange(h)[:, None] - lo + np.arange(s)[None, :], h)  # [h, s]
    ci = reflect_index(np.arange(w)[:, None] - lo + np.arange(s)[None, :], w)  # [w, s]
    win = a[ri[:, None, :, None], ci[None, :, None, :]].reshape(h, w, s * s)
    return np.sort(win, axis=-1)[..., (s * s) // 2]


# --------------------------------------------------------------------------------------
# a3  threshold / as_binary   (pylinac/core/image.py:785-815)
# --------------------------------------------------------------------------------------


def threshold(array: np.ndarray, threshold, kind: str = "high") -> np.ndarray:
    """``np.where(a >= t, a, 0)`` ('high') / ``np.where(a <= t, a, 0)``; image.py:797-800."""
    if kind == "high":
        return np.where(array >= threshold, array, 0)
    return np.where(array <= threshold, array, 0)


def as_binary(array: np.ndarray, threshold) -> np.ndarray:
    """image.py:802-815 -> int64 0/1 array."""
    return np.where(array >= threshold, 1, 0)


# --------------------------------------------------------------------------------------
# a5  ground / normalize / invert / stretch  (pylinac/core/array_utils.py:63-102, 141-168)
# --------------------------------------------------------------------------------------


def ground(array, value=0):
    return array - array.min() + value  # array_utils.py:102


def normalize(array, value=None):
    val = array.max() if value is None else value  # array_utils.py:66-71
    return array / val


def invert(array):
    return -array + array.max() + array.min()  # array_utils.py:77


def stretch(array, min=0, max=1):
    if max <= min:
        raise ValueError(f"Max must be larger than min. Passed max of {max} was <= {min}")
    return ground(normalize(ground(array)) * (max - min), value=min)  # array_utils.py:168


# --------------------------------------------------------------------------------------
# a4  Otsu  (skimage 0.18.3 filters/thresholding.py threshold_otsu + exposure.histogram;
#            call sites pylinac/ct.py:3323,3338-3340, pylinac/acr.py:1409)
# --------------------------------------------------------------------------------------


def histogram_like_skimage(image: np.ndarray, nbins: int = 256):
    """skimage.exposure.histogram(source_range='image'): integer image -> one bin per integer
    value in [min, max] (``_bincount_histogram``); float image -> ``np.histogram(bins=nbins)``
    with bin centres."""
    flat = image.ravel()
    if np.issubdtype(flat.dtype, np.integer):
        imin = int(flat.min())
        imax = int(flat.max())
        counts = np.bincount((flat.astype(np.int64) - imin), minlength=imax - imin + 1)
        centers = np.arange(imin, imax + 1)
        return counts, centers
    counts, edges = np.histogram(flat, bins=nbins, range=None)
    return counts, (edges[:-1] + edges[1:]) / 2.0


def threshold_otsu(image: np.ndarray, nbins: int = 256):
    """skimage 0.18.3 ``threshold_otsu``: constant image -> that value; otherwise argmax (first)
    of ``w1[:-1]*w2[1:]*(m1[:-1]-m2[1:])**2`` with float64 cumulative sums; returns the bin
    centre."""
    first = image.ravel()[0]
    if np.all(image == first):
        return first
    counts, centers = histogram_like_skimage(image, nbins)
    counts = counts.astype(float)
    weight1 = np.cumsum(counts)
    weight2 = np.cumsum(counts[::-1])[::-1]
    mean1 = np.cumsum(counts * centers) / weight1
    mean2 = (np.cumsum((counts * centers)[::-1]) / weight2[::-1])[::-1]
    variance12 = weight1[:-1] * weight2[1:] * (mean1[:-1] - mean2[1:]) ** 2
    idx = np.argmax(variance12)
    return centers[idx]


# --------------------------------------------------------------------------------------
# a6  percentiles  (np.percentile default 'linear'; call sites pylinac/core/image.py:899-926,
#                   pylinac/picketfence.py:229-238, pylinac/winston_lutz.py:775,1109-1133)
# --------------------------------------------------------------------------------------


def percentile(array: np.ndarray, q):
    return np.percentile(array, q)


def percentile_from_order_stats(lo_val, hi_val, frac):
    """numpy ``_lerp`` (lib/_function_base_impl.py): a + (b-a)*t, and b - (b-a)*(1-t) for
    t >= 0.5; used to rebuild np.percentile from two exact order statistics."""
    a = np.float64(lo_val)
    b = np.float64(hi_val)
    t = np.float64(frac)
    d = b - a
    r = a + d * t
    if t >= 0.5:
        r = b - d * (1 - t)
    if d == 0:
        r = a
    return r


# --------------------------------------------------------------------------------------
# a7  profile extraction  (no Image.profile() in the reference; pylinac/picketfence.py:747-750)
# --------------------------------------------------------------------------------------


def profile(array: np.ndarray, axis: int = 0, kind: str = "mean") -> np.ndarray:
    """``np.mean(image, axis)`` (picketfence.py:747-750), ``np.max(central, axis)``
    (starshot.py:216-217), ``np.sum(array, axis)`` (picketfence.py:1513-1514)."""
    if kind == "mean":
        return np.mean(array, axis=axis)
    if kind == "max":
        return np.max(array, axis=axis)
    if kind == "sum":
        return np.sum(array, axis=axis)
    if kind == "median":
        return np.median(array, axis=axis)
    raise ValueError(kind)


# --------------------------------------------------------------------------------------
# a8  find_peaks  (pylinac/core/profile.py:2545-2649 over scipy.signal.find_peaks)
# --------------------------------------------------------------------------------------


def parse_peak_args(peak_separation, search_region, threshold, values):
    """pylinac/core/profile.py:2626-2649 (note: int 0/1 also count as ratios)."""
    val_range = values.max() - values.min()
    if 0 <= threshold <= 1:
        threshold = values.min() + threshold * val_range
    if 0 <= peak_separation <= 1:
        peak_separation = max(int(peak_separation * len(values)), 1)
    if max(search_region) <= 1:
        shift_amount = int(search_region[0] * len(values))
        values = values[
            int(search_region[0] * len(values)) : int(search_region[1] * len(values))
        ]
    else:
        values = values[search_region[0] : search_region[1]]
        shift_amount = search_region[0]
    return peak_separation, shift_amount, threshold, values


def find_peaks(
    values,
    threshold=-np.inf,
    peak_separation=0,
    max_number=None,
    fwxm_height: float = 0.5,
    min_width: int = 0,
    search_region=(0.0, 1.0),
    peak_sort: str = "prominences",
    required_prominence=None,
    impl: str = "scipy",
):
    """pylinac/core/profile.py:2545-2623.  Only ``peak_idxs`` are shifted by the search-region
    offset (``left_ips``/``right_ips`` are not, profile.py:2613).  ``impl='restated'`` swaps
    scipy.signal.find_peaks for the pure-python restatement below."""
    values = np.asarray(values)
    peak_separation, shift_amount, threshold, trimmed = parse_peak_args(
        peak_separation, search_region, threshold, values
    )
    fp = signal.find_peaks if impl == "scipy" else scipy_find_peaks_restated
    peak_idxs, peak_props = fp(
        trimmed,
        rel_height=(1 - fwxm_height),
        width=min_width,
        height=threshold,
        distance=peak_separation,
        prominence=required_prominence,
    )
    peak_idxs = peak_idxs + shift_amount
    # Tie order of np.argsort(default quicksort) is implementation-defined;
    # the build pins it to 'stable' (== numpy's insertion sort for <=16 elements).
    order = list(np.argsort(peak_props[peak_sort], kind="stable"))[::-1][:max_number]
    largest = sorted(order)
    for key, vals in list(peak_props.items()):
        peak_props[key] = vals[largest]
    return peak_idxs[largest], peak_props


def scipy_find_peaks_restated(
    x, height=None, distance=None, prominence=None, width=None, rel_height=0.5
):
    """Pure-python restatement of scipy.signal.find_peaks (scipy/signal/_peak_finding.py and
    _peak_finding_utils.pyx) for the argument set pylinac uses.
    ``height``/``prominence``/``width`` are scalar minima (or None)."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    # -- _local_maxima_1d: strict rise, plateau -> midpoint, strict fall; ends never peaks
    peaks = []
    i = 1
    i_max = n - 1
    while i < i_max:
        if x[i - 1] < x[i]:
            i_ahead = i + 1
            while i_ahead < i_max and x[i_ahead] == x[i]:
                i_ahead += 1
            if x[i_ahead] < x[i]:
                peaks.append((i + i_ahead - 1) // 2)
                i = i_ahead
        i += 1
    peaks = np.array(peaks, dtype=np.intp)
    props = {}
    if height is not None:
        ph = x[peaks]
        keep = ph >= height
        peaks = peaks[keep]
        props["peak_heights"] = ph[keep]
    if distance is not None:
        # _select_by_peak_distance: highest first; ties -> stable argsort order
        d = math.ceil(distance)
        keep = np.ones(peaks.size, dtype=bool)
        order = np.argsort(x[peaks], kind="stable")
        for i in range(peaks.size - 1, -1, -1):
            j = order[i]
            if not keep[j]:
                continue
            k = j - 1
            while k >= 0 and peaks[j] - peaks[k] < d:
                keep[k] = False
                k -= 1
            k = j + 1
            while k < peaks.size and peaks[k] - peaks[j] < d:
                keep[k] = False
                k += 1
        peaks = peaks[keep]
        props = {k_: v[keep] for k_, v in props.items()}
    if prominence is not None or width is not None:
        prom = np.empty(peaks.size)
        lb = np.empty(peaks.size, dtype=np.intp)
        rb = np.empty(peaks.size, dtype=np.intp)
        for p, pk in enumerate(peaks):
            i = lb[p] = pk
            left_min = x[pk]
            while 0 <= i and x[i] <= x[pk]:
                if x[i] < left_min:
                    left_min = x[i]
                    lb[p] = i
                i -= 1
            i = rb[p] = pk
            right_min = x[pk]
            while i <= n - 1 and x[i] <= x[pk]:
                if x[i] < right_min:
                    right_min = x[i]
                    rb[p] = i
                i += 1
            prom[p] = x[pk] - max(left_min, right_min)
        props.update(prominences=prom, left_bases=lb, right_bases=rb)
    if prominence is not None:
        keep = props["prominences"] >= prominence
        peaks = peaks[keep]
        props = {k_: v[keep] for k_, v in props.items()}
    if width is not None:
        widths = np.empty(peaks.size)
        wh = np.empty(peaks.size)
        lips = np.empty(peaks.size)
        rips = np.empty(peaks.size)
        for p, pk in enumerate(peaks):
            i_min = props["left_bases"][p]
            i_max2 = props["right_bases"][p]
            h = wh[p] = x[pk] - props["prominences"][p] * rel_height
            i = pk
            while i_min < i and h < x[i]:
                i -= 1
            lip = float(i)
            if x[i] < h:
                lip += (h - x[i]) / (x[i + 1] - x[i])
            i = pk
            while i < i_max2 and h < x[i]:
                i += 1
            rip = float(i)
            if x[i] < h:
                rip -= (h - x[i]) / (x[i - 1] - x[i])
            widths[p] = rip - lip
            lips[p] = lip
            rips[p] = rip
        keep = widths >= width
        props.update(widths=widths, width_heights=wh, left_ips=lips, right_ips=rips)
        peaks = peaks[keep]
        props = {k_: v[keep] for k_, v in props.items()}
    return peaks, props


# --------------------------------------------------------------------------------------
# a9/a10  MultiProfile / FWXMProfile  (pylinac/core/profile.py:2050-2176, 578-611, 322-344)
# --------------------------------------------------------------------------------------


def multiprofile_find_peaks(values, threshold=0.3, min_distance=0.05, max_number=None,
                            search_region=(0.0, 1.0), peak_sort="prominences", impl="scipy"):
    """MultiProfile.find_peaks, profile.py:2050-2103 -> (idx, heights)."""
    idx, props = find_peaks(values, threshold=threshold, peak_separation=min_distance,
                            max_number=max_number, search_region=search_region,
                            peak_sort=peak_sort, impl=impl)
    return idx, props["peak_heights"]


def multiprofile_find_valleys(values, threshold=0.3, min_distance=0.05, max_number=None,
                              search_region=(0.0, 1.0), impl="scipy"):
    """MultiProfile.find_valleys, profile.py:2105-2133: peaks of ``-values``."""
    values = np.asarray(values)
    idx, _ = find_peaks(-values, threshold=threshold, peak_separation=min_distance,
                        max_number=max_number, search_region=search_region, impl=impl)
    return idx, values[idx]


def multiprofile_find_fwxm_peaks(values, threshold=0.3, min_distance=0.05, max_number=None,
                                 search_region=(0.0, 1.0), peak_sort="prominences",
                                 required_prominence=None, impl="scipy"):
    """MultiProfile.find_fwxm_peaks, profile.py:2135-2176: ``int(round(lt + (rt-lt)/2))``
    (python banker's rounding) and the value at that index."""
    values = np.asarray(values)
    _, props = find_peaks(values, threshold=threshold, peak_separation=min_distance,
                          max_number=max_number, search_region=search_region,
                          peak_sort=peak_sort, required_prominence=required_prominence,
                          impl=impl)
    idxs = [int(round(lt + (rt - lt) / 2)) for lt, rt in zip(props["left_ips"], props["right_ips"])]
    return np.array(idxs), np.array([values[i] for i in idxs])


def fwxm_edges(values, fwxm_height: float = 50, impl="scipy"):
    """FWXMProfile.field_edge_idx / center_idx / field_width_px with default x-values
    (profile.py:602-611, 322-327, 339-344).  ``x_at_x_idx`` over ``arange(len)`` with a k=1,s=0
    spline is the identity.  Raises IndexError when no peak exists,
    like the reference (profile.py:608)."""
    _, props = find_peaks(values, fwxm_height=fwxm_height / 100, max_number=1, impl=impl)
    left = float(props["left_ips"][0])
    right = float(props["right_ips"][0])
    center = abs(right - left) / 2 + left
    width = max(right, left) - min(right, left)
    return left, right, center, width


# --------------------------------------------------------------------------------------
# a12  circle profiles  (pylinac/core/profile.py:2244-2283, 2442-2483)
# --------------------------------------------------------------------------------------


def circle_radians(size, start_angle=0, ccw=True):
    """CircleProfile._radians, profile.py:2244-2252."""
    interval = (2 * np.pi) / size
    rads = np.arange(0 + start_angle, (2 * np.pi) + start_angle - interval, interval)
    if ccw:
        rads = rads[::-1]
    return rads


def circle_profile(image, center_xy, radius, start_angle=0, ccw=True, sampling_ratio=1.0):
    """CircleProfile._profile, profile.py:2279-2283."""
    rads = circle_radians(np.pi * radius * 2 * sampling_ratio, start_angle, ccw)
    x = np.cos(rads) * radius + center_xy[0]
    y = np.sin(rads) * radius + center_xy[1]
    return ndimage.map_coordinates(image, [y, x], order=0)


def collapsed_circle_profile(image, center_xy, radius, start_angle=0, ccw=True, sampling_ratio=1.0,
                             width_ratio=0.1, num_profiles=20):
    """CollapsedCircleProfile._profile, profile.py:2442-2483."""
    radii = np.linspace(start=radius * (1 - width_ratio), stop=radius * (1 + width_ratio), num=num_profiles)
    rads = circle_radians(np.pi * max(radii) * 2 * sampling_ratio, start_angle, ccw)
    cos, sin = np.cos(rads), np.sin(rads)
    profile = np.zeros(len(rads))
    for r in radii:
        profile += ndimage.map_coordinates(image, [sin * r + center_xy[1], cos * r + center_xy[0]], order=0)
    profile /= num_profiles
    return profile


def map_coordinates_nearest_restated(image, y, x):
    """scipy map_coordinates(order=0, mode='constant', cval=0): index floor(c+0.5); any coordinate
    outside [0, n-1] gives 0 (probed on scipy 1.15.3)."""
    h, w = image.shape
    ok = (x >= 0) & (x <= w - 1) & (y >= 0) & (y <= h - 1)
    xi = np.floor(np.where(ok, x, 0) + 0.5).astype(np.intp)
    yi = np.floor(np.where(ok, y, 0) + 0.5).astype(np.intp)
    return np.where(ok, image[yi, xi], 0).astype(image.dtype)


# --------------------------------------------------------------------------------------
# a15  Sobel  (pylinac/core/image.py:1006-1007)
# --------------------------------------------------------------------------------------


def sobel(image, axis):
    return ndimage.sobel(image, axis)


# --------------------------------------------------------------------------------------
# a14  Winston-Lutz field centroid  (pylinac/winston_lutz.py:711-712, 764-780)
# --------------------------------------------------------------------------------------


def wl_field_centroid(frame: np.ndarray):
    """ground -> normalize -> percentile threshold -> binary_fill_holes -> center_of_mass.
    Returns (x, y, filled pixel count)."""
    arr = ground(frame)                     # winston_lutz.py:711 (image.py:839-853)
    arr = normalize(arr)                    # winston_lutz.py:712
    mn, mx = np.percentile(arr, [5, 99.9])  # winston_lutz.py:775
    threshold_img = as_binary(arr, (mx - mn) / 2 + mn)
    filled = ndimage.binary_fill_holes(threshold_img)
    coords = ndimage.center_of_mass(filled)
    return coords[-1], coords[0], float(filled.sum())


def label_like_skimage(mask: np.ndarray, connectivity: int = 4):
    """skimage.measure.label numbering == scipy.ndimage.label numbering (raster order of the first
    pixel of each component); connectivity 4 -> cross structure, 8 -> full 3x3."""
    structure = ndimage.generate_binary_structure(2, 1 if connectivity == 4 else 2)
    return ndimage.label(mask, structure=structure)


# --------------------------------------------------------------------------------------
# a16  CatPhan slice localisation  (pylinac/ct.py:381-425, 3315-3348; skimage 0.18.3 semantics)
# --------------------------------------------------------------------------------------

_SCHARR_EDGE = np.array([1, 0, -1])
_SCHARR_SMOOTH = np.array([3, 10, 3]) / 16


def scharr_like_skimage(image: np.ndarray) -> np.ndarray:
    """skimage.filters.scharr (0.18.3 filters/edges.py:_generic_edge_filter): per axis
    ndi.convolve(image, edge (x) smooth, mode='reflect'), squared and summed, sqrt / sqrt(ndim)."""
    image = image.astype(float)
    out = np.zeros(image.shape, dtype=float)
    for edge_dim in (0, 1):
        k = _SCHARR_EDGE.reshape((3, 1) if edge_dim == 0 else (1, 3)) * \
            _SCHARR_SMOOTH.reshape((1, 3) if edge_dim == 0 else (3, 1))
        ax = ndimage.convolve(image, k, mode="reflect")
        ax *= ax
        out += ax
    return np.sqrt(out) / np.sqrt(2)


def gaussian_like_skimage(image: np.ndarray, sigma=1) -> np.ndarray:
    """skimage.filters.gaussian defaults: mode='nearest', truncate=4.0 on a float image."""
    return ndimage.gaussian_filter(image.astype(float), sigma, mode="nearest", truncate=4.0)


def disk_mask_like_skimage(center_rc, radius, shape) -> np.ndarray:
    """skimage.draw.disk(center, radius, shape=shape) (draw.py ellipse + _ellipse_in_shape,
    rotation 0) as a uint8 mask."""
    center = np.array(center_rc, dtype=float)
    radii = np.array([radius, radius], dtype=float)
    rot = 0.0 % np.pi
    r_rot = abs(radius * np.cos(rot)) + radius * np.sin(rot)
    c_rot = radius * np.sin(rot) + abs(radius * np.cos(rot))
    radii_rot = np.array([r_rot, c_rot])
    upper_left = np.maximum(np.ceil(center - radii_rot).astype(int), 0)
    lower_right = np.minimum(np.floor(center + radii_rot).astype(int), np.array(shape[:2]) - 1)
    shifted = center - upper_left
    bshape = lower_right - upper_left + 1
    r_lim, c_lim = np.ogrid[0:float(bshape[0]), 0:float(bshape[1])]
    sin_a, cos_a = np.sin(rot), np.cos(rot)
    r, c = (r_lim - shifted[0]), (c_lim - shifted[1])
    dist = ((r * cos_a + c * sin_a) / radii[0]) ** 2 + ((r * sin_a - c * cos_a) / radii[1]) ** 2
    rr, cc = np.nonzero(dist < 1)
    m = np.zeros(shape, np.uint8)
    m[rr + upper_left[0], cc + upper_left[1]] = 1
    return m


def clear_border_like_skimage(bw: np.ndarray, buffer_size: int) -> np.ndarray:
    """skimage.segmentation.clear_border(bw, buffer_size) (0.18.3 _clear_border.py): 8-connected
    labelling; every label owning a pixel in the (buffer_size+1)-wide border band is removed."""
    lab, _ = ndimage.label(bw, structure=np.ones((3, 3)))
    ext = buffer_size + 1
    band = np.zeros(bw.shape, bool)
    band[:ext, :] = band[-ext:, :] = True
    band[:, :ext] = band[:, -ext:] = True
    bad = np.unique(lab[band])
    out = bw.copy()
    out[np.isin(lab, bad[bad > 0])] = 0
    return out


def catphan_get_regions(arr: np.ndarray, mm_per_pixel: float, fill_holes: bool = True, clear_borders: bool = True):
    """pylinac/ct.py:3315-3348 for a Slice (default clip_in_localization=False, ct.py:2043):
    scharr -> gaussian(1) -> Otsu (256 bins) on the 110 mm disk * 0.8 -> '>' -> clear_border ->
    binary_fill_holes -> label (8-conn).  Returns (edges, bw, labels, n)."""
    edges = gaussian_like_skimage(scharr_like_skimage(arr), 1)
    cy, cx = arr.shape[0] / 2 - 0.5, arr.shape[1] / 2 - 0.5               # image.py:527-533
    disk = disk_mask_like_skimage((cy, cx), 110 / mm_per_pixel, edges.shape)
    thres = threshold_otsu(edges[disk.astype(bool)]) * 0.8
    bw = edges > thres
    if clear_borders:
        bw = clear_border_like_skimage(bw, min(int(max(bw.shape) / 100), 3))
    if fill_holes:
        bw = ndimage.binary_fill_holes(bw)
    lab, n = ndimage.label(bw, structure=np.ones((3, 3)))
    return edges, bw, lab, n


def region_table(lab: np.ndarray, n: int, intensity: np.ndarray) -> np.ndarray:
    """[n, 10]: area, bbox(r0,c0,r1,c1), centroid(r,c), filled_area, weighted_centroid(r,c) --
    skimage.measure.regionprops semantics (measure/_regionprops.py)."""
    rows = []
    for k in range(1, n + 1):
        m = lab == k
        rr, cc = np.nonzero(m)
        r0, r1, c0, c1 = rr.min(), rr.max() + 1, cc.min(), cc.max() + 1
        crop = m[r0:r1, c0:c1]
        filled = ndimage.binary_fill_holes(crop, np.ones((3, 3))).sum()
        w = intensity[m]
        rows.append([m.sum(), r0, c0, r1, c1, rr.mean(), cc.mean(), filled,
                     (w * rr).sum() / w.sum(), (w * cc).sum() / w.sum()])
    return np.array(rows, dtype=float).reshape(-1, 10)


def catphan_phantom_roi(arr: np.ndarray, mm_per_pixel: float, catphan_size: float):
    """Slice.phantom_roi (pylinac/ct.py:381-425): region whose filled_area is closest to the expected
    phantom size; ValueError when none / out of the 1.3x size window.  -> (label, table row)."""
    edges0 = scharr_like_skimage(arr)
    if np.max(edges0) < 0.1:
        raise ValueError("No edges were found in the image that look like the phantom")
    edges, bw, lab, n = catphan_get_regions(arr, mm_per_pixel)
    if n < 1:
        raise ValueError(f"The number of ROIs detected {n} was not the number expected (1)")
    tab = region_table(lab, n, edges)
    k = int(np.argsort(np.abs(tab[:, 7] - catphan_size), kind="stable")[0])
    if catphan_size * 1.3 < tab[k, 7] or tab[k, 7] < catphan_size / 1.3:
        raise ValueError("Unable to find ROI of expected size of the phantom")
    return k + 1, tab[k]


# --------------------------------------------------------------------------------------
# a13  BB finder: find_features + predicates  (pylinac/metrics/utils.py:66-190, metrics/features.py,
#      metrics/image.py:564-612; scikit-image 0.18.3 regionprops semantics)
# --------------------------------------------------------------------------------------

_PERIM_W = np.zeros(50)
_PERIM_W[[5, 7, 15, 17, 25, 27]] = 1
_PERIM_W[[21, 33]] = math.sqrt(2)
_PERIM_W[[13, 23]] = (1 + math.sqrt(2)) / 2


def perimeter_like_skimage(mask: np.ndarray) -> float:
    """skimage.measure.perimeter(image, neighbourhood=4) (measure/_regionprops_utils.py)."""
    img = mask.astype(np.uint8)
    eroded = ndimage.binary_erosion(img, ndimage.generate_binary_structure(2, 1), border_value=0)
    border = img - eroded
    codes = ndimage.convolve(border, np.array([[10, 2, 10], [2, 1, 2], [10, 2, 10]]), mode="constant", cval=0)
    hist = np.bincount(codes.ravel(), minlength=50)
    return float(_PERIM_W @ hist)


def _hull_ccw(points):
    """Andrew's monotone chain on integer points; strictly convex vertices, counter-clockwise in the
    (x0, x1) plane -- the orientation scipy.spatial.ConvexHull reports for 2-D input."""
    pts = sorted(set(map(tuple, points)))
    if len(pts) <= 2:
        return pts

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lower, upper = [], []
    for p in pts:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 0:
            lower.pop()
        lower.append(p)
    for p in reversed(pts):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 0:
            upper.pop()
        upper.append(p)
    return lower[:-1] + upper[:-1]


def convex_area_like_skimage(mask: np.ndarray) -> int:
    """np.sum(skimage.morphology.convex_hull_image(mask)) for a 2-D region image (0.18.3):
    hull of the pixel centres, each hull vertex expanded to the 4 mid-edge points (+-0.5), hull
    again, then every pixel centre tested for membership in the closed hull polygon."""
    rr, cc = np.nonzero(mask)
    base = _hull_ccw(zip((2 * rr).tolist(), (2 * cc).tolist()))           # doubled integer coordinates
    cand = []
    for r2, c2 in base:
        cand += [(r2, c2 - 1), (r2, c2 + 1), (r2 - 1, c2), (r2 + 1, c2)]
    hull = _hull_ccw(cand)
    # skimage's point_in_polygon (>= 0.17) labels boundary points VERTEX / EDGE and the mask keeps every
    # non-zero label: membership in the CLOSED polygon.  All coordinates are half-integers, so the
    # test is exact in doubled integer arithmetic: inside <=> on the left of (or on) every CCW edge.
    h, w = mask.shape
    x2 = 2 * np.arange(h, dtype=np.int64)[:, None]
    y2 = 2 * np.arange(w, dtype=np.int64)[None, :]
    inside = np.ones((h, w), bool)
    n = len(hull)
    for i in range(n):
        ax, ay = hull[i]
        bx, by = hull[(i + 1) % n]
        inside &= ((bx - ax) * (y2 - ay) - (by - ay) * (x2 - ax)) >= 0
    return int(inside.sum())


def region_props_like_skimage(lab: np.ndarray, k: int, intensity: np.ndarray) -> dict:
    """The regionprops the BB predicates read, for label k of a label image."""
    m = lab == k
    rr, cc = np.nonzero(m)
    r0, r1, c0, c1 = rr.min(), rr.max() + 1, cc.min(), cc.max() + 1
    crop = m[r0:r1, c0:c1]
    area = int(m.sum())
    filled = int(ndimage.binary_fill_holes(crop, np.ones((3, 3))).sum())
    convex = convex_area_like_skimage(crop)
    w = intensity[r0:r1, c0:c1] * crop
    m00 = w.sum()
    lr, lc = np.mgrid[0:crop.shape[0], 0:crop.shape[1]]
    return dict(label=k, area=area, filled_area=filled, bbox=(r0, c0, r1, c1), bbox_area=(r1 - r0) * (c1 - c0),
                perimeter=perimeter_like_skimage(crop), convex_area=convex, solidity=area / convex,
                weighted_centroid=((w * lr).sum() / m00 + r0, (w * lc).sum() / m00 + c0))


def bb_predicates(p: dict, dpmm: float, bb_size: float, tolerance: float) -> bool:
    """is_right_size_bb, is_round, is_right_circumference, is_symmetric, is_solid
    (pylinac/metrics/features.py:7-68), ANDed like find_features' filtering loop."""
    bb_area = p["filled_area"] / (dpmm**2)
    larger = np.pi * (bb_size + tolerance) ** 2
    smaller = max((np.pi * (bb_size - tolerance) ** 2, 2))
    if not (smaller < bb_area < larger):
        return False
    fill_ratio = p["filled_area"] / p["bbox_area"]
    if not (np.pi / 4 * 1.2 > fill_ratio > np.pi / 4 * 0.8):
        return False
    per = p["perimeter"] / dpmm
    if not (2 * np.pi * (bb_size + tolerance) > per > 2 * np.pi * (bb_size - tolerance)):
        return False
    ymin, xmin, ymax, xmax = p["bbox"]
    y, x = abs(ymax - ymin), abs(xmax - xmin)
    if x > max(y * 1.05, y + 3) or x < min(y * 0.95, y - 3):
        return False
    return p["solidity"] > 0.9


def find_features_restated(sample: np.ndarray, dpmm: float, radius_mm: float, radius_tolerance_mm: float,
                           max_number: int = 1, min_number: int = 1, min_separation_mm: float = 5):
    """pylinac/metrics/utils.py:66-190 with zero offsets: 50-step threshold sweep, 4-connected label,
    clear_border, predicates, weighted centroids (x, y).  Returns (points [(x, y)], level)."""
    s = stretch(sample, min=0, max=1)
    imin, imax = s.min(), s.max()
    step = (imax - imin) / 50
    cutoff = imin + step
    total, level, found_level = [], 0, -1
    while cutoff <= imax and len(total) < max_number:
        lab, n = ndimage.label(s > cutoff)                                       # connectivity=1
        border = np.unique(np.concatenate([lab[0], lab[-1], lab[:, 0], lab[:, -1]]))
        new = []
        for k in range(1, n + 1):
            if k in border:
                continue
            p = region_props_like_skimage(lab, k, s)
            if bb_predicates(p, dpmm, radius_mm, radius_tolerance_mm):
                new.append((p["weighted_centroid"][1], p["weighted_centroid"][0]))
        for pt in new:   # deduplicate_points_and_boundaries (utils.py:14-38): `combined_points` ALIASES
            # `original_points`, so a point accepted at this level already suppresses the next ones
            if all(math.hypot(pt[0] - q[0], pt[1] - q[1]) >= min_separation_mm * dpmm for q in total):
                total.append(pt)
        if new and found_level < 0:
            found_level = level
        cutoff += step
        level += 1
    if len(total) < min_number:
        raise ValueError(f"Couldn't find the minimum number of disks in the image. Found {len(total)}; required: {min_number}")
    return total, found_level


# --------------------------------------------------------------------------------------
# config #3  Picket fence per-image measurement  (pylinac/picketfence.py:67-100, 745-803, 847-912,
#            1605-1628).  Input: the frame AFTER the constructor's crop/ground/normalize (float64).
# --------------------------------------------------------------------------------------

MLC_ARRANGEMENTS = {  # picketfence.py:103-135
    "MILLENNIUM": [(10, 10), (40, 5), (10, 10)],
    "HD_MILLENNIUM": [(14, 5), (32, 2.5), (14, 5)],
    "BMOD": [(40, 4)],
    "AGILITY": [(80, 5)],
    "MLCI": [(40, 10)],
    "HALCYON_DISTAL": [(28, 10)],
    "HALCYON_PROXIMAL": [(29, 10)],
}


def mlc_arrangement(leaf_arrangement, offset: float = 0):
    """MLCArrangement.__init__/leaves, picketfence.py:67-100 -> (leaf numbers, centers mm, widths mm)."""
    centers, widths = [], []
    rolling_edge = 0
    for leaf_num, width in leaf_arrangement:
        centers += np.arange(start=rolling_edge + width / 2, stop=leaf_num * width + rolling_edge + width / 2,
                             step=width).tolist()
        rolling_edge = centers[-1] + width / 2
        widths += [width] * leaf_num
    mean = np.mean(centers)
    centers = [c - mean + offset for c in centers]
    leaves = np.arange(1, len(centers) + 1, dtype=int)[::-1].tolist()
    return leaves, centers, widths


def pf_leaves_in_view(shape, dpmm, leaves, centers, widths, analysis_width=0.4):
    """PicketFence._leaves_in_view (UP_DOWN), picketfence.py:888-912."""
    pixel_range = shape[0] / 2
    pixel_range -= max(widths[0] * analysis_width, widths[-1] * analysis_width) * dpmm
    return [(n, c, w) for n, c, w in zip(leaves, centers, widths) if abs(c) < pixel_range / dpmm]


def pf_measure(image: np.ndarray, dpmm: float, mlc="MILLENNIUM", num_pickets=None, leaf_analysis_width_ratio=0.4,
               picket_spacing=None, height_threshold=0.5, edge_threshold=1.5, peak_sort="peak_heights",
               required_prominence=0.2, fwxm=50, orientation="UP_DOWN", separate_leaves=False):
    """The per-image measurement loop of PicketFence.analyze (picketfence.py:745-803) for either orientation: returns
    dict(peak_idxs, peak_vals, spacing, leaves [(num, center, width)], position [n_leaves, P] float64 (NaN where the window
    failed _is_mlc_peak_in_window)[, left, right: the two leaf-end positions of separate_leaves, :1616-1623])."""
    ud = orientation == "UP_DOWN"
    leaf_prof = np.mean(image, 0) if ud else np.mean(image, 1)           # :746-749
    leaf_prof = normalize(leaf_prof)                                     # MultiProfile.normalize :752
    peak_idxs, peak_vals = multiprofile_find_fwxm_peaks(                 # :753-759
        leaf_prof, min_distance=0.02, threshold=height_threshold, max_number=num_pickets,
        peak_sort=peak_sort, required_prominence=required_prominence)
    if len(peak_idxs) == 0:
        raise ValueError("No pickets were found.")
    if picket_spacing is None:
        picket_spacing = np.median(np.diff(np.sort(peak_idxs)))          # :766-767
    leaves, centers, widths = mlc_arrangement(MLC_ARRANGEMENTS[mlc])
    across = image.shape[0] if ud else image.shape[1]                     # the axis the leaves are stacked along
    pixel_range = across / 2                                             # _leaves_in_view :888-912
    pixel_range -= max(widths[0] * leaf_analysis_width_ratio, widths[-1] * leaf_analysis_width_ratio) * dpmm
    in_view = [(n, c, w) for n, c, w in zip(leaves, centers, widths) if abs(c) < pixel_range / dpmm]
    pos = np.full((len(in_view), len(peak_idxs)), np.nan)
    left_pos, right_pos = pos.copy(), pos.copy()
    for li, (leaf_num, center, width) in enumerate(in_view):
        leaf_width_px = width * dpmm                                     # _get_mlc_window :859-886
        leaf_center_px = center * dpmm + across / 2
        lo = max(int(leaf_center_px - leaf_width_px / 2), 0)
        hi = min(int(leaf_center_px + leaf_width_px / 2), across)
        for pi, (approx_idx, peak_val) in enumerate(zip(peak_idxs, peak_vals)):
            t0 = max(int(approx_idx - picket_spacing / 2), 0)
            t1 = min(int(approx_idx + picket_spacing / 2), image.shape[1] if ud else image.shape[0])
            window = image[lo:hi, t0:t1] if ud else image[t0:t1, lo:hi]
            std = np.std(window, axis=1 if ud else 0)                    # _is_mlc_peak_in_window :847-857
            if not (np.max(window) > height_threshold * peak_val and max(std) < edge_threshold * np.median(std)):
                continue
            pix_vals = np.median(window, axis=0 if ud else 1)            # MLCValue.get_peak_positions :1605-1628
            vals = ground(pix_vals)                                      # FWXMProfilePhysical(ground=True,
            vals = normalize(vals)                                       #   normalization=MAX)
            le, re, centre, _ = fwxm_edges(vals, fwxm)
            off = max(approx_idx - picket_spacing / 2, 0)
            pos[li, pi] = centre + off
            left_pos[li, pi], right_pos[li, pi] = le + off, re + off     # separate_leaves :1616-1623
    out = dict(peak_idxs=np.asarray(peak_idxs), peak_vals=np.asarray(peak_vals), spacing=float(picket_spacing),
               leaves=in_view, position=pos)
    if separate_leaves:
        out.update(left=left_pos, right=right_pos)
    return out


# --------------------------------------------------------------------------------------
# f2 (first half): skimage.feature.canny 0.18.3 restated on scipy.ndimage (float64 image, mask=None)
# --------------------------------------------------------------------------------------
def canny(image: np.ndarray, sigma=1.0, low_threshold=None, high_threshold=None, use_quantiles=False) -> np.ndarray:
    """skimage/feature/_canny.py (0.18.3; third-party, absent from /root/reference) as called at
    pylinac/planar_imaging.py:577-583.  Gaussian(mode='constant') normalised by the smoothed all-ones mask, Sobel,
    hypot, interior mask, four-sector interpolated non-maximum suppression, thresholds (absolute / np.percentile of the
    magnitude), hysteresis over 8-connected segments of the low mask."""
    # integer images: skimage.filters.gaussian runs img_as_float first (util/dtype.py _convert: unsigned x * (1 / imax),
    # signed (x + 0.5) * (2 / (imax - imin)), float64) and absolute thresholds are divided by dtype_max = imax
    dtype_max = 1.0
    if image.dtype.kind == "u":
        dtype_max = float(np.iinfo(image.dtype).max)
        image = np.multiply(image, 1.0 / dtype_max, dtype=np.float64)
    elif image.dtype.kind == "i":
        info = np.iinfo(image.dtype)
        dtype_max = float(info.max)
        image = np.add(image, 0.5, dtype=np.float64)
        image *= 2 / (float(info.max) - info.min)
    scale = 1.0 if use_quantiles else dtype_max
    low_threshold = 0.1 if low_threshold is None else low_threshold / scale
    high_threshold = 0.2 if high_threshold is None else high_threshold / scale
    g = lambda a: ndimage.gaussian_filter(a, sigma, mode="constant", cval=0, truncate=4.0)   # noqa: E731
    smoothed = g(image.astype(float)) / (g(np.ones(image.shape)) + np.finfo(float).eps)
    js, is_ = ndimage.sobel(smoothed, axis=1), ndimage.sobel(smoothed, axis=0)
    ai, aj, mag = np.abs(is_), np.abs(js), np.hypot(is_, js)
    inner = np.zeros(image.shape, bool)
    inner[1:-1, 1:-1] = True
    inner &= mag > 0
    pad = np.pad(mag, 1)                                     # neighbours of border pixels are never used (inner)
    nb = lambda dr, dc: pad[1 + dr: 1 + dr + image.shape[0], 1 + dc: 1 + dc + image.shape[1]]   # noqa: E731
    same = ((is_ >= 0) & (js >= 0)) | ((is_ <= 0) & (js <= 0))
    opp = ((is_ <= 0) & (js >= 0)) | ((is_ >= 0) & (js <= 0))
    local = np.zeros(image.shape, bool)
    with np.errstate(all="ignore"):
        sectors = [  # (selector, weight, (+c2, +c1), (-c2, -c1))
            (same & (ai >= aj), aj / ai, ((1, 1), (1, 0)), ((-1, -1), (-1, 0))),
            (same & (ai <= aj), ai / aj, ((1, 1), (0, 1)), ((-1, -1), (0, -1))),
            (opp & (ai <= aj), ai / aj, ((-1, 1), (0, 1)), ((1, -1), (0, -1))),
            (opp & (ai >= aj), aj / ai, ((-1, 1), (-1, 0)), ((1, -1), (1, 0))),
        ]
        for sel, wgt, (p2, p1), (m2, m1) in sectors:
            pts = inner & sel
            cp = nb(*p2) * wgt + nb(*p1) * (1 - wgt) <= mag
            cm = nb(*m2) * wgt + nb(*m1) * (1 - wgt) <= mag
            local[pts] = (cp & cm)[pts]
    if use_quantiles:
        high_threshold = np.percentile(mag, 100.0 * high_threshold)
        low_threshold = np.percentile(mag, 100.0 * low_threshold)
    high_mask, low_mask = local & (mag >= high_threshold), local & (mag >= low_threshold)
    labels, count = ndimage.label(low_mask, np.ones((3, 3), bool))
    if count == 0:
        return low_mask
    sums = np.atleast_1d(ndimage.sum(high_mask, labels, np.arange(count, dtype=np.int32) + 1))
    good = np.zeros(count + 1, bool)
    good[1:] = sums > 0
    return good[labels]


def hough_line(image: np.ndarray, theta=None):
    """skimage.transform.hough_line of scikit-image 0.18.3 (compiled _hough_transform; behaviour pinned by probing the
    installed build, tests/golden/hough.npz): votes at round(cos*x + sin*y) + offset with C rounding, 2*offset rows."""
    if theta is None:
        theta = np.linspace(-np.pi / 2, np.pi / 2, 180)
    h, w = image.shape
    off = int(np.ceil(np.sqrt(h * h + w * w)))
    acc = np.zeros((2 * off, len(theta)), np.uint64)
    ys, xs = np.nonzero(image)
    v = np.cos(theta)[None, :] * xs[:, None] + np.sin(theta)[None, :] * ys[:, None]
    idx = (np.sign(v) * np.floor(np.abs(v) + 0.5)).astype(int) + off
    for j in range(len(theta)):
        np.add.at(acc[:, j], idx[:, j], 1)
    return acc, theta, np.linspace(-off, off, 2 * off)


# --------------------------------------------------------------------------------------
# f1: XIM compressed-pixel decoding (pylinac/core/image.py:1180-1296)
# --------------------------------------------------------------------------------------
XIM_DTYPES = {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}


def xim_encode(pixels: np.ndarray):
    """Inverse of the reference's decoder, for building test streams: -> (lookup_table_bytes uint8, byte_stream uint8).
    The stream is what follows the 4-byte buffer size in the file: (W + 1) int32 values (first row and the first pixel of
    the second row), then one 1 / 2 / 4-byte little-endian difference per remaining pixel,
    diff = P[i] - P[i-1] - P[i-W] + P[i-W-1] on the FLAT pixel index; the lookup table holds 2-bit size codes, 4 per
    byte, low bits first."""
    h, w = pixels.shape
    flat = pixels.astype(np.int64).ravel()
    i = np.arange(w + 1, h * w)
    diffs = flat[i] - flat[i - 1] - flat[i - w] + flat[i - w - 1]
    codes = np.where((diffs >= -128) & (diffs <= 127), 0, np.where((diffs >= -32768) & (diffs <= 32767), 1, 2)).astype(np.uint8)
    pad = (-len(codes)) % 4
    c4 = np.concatenate([codes, np.zeros(pad, np.uint8)]).reshape(-1, 4)
    lut = (c4[:, 0] | (c4[:, 1] << 2) | (c4[:, 2] << 4) | (c4[:, 3] << 6)).astype(np.uint8)
    parts = [flat[: w + 1].astype("<i4").tobytes()]
    for d, c in zip(diffs, codes):
        parts.append(int(d).to_bytes(1 << int(c), "little", signed=True))
    return lut, np.frombuffer(b"".join(parts), dtype=np.uint8).copy()


def xim_decode(lookup_table_bytes: np.ndarray, stream: np.ndarray, width: int, height: int, bytes_per_pixel: int):
    """XIM._parse_lookup_table (image.py:1180-1204) + _get_diffs (:1258-1296) + _parse_compressed_bytes (:1206-1256),
    restated without the per-row Python loop: with S_r = row-wise prefix sums of the raw differences, the reference's
    recurrence is P[r] = P[r-1] + S_r + c_r, c_1 = -P[0][0], c_r = c_{r-1} + S_{r-1}[W-1], all in the array dtype's
    wrap-around arithmetic."""
    dtype = XIM_DTYPES[bytes_per_pixel]
    codes = ((lookup_table_bytes[:, None] >> np.array([0, 2, 4, 6])[None, :]) & 3).ravel()
    n = width * height - width - 1
    codes = codes[:n].astype(np.int64)
    if (codes > 2).any():
        raise KeyError(3)                                     # LOOKUP_CONVERSION has no entry for code 3
    sizes = 1 << codes
    offs = (width + 1) * 4 + np.concatenate([[0], np.cumsum(sizes)[:-1]])
    b = stream.astype(np.int64)
    val = np.zeros(n, np.int64)
    for k in range(4):
        use = sizes > k
        val[use] |= b[offs[use] + k] << (8 * k)
    bits = 8 * sizes
    val = np.where(val >= (1 << (bits - 1)), val - (1 << bits), val)          # sign extension
    a = np.zeros(width * height, np.int64)
    a[: width + 1] = stream[: (width + 1) * 4].view("<i4")
    a[width + 1:] = val
    a = a.astype(dtype).astype(np.int64).reshape(height, width)               # stored in the array dtype
    mask = (1 << (8 * bytes_per_pixel)) - 1

    def wrap(v):
        v = v & mask
        return np.where(v >= (mask + 1) // 2, v - (mask + 1), v)

    s_rows = np.cumsum(a[1:], axis=1)
    tot = s_rows[:, -1]
    c = np.concatenate([[-a[0, 0]], -a[0, 0] + np.cumsum(tot[:-1])])
    e = s_rows + c[:, None]
    out = np.vstack([a[0:1], a[0:1] + np.cumsum(e, axis=0)])
    return wrap(out).astype(dtype)


def xim_file_bytes(pixels: np.ndarray, bytes_per_pixel: int = 4, properties=None, histogram=()) -> bytes:
    """A complete compressed .xim file in the layout the reference's reader walks (image.py:1123-1178): header, lookup
    table, pixel buffer, histogram, typed properties.  Test infrastructure (the reference has no writer)."""
    import struct

    h, w = pixels.shape
    lut, stream = xim_encode(pixels)
    out = [b"VMS.XI\x00\x00", struct.pack("<6i", 1, w, h, 8 * bytes_per_pixel, bytes_per_pixel, 1),
           struct.pack("<i", len(lut)), lut.tobytes(), struct.pack("<i", len(stream)), stream.tobytes(),
           struct.pack("<i", w * h * bytes_per_pixel), struct.pack("<i", len(histogram)),
           struct.pack(f"<{len(histogram)}i", *histogram)]
    props = properties or {}
    out.append(struct.pack("<i", len(props)))
    for name, value in props.items():
        nb = name.encode()
        out.append(struct.pack("<i", len(nb)) + nb)
        if isinstance(value, (int, np.integer)):
            out.append(struct.pack("<ii", 0, int(value)))
        elif isinstance(value, float):
            out.append(struct.pack("<id", 1, value))
        elif isinstance(value, str):
            vb = value.encode()
            out.append(struct.pack("<ii", 2, len(vb)) + vb)
        elif np.asarray(value).dtype.kind == "f":
            v = np.asarray(value, dtype="<f8")
            out.append(struct.pack("<ii", 4, v.nbytes) + v.tobytes())
        else:
            v = np.asarray(value, dtype="<i4")
            out.append(struct.pack("<ii", 5, v.nbytes) + v.tobytes())
    return b"".join(out)


# --------------------------------------------------------------------------------------
# a6: percentile-driven decisions, restated line by line
# --------------------------------------------------------------------------------------
def has_noise(a: np.ndarray) -> bool:
    """pylinac/picketfence.py:229-238."""
    mn, mx = a.min(), a.max()
    near_min, near_max = np.percentile(a, [0.5, 99.5])
    return bool(mx > near_max * 1.25 or ((mn < near_min * 0.75) and (abs(mn - near_min) > 0.1 * (near_max - near_min))))


def pf_orientation(a: np.ndarray) -> str:
    """pylinac/picketfence.py:1501-1526."""
    temp = a.copy()
    temp[temp < np.median(temp)] = np.median(temp)
    row_sum, col_sum = np.sum(temp, 0), np.sum(temp, 1)
    row80, row90 = np.percentile(row_sum, [85, 99])
    col80, col90 = np.percentile(col_sum, [85, 99])
    return "Left-Right" if (row90 - row80) < (col90 - col80) else "Up-Down"


def corners_inverted(a: np.ndarray, box_size=20, position=(0.0, 0.0)) -> bool:
    """pylinac/core/image.py:868-897 (the comparison; the reference then inverts)."""
    row_pos = max(int(position[0] * a.shape[0]), 1)
    col_pos = max(int(position[1] * a.shape[1]), 1)
    boxes = (a[row_pos: row_pos + box_size, col_pos: col_pos + box_size],
             a[-row_pos - box_size: -row_pos, col_pos: col_pos + box_size],
             a[row_pos: row_pos + box_size, -col_pos - box_size: -col_pos],
             a[-row_pos - box_size: -row_pos, -col_pos - box_size: -col_pos])
    return bool(np.mean(boxes) > np.mean(a.flatten()))


def clean_edges(a: np.ndarray, window_size=2) -> np.ndarray:
    """pylinac/winston_lutz.py:1109-1133."""
    safety_stop = np.min(a.shape) / 10
    while safety_stop > 0:
        near_min, near_max = np.percentile(a, [5, 99.5])
        rng = near_max - near_min
        ws = window_size
        edge = np.concatenate((a[:ws, :].flatten(), a[:, :ws].flatten(), a[-ws:, :].flatten(), a[:, -ws:].flatten()))
        if not (edge.min() < (near_min - rng / 10) or edge.max() > (near_max + rng / 10)):
            break
        a = a[ws:-ws, ws:-ws]
        safety_stop -= 1
    return a


# --------------------------------------------------------------------------------------
# a15: BaseImage.gamma (Bakai gamma map), pylinac/core/image.py:929-1016
# --------------------------------------------------------------------------------------
def bakai_gamma(reference: np.ndarray, comparison: np.ndarray, dpmm: float, doseTA=1, distTA=1, threshold=0.1,
                ground_images=True, normalize_images=True) -> np.ndarray:
    """image.py:981-1016 restated on plain arrays: inversion check (:899-926), ground, normalize, NaN below the dose
    threshold, float32 Sobel gradient, |comp - ref| / sqrt(doseTA^2 + distTA_px^2 * grad^2)."""
    def prep(a):
        a = np.array(a, copy=True)
        p5, p50, p95 = np.percentile(a, [5, 50, 95])
        if abs(p50 - p5) > abs(p50 - p95):
            a = invert(a)
        if ground_images:
            a = ground(a)
        if normalize_images:
            a = normalize(a)
        return a

    ref, comp = prep(reference), prep(comparison)
    ref[ref < threshold * np.max(ref)] = np.nan
    img_x = ndimage.sobel(ref.astype(np.float32), 1)
    img_y = ndimage.sobel(ref.astype(np.float32), 0)
    grad = np.hypot(img_x, img_y)
    denominator = np.sqrt(((doseTA / 100.0) ** 2) + (((dpmm * distTA) ** 2) * (grad**2)))
    return np.abs(comp - ref) / denominator


# --------------------------------------------------------------------------------------
# f4 (gamma part): gamma_2d restated with array operations (the reference loops over pixels in Python)
# --------------------------------------------------------------------------------------
def gamma_2d(reference, evaluation, dose_to_agreement=1, distance_to_agreement=1, gamma_cap_value=2,
             global_dose=True, dose_threshold=5, fill_value=np.nan):
    """pylinac/core/gamma.py:229-330: same operations per pixel, the disk offsets iterated instead of the pixels."""
    if reference.ndim != 2 or evaluation.ndim != 2:
        raise ValueError("Reference and evaluation arrays must be 2D.")
    dose_ta = dose_to_agreement / 100 * (reference.max() if global_dose else reference)
    with np.errstate(all="ignore"):
        ev = np.pad(evaluation / dose_ta, distance_to_agreement, mode="edge")
        rn = reference / dose_ta
    r = distance_to_agreement + 1                                 # skimage.draw.disk((0, 0), DTA + 1)
    grid = np.arange(-r, r + 1, dtype=float)
    rr, cc = np.nonzero((grid[:, None] / r) ** 2 + (grid[None, :] / r) ** 2 < 1)
    rr, cc = rr - r, cc - r
    h, w = reference.shape
    best = np.full(reference.shape, np.nan)
    for dr, dc in zip(rr, cc):
        d2 = (dr / distance_to_agreement) ** 2 + (dc / distance_to_agreement) ** 2
        roi = ev[distance_to_agreement + dr: distance_to_agreement + dr + h, distance_to_agreement + dc: distance_to_agreement + dc + w]
        dd = roi - rn
        best = np.fmin(best, d2 + dd * dd)                        # nanmin over the disk
    gamma = np.full(reference.shape, float(gamma_cap_value))
    with np.errstate(all="ignore"):
        calc = ~(best >= gamma_cap_value**2)
        gamma[calc] = np.sqrt(best[calc])
        gamma[np.isnan(rn) | (rn < dose_threshold / 100)] = fill_value
    return gamma


def gamma_1d(reference, evaluation, reference_coordinates=None, evaluation_coordinates=None, dose_to_agreement=1,
             distance_to_agreement=1, gamma_cap_value=2, global_dose=True, dose_threshold=5, resolution_factor=3,
             fill_value=np.nan):
    """pylinac/core/gamma.py:333-455 (validation :399-428 omitted): per reference point a linspace of search
    positions, the evaluation read through a linear extrapolating interp1d, Gamma minimum capped."""
    from scipy.interpolate import interp1d

    if reference_coordinates is None:
        reference_coordinates = np.arange(len(reference), dtype=float)
    if evaluation_coordinates is None:
        evaluation_coordinates = np.arange(len(evaluation), dtype=float)
    threshold = reference.max() / 100 * dose_threshold
    dose_ta = dose_to_agreement / 100 * reference.max()
    f = interp1d(evaluation_coordinates, evaluation, kind="linear", fill_value="extrapolate")
    vals, xs, gamma = [], [], []
    for ref_x, ref_point in zip(reference_coordinates, reference):
        if ref_point < threshold:
            gamma.append(fill_value)
            continue
        eval_xs = np.linspace(ref_x - distance_to_agreement, ref_x + distance_to_agreement,
                              num=int(distance_to_agreement * resolution_factor * 2 + 1))
        eval_vals = f(eval_xs)
        xs.extend(eval_xs)
        vals.extend(eval_vals)
        cgs = []
        for eval_x, eval_point in zip(eval_xs, eval_vals):
            dist = abs(ref_x - eval_x)
            dose = float(ref_point) - float(eval_point)
            if not global_dose:
                dose_ta = dose_to_agreement / 100 * ref_point
            cgs.append(math.sqrt(dist**2 / distance_to_agreement**2 + dose**2 / dose_ta**2))
        gamma.append(min(min(cgs), gamma_cap_value))
    return np.asarray(gamma), np.asarray(vals), np.asarray(xs)


def gamma_geometric(reference, evaluation, reference_coordinates=None, evaluation_coordinates=None, dose_to_agreement=1,
                    distance_to_agreement=1, gamma_cap_value=2, dose_threshold=5, fill_value=np.nan):
    """pylinac/core/gamma.py:105-227 (+ _construct_matrices / _calculate_weights / _compute_distance :16-102; validation
    omitted): per reference point the smallest distance to the segments of the normalised evaluation curve between the
    samples nearest (x - DTA) and (x + DTA), projection weights through the 1 x 1 pseudo-inverse."""
    if reference_coordinates is None:
        reference_coordinates = np.arange(len(reference), dtype=float)
    if evaluation_coordinates is None:
        evaluation_coordinates = np.arange(len(evaluation), dtype=float)
    threshold = float(dose_threshold) / float(dose_to_agreement)
    nref = reference.astype(float) * 100 / (reference.max() * dose_to_agreement)
    nev = evaluation.astype(float) * 100 / (reference.max() * dose_to_agreement)
    nrx = np.asarray(reference_coordinates) / distance_to_agreement
    nex = np.asarray(evaluation_coordinates) / distance_to_agreement
    decreasing = bool(np.all(np.diff(nex) < 0))
    gamma = np.full(len(reference), fill_value)
    for idx, (rx, rp) in enumerate(zip(nrx, nref)):
        if rp < threshold:
            continue
        left_d, right_d = np.abs(nex - (rx - distance_to_agreement)), np.abs(nex - (rx + distance_to_agreement))
        if decreasing:
            left_d, right_d = right_d, left_d
        lo = max(np.argmin(left_d) - 1, 0)
        hi = min(np.argmin(right_d) + 1, len(nev) - 1)
        p = np.array([rx, rp])
        dists = []
        for j in range(lo, hi):
            v1, v2 = np.array([nex[j], nev[j]]), np.array([nex[j + 1], nev[j + 1]])
            V, P = (v1 - v2)[:, None], p - v2                  # V: 2 x 1
            w = np.dot(np.linalg.pinv(np.dot(V.T, V)), np.dot(V.T, P))
            weights = np.append(w, 1 - np.sum(w))
            if np.any(weights < 0):
                dists.append(min(math.dist(p, v1), math.dist(p, v2)))
            else:
                dists.append(np.linalg.norm(p - np.sum(weights[:, None] * np.array([v1, v2]), axis=0)))
        gamma[idx] = min(min(dists), gamma_cap_value)
    return gamma


# --------------------------------------------------------------------------------------
# f3: DiskROI statistics (skimage.draw.disk restated by disk_mask_like_skimage)
# --------------------------------------------------------------------------------------
def disk_roi_stats(arr: np.ndarray, cx: float, cy: float, radius: float) -> np.ndarray:
    """pylinac/core/roi.py:104-138: circle_mask() = array[draw.disk((cy, cx), radius)] -> count, mean, std, min, max,
    median (pixel_value)."""
    v = arr[disk_mask_like_skimage((cy, cx), radius, arr.shape).astype(bool)]
    return np.array([v.size, np.mean(v), np.std(v), np.min(v), np.max(v), np.median(v)], dtype=float)


# --------------------------------------------------------------------------------------
# a13 (fields): GlobalSizedFieldLocator -- skimage 0.18.3 regionprops restated, scipy labelling
# --------------------------------------------------------------------------------------
def find_fields_restated(sample: np.ndarray, dpmm: float, field_width_mm: float, field_height_mm: float,
                         field_tolerance_mm: float, max_number: int | None = None, min_number: int = 1):
    """pylinac/metrics/image.py:817-897 (from_physical): threshold ladder from 10 % height, clear_border(3),
    8-connected label, is_right_square_perimeter + is_right_area_square (metrics/features.py:69-101), unweighted
    centroid (x, y), de-duplication radius max(equivalent_diameter) / dpmm.  Returns (points, first level)."""
    max_number = max_number or 1e6
    imin, imax = sample.min(), sample.max()
    step = (imax - imin) / 50
    cutoff = imin + step * 5
    fields, level, found = [], 0, -1
    fw, fh, ft = field_width_mm, field_height_mm, field_tolerance_mm
    while cutoff <= imax and len(fields) < max_number:
        bw = clear_border_like_skimage(sample > cutoff, 3)
        lab, n = ndimage.label(bw, structure=np.ones((3, 3)))
        hits = []
        for k in range(1, n + 1):
            rr, cc = np.nonzero(lab == k)
            crop = (lab == k)[rr.min(): rr.max() + 1, cc.min(): cc.max() + 1]
            filled = ndimage.binary_fill_holes(crop, structure=np.ones((3, 3))).sum()
            per_mm = perimeter_like_skimage(crop) / dpmm
            upper = 1.20 * 2 * (fw + ft) + 2 * (fh + ft)
            lower = 2 * (fw - ft) + 2 * (fh - ft)
            area_mm = filled / (dpmm**2)
            if upper > per_mm > lower and (fw - ft) * (fh - ft) < area_mm < (fw + ft) * (fh + ft):
                hits.append((cc.mean(), rr.mean(), math.sqrt(4 * len(rr) / math.pi)))
        if hits:
            sep = max(hh[2] for hh in hits) / dpmm
            for x, y, _ in hits:
                if all(math.hypot(x - q[0], y - q[1]) >= sep for q in fields):
                    fields.append((x, y))
                    if found < 0:
                        found = level
        cutoff += step
        level += 1
    if len(fields) < min_number:
        raise ValueError(f"Couldn't find the minimum number of fields in the image. Found {len(fields)}; required: {min_number}")
    return fields, found


# --------------------------------------------------------------------------------------
# a11: SingleProfile (FWHM edge method) -- scipy interp1d / find_peaks / linregress / minimize as the reference
# --------------------------------------------------------------------------------------
def hill_func(x, a, b, c, d):
    """pylinac/core/hill.py:67-78"""
    return a + (b - a) / (1.0 + (c / x) ** d)


def hill_fit(x_data, y_data) -> np.ndarray:
    """Hill.fit (pylinac/core/hill.py:19-31)."""
    from scipy.optimize import curve_fit

    params, _ = curve_fit(hill_func, x_data, y_data, p0=(min(y_data), max(y_data), np.median(x_data), 0))
    return params


def hill_inflection(p) -> float:
    """Hill.inflection_idx (hill.py:33-38)"""
    return p[2] * math.pow((p[3] - 1) / (p[3] + 1), 1 / p[3])


def hill_y(p, x) -> float:
    """Hill.y (hill.py:61-65)"""
    return p[0] + (p[1] - p[0]) / (1 + (p[2] / x) ** p[3])


def hill_x(p, y) -> float:
    """Hill.x (hill.py:55-59)"""
    return p[2] * math.pow((y - p[0]) / (p[1] - y), 1 / p[3])


def hill_gradient(p, x) -> float:
    """Hill.gradient_at (hill.py:47-53)"""
    cxd = math.pow(p[2] / x, p[3])
    return (p[1] - p[0]) * p[3] * cxd / (math.pow(cxd + 1, 2) * x)


class SingleProfileRestated:
    """pylinac/core/profile.py:1118-1633 restated for Edge.FWHM: __init__ :1125-1215, _interpolate :1306-1360,
    _normalize :1362-1371, fwxm_data :1411-1461, _sample_points_in_physical_window :1237-1283,
    field_data :1463-1633, field_calculation :1910-1937."""

    def __init__(self, values, dpmm=None, interpolation="Linear", ground=True, interpolation_resolution_mm=0.1,
                 interpolation_factor=10, normalization_method="Beam center", x_values=None,
                 centering="Beam center", edge_detection_method="FWHM", edge_smoothing_ratio=0.003,
                 hill_window_ratio=0.1):
        from scipy.interpolate import interp1d

        self._hill_window_ratio = hill_window_ratio

        values = np.asarray(values, dtype=float)
        self.dpmm, self._centering = dpmm, centering
        self._edge, self._smooth = edge_detection_method, edge_smoothing_ratio
        if x_values is None:
            x_values = np.array(range(len(values)))
        if np.diff(x_values).min() < 0:
            raise ValueError("Profile values must be monotonically increasing")
        if interpolation is None:
            fitted, x_indices = values.copy(), np.asarray(x_values)
        else:
            if dpmm is not None:
                samples = int(round(len(x_values) / (dpmm * interpolation_resolution_mm)))
            else:
                samples = int(round(len(x_values) * interpolation_factor))
            offset = 0.5 - 1 / (2 * (samples / len(values)))
            f = interp1d(x_values, values, kind="linear" if interpolation == "Linear" else "cubic",
                         bounds_error=False, fill_value="extrapolate")
            x_indices = np.linspace(x_values[0] - offset, x_values[-1] + offset, num=samples)
            fitted = f(x_indices)
        self.x_indices = x_indices
        self._x = interp1d(list(range(len(x_indices))), x_indices)
        if ground:
            fitted = fitted - fitted.min()
        self._set(fitted)
        if normalization_method == "Max":
            self._set(fitted / fitted.max())
        elif normalization_method == "Geometric center":
            n = len(fitted)
            cv = (fitted[n // 2] + fitted[n // 2 - 1]) / 2.0 if n % 2 == 0 else fitted[(n - 1) // 2]
            self._set(fitted / cv)
        elif normalization_method == "Beam center":
            self._set(fitted / self.beam_center()["value (@rounded)"])

    def _set(self, v):
        from scipy.interpolate import interp1d

        self.values = v
        self._y = interp1d(self.x_indices, v, bounds_error=False, fill_value="extrapolate")

    def _yat(self, loc):
        y = self._y(loc)
        return float(y) if np.size(loc) == 1 else y

    def fwxm_data(self, x=50):
        _, props = find_peaks(self.values, fwxm_height=x / 100, max_number=1)
        left, right = float(self._x(props["left_ips"][0])), float(self._x(props["right_ips"][0]))
        center = (right - left) / 2 + left
        return {"width (exact)": right - left, "center index (exact)": center,
                "center value (@rounded)": self._yat(int(round(center))), "left index (exact)": left,
                "left value (@rounded)": self._yat(int(round(left))), "right index (exact)": right,
                "right value (@rounded)": self._yat(int(round(right))),
                "field values": self._yat(self.x_indices[int(round(left)): int(round(right))])}

    def inflection_data(self):
        """profile.py:1635-1670 (INFLECTION_DERIVATIVE; MultiProfile.find_peaks / find_valleys :2050-2133)."""
        d1 = np.gradient(ndimage.gaussian_filter1d(self.values, sigma=self._smooth * len(self.values)))
        pk, _ = find_peaks(d1, threshold=0.8, peak_separation=0.05)
        vl, _ = find_peaks(-d1, threshold=0.8, peak_separation=0.05)
        left, right = float(self._x(pk[0])), float(self._x(vl[-1]))
        if self._edge != "Inflection Hill":
            return {"left index (exact)": left, "right index (exact)": right,
                    "left value (@rounded)": self._yat(int(round(left))), "left value (@exact)": self._yat(left),
                    "right value (@rounded)": self._yat(int(round(right))), "right value (@exact)": self._yat(right)}
        # profile.py:1675-1721: a Hill function fitted to a window about each derivative extremum
        # (pylinac/core/hill.py:19-36; scipy.optimize.curve_fit = MINPACK lmdif, third-party, called like the reference)
        half = int(round(self._hill_window_ratio * abs(right - left) / 2))
        xl = np.array([x for x in np.arange(left - half, left + half) if x >= 0])
        lp = hill_fit(xl, self._yat(xl))
        xr = np.array([x for x in np.arange(right - half, right + half) if x < len(d1)])
        rp = hill_fit(xr, self._yat(xr))
        li, ri = hill_inflection(lp), hill_inflection(rp)
        return {"left index (rounded)": int(round(li)), "left index (exact)": li,
                "right index (rounded)": int(round(ri)), "right index (exact)": ri,
                "left value (@exact)": hill_y(lp, li), "right value (@exact)": hill_y(rp, ri),
                "left Hill params": lp, "right Hill params": rp}

    def penumbra(self, lower=20, upper=80):
        """profile.py:1723-1908."""
        if lower > upper:
            raise ValueError("Upper penumbra value must be larger than the lower penumbra value")
        if self._edge == "FWHM":
            up, lo = self.fwxm_data(upper), self.fwxm_data(lower)
            data = {f"left {lower}% index (exact)": lo["left index (exact)"],
                    f"left {lower}% value (@rounded)": lo["left value (@rounded)"],
                    f"left {upper}% index (exact)": up["left index (exact)"],
                    f"left {upper}% value (@rounded)": up["left value (@rounded)"],
                    f"right {lower}% index (exact)": lo["right index (exact)"],
                    f"right {lower}% value (@rounded)": lo["right value (@rounded)"],
                    f"right {upper}% index (exact)": up["right index (exact)"],
                    f"right {upper}% value (@rounded)": up["right value (@rounded)"],
                    "left values": self.values[int(round(lo["left index (exact)"])):int(round(up["left index (exact)"]))],
                    "right values": self.values[int(round(up["right index (exact)"])):int(round(lo["right index (exact)"]))],
                    "left penumbra width (exact)": abs(up["left index (exact)"] - lo["left index (exact)"]),
                    "right penumbra width (exact)": abs(up["right index (exact)"] - lo["right index (exact)"])}
        elif self._edge == "Inflection Derivative":
            infl = self.inflection_data()
            vmax = self.values.max()
            ll = self.fwxm_data(max(infl["left value (@exact)"] / vmax * lower / 50 * 100, 1))
            ul = self.fwxm_data(min(infl["left value (@exact)"] / vmax * upper / 50 * 100, 99))
            lr = self.fwxm_data(max(infl["right value (@exact)"] / vmax * lower / 50 * 100, 1))
            ur = self.fwxm_data(min(infl["right value (@exact)"] / vmax * upper / 50 * 100, 99))
            data = {f"left {lower}% index (exact)": ll["left index (exact)"],
                    f"left {upper}% index (exact)": ul["left index (exact)"],
                    f"right {lower}% index (exact)": lr["right index (exact)"],
                    f"right {upper}% index (exact)": ur["right index (exact)"],
                    "left values": self._yat(np.arange(int(round(ll["left index (exact)"])), int(round(ul["left index (exact)"])))),
                    "right values": self._yat(np.arange(int(round(ur["right index (exact)"])), int(round(lr["right index (exact)"])))),
                    "left penumbra width (exact)": abs(ul["left index (exact)"] - ll["left index (exact)"]),
                    "right penumbra width (exact)": abs(ur["right index (exact)"] - lr["right index (exact)"])}
        else:
            infl = self.inflection_data()
            lp, rp = infl["left Hill params"], infl["right Hill params"]
            llv, ulv = infl["left value (@exact)"] * lower / 50, infl["left value (@exact)"] * upper / 50
            lrv, urv = infl["right value (@exact)"] * lower / 50, infl["right value (@exact)"] * upper / 50
            lli, uli, lri, uri = hill_x(lp, llv), hill_x(lp, ulv), hill_x(rp, lrv), hill_x(rp, urv)
            data = {f"left {lower}% index (exact)": lli, f"left {lower}% value (exact)": llv,
                    f"left {upper}% index (exact)": uli, f"left {upper}% value (exact)": ulv,
                    f"right {lower}% index (exact)": lri, f"right {lower}% value (exact)": lrv,
                    f"right {upper}% index (exact)": uri, f"right {upper}% value (exact)": urv,
                    "left values": self.values[int(round(lli)):int(round(uli))],
                    "right values": self.values[int(round(uri)):int(round(lri))],
                    "left penumbra width (exact)": abs(uli - lli), "right penumbra width (exact)": abs(uri - lri),
                    "left gradient (exact)": hill_gradient(lp, infl["left index (exact)"]),
                    "right gradient (exact)": hill_gradient(rp, infl["right index (exact)"])}
            if self.dpmm:
                data["left gradient (exact) %/mm"] = data["left gradient (exact)"] * self.dpmm * 100
                data["right gradient (exact) %/mm"] = data["right gradient (exact)"] * self.dpmm * 100
        if self.dpmm:
            data["left penumbra width (exact) mm"] = data["left penumbra width (exact)"] / self.dpmm
            data["right penumbra width (exact) mm"] = data["right penumbra width (exact)"] / self.dpmm
        return data

    def beam_center(self):
        """profile.py:1390-1409."""
        if self._edge == "FWHM":
            d = self.fwxm_data(50)
            return {"index (exact)": d["center index (exact)"], "value (@rounded)": d["center value (@rounded)"]}
        infl = self.inflection_data()
        mid = infl["left index (exact)"] + (infl["right index (exact)"] - infl["left index (exact)"]) / 2
        return {"index (exact)": mid, "value (@rounded)": self._yat(int(round(mid)))}

    def _window(self, a, b):
        lower, upper = sorted((a, b))
        xi = self.x_indices
        start = int(np.searchsorted(xi, lower, side="left"))
        stop = int(np.searchsorted(xi, upper, side="right"))
        if stop - start < 3:
            li, ri = int(np.abs(xi - lower).argmin()), int(np.abs(xi - upper).argmin())
            start, stop = min(li, ri), max(li, ri) + 1
        if stop - start < 3:
            c = int(np.abs(xi - (lower + upper) / 2).argmin())
            start = max(0, c - 1)
            stop = min(len(xi), start + 3)
            start = max(0, stop - 3)
        return xi[start:stop], self._yat(xi[start:stop])

    def field_data(self, in_field_ratio=0.8, slope_exclusion_ratio=0.2):
        from scipy.optimize import minimize
        from scipy.stats import linregress

        if slope_exclusion_ratio >= in_field_ratio:
            raise ValueError("The exclusion region must be smaller than the field ratio")
        if self._edge == "FWHM":
            d = self.fwxm_data(50)
            beam, full = d["center index (exact)"], d["width (exact)"]
        else:
            d = self.inflection_data()
            beam, full = self.beam_center()["index (exact)"], d["right index (exact)"] - d["left index (exact)"]
        cax = float(self._x((len(self.values) - 1) / 2.0))
        center = cax if self._centering == "Geometric center" else beam
        fl, fr = center - in_field_ratio * full / 2, center + in_field_ratio * full / 2
        fw = fr - fl
        il, ir = center - slope_exclusion_ratio * fw / 2, center + slope_exclusion_ratio * fw / 2
        lfit = linregress(*self._window(fl, il))
        rfit = linregress(*self._window(ir, fr))
        tx, ty = self._window(il, ir)
        prm = np.polyfit(tx, ty, deg=2)
        width = abs(tx[-1] - tx[0])
        mf = minimize(lambda x: -(prm[0] * (x**2) + prm[1] * x + prm[2]), x0=(tx[0] + width / 2,),
                      bounds=((tx[0], tx[-1]),))
        shifted = self.x_indices + (center - int(round(center)))
        i0, i1 = int(np.abs(shifted - fl).argmin()), int(np.abs(shifted - fr).argmin())
        return {"width (exact)": fw, "beam center index (exact)": beam,
                "beam center value (@rounded)": self._yat(round(beam)), "cax index (exact)": cax,
                "cax value (@rounded)": self._yat(round(cax)), "left index (exact)": fl,
                "left value (@rounded)": self._yat(round(fl)), "left slope": lfit.slope,
                "left intercept": lfit.intercept, "right slope": rfit.slope, "right intercept": rfit.intercept,
                "left inner index (exact)": il, "right inner index (exact)": ir,
                '"top" index (exact)': mf.x[0], '"top" value (@exact)': -mf.fun, "top params": prm,
                "right index (exact)": fr, "right value (@rounded)": self._yat(round(fr)),
                "field values": self._yat(shifted[i0: i1 + 1])}

    def field_calculation(self, in_field_ratio=0.8, calculation="mean", slope_exclusion_ratio=0.2):
        fv = self.field_data(in_field_ratio, slope_exclusion_ratio)["field values"]
        return {"mean": fv.mean, "median": lambda: float(np.median(fv)), "max": fv.max, "min": fv.min}[calculation]()


# --------------------------------------------------------------------------------------
# a18: noise power spectrum / radial average / ESF-FFT MTF (numpy pocketfft, as the reference calls it)
# --------------------------------------------------------------------------------------
def radial_average(arr: np.ndarray) -> np.ndarray:
    """pylinac/core/nps.py:12-32."""
    center = np.floor(np.array(arr.shape) / 2)
    y, x = np.indices(arr.shape)
    r = np.sqrt((x - center[1]) ** 2 + (y - center[0]) ** 2).astype(int)
    tbin = np.bincount(r.ravel(), arr.ravel())
    nr = np.bincount(r.ravel())
    nonzero = nr != 0
    radial_mean = np.zeros(nr.shape)
    radial_mean[nonzero] = tbin[nonzero] / nr[nonzero]
    return radial_mean


def noise_power_spectrum_2d(pixel_size: float, rois) -> np.ndarray:
    """pylinac/core/nps.py:35-79."""
    length = min(min(roi.shape) for roi in rois)
    ffts = np.zeros((length, length, len(rois)))
    for idx, roi in enumerate(rois):
        rroi = roi[0:length, 0:length]
        b = np.abs(np.fft.fft2(rroi - np.mean(rroi))) ** 2
        ffts[:, :, idx] = np.fft.fftshift(b)
    return pixel_size**2 / length**2 * np.mean(ffts, axis=-1)


def average_power(nps1d: np.ndarray) -> float:
    """pylinac/core/nps.py:99-115."""
    return float(np.average(np.linspace(0, 1, len(nps1d)), weights=nps1d))


def max_frequency(nps1d: np.ndarray) -> float:
    """pylinac/core/nps.py:118-121."""
    return float(np.argmax(nps1d) / len(nps1d))


def esf_mtf(esf_list, sample_spacing=None, padding_mode="auto", num_samples=1024, windowing=None, **kwargs):
    """pylinac/core/mtf.py:336-376 + _compute_esf_mtf :448-456 -> (freq, mtf, [mtf per esf]).
    ``windowing``: callable(len, **kwargs); None = boxcar (the reference's default is scipy's hann)."""
    windowing = windowing or (lambda m: np.ones(m))
    len_esf = np.unique([len(e) for e in esf_list])
    if padding_mode == "none":
        if len(len_esf) > 1:
            raise ValueError("If padding_mode='none', all ESF samples must have the same size")
        num_samples = len_esf[0]
    elif padding_mode == "fixed":
        if num_samples < max(len_esf):
            raise ValueError("num_samples must be larger than the largest array")
    elif padding_mode == "auto":
        num_samples = int(max(max(2 ** np.ceil(np.log2(len_esf))), num_samples))
    from scipy.fft import fft, fftfreq   # what pylinac/core/mtf.py:13 imports (not numpy.fft)

    pixel_spacing = 1 if sample_spacing is None else sample_spacing
    freq = fftfreq(num_samples, d=pixel_spacing)[: num_samples // 2]
    each = []
    for e in esf_list:
        lsf = np.gradient(e)
        m = np.abs(fft(lsf * windowing(len(e), **kwargs), num_samples))
        m /= m[0]
        each.append(m[: num_samples // 2])
    return freq, np.mean(np.array(each), axis=0), each


# --------------------------------------------------------------------------------------
# BASELINE config #2 + profile/peak: the pipeline bench.py measures
# --------------------------------------------------------------------------------------

EPID_RECORD_FIELDS = (
    "otsu_threshold", "n_peaks", "peak_idx", "peak_height", "prominence",
    "left_edge", "right_edge", "center", "width",
)


def epid_pipeline_frame(frame: np.ndarray, sigma=5, median_size=3, fwxm_height=50):
    """filter(5,'gaussian') -> filter(3,'median') -> Otsu -> threshold(t,'high') ->
    np.mean(axis=0) profile -> FWXM peak (SURVEY.md section 8d config #2 + a7/a8/a10).
    Returns (thresholded u16 frame, profile f64[W], record f64[9])."""
    g = filter(frame, sigma, "gaussian")
    m = filter(g, median_size, "median")
    t = threshold_otsu(m)
    th = threshold(m, t, "high").astype(frame.dtype)
    prof = np.mean(th, axis=0)
    rec = np.full(len(EPID_RECORD_FIELDS), np.nan)
    rec[0] = float(t)
    try:
        idx, props = find_peaks(prof, fwxm_height=fwxm_height / 100, max_number=1)
    except (IndexError, ValueError):
        idx, props = np.array([], dtype=int), None
    rec[1] = float(len(idx))
    if len(idx):
        left = float(props["left_ips"][0])
        right = float(props["right_ips"][0])
        rec[2] = float(idx[0])
        rec[3] = float(props["peak_heights"][0])
        rec[4] = float(props["prominences"][0])
        rec[5] = left
        rec[6] = right
        rec[7] = abs(right - left) / 2 + left
        rec[8] = max(right, left) - min(right, left)
    return th, prof, rec


def epid_pipeline(frames: np.ndarray, **kw):
    outs, profs, recs = [], [], []
    for f in frames:
        th, prof, rec = epid_pipeline_frame(f, **kw)
        outs.append(th)
        profs.append(prof)
        recs.append(rec)
    return np.stack(outs), np.stack(profs), np.stack(recs)


# --------------------------------------------------------------------------------------
# "next" row f2, second half: Hough peaks and the planar phantom outline (pylinac/planar_imaging.py:300-341, 3136-3179)
# --------------------------------------------------------------------------------------

def prominent_peaks(image: np.ndarray, min_xdistance=1, min_ydistance=1, threshold=None, num_peaks=np.inf):
    """skimage.feature.peak._prominent_peaks (scikit-image 0.18.3; third-party, absent from /root/reference), the
    engine of transform.hough_line_peaks: separable maximum filter with zero padding, candidates = pixels equal to
    their window maximum and above the threshold, 8-connected candidate groups ranked by height (ties: the later
    label first, because the stable ascending sort is reversed), then greedy suppression of a (2*min_y+1) x
    (2*min_x+1) neighbourhood around each accepted group's rounded centroid -- rows do not wrap and row 0 is never
    suppressed (`ycoords_nh > 0`), columns wrap with the row mirrored (angles are periodic).
    -> (heights, x (column) indices, y (row) indices)."""
    img = image.copy()
    rows, cols = img.shape
    if threshold is None:
        threshold = 0.5 * np.max(img)
    img_max = ndimage.maximum_filter1d(img, size=2 * min_ydistance + 1, axis=0, mode="constant", cval=0)
    img_max = ndimage.maximum_filter1d(img_max, size=2 * min_xdistance + 1, axis=1, mode="constant", cval=0)
    candidates = (img == img_max) & (img > threshold)
    lab, nlab = ndimage.label(candidates, structure=np.ones((3, 3)))
    groups = []
    for k in range(1, nlab + 1):
        rr, cc = np.nonzero(lab == k)
        groups.append((img_max[rr, cc].max(), np.round(rr.mean()).astype(int), np.round(cc.mean()).astype(int)))
    groups = sorted(groups, key=lambda g: g[0])[::-1]
    heights, xs, ys = [], [], []
    dy, dx = np.mgrid[-min_ydistance:min_ydistance + 1, -min_xdistance:min_xdistance + 1]
    for _, y0, x0 in groups:
        accum = img_max[y0, x0]
        if not accum > threshold:
            continue
        yn, xn = y0 + dy, x0 + dx
        keep = (yn > 0) & (yn < rows)
        yn, xn = yn[keep], xn[keep]
        low = xn < 0
        yn[low] = rows - yn[low]
        xn[low] += cols
        high = xn >= cols
        yn[high] = rows - yn[high]
        xn[high] -= cols
        img_max[yn, xn] = 0
        heights.append(accum)
        xs.append(x0)
        ys.append(y0)
    heights, xs, ys = np.array(heights), np.array(xs), np.array(ys)
    if num_peaks < len(heights):
        top = np.argsort(heights)[::-1][:num_peaks]
        heights, xs, ys = heights[top], xs[top], ys[top]
    return heights, xs, ys


def hough_line_peaks(hspace, angles, dists, min_distance=9, min_angle=10, threshold=None, num_peaks=np.inf):
    """skimage.transform.hough_line_peaks (0.18.3) as called at pylinac/planar_imaging.py:3160-3166."""
    min_angle = min(min_angle, hspace.shape[1])
    h, a, d = prominent_peaks(hspace, min_xdistance=min_angle, min_ydistance=min_distance, threshold=threshold,
                              num_peaks=num_peaks)
    if a.any():
        return h, angles[a], dists[d]
    return h, np.array([]), np.array([])


def region_bboxes(edges: np.ndarray):
    """measure.label(canny_img) (2-D default: 8-connected) + regionprops(...).bbox for every region, in label order
    (pylinac/planar_imaging.py:585-587) -> int array [n_regions, 4] = (min_row, min_col, max_row, max_col), half-open."""
    lab, nlab = ndimage.label(edges, structure=np.ones((3, 3)))
    out = np.zeros((nlab, 4), dtype=np.int64)
    for k, sl in enumerate(ndimage.find_objects(lab)):
        out[k] = (sl[0].start, sl[1].start, sl[0].stop, sl[1].stop)
    return lab, out


def select_phantom_region(bboxes: np.ndarray, image_shape, phantom_bbox_size_px: float,
                          conditions=("is_centered", "is_right_size"), roi_match_condition="max"):
    """ImagePhantomBase.phantom_ski_region (pylinac/planar_imaging.py:300-341) on the bbox table of the canny regions:
    keep area_bbox > 100, order by area_bbox descending (stable), apply the detection conditions
    (:115-137: is_square rel_tol 0.2 via math.isclose; is_centered np.allclose(rtol 0.3) of the bbox middle against
    the image centre; is_right_size np.isclose(rtol 0.1) against phantom_bbox_size_px), then the biggest / the
    closest-in-size passing region.  -> index into `bboxes` (label - 1).  Raises ValueError like the reference."""
    b = np.asarray(bboxes)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    cand = [i for i in range(len(b)) if area[i] > 100]
    cand = sorted(cand, key=lambda i: area[i], reverse=True)
    # image.center (pylinac/core/image.py:526-533): Point(x = shape[1] / 2 - 0.5, y = shape[0] / 2 - 0.5)
    centre = (image_shape[0] / 2 - 0.5, image_shape[1] / 2 - 0.5)
    passing = []
    for i in cand:
        r0, c0, r1, c1 = (int(v) for v in b[i])
        ok = True
        for cond in conditions:
            if cond == "is_square":
                ok &= math.isclose((r1 - r0) / (c1 - c0), 1, rel_tol=0.2)
            elif cond == "is_centered":
                ok &= bool(np.allclose(((r1 - r0) / 2 + r0, (c1 - c0) / 2 + c0), centre, rtol=0.3))
            elif cond == "is_right_size":
                ok &= bool(np.isclose(area[i], phantom_bbox_size_px, rtol=0.1))
            else:
                raise ValueError(cond)
        if ok:
            passing.append(i)
    if not passing:
        raise ValueError("Unable to find the phantom in the image.")
    if roi_match_condition == "max":
        best = np.argsort([area[i] for i in passing])[-1]
    elif roi_match_condition == "closest":
        best = np.argsort([abs(area[i] - phantom_bbox_size_px) for i in passing])[0]
    else:
        raise ValueError(roi_match_condition)
    return passing[best]


# --------------------------------------------------------------------------------------
# "next" row f3, second half: RectangleROI (pylinac/core/roi.py:481-704, pylinac/core/geometry.py:692-724)
# --------------------------------------------------------------------------------------

def point_in_polygon(xp, yp, x: float, y: float) -> int:
    """skimage._shared.geometry.point_in_polygon (0.18.3; compiled, behaviour pinned by tests/golden/rect.npz): the
    crossing test of Hao et al. 2018 with left and right rays -- 0 outside, 1 inside, 2 on a vertex (|dx|, |dy| <
    1e-12), 3 on an edge (the two rays disagree in parity)."""
    eps = 1e-12
    r_cross = l_cross = 0
    x1, y1 = xp[-1] - x, yp[-1] - y
    for i in range(len(xp)):
        x0, y0 = x1, y1
        x1, y1 = xp[i] - x, yp[i] - y
        if -eps < x0 < eps and -eps < y0 < eps:
            return 2
        if (y0 > 0) != (y1 > 0) and (x1 * y0 - x0 * y1) / (y0 - y1) > 0:
            r_cross += 1
        if (y0 < 0) != (y1 < 0) and (x1 * y0 - x0 * y1) / (y0 - y1) < 0:
            l_cross += 1
    if (r_cross & 1) != (l_cross & 1):
        return 3
    return r_cross & 1


def polygon_pixels(r, c, shape=None):
    """skimage.draw.polygon (scikit-image 0.18.3: _draw._polygon; third-party, compiled, absent from /root/reference --
    behaviour pinned by tests/golden/rect.npz): the bounding box int(max(0, min)) .. int(ceil(max)) clipped to `shape`,
    every integer point kept unless point_in_polygon says "outside" (so vertices and edge points belong to the
    polygon); pixels in raster order."""
    r = np.atleast_1d(np.asarray(r, dtype=np.float64))
    c = np.atleast_1d(np.asarray(c, dtype=np.float64))
    minr, maxr = int(max(0, r.min())), int(math.ceil(r.max()))
    minc, maxc = int(max(0, c.min())), int(math.ceil(c.max()))
    if shape is not None:
        maxr, maxc = min(shape[0] - 1, maxr), min(shape[1] - 1, maxc)
    rr, cc = [], []
    for y in range(minr, maxr + 1):
        for x in range(minc, maxc + 1):
            if point_in_polygon(c, r, float(x), float(y)):
                rr.append(y)
                cc.append(x)
    return np.array(rr, dtype=np.int64), np.array(cc, dtype=np.int64)


def rectangle_vertices(width: float, height: float, cx: float, cy: float, rotation: float = 0.0) -> np.ndarray:
    """Rectangle.vertices (pylinac/core/geometry.py:692-704) -> [4, 2] (x, y): tl, tr, br, bl of the unrotated
    rectangle, rotated by `rotation` degrees about the origin, then translated to the centre (the homogeneous matrix
    product of skimage's EuclideanTransform written out: x' = cos*x - sin*y + tx, y' = sin*x + cos*y + ty)."""
    square = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]])
    scaled = square @ np.diag((width, height)) / 2
    a = np.deg2rad(rotation)
    m = np.array([[math.cos(a), -math.sin(a), cx], [math.sin(a), math.cos(a), cy], [0, 0, 1]])
    src = np.vstack((scaled[:, 0], scaled[:, 1], np.ones(4)))
    dst = src.T @ m.T
    return dst[:, :2] / dst[:, 2:3]


def rectangle_roi_pixels(arr: np.ndarray, width, height, cx, cy, rotation=0.0) -> np.ndarray:
    """RectangleROI.pixels_flat (pylinac/core/roi.py:644-662): the polygon through (bl.x, bl.y - 1), (br.x - 1,
    br.y - 1), (tr.x - 1, tr.y), (tl.x, tl.y), clipped to the array."""
    tl, tr, br, bl = rectangle_vertices(width, height, cx, cy, rotation)
    corners = np.array([(bl[0], bl[1] - 1), (br[0] - 1, br[1] - 1), (tr[0] - 1, tr[1]), (tl[0], tl[1])])
    rr, cc = polygon_pixels(corners[:, 1], corners[:, 0], arr.shape)
    return arr[rr, cc]


def rectangle_roi_stats(arr: np.ndarray, width, height, cx, cy, rotation=0.0) -> np.ndarray:
    """-> count, mean, std, min, max, median of pixels_flat (roi.py:683-704; pixel_value is the MEAN for rectangles)."""
    v = rectangle_roi_pixels(arr, width, height, cx, cy, rotation)
    return np.array([v.size, np.mean(v), np.std(v), np.min(v), np.max(v), np.median(v)], dtype=float)


# --------------------------------------------------------------------------------------
# Starshot per-image measurement (pylinac/starshot.py:197-401, 701-834)
# --------------------------------------------------------------------------------------

class _Pt:
    def __init__(self, x=0.0, y=0.0, z=0.0, idx=None, value=None):
        self.x, self.y, self.z, self.idx, self.value = x, y, z, idx, value


def _line_distance(p1, p2, pt) -> float:
    """Line.distance_to (pylinac/core/geometry.py:569-584)"""
    lp1, lp2, p = (np.array([q.x, q.y, q.z], dtype=float) for q in (p1, p2, pt))
    return np.sqrt(np.sum(np.power(np.cross(lp2 - lp1, lp1 - p), 2))) / np.sqrt(np.sum(np.power(lp2 - lp1, 2)))


class StarshotRestated:
    """Starshot.analyze for an array: check_inversion_by_histogram([4, 50, 96]) (image.py:899-926), ground, start point
    (starshot.py:197-227), StarProfile (:765-814), LineManager (:701-762), Nelder-Mead wobble (:378-401), the retry
    sweep (:306-376) and calculate_angles (:817-834)."""

    def __init__(self, array, dpi, sid=1000):
        self.array = np.array(array)
        self.dpmm = dpi * (sid / 1000) / 25.4
        self.tolerance = 1

    def _get_reasonable_start_point(self):
        a = self.array
        t3 = int(a.shape[0] / 3)
        l3 = int(a.shape[1] / 3)
        central = a[t3:int(t3 * 2), l3:int(l3 * 2)]
        cx = round(fwxm_edges(np.max(central, 0), 80)[2]) + l3
        cy = round(fwxm_edges(np.max(central, 1), 80)[2]) + t3
        return _Pt(cx, cy), np.percentile(central, 90)

    def _star_profile(self, centre, radius_ratio, min_height, fwhm):
        rows, cols = self.array.shape
        radius = min(rows - centre.y, cols - centre.x, centre.y, centre.x) * radius_ratio
        if self.array.shape[1] < radius + centre.x or self.array.shape[0] < radius + centre.y:
            raise ValueError("Array size not large enough to compute profile")
        radii = np.linspace(radius * 0.9, radius * 1.1, 20)
        rads = circle_radians(np.pi * max(radii) * 2 * 3)
        values = collapsed_circle_profile(self.array, (centre.x, centre.y), radius, sampling_ratio=3)
        xs, ys = np.cos(rads) * radius + centre.x, np.sin(rads) * radius + centre.y
        roll = np.where(values == values.min())[0][0]
        values, xs, ys = np.roll(values, -roll), np.roll(xs, -roll), np.roll(ys, -roll)
        values = filter(values, size=0.003, kind="gaussian")
        values = ground(values)
        if fwhm:
            idx, vals = multiprofile_find_fwxm_peaks(values, threshold=min_height, min_distance=0.02)
        else:
            idx, vals = multiprofile_find_peaks(values, threshold=min_height, min_distance=0.02)
        peaks = [_Pt(xs[int(i)], ys[int(i)], idx=i, value=v) for i, v in zip(idx, vals)]
        return values, radius, peaks

    def analyze(self, radius=0.85, min_peak_height=0.25, max_wobble_diameter=2.0, tolerance=1.0, start_point=None,
                fwhm=True, recursive=True, invert_image=False):
        from itertools import product

        from scipy import optimize

        self.tolerance = tolerance
        p_low, p_mid, p_high = (np.percentile(self.array, q) for q in (4, 50, 96))
        if abs(p_mid - p_low) > abs(p_mid - p_high):
            self.array = invert(self.array)
        self.array = ground(self.array)
        if invert_image:
            self.array = invert(self.array)
        auto, local_max = self._get_reasonable_start_point()
        focus = auto if start_point is None else _Pt(start_point[0], start_point[1])
        gen = product(np.append(radius, np.linspace(0.95, 0.1, 10)), np.append(min_peak_height, np.linspace(0.05, 0.95, 10)))
        while True:
            try:
                self.profile, self.radius, self.peaks = self._star_profile(focus, radius, min_peak_height * local_max, fwhm)
                self.centre = focus
                n = len(self.peaks)
                if n < 6 or n % 2:
                    if not recursive:
                        raise RuntimeError("The algorithm was unable to properly detect the radiation lines.")
                    raise ValueError
                half = int(n / 2)
                self.lines = [(self.peaks[k], self.peaks[k + half]) for k in range(half)]
                if any(_line_distance(a, b, focus) > 10 * self.dpmm for a, b in self.lines):
                    raise ValueError
                res = optimize.minimize(
                    lambda p, lines: max(_line_distance(a, b, _Pt(p[0], p[1])) for a, b in lines),
                    np.array([focus.x, focus.y, focus.z]), args=(self.lines,), method="Nelder-Mead",
                    options={"fatol": 0.001})
                self.wobble_radius, self.wobble_centre = res.fun, (res.x[0], res.x[1])
                self.wobble_radius_mm = res.fun / self.dpmm
                near = math.sqrt((res.x[0] - focus.x) ** 2 + (res.x[1] - focus.y) ** 2) < 10 * self.dpmm
                if (self.wobble_radius_mm * 2 < max_wobble_diameter and near) or not recursive:
                    break
                raise ValueError
            except ValueError:
                try:
                    radius, min_peak_height = next(gen)
                except StopIteration:
                    raise RuntimeError("The algorithm was unable to determine a reasonable wobble.")
        self.angles = []
        for a, b in self.lines:
            with np.errstate(divide="ignore"):
                m = (a.y - b.y) / (a.x - b.x)
            phi = math.degrees(math.atan(m)) - 90
            phi = phi - 180 if phi > 90 else (phi + 180 if phi <= -90 else phi)
            self.angles.append(phi)
        self.passed = bool(self.wobble_radius_mm * 2 < self.tolerance)


# --------------------------------------------------------------------------------------
# "next" row f1, DICOM half: _rescale_dicom_values (pylinac/core/image.py:363-389).  The rescale itself is pydicom's
# pixels.apply_rescale (third party: pydicom>=2.0,<3 per the reference's pyproject.toml:40; not installed here), restated from
# its documented behaviour and pinned on the known answers of the reference's own tests (tests_basic/core/test_image.py:131-229:
# pass-through cases, slope * array + intercept, the inversion cases, no overflow when inverting); no pydicom-produced vector.
# --------------------------------------------------------------------------------------

def rescale_dicom_values(unscaled: np.ndarray, rescale_slope=None, rescale_intercept=None,
                         pixel_intensity_relationship_sign=None, raw_pixels=False, invert_pixels=None) -> np.ndarray:
    if raw_pixels:
        return unscaled
    scaled = unscaled
    if rescale_slope is not None and rescale_intercept is not None:
        scaled = unscaled.astype(np.float64) * float(rescale_slope)
        scaled += float(rescale_intercept)
    if invert_pixels or (invert_pixels is None and pixel_intensity_relationship_sign == -1):
        scaled = scaled.max() - scaled + scaled.min()
    return scaled


# --------------------------------------------------------------------------------------
# "next" row f4: the new-style edge profiles (pylinac/core/profile.py:612-740)
# --------------------------------------------------------------------------------------

class EdgeProfileRestated:
    """ProfileBase (:195-344) + InflectionDerivativeProfile.field_edge_idx (:656-670) + HillProfile.field_edge_idx
    (:708-728), calling the same scipy routines as the reference (gaussian_filter1d, interp1d cubic, minimize, curve_fit)."""

    def __init__(self, values, x_values=None, ground_profile=False, normalization=None, edge_smoothing_ratio=0.003,
                 hill_window_ratio=None):
        values = np.asarray(values, dtype=float)
        x_values = np.arange(len(values)) if x_values is None else np.asarray(x_values)
        order = np.argsort(x_values)
        self.x_values, self.values = x_values[order], values[order]
        self.smooth, self.hill_window_ratio = edge_smoothing_ratio, hill_window_ratio
        if ground_profile:
            self.values = ground(self.values)
        if normalization == "Max":
            self.values = self.values / self.values.max()
        elif normalization == "Beam center":
            c = self.center_idx
            self.values = self.values / float(np.interp(c, self.x_values, self.values))

    def x_at_x_idx(self, idx):
        return float(np.interp(idx, np.arange(len(self.x_values)), self.x_values))

    def _inflection(self, side):
        from scipy.interpolate import interp1d
        from scipy.optimize import minimize

        diff = np.gradient(ndimage.gaussian_filter1d(self.values, sigma=self.smooth * len(self.values)))
        f_diff = interp1d(x=self.x_values, y=diff, kind="cubic")
        if side == "left":
            return minimize(lambda x: -f_diff(x), x0=self.x_at_x_idx(np.argmax(diff))).x[0]
        return minimize(f_diff, x0=self.x_at_x_idx(np.argmin(diff))).x[0]

    def field_edge_idx(self, side):
        if self.hill_window_ratio is None:
            return self._inflection(side)
        li, ri = self._inflection("left"), self._inflection("right")
        win = (ri - li) * self.hill_window_ratio
        mid = li if side == "left" else ri
        a = int(np.argmin(np.abs(self.x_values - (mid - win))))
        b = int(np.argmin(np.abs(self.x_values - (mid + win))))
        return hill_inflection(hill_fit(self.x_values[a:b + 1], self.values[a:b + 1]))

    @property
    def center_idx(self):
        left, right = self.field_edge_idx("left"), self.field_edge_idx("right")
        return abs(right - left) / 2 + left

    @property
    def field_width_px(self):
        left, right = self.field_edge_idx("left"), self.field_edge_idx("right")
        return max(right, left) - min(right, left)


def zoom1d_cubic_nearest(values, factor: float, grid_mode: bool = False) -> np.ndarray:
    """scipy.ndimage.zoom(values, factor, order=3, mode="nearest", grid_mode=False) (scipy/ndimage/_interpolation.py zoom,
    src/ni_splines.c) restated for 1-D float64 input, as ProfileBase.as_resampled calls it (pylinac/core/profile.py:
    370-376): 12 edge samples of padding, cubic B-spline prefilter with mirror initialisation, four-tap evaluation at
    i * (n - 1) / (m - 1) + 12 with clamped indices.  Pinned against scipy itself (tests/test_oracle_golden.py)."""
    values = np.asarray(values, dtype=np.float64)
    n = len(values)
    m = int(round(n * factor))
    npad = 12
    c = np.pad(values, npad, mode="edge")
    ln = len(c)
    z = math.sqrt(3.0) - 2.0
    c = c * ((1 - z) * (1 - 1 / z))
    z_i, z_n_1 = z, math.pow(z, ln - 1)
    c0 = c[0] + z_n_1 * c[ln - 1]
    for i in range(1, ln - 1):
        c0 += z_i * (c[i] + z_n_1 * c[ln - 1 - i])
        z_i *= z
    c[0] = c0 / (1 - z_n_1 * z_n_1)
    for i in range(1, ln):
        c[i] += z * c[i - 1]
    c[ln - 1] = (z * c[ln - 2] + c[ln - 1]) * z / (z * z - 1)
    for i in range(ln - 2, -1, -1):
        c[i] = z * (c[i + 1] - c[i])
    if grid_mode:
        zoom = n / m
        shift = 0.5 * zoom - 0.5
    else:
        zoom, shift = ((n - 1) / (m - 1) if m > 1 else 1.0), 0.0
    out = np.empty(m)
    for i in range(m):
        cc = zoom * i + shift + npad
        fl = math.floor(cc)
        y = cc - fl
        zz = 1 - y
        w = [zz * zz * zz / 6.0, (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0, (zz * zz * (zz - 2.0) * 3.0 + 4.0) / 6.0]
        w.append(1.0 - w[0] - w[1] - w[2])
        out[i] = sum(c[min(max(fl - 1 + k, 0), ln - 1)] * w[k] for k in range(4))
    return out


# --------------------------------------------------------------------------------------
# config #4  Winston-Lutz per-image sequence  (pylinac/winston_lutz.py:709-725, 764-806, 1109-1133;
#            pylinac/metrics/image.py:564-612).  CPU baseline of bench.py and restatement pinned to the
#            reference's own sequence (tests/golden/wl.npz, tests/test_oracle_golden.py).
# --------------------------------------------------------------------------------------
def wl_analyze_frame(frame: np.ndarray, dpmm: float, bb_diameter_mm: float = 5.0, low_density: bool = False):
    """check_inversion_by_histogram((0.01, 50, 99.99)) -> _clean_edges -> ground -> normalize -> find_field_centroids ->
    find_bb_centroids.  -> (field_x, field_y, bb_x, bb_y, inverted, crop) in the cleaned frame's coordinates."""
    a = np.asarray(frame)
    p_low, p_mid, p_high = (np.percentile(a, q) for q in (0.01, 50, 99.99))           # image.py:899-926
    inverted = bool(abs(p_mid - p_low) > abs(p_mid - p_high))
    if inverted:
        a = invert(a)
    h0 = a.shape[0]
    a = clean_edges(a)
    crop = (h0 - a.shape[0]) // 2
    fx, fy, _ = wl_field_centroid(a)
    arr = normalize(ground(a))                                                        # winston_lutz.py:711-712
    tol = float(np.interp(bb_diameter_mm, (1.5, 30), (2, 4)))                         # _calculate_bb_tolerance
    win = (40 + bb_diameter_mm) * dpmm
    ex, ey = a.shape[1] / 2, a.shape[0] / 2                                           # from_center_physical, (0, 0) mm
    left, right = max(math.floor(ex - win / 2), 0), math.ceil(ex + win / 2)
    top, bottom = max(math.floor(ey - win / 2), 0), math.ceil(ey + win / 2)
    sample = arr[top:bottom, left:right]
    if not low_density:
        sample = invert(sample)
    try:
        pts, _ = find_features_restated(sample, dpmm, bb_diameter_mm / 2, tol)
        bx, by = pts[0][0] + left, pts[0][1] + top
    except ValueError:
        bx = by = np.nan
    return fx, fy, bx, by, inverted, crop


# --------------------------------------------------------------------------------------
# config #5  CTP528 per slice  (pylinac/ct.py:1511-1580, 3351-3386)
# --------------------------------------------------------------------------------------
CTP528_REGIONS = (
    (0, 0.107, 2, 1, 0.021, 0.1), (0.107, 0.173, 3, 2, 0.01, 0.2), (0.173, 0.236, 4, 3, 0.006, 0.3),
    (0.236, 0.286, 4, 3, 0.00557, 0.4), (0.286, 0.335, 4, 3, 0.004777, 0.5), (0.335, 0.387, 5, 4, 0.00398, 0.6),
    (0.387, 0.434, 5, 4, 0.00358, 0.7), (0.434, 0.479, 5, 4, 0.0027866, 0.8),
)


def ctp528_slice(volume: np.ndarray, s: int, center_xy, mm_per_pixel: float, roll_deg: float = 0.0):
    """combine_surrounding_slices(+-3, "max") -> CollapsedCircleProfile(20 radii, +-4 %, 2x sampling, start pi, ccw) ->
    filter(0.001, "gaussian") -> ground -> per region find_peaks / find_valleys -> relative MTF.
    -> (profile float64 [L], rmtf float64 [8] NaN beyond the regions found)."""
    # combine_surrounding_slices (pylinac/ct.py:3375-3385): dicomstack[q] for q in range(s - 3, s + 4) -- a negative q wraps
    # to the end of the stack, q >= len raises IndexError, exactly like indexing the volume array here
    arr = np.max(np.dstack([volume[q] for q in range(s - 3, s + 4)]), 2)
    radius = 47 / mm_per_pixel
    prof = collapsed_circle_profile(arr, center_xy, radius, start_angle=np.pi + np.deg2rad(roll_deg), ccw=True,
                                    sampling_ratio=2, width_ratio=0.04, num_profiles=20)
    prof = filter(prof, 0.001, "gaussian")
    prof = ground(prof)
    maxs, mins = [], []
    for start, end, npk, nval, spacing, _ in CTP528_REGIONS:
        idx, vals = multiprofile_find_peaks(prof, min_distance=spacing, max_number=npk, search_region=(start, end))
        if len(vals) != npk:
            break
        maxs.append(vals.mean())
        _, vvals = multiprofile_find_valleys(prof, min_distance=spacing, max_number=nval,
                                             search_region=(min(idx), max(idx)))
        mins.append(vvals.mean())
    rmtf = np.full(len(CTP528_REGIONS), np.nan)
    if maxs:
        mtf = [(a - b) / (a + b) for a, b in zip(maxs, mins)]
        rmtf[: len(mtf)] = np.array(mtf) / mtf[0]
    return prof, rmtf


# --------------------------------------------------------------------------------------
# f1, the DICOM half: `DicomImage.__init__` (pylinac/core/image.py:1431-1444) = pydicom's `pixel_array` [+ astype] +
# `_rescale_dicom_values` (:363-389).  PARITY UNPINNED against pydicom itself: pydicom (pyproject.toml:40, >=2.0,<3) is
# in no environment this build reaches and its source is not under /root/reference.  Restated from its published native
# path (pixel_data_handlers/numpy_handler.py `get_pixeldata`, util.py `pixel_dtype` / `get_expected_length` /
# `reshape_pixel_array`, `apply_modality_lut`) and pinned to (a) the layout the reference's OWN writer states
# (pylinac/core/array_utils.py:291-297: `PixelData = array.tobytes()`, BitsAllocated = itemsize * 8, Explicit VR Little
# Endian) through the fixtures of tests/golden/make_dicom_golden.py, (b) the identities the reference's tests state for
# `_rescale_dicom_values` (tests_basic/core/test_image.py:131-200).  The Part-10 walk below is written independently of
# pylinac_amd/dicom.py (a generator over raw elements; tags by number).
# --------------------------------------------------------------------------------------
def _dicom_elements(buf: bytes):
    """(group, element, VR or None, value bytes or None for a skipped undefined-length value, value offset)"""
    import struct

    n, pos = len(buf), 0
    if n >= 132 and buf[128:132] == b"DICM":
        pos = 132
    syntax, explicit, big = None, pos != 0, False
    long_vrs = (b"OB", b"OD", b"OF", b"OL", b"OV", b"OW", b"SQ", b"UC", b"UN", b"UR", b"UT", b"SV", b"UV")

    def read_at(p, expl, be):
        e = ">" if be else "<"
        g, el = struct.unpack_from(e + "HH", buf, p)
        vr = None
        if expl and g != 0xFFFE:
            vr = buf[p + 4:p + 6]
            if vr in long_vrs:
                ln, p = struct.unpack_from(e + "I", buf, p + 8)[0], p + 12
            else:
                ln, p = struct.unpack_from(e + "H", buf, p + 6)[0], p + 8
        else:
            ln, p = struct.unpack_from(e + "I", buf, p + 4)[0], p + 8
        return g, el, vr, ln, p

    def skip_items(p, expl, be):                            # PS3.5 section 7.5
        while True:
            g, el, _, ln, p = read_at(p, expl, be)
            if (g, el) == (0xFFFE, 0xE0DD):
                return p
            if ln != 0xFFFFFFFF:
                p += ln
                continue
            while True:
                g2, el2, _, ln2, q = read_at(p, expl, be)
                if (g2, el2) == (0xFFFE, 0xE00D):
                    p = q
                    break
                p = skip_items(q, expl, be) if ln2 == 0xFFFFFFFF else q + ln2

    in_meta = pos != 0
    while pos + 8 <= n:
        if in_meta and struct.unpack_from("<H", buf, pos)[0] != 0x0002:
            in_meta = False
            explicit = syntax != "1.2.840.10008.1.2"
            big = syntax == "1.2.840.10008.1.2.2"
        g, el, vr, ln, start = read_at(pos, True if in_meta else explicit, False if in_meta else big)
        if ln == 0xFFFFFFFF:
            pos = skip_items(start, explicit, big)
            yield g, el, vr, None, start, big
            continue
        val = buf[start:start + ln]
        if (g, el) == (0x0002, 0x0010):
            syntax = val.decode().rstrip(" \x00")
        yield g, el, vr, val, start, (False if in_meta else big)
        pos = start + ln


def dicom_pixel_array(file_bytes, correct_unused_bits: bool = False):
    """pydicom 2.x `Dataset.pixel_array` for native Pixel Data -> (array, dict of the numeric tags used, value offset).
    numpy_handler.get_pixeldata: `np.frombuffer(PixelData[:expected_len], pixel_dtype)`; util.pixel_dtype: byte order by
    transfer syntax, 'u' / 'i' by PixelRepresentation, BitsAllocated // 8 bytes; reshape_pixel_array: (frames, rows, cols)
    for NumberOfFrames > 1, else (rows, cols).  ``correct_unused_bits`` = pydicom >= 3's default for native data."""
    import struct

    buf = bytes(np.asarray(file_bytes, dtype=np.uint8).tobytes())
    tags, pixel = {}, None
    for g, el, vr, val, start, big in _dicom_elements(buf):
        e = ">" if big else "<"
        if val is None:
            continue
        if (g, el) == (0x7FE0, 0x0010):
            pixel = (val, start, big)
        elif g == 0x0028 and el in (0x0002, 0x0010, 0x0011, 0x0100, 0x0101, 0x0102, 0x0103):
            tags[el] = struct.unpack(e + "H", val[:2])[0]
        elif (g, el) == (0x0028, 0x1041):
            tags[el] = struct.unpack(e + "h", val[:2])[0]
        elif (g, el) in ((0x0028, 0x0008), (0x0028, 0x1052), (0x0028, 0x1053)):
            tags[el] = float(val.decode().strip(" \x00"))
    rows, cols, bits, rep = tags[0x0010], tags[0x0011], tags[0x0100], tags[0x0103]
    frames = int(tags.get(0x0008, 1))
    val, start, big = pixel
    dt = np.dtype((">" if big else "<") + ("i" if rep else "u") + str(bits // 8))
    expected = rows * cols * frames * (bits // 8)
    if len(val) < expected:
        raise ValueError("The length of the pixel data in the dataset doesn't match the expected length")
    arr = np.frombuffer(val[:expected], dtype=dt).astype(dt.newbyteorder("="))
    stored = tags.get(0x0101, bits)
    if correct_unused_bits and stored < bits:
        if rep:
            sh = bits - stored
            arr = np.right_shift(np.left_shift(arr, sh), sh)          # arithmetic shift on a signed dtype
        else:
            arr = arr & arr.dtype.type((1 << stored) - 1)
    arr = arr.reshape((frames, rows, cols) if frames > 1 else (rows, cols))
    return arr, tags, start


def dicom_image_array(file_bytes, dtype=None, raw_pixels: bool = False, invert_pixels=None):
    """`DicomImage.__init__`'s array (image.py:1431-1444)"""
    arr, tags, _ = dicom_pixel_array(file_bytes)
    arr = arr.astype(dtype) if dtype is not None else arr.copy()
    return rescale_dicom_values(arr, tags.get(0x1053), tags.get(0x1052), tags.get(0x1041), raw_pixels, invert_pixels)
