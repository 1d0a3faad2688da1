"""Canny edges on the device (SURVEY.md section 8 "next" row f2): ``skimage.feature.canny`` as pylinac's planar
phantom finder calls it (pylinac/planar_imaging.py:574-588: ``feature.canny(image, sigma, low_threshold,
high_threshold, use_quantiles=True)``; scikit-image 0.18.3 semantics, skimage/feature/_canny.py).

Float64 and uint8 / uint16 / int16 images (the latter through skimage's ``img_as_float`` scaling), with or without ``mask``.  Every stage is a kernel: Gaussian smoothing with zero padding normalised by the
smoothed all-ones frame, the two Sobel gradients, hypot + interpolated non-maximum suppression, exact float64 order
statistics for the quantile thresholds, and hysteresis by 8-connected labelling.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, ops
from ._lib import check


# integer dtypes skimage's img_as_float rescales: (imin, imax)
_IMG_AS_FLOAT = {torch.uint8: (0, 255), torch.uint16: (0, 65535), torch.int16: (-32768, 32767)}


def canny(image, sigma: float = 1.0, low_threshold=None, high_threshold=None, mask=None,
          use_quantiles: bool = False, device=None) -> torch.Tensor:
    """-> uint8 edge map(s) with the shape of ``image`` ([H, W] or a batch [N, H, W])."""
    t = image if isinstance(image, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(image))
    if t.dtype not in _IMG_AS_FLOAT and t.dtype != torch.float64:
        raise TypeError("canny needs a float64, uint8, uint16 or int16 image")
    dev = t.device if t.is_cuda else (torch.device(device) if device is not None
                                      else torch.device("cuda", torch.cuda.current_device()))
    t = t.to(dev).contiguous()
    if t.ndim not in (2, 3):
        raise ValueError("The parameter `image` must be a 2-dimensional array")      # check_nD(image, 2)
    if t.dtype in _IMG_AS_FLOAT:
        # skimage.filters.gaussian converts integer images with img_as_float before smoothing (util/dtype.py _convert):
        # unsigned: x * (1 / imax); signed: (x + 0.5) * (2 / (imax - imin)), in float64.  The division by 1 is the exact
        # integer -> float64 conversion; the thresholds of non-quantile calls are divided by dtype_max = imax.
        imin, imax = _IMG_AS_FLOAT[t.dtype]
        frames = t if t.ndim == 3 else t[None]
        f = ops.normalize(frames, 1.0)
        if imin < 0:
            f = ops.ground(f, value=0.5, mn=torch.zeros(f.shape[0], dtype=torch.float64, device=dev))
            f = ops.scale(f, 2.0 / (float(imax) - float(imin)))
        else:
            f = ops.scale(f, 1.0 / imax)
        t = f if t.ndim == 3 else f[0]
        dtype_max = float(imax)
    else:
        dtype_max = 1.0                       # dtype_limits(float image, clip_negative=False)[1]
    batched = t.ndim == 3
    x = t if batched else t[None]
    n, h, w = x.shape
    if low_threshold is None:
        low_threshold = 0.1
    elif use_quantiles:
        if not 0.0 <= low_threshold <= 1.0:
            raise ValueError("Quantile thresholds must be between 0 and 1.")
    else:
        low_threshold = low_threshold / dtype_max
    if high_threshold is None:
        high_threshold = 0.2
    elif use_quantiles:
        if not 0.0 <= high_threshold <= 1.0:
            raise ValueError("Quantile thresholds must be between 0 and 1.")
    else:
        high_threshold = high_threshold / dtype_max
    lib, st = _lib.load(), torch.cuda.current_stream(dev).cuda_stream
    smoothed = torch.empty_like(x)
    mk = None
    if mask is None:
        g_img = ops.gaussian_filter_mode(x, sigma, mode="constant")
        g_one = ops.gaussian_filter_mode(torch.ones((1, h, w), dtype=torch.float64, device=dev), sigma, mode="constant")
        check(lib.pl_canny_normalise(g_img.data_ptr(), g_one.data_ptr(), n, h * w, smoothed.data_ptr(), st),
              "pl_canny_normalise")
    else:
        # smooth_with_function_and_mask (_canny.py): G(image with 0 outside the mask) / (G(mask) + eps), per frame; the mask
        # is one [H, W] plane for every frame or one per frame
        mk = mask if isinstance(mask, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(mask))
        mk = (mk.to(dev) != 0).to(torch.uint8).contiguous()
        if tuple(mk.shape) not in ((h, w), (n, h, w)):
            raise ValueError("mask must have the shape of the image")
        per_frame = 1 if mk.ndim == 3 else 0
        masked, mask_f = torch.empty_like(x), torch.empty_like(x)
        check(lib.pl_canny_mask_prepare(x.data_ptr(), mk.data_ptr(), per_frame, n, h * w, masked.data_ptr(), mask_f.data_ptr(), st),
              "pl_canny_mask_prepare")
        g_img = ops.gaussian_filter_mode(masked, sigma, mode="constant")
        g_msk = ops.gaussian_filter_mode(mask_f, sigma, mode="constant")
        check(lib.pl_canny_normalise(g_img.data_ptr(), g_msk.data_ptr(), 1, n * h * w, smoothed.data_ptr(), st),
              "pl_canny_normalise")
    jsobel = ops.sobel(smoothed, 1)
    isobel = ops.sobel(smoothed, 0)
    mag = torch.empty_like(x)
    local_max = torch.empty((n, h, w), dtype=torch.uint8, device=dev)
    if mk is None:
        check(lib.pl_canny_nms(isobel.data_ptr(), jsobel.data_ptr(), n, h, w, mag.data_ptr(), local_max.data_ptr(), st),
              "pl_canny_nms")
    else:
        check(lib.pl_canny_nms_masked(isobel.data_ptr(), jsobel.data_ptr(), n, h, w, mk.data_ptr(), 1 if mk.ndim == 3 else 0,
                                      mag.data_ptr(), local_max.data_ptr(), st), "pl_canny_nms_masked")
    if use_quantiles:
        thr = torch.stack([_percentile_f64(mag, 100.0 * low_threshold), _percentile_f64(mag, 100.0 * high_threshold)],
                          dim=1).contiguous()
    else:
        thr = torch.tensor([[low_threshold, high_threshold]] * n, dtype=torch.float64, device=dev)
    low = torch.empty_like(local_max)
    high = torch.empty_like(local_max)
    check(lib.pl_canny_hysteresis(local_max.data_ptr(), mag.data_ptr(), thr.data_ptr(), n, h, w, low.data_ptr(),
                                  high.data_ptr(), None, None, None, 0, st), "pl_canny_hysteresis")
    labels, _ = ops.label(low, 8)
    good = torch.empty((n, h, w), dtype=torch.int32, device=dev)
    out = torch.empty_like(local_max)
    check(lib.pl_canny_hysteresis(None, None, None, n, h, w, None, high.data_ptr(), labels.data_ptr(), good.data_ptr(),
                                  out.data_ptr(), 1, st), "pl_canny_hysteresis")
    return out if batched else out[0]


def _percentile_f64(values: torch.Tensor, q: float) -> torch.Tensor:
    """``np.percentile(frame, q)`` (linear interpolation, numpy's ``_lerp``) per frame of a float64 batch."""
    x = values.reshape(values.shape[0], -1).contiguous()
    n, count = x.shape
    _, lo, hi, frac = ops._percentile_plan(count, [q])
    ranks = torch.tensor([int(lo[0]), int(hi[0])], dtype=torch.int64, device=x.device)
    st = torch.empty((n, 2), dtype=torch.float64, device=x.device)
    check(_lib.load().pl_order_stats_f64(x.data_ptr(), n, count, ranks.data_ptr(), 2, st.data_ptr(),
                                         torch.cuda.current_stream(x.device).cuda_stream), "pl_order_stats_f64")
    t = torch.full((n,), float(frac[0]), dtype=torch.float64, device=x.device)
    return ops.lerp_like_numpy(st[:, 0].contiguous(), st[:, 1].contiguous(), t)


def hough_line(image, theta=None, device=None):
    """``skimage.transform.hough_line(image, theta)`` (pylinac/planar_imaging.py:3158) -> (hspace uint64 tensor
    [n_dist, n_theta] on the device, theta, dists) -- the accumulation runs on the GPU, one lane per (pixel, angle)."""
    t = image if isinstance(image, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(image))
    if t.ndim != 2:
        raise ValueError("The input image `image` must be 2D.")
    dev = t.device if t.is_cuda else (torch.device(device) if device is not None
                                      else torch.device("cuda", torch.cuda.current_device()))
    img = (t.to(dev) != 0).to(torch.uint8).contiguous()
    if theta is None:
        theta = np.linspace(-np.pi / 2, np.pi / 2, 180)
    theta = np.asarray(theta, dtype=np.float64)
    h, w = img.shape
    offset = int(np.ceil(np.sqrt(h * h + w * w)))
    n_dist = 2 * offset                     # scikit-image 0.18.3 (>= 0.19 allocates one more row)
    ct = torch.from_numpy(np.cos(theta)).to(dev)
    sn = torch.from_numpy(np.sin(theta)).to(dev)
    accum = torch.empty((n_dist, len(theta)), dtype=torch.int64, device=dev)      # uint64 counts, int64 storage
    check(_lib.load().pl_hough_line(img.data_ptr(), h, w, ct.data_ptr(), sn.data_ptr(), len(theta), accum.data_ptr(),
                                    torch.cuda.current_stream(dev).cuda_stream), "pl_hough_line")
    return accum, theta, np.linspace(-offset, offset, n_dist)
