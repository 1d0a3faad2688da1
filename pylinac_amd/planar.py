"""Planar-phantom outline search on the device (SURVEY.md section 8 "next" row f2, second half).

Mirrors, for one frame or a batch:
  * ``ImagePhantomBase._get_canny_regions`` (pylinac/planar_imaging.py:574-588): ``feature.canny`` with the phantom's
    ``detection_canny_settings`` -> ``measure.label`` (8-connected) -> the regions' bounding boxes,
  * ``ImagePhantomBase.phantom_ski_region`` (:300-341) with the detection conditions ``is_centered`` /
    ``is_right_size`` / ``is_square`` (:115-137) and ``roi_match_condition`` "max" / "closest",
  * ``transform.hough_line_peaks`` (scikit-image 0.18.3) as the MC2 roll estimate calls it (:3158-3166).

Dense work (canny, labelling, region tables, Hough accumulation, the separable maximum filter and the candidate mask)
runs in kernels; what stays on the host is bookkeeping over a table of a few hundred bounding boxes and the greedy
walk over a handful of Hough peaks.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib, canny as _canny, ops, regionprops as _rp
from ._lib import check


# ------------------------------------------------------------------------------------------------ Hough peaks
def _device_of(t, device):
    if isinstance(t, torch.Tensor) and t.is_cuda:
        return t.device
    return torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())


def max_filter1d(frames: torch.Tensor, half: int, axis: int) -> torch.Tensor:
    """``ndimage.maximum_filter1d(frame, size=2*half+1, axis=axis, mode="constant", cval=0)`` per frame."""
    x = ops._frames(frames)
    n, h, w = x.shape
    out = torch.empty_like(x)
    check(_lib.load().pl_max_filter1d(x.data_ptr(), out.data_ptr(), ops._dt(x), n, h, w, int(axis), int(half),
                                      torch.cuda.current_stream(x.device).cuda_stream), "pl_max_filter1d")
    return out


def prominent_peaks(image, min_xdistance: int = 1, min_ydistance: int = 1, threshold=None, num_peaks=np.inf,
                    device=None):
    """``skimage.feature.peak._prominent_peaks`` (0.18.3) -> (heights, x indices, y indices) as numpy arrays."""
    t = image if isinstance(image, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(image))
    if t.ndim != 2:
        raise ValueError("prominent_peaks needs a 2-D accumulator")
    if t.dtype == torch.uint64:
        t = t.view(torch.int64)                       # counts < 2**63
    elif t.dtype not in (torch.int64, torch.float64, torch.int32, torch.float32):
        t = t.to(torch.int64)
    dev = _device_of(t, device)
    img = t.to(dev).contiguous()[None]
    rows, cols = img.shape[1:]
    if threshold is None:
        threshold = 0.5 * float(ops.minmax(img)[1][0])
    threshold = float(threshold)
    img_max = max_filter1d(max_filter1d(img, min_ydistance, 0), min_xdistance, 1)
    mask = torch.empty(img.shape, dtype=torch.uint8, device=dev)
    check(_lib.load().pl_peak_candidates(img.data_ptr(), img_max.data_ptr(), ops._dt(img), img.numel(), threshold,
                                         mask.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
          "pl_peak_candidates")
    labels, _ = ops.label(mask, connectivity=8)
    rc = torch.nonzero(mask[0])                        # raster order, like np.nonzero
    empty = (np.array([]), np.array([]), np.array([]))
    if rc.shape[0] == 0:
        return empty
    lab = labels[0][rc[:, 0], rc[:, 1]].cpu().numpy()
    val = img_max[0][rc[:, 0], rc[:, 1]].cpu().numpy()
    rc = rc.cpu().numpy()
    # regionprops per candidate group, in label order: max_intensity, round(centroid)
    order = np.argsort(lab, kind="stable")
    lab, val, rc = lab[order], val[order], rc[order]
    starts = np.flatnonzero(np.r_[True, lab[1:] != lab[:-1]])
    ends = np.r_[starts[1:], len(lab)]
    heights = np.array([val[s:e].max() for s, e in zip(starts, ends)])
    cy = np.array([np.round(rc[s:e, 0].mean()) for s, e in zip(starts, ends)]).astype(int)
    cx = np.array([np.round(rc[s:e, 1].mean()) for s, e in zip(starts, ends)]).astype(int)
    rank = np.argsort(heights, kind="stable")[::-1]    # sorted(props, key=max_intensity)[::-1]
    cy, cx = cy[rank], cx[rank]
    at_centre = img_max[0][torch.from_numpy(cy).to(dev), torch.from_numpy(cx).to(dev)].cpu().numpy()
    my, mx = int(min_ydistance), int(min_xdistance)
    acc_h, acc_x, acc_y = [], [], []
    for y0, x0, accum in zip(cy, cx, at_centre):
        # has an accepted peak zeroed img_max[y0, x0]?  (rows: 0 < y < rows, no wrap; columns wrap with the row mirrored)
        gone = False
        for ya, xa in zip(acc_y, acc_x):
            if abs(x0 - xa) <= mx and abs(y0 - ya) <= my and y0 > 0:
                gone = True
            ym = rows - y0                             # the row a wrapped neighbour came from
            if 0 < ym < rows and abs(ym - ya) <= my and (abs(x0 - cols - xa) <= mx or abs(x0 + cols - xa) <= mx):
                gone = True
            if gone:
                break
        if gone:
            accum = accum * 0
        if not accum > threshold:
            continue
        acc_h.append(accum)
        acc_x.append(int(x0))
        acc_y.append(int(y0))
    acc_h, acc_x, acc_y = np.array(acc_h), np.array(acc_x), np.array(acc_y)
    if num_peaks < len(acc_h):
        top = np.argsort(acc_h)[::-1][:num_peaks]
        acc_h, acc_x, acc_y = acc_h[top], acc_x[top], acc_y[top]
    return acc_h, acc_x, acc_y


def hough_line_peaks(hspace, angles, dists, min_distance: int = 9, min_angle: int = 10, threshold=None,
                     num_peaks=np.inf, device=None):
    """``skimage.transform.hough_line_peaks`` (0.18.3; pylinac/planar_imaging.py:3160-3166) -> (heights, angles,
    dists).  ``hspace`` may be the device tensor returned by ``canny.hough_line``."""
    min_angle = min(min_angle, hspace.shape[1])
    h, a, d = prominent_peaks(hspace, min_xdistance=min_angle, min_ydistance=min_distance, threshold=threshold,
                              num_peaks=num_peaks, device=device)
    angles, dists = np.asarray(angles), np.asarray(dists)
    if a.any():
        return h, angles[a], dists[d]
    return h, np.array([]), np.array([])


# ------------------------------------------------------------------------------------------------ phantom outline
def is_square(bbox, image_shape, phantom_bbox_size_px, rtol: float = 0.2) -> bool:
    """planar_imaging.py:115-119"""
    return math.isclose((bbox[2] - bbox[0]) / (bbox[3] - bbox[1]), 1, rel_tol=rtol)


def is_centered(bbox, image_shape, phantom_bbox_size_px, rtol: float = 0.3) -> bool:
    """planar_imaging.py:122-126 (image.center = shape / 2 - 0.5, pylinac/core/image.py:526-533)"""
    middle = ((bbox[2] - bbox[0]) / 2 + bbox[0], (bbox[3] - bbox[1]) / 2 + bbox[1])
    return bool(np.allclose(middle, (image_shape[0] / 2 - 0.5, image_shape[1] / 2 - 0.5), rtol=rtol))


def is_right_size(bbox, image_shape, phantom_bbox_size_px, rtol: float = 0.1) -> bool:
    """planar_imaging.py:129-137"""
    return bool(np.isclose((bbox[2] - bbox[0]) * (bbox[3] - bbox[1]), phantom_bbox_size_px, rtol=rtol))


def select_phantom_region(bboxes, image_shape, phantom_bbox_size_px: float,
                          conditions=(is_centered, is_right_size), roi_match_condition: str = "max") -> int:
    """``phantom_ski_region`` (planar_imaging.py:300-341) over the bbox table -> row index (= label - 1)."""
    b = np.asarray(bboxes, dtype=np.int64)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    ranked = sorted((i for i in range(len(b)) if area[i] > 100), key=lambda i: area[i], reverse=True)
    blobs = [i for i in ranked if all(c(tuple(int(v) for v in b[i]), image_shape, phantom_bbox_size_px)
                                      for c in conditions)]
    if not blobs:
        raise ValueError(
            "Unable to find the phantom in the image. Potential solutions: check the SSD was passed correctly, check "
            "that the phantom isn't at the edge of the field, check that the phantom is centered along the CAX.")
    if roi_match_condition == "max":
        best = np.argsort([area[i] for i in blobs])[-1]
    elif roi_match_condition == "closest":
        best = np.argsort([abs(area[i] - phantom_bbox_size_px) for i in blobs])[0]
    else:
        raise ValueError("roi_match_condition must be 'max' or 'closest'")
    return blobs[best]


def canny_regions(image, sigma: float = 2, percentiles=(0.001, 0.01), device=None):
    """``_get_canny_regions`` (planar_imaging.py:574-588) -> (edges uint8 [N,H,W], labels int32 [N,H,W], list of int64
    bbox tables [n_regions, 4] = (min_row, min_col, max_row, max_col) half-open, one per frame)."""
    edges = _canny.canny(image, sigma=sigma, low_threshold=percentiles[0], high_threshold=percentiles[1],
                         use_quantiles=True, device=device)
    e = edges if edges.ndim == 3 else edges[None]
    labels, count = ops.label(e, connectivity=8)
    counts = count.cpu().numpy()
    cap = max(int(counts.max()), 1)
    stats, overflow = ops.region_stats(labels, None, cap)
    if int(overflow.max()):
        raise RuntimeError("region table overflow")       # cannot happen: cap is the label count
    table = stats[:, :, 1:5].cpu().numpy()
    return e, labels, [table[i, : counts[i]].astype(np.int64) for i in range(len(counts))]


@dataclass
class PhantomRegion:
    """The attributes of the skimage region that the planar analyses read (planar_imaging.py:1147-1154, 2348, 3336)."""

    label: int
    bbox: tuple
    image: torch.Tensor            # bool [bbox rows, bbox cols] on the device: region.image
    _raw: tuple = None

    @property
    def bbox_area(self) -> int:
        return (self.bbox[2] - self.bbox[0]) * (self.bbox[3] - self.bbox[1])

    area_bbox = bbox_area

    def _raw_moments(self):
        """exact integer raw moments (m00, m10, m01, m20, m02, m11) of the region in bbox-local (row, col) coordinates
        (``pl_region_moments``); see ``regionprops.py`` for how scikit-image's quantities are formed from them"""
        if getattr(self, "_raw", None) is None:
            mom, _ = ops.region_moments(self.image.to(torch.int32).contiguous()[None], 1)
            self._raw = tuple(int(v) for v in mom[0, 0].cpu().tolist())
        return self._raw

    @property
    def centroid(self) -> tuple:
        """regionprops.centroid: (row, col) in image coordinates"""
        r, c = _rp.centroid(self._raw_moments())
        return (r + self.bbox[0], c + self.bbox[1])

    @property
    def orientation(self) -> float:
        """regionprops.orientation (scikit-image 0.18.3, used at planar_imaging.py:2348, 2498): the angle between the
        row axis and the major axis of the region's inertia ellipse, from the inertia tensor
        [[mu02, -mu11], [-mu11, mu20]] / n built from exact integer moments."""
        return _rp.orientation(self._raw_moments())

    @property
    def eccentricity(self) -> float:
        return _rp.eccentricity(self._raw_moments())

    @property
    def bbox_center_xy(self) -> tuple:
        """core/roi.py bbox_center: Point(x, y) of the bbox middle"""
        return ((self.bbox[3] - self.bbox[1]) / 2 + self.bbox[1], (self.bbox[2] - self.bbox[0]) / 2 + self.bbox[0])


def find_phantom_region(image, phantom_bbox_size_px: float, sigma: float = 2, percentiles=(0.001, 0.01),
                        conditions=(is_centered, is_right_size), roi_match_condition: str = "max",
                        device=None) -> PhantomRegion:
    """canny -> label -> region table -> ``phantom_ski_region`` for ONE frame."""
    t = image if isinstance(image, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(image))
    if t.ndim != 2:
        raise ValueError("find_phantom_region takes one 2-D frame")
    _, labels, tables = canny_regions(t, sigma=sigma, percentiles=percentiles, device=device)
    row = select_phantom_region(tables[0], tuple(t.shape), phantom_bbox_size_px, conditions, roi_match_condition)
    r0, c0, r1, c1 = (int(v) for v in tables[0][row])
    return PhantomRegion(label=row + 1, bbox=(r0, c0, r1, c1), image=labels[0, r0:r1, c0:c1] == row + 1)
