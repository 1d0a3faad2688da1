"""Batched picket-fence measurement (BASELINE config #3).

Mirrors the per-image part of ``PicketFence.analyze`` for UP_DOWN pickets
(pylinac/picketfence.py:745-803): leaf profile -> FWXM picket positions -> picket spacing -> for every
leaf pair in view and every picket the MLC window, the ``_is_mlc_peak_in_window`` test and the FWXM
centre of the window's median profile.  The input batch is uint16 ``[N,H,W]`` frames AFTER the
constructor's crop (picketfence.py:214-215); ground()/normalize() (:322-323) are folded into the
kernels as the float64 quotient ``(a - min) / (max - min)``.

The per-dataset post-processing of the reference (dropping leaf rows without the modal number of
kisses :810-824, per-picket line fits :831-843, error in mm :1701-1718) works on the returned
``[N, leaves, P]`` positions and stays on the host -- a few thousand flops per image.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from . import _lib, ops
from ._lib import check

MLC_ARRANGEMENTS = {  # pylinac/picketfence.py:103-135
    "MILLENNIUM": [(10, 10), (40, 5), (10, 10)],
    "HD_MILLENNIUM": [(14, 5), (32, 2.5), (14, 5)],
    "BMOD": [(40, 4)],
    "AGILITY": [(80, 5)],
    "MLCI": [(40, 10)],
    "HALCYON_DISTAL": [(28, 10)],
    "HALCYON_PROXIMAL": [(29, 10)],
}


def mlc_arrangement(leaf_arrangement, offset: float = 0):
    """``MLCArrangement`` (picketfence.py:67-100) -> (leaf numbers, centers mm, widths mm)."""
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


def leaves_in_view(shape, dpmm, leaves, centers, widths, analysis_width=0.4, orientation: str = "UP_DOWN"):
    """``PicketFence._leaves_in_view`` (picketfence.py:888-912): the leaves run across the rows (UP_DOWN pickets) or across the
    columns (LEFT_RIGHT)."""
    pixel_range = (shape[0] if orientation == "UP_DOWN" else shape[1]) / 2
    pixel_range -= max(widths[0] * analysis_width, widths[-1] * analysis_width) * dpmm
    return [(n, c, w) for n, c, w in zip(leaves, centers, widths) if abs(c) < pixel_range / dpmm]


def pairwise_plan(n: int):
    """numpy's pairwise summation tree for a contiguous run of ``n`` float64 values (numpy/_core/src/umath/loops_utils.h.src:
    more than 128 values are halved, the first half rounded down to a multiple of 8) -> (leaf starts, leaf lengths, postfix
    program: k >= 0 = leaf k's sum, -1 = add the two sums on top).  Feeds ``pl_scaled_rowmean``."""
    starts, lens, prog = [], [], []

    def rec(s, m):
        if m <= 128:
            prog.append(len(starts))
            starts.append(s)
            lens.append(m)
            return
        n2 = m // 2
        n2 -= n2 % 8
        rec(s, n2)
        rec(s + n2, m - n2)
        prog.append(-1)

    rec(0, int(n))
    return np.asarray(starts, np.int32), np.asarray(lens, np.int32), np.asarray(prog, np.int32)


@dataclass
class PFBatchResult:
    leaf_nums: list            # leaf numbers in view, row order of `position`
    picket_idx: torch.Tensor   # int32 [N, cap]   FWXM picket centres (pixel index)
    picket_count: torch.Tensor # int32 [N]
    spacing: torch.Tensor      # float64 [N]      picket spacing in pixels
    position: torch.Tensor     # float64 [N, leaves, cap] MLC positions in pixels (NaN = no measurement): the leaf-pair centre
    status: torch.Tensor       # int32 [N, leaves, cap]   0 ok, 1 no picket, 2 rejected window, 3 unsupported
    left: torch.Tensor | None = None    # separate_leaves: float64 [N, leaves, cap] positions of the two banks' leaf ends
    right: torch.Tensor | None = None   # (MLCValue.get_peak_positions' (left, right), picketfence.py:1616-1623)


_PLAN_CACHE: dict = {}


def analyze_batch(frames: torch.Tensor, dpmm: float, mlc: str = "MILLENNIUM", num_pickets: int | None = None,
                  leaf_analysis_width_ratio: float = 0.4, height_threshold: float = 0.5,
                  edge_threshold: float = 1.5, peak_sort: str = "peak_heights",
                  required_prominence: float = 0.2, fwxm: int = 50, cap: int | None = None,
                  orientation: str = "UP_DOWN", separate_leaves: bool = False, exact_deviation: bool = False) -> PFBatchResult:
    """The per-image measurement of ``PicketFence.analyze`` (picketfence.py:745-803, 1605-1628) for a resident batch, in five
    launches: min / max (ground + normalize folded into every later read), leaf profile, picket peaks, picket table, and ONE
    kernel for all leaf x picket windows (window test, median profile, FWXM search, position).  ``orientation``: "UP_DOWN"
    (pickets run up-down: the leaf profile is ``np.mean(image, 0)``) or "LEFT_RIGHT" (``np.mean(image, 1)``, windows
    transposed).  ``separate_leaves``: also return both leaf-end positions per window.  ``cap`` = picket slots per frame
    (default ``num_pickets`` when given, else 16).  ``exact_deviation=True`` makes every window evaluate numpy's float64
    ``np.std`` sequence for the edge test instead of deciding it from exact integer row moments where the margin allows
    (identical results; a test knob)."""
    x = ops._frames(frames)
    unfit = None
    if x.dtype in (torch.int16, torch.float64):
        # what the reference's loader may hand over instead of uint16: a signed panel's int16, or float64 holding integers
        # (``dtype=float``; rescale tags with an integer slope and intercept).  ground() / normalize() (picketfence.py:322-323)
        # only see a - min, which pl_to_u16_exact forms exactly; frames it cannot represent -- non-integer float64 values,
        # an int16 range beyond 32767 where the reference's own ground() wraps -- come back with status 3 in every window
        x, unfit = ops.to_u16_exact(x)
    elif x.dtype != torch.uint16:
        raise TypeError("analyze_batch takes uint16 frames, int16 frames, or float64 frames holding integers")
    if orientation not in ("UP_DOWN", "LEFT_RIGHT"):
        raise ValueError("orientation must be 'UP_DOWN' or 'LEFT_RIGHT'")
    lr = orientation == "LEFT_RIGHT"
    if cap is None:
        cap = int(num_pickets) if num_pickets else 16
    n, h, w = x.shape
    dev = x.device
    lib, st = _lib.load(), torch.cuda.current_stream().cuda_stream
    # ground()/normalize(): q = (a - min) / (max - min)
    vmin, vmax = ops.minmax(x)
    gmax = vmax - vmin
    travel = h if lr else w                               # length of the leaf profile = the pickets' travel axis
    leaf_prof = torch.empty((n, travel), dtype=torch.float64, device=dev)
    if lr:
        key = (w, str(dev))
        if key not in _PLAN_CACHE:
            _PLAN_CACHE[key] = tuple(torch.from_numpy(a).to(dev) for a in pairwise_plan(w))
        ls, ll, pg = _PLAN_CACHE[key]
        check(lib.pl_scaled_rowmean(x.data_ptr(), n, h, w, vmin.data_ptr(), gmax.data_ptr(), ls.data_ptr(), ll.data_ptr(),
                                    ls.numel(), pg.data_ptr(), pg.numel(), leaf_prof.data_ptr(), st), "pl_scaled_rowmean")
    else:
        check(lib.pl_scaled_colmean(x.data_ptr(), n, h, w, vmin.data_ptr(), gmax.data_ptr(), leaf_prof.data_ptr(), st),
              "pl_scaled_colmean")
    leaf_prof = ops.normalize(leaf_prof.unsqueeze(1)).squeeze(1).contiguous()      # MultiProfile.normalize()
    peaks = ops.find_peaks_batch(leaf_prof, cap=cap, threshold=height_threshold, peak_separation=0.02,
                                 max_number=num_pickets, peak_sort=peak_sort,
                                 required_prominence=required_prominence)
    pk_idx = torch.empty((n, cap), dtype=torch.int32, device=dev)
    pk_val = torch.empty((n, cap), dtype=torch.float64, device=dev)
    spacing = torch.empty(n, dtype=torch.float64, device=dev)
    check(lib.pl_pf_pickets(peaks.count.data_ptr(), peaks.props.data_ptr(), cap, leaf_prof.data_ptr(), travel, n,
                            pk_idx.data_ptr(), pk_val.data_ptr(), spacing.data_ptr(), st), "pl_pf_pickets")
    # leaf geometry is image-independent: _get_mlc_window (picketfence.py:859-886): rows for UP_DOWN, columns for LEFT_RIGHT
    leaves, centers, widths = mlc_arrangement(MLC_ARRANGEMENTS[mlc])
    view = leaves_in_view((h, w), dpmm, leaves, centers, widths, leaf_analysis_width_ratio, orientation)
    across = w if lr else h                               # the axis the leaves are stacked along
    los, his = [], []
    for _, center, width in view:
        leaf_width_px = width * dpmm
        leaf_center_px = center * dpmm + across / 2
        los.append(max(int(leaf_center_px - leaf_width_px / 2), 0))
        his.append(min(int(leaf_center_px + leaf_width_px / 2), across))
    nl = len(view)
    if nl == 0:
        raise ValueError("no leaves in view")
    key = (tuple(los), tuple(his), str(dev))
    if key not in _PLAN_CACHE:
        _PLAN_CACHE[key] = (torch.tensor(los, dtype=torch.int32, device=dev), torch.tensor(his, dtype=torch.int32, device=dev))
    d_lo, d_hi = _PLAN_CACHE[key]
    m = n * nl * cap
    # the widest leaf in pixels: the kernel's LDS per wave follows it.  48 is the limit (status 3 beyond): 10 mm leaves (the
    # widest of every supported bank) on the finest supported panel at isocentre scale (aS1200, 0.336 mm at SID 1500) are 45
    max_rows = min(max([b - t for t, b in zip(los, his)] + [1]), 48)
    fw = ops.make_peak_params(128, fwxm_height=fwxm / 100, max_number=1)                 # FWXMProfile.field_edge_idx
    rec = torch.empty((m, 3), dtype=torch.float64, device=dev)
    status = torch.empty(m, dtype=torch.int32, device=dev)
    import ctypes as C
    check(lib.pl_pf_measure(x.data_ptr(), n, h, w, 1 if lr else 0, vmin.data_ptr(), gmax.data_ptr(), peaks.count.data_ptr(),
                            pk_idx.data_ptr(), pk_val.data_ptr(), cap, spacing.data_ptr(), d_lo.data_ptr(), d_hi.data_ptr(), nl,
                            max_rows, float(height_threshold), float(edge_threshold), 1 if exact_deviation else 0, C.byref(fw),
                            rec.data_ptr(),
                            status.data_ptr(), 0, 0, st), "pl_pf_measure")
    rec = rec.view(n, nl, cap, 3)
    if unfit is not None:                                   # (device-side selects: no synchronisation)
        bad = (unfit != 0).view(n, 1, 1)
        status = torch.where(bad.expand(n, nl, cap).reshape(-1), torch.full_like(status, 3), status)
        rec = torch.where(bad.unsqueeze(-1), torch.full_like(rec, float("nan")), rec)
    return PFBatchResult([v[0] for v in view], pk_idx, peaks.count, spacing, rec[..., 0], status.view(n, nl, cap),
                         rec[..., 1] if separate_leaves else None, rec[..., 2] if separate_leaves else None)
