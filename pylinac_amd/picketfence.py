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


def leaves_in_view(shape, dpmm, leaves, centers, widths, analysis_width=0.4):
    """``PicketFence._leaves_in_view`` for UP_DOWN pickets (picketfence.py:888-912)."""
    pixel_range = shape[0] / 2
    pixel_range -= max(widths[0] * analysis_width, widths[-1] * analysis_width) * dpmm
    return [(n, c, w) for n, c, w in zip(leaves, centers, widths) if abs(c) < pixel_range / dpmm]


@dataclass
class PFBatchResult:
    leaf_nums: list            # leaf numbers in view, row order of `position`
    picket_idx: torch.Tensor   # int32 [N, cap]   FWXM picket centres (pixel index)
    picket_count: torch.Tensor # int32 [N]
    spacing: torch.Tensor      # float64 [N]      picket spacing in pixels
    position: torch.Tensor     # float64 [N, leaves, cap] MLC positions in pixels (NaN = no measurement)
    status: torch.Tensor       # int32 [N, leaves, cap]   0 ok, 1 no picket, 2 rejected window, 3 unsupported


def analyze_batch(frames: torch.Tensor, dpmm: float, mlc: str = "MILLENNIUM", num_pickets: int | None = None,
                  leaf_analysis_width_ratio: float = 0.4, height_threshold: float = 0.5,
                  edge_threshold: float = 1.5, peak_sort: str = "peak_heights",
                  required_prominence: float = 0.2, fwxm: int = 50, cap: int = 16) -> PFBatchResult:
    x = ops._frames(frames)
    if x.dtype != torch.uint16:
        raise TypeError("analyze_batch needs uint16 frames (the reference's int16 ground() overflows)")
    n, h, w = x.shape
    dev = x.device
    lib, st = _lib.load(), torch.cuda.current_stream().cuda_stream
    # ground()/normalize(): q = (a - min) / (max - min)
    vmin, vmax = ops.minmax(x)
    gmax = vmax - vmin
    leaf_prof = torch.empty((n, w), dtype=torch.float64, device=dev)
    check(lib.pl_scaled_colmean(x.data_ptr(), n, h, w, vmin.data_ptr(), gmax.data_ptr(), leaf_prof.data_ptr(), st),
          "pl_scaled_colmean")
    leaf_prof = ops.normalize(leaf_prof.unsqueeze(1)).squeeze(1).contiguous()      # MultiProfile.normalize()
    peaks = ops.find_peaks_batch(leaf_prof, cap=cap, threshold=height_threshold, peak_separation=0.02,
                                 max_number=num_pickets, peak_sort=peak_sort,
                                 required_prominence=required_prominence)
    pk_idx = torch.empty((n, cap), dtype=torch.int32, device=dev)
    pk_val = torch.empty((n, cap), dtype=torch.float64, device=dev)
    spacing = torch.empty(n, dtype=torch.float64, device=dev)
    check(lib.pl_pf_pickets(peaks.count.data_ptr(), peaks.props.data_ptr(), cap, leaf_prof.data_ptr(), w, n,
                            pk_idx.data_ptr(), pk_val.data_ptr(), spacing.data_ptr(), st), "pl_pf_pickets")
    # leaf geometry is image-independent: _get_mlc_window rows (picketfence.py:859-886)
    leaves, centers, widths = mlc_arrangement(MLC_ARRANGEMENTS[mlc])
    view = leaves_in_view((h, w), dpmm, leaves, centers, widths, leaf_analysis_width_ratio)
    tops, bottoms = [], []
    for _, center, width in view:
        leaf_width_px = width * dpmm
        leaf_center_px = center * dpmm + h / 2
        tops.append(max(int(leaf_center_px - leaf_width_px / 2), 0))
        bottoms.append(min(int(leaf_center_px + leaf_width_px / 2), h))
    nl = len(view)
    d_top = torch.tensor(tops, dtype=torch.int32, device=dev)
    d_bot = torch.tensor(bottoms, dtype=torch.int32, device=dev)
    m = n * nl * cap
    lmax = 128
    prof = torch.zeros((m, lmax), dtype=torch.float64, device=dev)
    lens = torch.empty(m, dtype=torch.int32, device=dev)
    offset = torch.empty(m, dtype=torch.float64, device=dev)
    status = torch.empty(m, dtype=torch.int32, device=dev)
    max_rows = min(max([b - t for t, b in zip(tops, bottoms)] + [1]), 48)      # taller windows: status 3, as before
    check(lib.pl_pf_windows_rows(x.data_ptr(), n, h, w, vmin.data_ptr(), gmax.data_ptr(), peaks.count.data_ptr(),
                            pk_idx.data_ptr(), pk_val.data_ptr(), cap, spacing.data_ptr(), d_top.data_ptr(),
                            d_bot.data_ptr(), nl, max_rows, float(height_threshold), float(edge_threshold), lmax,
                            prof.data_ptr(), lens.data_ptr(), offset.data_ptr(), status.data_ptr(), st),
          "pl_pf_windows_rows")
    wpk = ops.find_peaks_batch(prof, cap=1, lens=lens, fwxm_height=fwxm / 100, max_number=1)   # FWXMProfile edges
    rec = ops.fwxm_record(wpk)
    pos = torch.empty(m, dtype=torch.float64, device=dev)
    check(lib.pl_pf_positions(status.data_ptr(), rec.data_ptr(), offset.data_ptr(), m, pos.data_ptr(), st),
          "pl_pf_positions")
    return PFBatchResult([v[0] for v in view], pk_idx, peaks.count, spacing, pos.view(n, nl, cap),
                         status.view(n, nl, cap))
