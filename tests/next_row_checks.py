"""Parity checks of the "next"-row host functions that are shared by the `-m gpu` tests (real MI355X) and the
emulated-device tests (tests/emu_backend.py, CPU).  Each takes the golden loader and the device."""
from __future__ import annotations

import numpy as np
import torch


def _peaks_kwargs(row):
    md, ma, thr, npk = row
    kw = dict(min_distance=int(md), min_angle=int(ma))
    if thr >= 0:
        kw["threshold"] = float(thr)
    if npk >= 0:
        kw["num_peaks"] = int(npk)
    return kw


def check_hough_line_peaks(golden, dev):
    """planar.hough_line_peaks == scikit-image 0.18.3 transform.hough_line_peaks (tests/golden/planar.npz)."""
    from pylinac_amd import planar

    g = golden("planar")
    for k in range(4):
        hs, an, di = g[f"acc{k}.hspace"], g[f"acc{k}.angles"], g[f"acc{k}.dists"]
        for j in range(4):
            h, a, d = planar.hough_line_peaks(torch.from_numpy(hs.astype(np.int64)).to(dev), an, di,
                                              **_peaks_kwargs(g[f"acc{k}.kw{j}"]))
            assert np.array_equal(h, g[f"acc{k}.p{j}.h"]), (k, j)
            if (k, j) == (2, 2):
                # the top-`num_peaks` cut falls inside a run of equal heights: np.argsort (unstable) decides, and its
                # tie order differs between numpy versions -- the reference's choice is not defined there
                continue
            assert np.array_equal(a, g[f"acc{k}.p{j}.a"]) and np.array_equal(d, g[f"acc{k}.p{j}.d"]), (k, j)
    for n in g["names"]:
        hs, an, di = g[f"{n}.hspace"], g[f"{n}.theta"], g[f"{n}.dists"]
        for md in (17, 9):
            for npk, tag in ((2, "2"), (np.inf, "inf")):
                h, a, d = planar.hough_line_peaks(hs, an, di, min_distance=md, num_peaks=npk, device=dev)
                t = f"{n}.peaks.md{md}.n{tag}"
                assert np.array_equal(h, g[t + ".h"]) and np.array_equal(a, g[t + ".a"]) and \
                    np.array_equal(d, g[t + ".d"]), t


def check_phantom_outline(golden, dev, names=None):
    """canny -> label -> bbox table -> phantom_ski_region -> region.image -> hough_line, against scikit-image's own
    regionprops / hough_line on the same frames."""
    from pylinac_amd import canny, planar

    g = golden("planar")
    for n in (names or g["names"]):
        img, (sigma, lo, hi) = g[f"{n}.img"], g[f"{n}.kw"]
        edges, labels, tables = planar.canny_regions(torch.from_numpy(img).to(dev), sigma=sigma, percentiles=(lo, hi))
        assert np.array_equal(edges[0].cpu().numpy().astype(bool), g[f"{n}.edges"]), n
        assert np.array_equal(tables[0], g[f"{n}.bbox"]), n
        big = int(g[f"{n}.big"])
        size = float(g[f"{n}.bbox_area"][big])
        region = planar.find_phantom_region(torch.from_numpy(img).to(dev), size, sigma=sigma, percentiles=(lo, hi))
        assert region.label == big + 1 and region.bbox == tuple(int(v) for v in g[f"{n}.bbox"][big]), n
        assert region.bbox_area == size
        assert np.array_equal(region.image.cpu().numpy(), g[f"{n}.region_image"]), n
        hs, _, di = canny.hough_line(region.image, theta=g[f"{n}.theta"])
        assert np.array_equal(hs.cpu().numpy().astype(np.uint64), g[f"{n}.hspace"]), n
        assert np.array_equal(di, g[f"{n}.dists"]), n
