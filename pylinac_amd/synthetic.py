"""Deterministic synthetic inputs for the BASELINE configs.

Restates the reference's synthetic generator in closed form -- an open field with Gaussian
penumbra and horns (``FilteredFieldLayer`` + ``GaussianFilterLayer``,
pylinac/core/image_generator/layers.py:244-303, 383-393) plus seeded Gaussian noise
(``RandomNoiseLayer``, layers.py:396-407, which the reference leaves UNSEEDED) -- so that the GPU
run and the CPU oracle see the same frames.  Input generation is not part of the measured path.
"""
from __future__ import annotations

import math


def _to_u16(x):
    import torch


    x = x.round().clamp_(0, 65535).to(torch.int32)
    return (x & 0xFFFF).to(torch.int16).view(torch.uint16)


def epid_open_field_frames(n: int, h: int = 1024, w: int = 1024, seed0: int = 1000,
                           device="cpu", pixel_mm: float = 0.336, field_mm: float = 200.0,
                           background: float = 2000.0, plateau: float = 40000.0,
                           blur_mm: float = 2.0, noise_frac: float = 0.01,
                           bad_pixels: int = 8):
    """Config #2: n frames h x w uint16; frame i uses seed ``seed0 + i``: background 2 000, a
    20 cm square field (plateau 40 000, 3 % Gaussian horn dip sigma 32 mm, 2 mm Gaussian
    penumbra), 1 % Gaussian noise, ``bad_pixels`` dead/hot pixels (0 / 65535)."""
    import torch

    device = torch.device(device)
    ys = torch.arange(h, dtype=torch.float32, device=device)
    xs = torch.arange(w, dtype=torch.float32, device=device)
    out = torch.empty((n, h, w), dtype=torch.uint16, device=device)
    s = blur_mm / pixel_mm * math.sqrt(2.0)
    half = field_mm / pixel_mm / 2
    for i in range(n):
        g = torch.Generator(device=device)
        g.manual_seed(seed0 + i)
        off = (torch.rand(2, generator=g, device=device) - 0.5) * 10.0  # +-5 px centre jitter
        cy, cx = (h - 1) / 2 + off[0], (w - 1) / 2 + off[1]
        fy = 0.5 * (torch.erf((ys - (cy - half)) / s) - torch.erf((ys - (cy + half)) / s))
        fx = 0.5 * (torch.erf((xs - (cx - half)) / s) - torch.erf((xs - (cx + half)) / s))
        r2 = ((ys - cy)[:, None] ** 2 + (xs - cx)[None, :] ** 2) * pixel_mm**2
        horn = 1.0 - (0.03 * 65535.0 / plateau) * torch.exp(-r2 / (2 * 32.0**2))
        img = background + (plateau - background) * fy[:, None] * fx[None, :] * horn
        img = img + torch.randn((h, w), generator=g, device=device) * (noise_frac * plateau)
        if bad_pixels:
            pos = torch.randint(0, h * w, (bad_pixels,), generator=g, device=device)
            val = (torch.rand(bad_pixels, generator=g, device=device) > 0.5).float() * 65535.0
            img.view(-1)[pos] = val
        out[i] = _to_u16(img)
    return out


def wl_frames(n: int, h: int = 1024, w: int = 1024, seed0: int = 3000, pixel_mm: float = 0.336,
              field_mm: float = 20.0, bb_mm: float = 5.0, bb_alpha: float = -0.8, blur_mm: float = 1.5,
              max_offset_mm: float = 1.0, return_truth: bool = False, noise_sigma: float = 0.0):
    """Config #4: n Winston-Lutz frames h x w uint16 on the HOST (numpy), frame i from
    ``np.random.default_rng(seed0 + i)``.  Closed-form restatement of the reference's generator recipe
    (``generate_winstonlutz`` with ``PerfectFieldLayer`` 20 x 20 mm, ``PerfectBBLayer`` 5 mm ``alpha=-0.8`` at a random
    sub-pixel offset of at most 1 mm, ``GaussianFilterLayer(1.5 mm)``; pylinac/core/image_generator/utils.py:139-263,
    layers.py:80-134, 187-243, 365-393; the ``SyntheticWLMixin`` recipe, tests_basic/test_winstonlutz.py:1244-1300):
    zeros -> field rectangle = 65535 (polygon fill: pixel centres inside the half-integer bounds) -> BB disk adds
    int(65535 * alpha) with clipping (``draw.disk``: strict ellipse inequality) -> ``skimage.filters.gaussian`` =
    ``ndimage.gaussian_filter(float image, sigma, mode="nearest", truncate=4)`` -> truncation to uint16.
    The blur only touches the neighbourhood of the field, so it is evaluated on a centred crop.
    ``noise_sigma`` > 0 adds the reference's dark-current layer on top (``RandomNoiseLayer(sigma)``, layers.py:396-407:
    N(0, sigma * 65535) over the whole frame through ``clip_add``), drawn from the frame's own seeded generator AFTER the
    offsets, so the noise-free and the noisy variant of a seed share their geometry."""
    import numpy as np

    def blur(img, sigma):
        """ndimage.gaussian_filter(img, sigma, mode="nearest", truncate=4): scipy's kernel formula, one pass per axis"""
        lw = int(4.0 * sigma + 0.5)
        k = np.exp(-0.5 / (sigma * sigma) * np.arange(-lw, lw + 1) ** 2)
        k /= k.sum()
        for axis in (0, 1):
            p = np.pad(img, [(lw, lw) if a == axis else (0, 0) for a in (0, 1)], mode="edge")
            img = sum(k[j] * np.take(p, range(j, j + img.shape[axis]), axis=axis) for j in range(2 * lw + 1))
        return img

    out = np.zeros((n, h, w), dtype=np.uint16)
    truth = np.zeros((n, 4), dtype=np.float64)          # field x, field y, bb x, bb y (generator's nominal centres)
    sigma = blur_mm / pixel_mm
    half = int(round(field_mm / pixel_mm)) // 2 + int(4 * sigma + 0.5) + int(max_offset_mm / pixel_mm) + 8
    ext = int(round(field_mm / pixel_mm))
    ext += ext % 2                                       # even_round
    rad = bb_mm / 2 / pixel_mm
    for i in range(n):
        rng = np.random.default_rng(seed0 + i)
        f_off = rng.uniform(-1.0, 1.0, 2)                # field centre jitter (pixels)
        b_off = rng.uniform(-max_offset_mm, max_offset_mm, 2) / pixel_mm
        fcy, fcx = h / 2 - 0.5 + f_off[0], w / 2 - 0.5 + f_off[1]
        bcy, bcx = h / 2 - 0.5 + b_off[0], w / 2 - 0.5 + b_off[1]
        r0, c0 = h // 2 - half, w // 2 - half
        yy, xx = np.mgrid[r0:r0 + 2 * half, c0:c0 + 2 * half].astype(np.float64)
        img = np.zeros((2 * half, 2 * half), dtype=np.float64)
        img[(yy >= fcy - ext / 2) & (yy <= fcy + ext / 2) & (xx >= fcx - ext / 2) & (xx <= fcx + ext / 2)] = 65535.0
        disk = ((yy - bcy) / rad) ** 2 + ((xx - bcx) / rad) ** 2 < 1
        img[disk] = np.clip(img[disk] + float(int(65535 * bb_alpha)), 0, 65535)
        img = blur(img, sigma)
        out[i, r0:r0 + 2 * half, c0:c0 + 2 * half] = img.astype(np.uint16)
        if noise_sigma > 0:
            noise = rng.normal(0.0, noise_sigma * 65535.0, size=(h, w))
            out[i] = np.clip(out[i].astype(float) + noise, 0, 65535).astype(np.uint16)     # clip_add (layers.py:12-18)
        truth[i] = (fcx, fcy, bcx, bcy)
    return (out, truth) if return_truth else out


# CTP528 line-pair regions (pylinac/ct.py:1417-1503): start / end as fractions of the circle profile, bars, gap (cm)
_CTP528_REGIONS = ((0.0, 0.107, 2, 0.5), (0.107, 0.173, 3, 0.25), (0.173, 0.236, 4, 0.167), (0.236, 0.286, 4, 0.125),
                   (0.286, 0.335, 4, 0.1), (0.335, 0.387, 5, 0.083), (0.387, 0.434, 5, 0.071), (0.434, 0.479, 5, 0.063))


def catphan_volume(seed: int = 4000, n_slices: int = 80, size: int = 512, mm_per_pixel: float = 0.5,
                   noise_hu: float = 8.0, return_truth: bool = False):
    """Config #5: one analytic CatPhan-504-like volume [n_slices, size, size] int16 on the HOST
    from ``np.random.default_rng(seed)``: air (-1000 HU) around a 200 mm cylinder (90 HU) whose centre drifts linearly
    with z (a slightly tilted phantom); a HU module (eight inserts on the 58.4 mm circle incl. two air bubbles) around
    z = 0.3 n; a spatial-resolution module around z = 0.55 n with the eight CTP528 line-pair groups (0.1 ... 0.8 lp/mm,
    bar = gap widths of pylinac/ct.py:1417-1503) as 1000 HU bars on the 47 mm circle, each group centred in its
    region of the reference's circle profile (start angle pi, counter-clockwise); a couch bar below; Gaussian noise."""
    import numpy as np

    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:size, 0:size].astype(np.float64)
    c0 = size / 2 - 0.5 + rng.uniform(-4, 4, 2)                    # (row, col) centre at z = 0
    tilt = rng.uniform(-0.06, 0.06, 2)                             # pixels per slice
    hu_c, res_c = int(round(0.3 * n_slices)), int(round(0.55 * n_slices))
    hu_half, res_half = max(n_slices // 16, 2), max(n_slices // 14, 3)
    out = np.empty((n_slices, size, size), dtype=np.int16)
    circ_mm = 2 * np.pi * 47.0
    for z in range(n_slices):
        cy, cx = c0[0] + tilt[0] * z, c0[1] + tilt[1] * z
        dy, dx = (y - cy) * mm_per_pixel, (x - cx) * mm_per_pixel
        r = np.hypot(dy, dx)
        img = np.full((size, size), -1000.0)
        img[r < 100.0] = 90.0
        if abs(z - hu_c) <= hu_half:
            for k, hu in enumerate((-1000, 340, -200, 950, -100, 120, -1000, 990)):
                a = k * np.pi / 4 + np.pi / 2                       # air bubbles at the top and the bottom
                img[np.hypot(dy - 58.4 * np.sin(a), dx - 58.4 * np.cos(a)) < 6.0] = hu
        if abs(z - res_c) <= res_half:
            # position along the reference's circle profile: sample i sits at angle pi + 2 pi (1 - i / L) (ccw reversal)
            frac = np.mod(1.0 - (np.arctan2(dy, dx) - np.pi) / (2 * np.pi), 1.0)
            ring = np.abs(r - 47.0) < 4.0
            for lo, hi, bars, gap_cm in _CTP528_REGIONS:
                width = gap_cm * 10.0 / circ_mm                       # bar (= gap) width as a fraction of the circle
                start = (lo + hi) / 2 - (2 * bars - 1) * width / 2
                for b in range(bars):
                    s = start + 2 * b * width
                    img[ring & (frac >= s) & (frac < s + width)] = 1000.0
        img[int(size * 0.955):int(size * 0.975), int(size * 0.15):int(size * 0.85)] = 200.0     # couch
        img += rng.normal(0, noise_hu, img.shape)
        out[z] = np.clip(np.round(img), -1024, 3000).astype(np.int16)
    if return_truth:
        return out, dict(center0=c0, tilt=tilt, hu_slice=hu_c, resolution_slice=res_c)
    return out


def pf_frames(n: int, h: int = 768, w: int = 1024, seed0: int = 2000, device="cpu", pixel_mm: float = 0.390625,
              pickets: int = 10, picket_spacing_mm: float = 15.0, gap_mm: float = 2.0, blur_mm: float = 2.0,
              offset_sigma_mm: float = 0.2, background: float = 2000.0, peak: float = 50000.0,
              noise_frac: float = 0.001):
    """Config #3: n picket-fence frames h x w uint16 (AS1000 geometry, SID 1000), UP_DOWN pickets:
    ``pickets`` strips of ``gap_mm`` every ``picket_spacing_mm`` with a per-picket offset N(0, 0.2 mm), blurred by a 2 mm
    Gaussian (closed form: the difference of two error functions -- ``FilteredFieldLayer`` strips + ``GaussianFilterLayer``
    of the reference's ``generate_picketfence``, pylinac/core/image_generator/utils.py:78-136), + N(0, 0.001 * 65535)
    noise (``RandomNoiseLayer``), frame i from seed ``seed0 + i``."""
    import torch

    device = torch.device(device)
    xs = (torch.arange(w, dtype=torch.float64, device=device) - (w / 2 - 0.5)) * pixel_mm
    out = torch.empty((n, h, w), dtype=torch.uint16, device=device)
    s = blur_mm * math.sqrt(2.0)
    for i in range(n):
        g = torch.Generator(device=device)
        g.manual_seed(seed0 + i)
        off = torch.randn(pickets, generator=g, device=device, dtype=torch.float64) * offset_sigma_mm
        centres = (torch.arange(pickets, dtype=torch.float64, device=device) - (pickets - 1) / 2) * picket_spacing_mm + off
        prof = (0.5 * (torch.erf((xs[None, :] - centres[:, None] + gap_mm / 2) / s)
                       - torch.erf((xs[None, :] - centres[:, None] - gap_mm / 2) / s))).sum(dim=0)
        img = background + peak * prof[None, :].expand(h, w)
        img = img + torch.randn((h, w), generator=g, device=device, dtype=torch.float64) * (noise_frac * 65535.0)
        out[i] = _to_u16(img)
    return out
