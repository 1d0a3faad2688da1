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
              max_offset_mm: float = 1.0, return_truth: bool = False):
    """Config #4: n Winston-Lutz frames h x w uint16 on the HOST (numpy), frame i from
    ``np.random.default_rng(seed0 + i)``.  Closed-form restatement of the reference's generator recipe
    (``generate_winstonlutz`` with ``PerfectFieldLayer`` 20 x 20 mm, ``PerfectBBLayer`` 5 mm ``alpha=-0.8`` at a random
    sub-pixel offset of at most 1 mm, ``GaussianFilterLayer(1.5 mm)``; pylinac/core/image_generator/utils.py:139-263,
    layers.py:80-134, 187-243, 365-393; the ``SyntheticWLMixin`` recipe, tests_basic/test_winstonlutz.py:1244-1300):
    zeros -> field rectangle = 65535 (polygon fill: pixel centres inside the half-integer bounds) -> BB disk adds
    int(65535 * alpha) with clipping (``draw.disk``: strict ellipse inequality) -> ``skimage.filters.gaussian`` =
    ``ndimage.gaussian_filter(float image, sigma, mode="nearest", truncate=4)`` -> truncation to uint16.
    The blur only touches the neighbourhood of the field, so it is evaluated on a centred crop."""
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
        truth[i] = (fcx, fcy, bcx, bcy)
    return (out, truth) if return_truth else out
