"""Deterministic synthetic inputs for the BASELINE configs.

Restates the reference's synthetic generator in closed form -- an open field with Gaussian
penumbra and horns (``FilteredFieldLayer`` + ``GaussianFilterLayer``,
pylinac/core/image_generator/layers.py:244-303, 383-393) plus seeded Gaussian noise
(``RandomNoiseLayer``, layers.py:396-407, which the reference leaves UNSEEDED) -- so that the GPU
run and the CPU oracle see the same frames.  Input generation is not part of the measured path.
"""
from __future__ import annotations

import math

import torch


def _to_u16(x: torch.Tensor) -> torch.Tensor:
    x = x.round().clamp_(0, 65535).to(torch.int32)
    return (x & 0xFFFF).to(torch.int16).view(torch.uint16)


def epid_open_field_frames(n: int, h: int = 1024, w: int = 1024, seed0: int = 1000,
                           device="cpu", pixel_mm: float = 0.336, field_mm: float = 200.0,
                           background: float = 2000.0, plateau: float = 40000.0,
                           blur_mm: float = 2.0, noise_frac: float = 0.01,
                           bad_pixels: int = 8) -> torch.Tensor:
    """Config #2: n frames h x w uint16; frame i uses seed ``seed0 + i``: background 2 000, a
    20 cm square field (plateau 40 000, 3 % Gaussian horn dip sigma 32 mm, 2 mm Gaussian
    penumbra), 1 % Gaussian noise, ``bad_pixels`` dead/hot pixels (0 / 65535)."""
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
