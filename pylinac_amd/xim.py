"""Varian XIM images (SURVEY.md section 8 "next" row f1): device decoding of the compressed pixel stream.

``decode_xim_pixels`` is the kernel-level entry (``pl_xim_decode``); ``XIM`` mirrors ``pylinac.core.image.XIM``'s
reader (pylinac/core/image.py:1105-1296): same attributes (``img_width_px``, ``img_height_px``, ``bytes_per_pixel``,
``compression``, ``lookup_table``, ``histogram``, ``properties``, ``dpmm``) and the pixel ``array`` as a device
tensor.  The header / property parsing is host I/O like the reference's ``decode_binary`` calls.
"""
from __future__ import annotations

import struct

import numpy as np
import torch

from . import _lib
from ._lib import check

_DTYPES = {1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}
XIM_PROP_INT, XIM_PROP_DOUBLE, XIM_PROP_STRING, XIM_PROP_DOUBLE_ARRAY, XIM_PROP_INT_ARRAY = 0, 1, 2, 4, 5


def decode_xim_pixels(lookup_table_bytes, stream, width: int, height: int, bytes_per_pixel: int,
                      device=None) -> torch.Tensor:
    """XIM._parse_lookup_table + _get_diffs + _parse_compressed_bytes (image.py:1180-1296) on the GPU.
    ``lookup_table_bytes`` / ``stream``: uint8 arrays or tensors (the file's lookup table and the compressed pixel
    buffer that follows its 4-byte length).  -> int8/16/32/64 [height, width] device tensor."""
    if bytes_per_pixel not in _DTYPES:
        raise ValueError("The XIM image has an unsupported bytes per pixel value. "
                         "Raise a ticket on the pylinac Github with this file.")
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())

    def up(a):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint8))
        return t.to(device=dev, dtype=torch.uint8).contiguous()

    lut, buf = up(lookup_table_bytes), up(stream)
    lib = _lib.load()
    work = torch.empty(int(lib.pl_xim_work_bytes(width, height)), dtype=torch.uint8, device=dev)
    out = torch.empty((height, width), dtype=_DTYPES[bytes_per_pixel], device=dev)
    check(lib.pl_xim_decode(lut.data_ptr(), lut.numel(), buf.data_ptr(), buf.numel(), width, height, bytes_per_pixel,
                            out.data_ptr(), work.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
          "pl_xim_decode")
    status = int(work[:4].view(torch.int32)[0])
    if status & 1:
        raise KeyError(3)          # the reference's LOOKUP_CONVERSION has no entry for size code 3
    if status & 2:
        raise ValueError("XIM pixel buffer is shorter than its lookup table implies")
    return out


def _read(f, fmt: str, n: int = 1):
    vals = struct.unpack("<" + fmt * n, f.read(struct.calcsize("<" + fmt) * n))
    return vals[0] if n == 1 else np.array(vals)


def _read_str(f, n: int) -> str:
    return "".join(chr(b) for b in f.read(n) if b != 0)


class XIM:
    """pylinac/core/image.py:1105-1178 (reader) with the pixel decoding on the GPU."""

    def __init__(self, file_path, read_pixels: bool = True, device=None):
        self.path = file_path
        with open(file_path, "rb") as xim:
            self.format_id = _read_str(xim, 8)
            self.format_version = _read(xim, "i")
            self.img_width_px = _read(xim, "i")
            self.img_height_px = _read(xim, "i")
            self.bits_per_pixel = _read(xim, "i")
            self.bytes_per_pixel = _read(xim, "i")
            self.compression = _read(xim, "i")
            if not self.compression:
                pixel_buffer_size = _read(xim, "i")
                raw = np.frombuffer(xim.read(pixel_buffer_size), dtype=np.uint8)
                if read_pixels:
                    dt = {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}[self.bytes_per_pixel]
                    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
                    self.array = torch.from_numpy(raw.view(dt).reshape(self.img_height_px, self.img_width_px).copy()).to(dev)
            else:
                lookup_table_size = _read(xim, "i")
                self.lookup_table = np.frombuffer(xim.read(lookup_table_size), dtype=np.uint8)
                comp_pixel_buffer_size = _read(xim, "i")
                stream = np.frombuffer(xim.read(comp_pixel_buffer_size), dtype=np.uint8)
                if read_pixels:
                    self.array = decode_xim_pixels(self.lookup_table, stream, self.img_width_px, self.img_height_px,
                                                   self.bytes_per_pixel, device=device)
                _read(xim, "i")                                           # uncompressed size (unused by the reference)
            self.num_hist_bins = _read(xim, "i")
            self.histogram = _read(xim, "i", self.num_hist_bins) if self.num_hist_bins else np.array([], dtype=int)
            self.num_properties = _read(xim, "i")
            self.properties = {}
            for _ in range(self.num_properties):
                name = _read_str(xim, _read(xim, "i"))
                tipe = _read(xim, "i")
                if tipe == XIM_PROP_INT:
                    value = _read(xim, "i")
                elif tipe == XIM_PROP_DOUBLE:
                    value = _read(xim, "d")
                elif tipe == XIM_PROP_STRING:
                    value = _read_str(xim, _read(xim, "i"))
                elif tipe == XIM_PROP_DOUBLE_ARRAY:
                    value = _read(xim, "d", int(_read(xim, "i") // 8))
                elif tipe == XIM_PROP_INT_ARRAY:
                    value = _read(xim, "i", int(_read(xim, "i") // 4))
                else:
                    raise ValueError(f"unknown XIM property type {tipe}")
                self.properties[name] = value

    @property
    def dpmm(self) -> float:
        """image.py:1298-1305."""
        if self.properties["PixelWidth"] != self.properties["PixelHeight"]:
            raise ValueError("The XIM image does not have the same pixel height and width")
        return 1 / (10 * self.properties["PixelHeight"])
