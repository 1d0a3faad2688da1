"""DICOM native Pixel Data on the device (the DICOM half): the step before the hot path.

``DicomImage`` mirrors the array-producing part of ``pylinac.core.image.DicomImage`` (pylinac/core/image.py:1383-1444, the
properties :1491-1578); ``load_frames`` is the batched form the reference does not have: every file's bytes go to the GPU as
they are, ONE ``pl_dicom_decode`` launch turns the (arbitrarily aligned) Pixel Data values into ``[N, H, W]`` frames -- the
container dtype, ``astype(dtype)`` or the rescaled float64 of ``_rescale_dicom_values`` (image.py:363-389) -- and nothing
comes back to the host.

The reference gets ``pixel_array`` from pydicom (``pydicom>=2.0,<3``, pyproject.toml:40; absent from every environment this
build reaches).  What is restated here is pydicom's documented native path -- ``numpy_handler.get_pixeldata``:
``np.frombuffer(PixelData[:expected_len], pixel_dtype(ds))`` reshaped to (NumberOfFrames, Rows, Columns) -- and the reading
of a Part-10 stream as far as that path needs it (PS3.10 section 7.1 preamble + ``DICM``, PS3.5 section 7.1 data elements,
explicit / implicit VR little endian and explicit VR big endian; sequences are skipped, never entered).  Compressed
(encapsulated) transfer syntaxes belong to pydicom's codec plug-ins, not to this path: ``NotImplementedError``.
"""
from __future__ import annotations

import io
import struct
from pathlib import Path

import numpy as np
import torch

from . import _lib
from ._lib import PL_F32, PL_F64, PL_I16, PL_I32, PL_U8, PL_U16, check
from .geometry import Point

MM_PER_INCH = 25.4

IMPLICIT_LE, EXPLICIT_LE, EXPLICIT_BE = "1.2.840.10008.1.2", "1.2.840.10008.1.2.1", "1.2.840.10008.1.2.2"
_LONG_VRS = {b"OB", b"OD", b"OF", b"OL", b"OV", b"OW", b"SQ", b"UC", b"UN", b"UR", b"UT", b"SV", b"UV"}

# (group, element) -> (keyword, VR): the elements the image classes read (the VR column serves implicit-VR streams)
_TAGS = {
    (0x0002, 0x0010): ("TransferSyntaxUID", "UI"), (0x0008, 0x0016): ("SOPClassUID", "UI"), (0x0008, 0x0018): ("SOPInstanceUID", "UI"),
    (0x0008, 0x0060): ("Modality", "CS"), (0x0008, 0x0070): ("Manufacturer", "LO"), (0x0008, 0x0022): ("AcquisitionDate", "DA"),
    (0x0008, 0x0020): ("StudyDate", "DA"), (0x0008, 0x0023): ("ContentDate", "DA"), (0x0008, 0x0008): ("ImageType", "CS"),
    (0x0018, 0x0050): ("SliceThickness", "DS"), (0x0018, 0x0088): ("SpacingBetweenSlices", "DS"),
    (0x0018, 0x1110): ("DistanceSourceToDetector", "DS"),
    (0x0020, 0x000E): ("SeriesInstanceUID", "UI"), (0x0020, 0x0013): ("InstanceNumber", "IS"),
    (0x0020, 0x0032): ("ImagePositionPatient", "DS"), (0x0020, 0x1041): ("SliceLocation", "DS"),
    (0x0028, 0x0002): ("SamplesPerPixel", "US"), (0x0028, 0x0004): ("PhotometricInterpretation", "CS"),
    (0x0028, 0x0008): ("NumberOfFrames", "IS"), (0x0028, 0x0010): ("Rows", "US"), (0x0028, 0x0011): ("Columns", "US"),
    (0x0028, 0x0030): ("PixelSpacing", "DS"), (0x0028, 0x0100): ("BitsAllocated", "US"), (0x0028, 0x0101): ("BitsStored", "US"),
    (0x0028, 0x0102): ("HighBit", "US"), (0x0028, 0x0103): ("PixelRepresentation", "US"),
    (0x0028, 0x1040): ("PixelIntensityRelationship", "CS"), (0x0028, 0x1041): ("PixelIntensityRelationshipSign", "SS"),
    (0x0028, 0x1052): ("RescaleIntercept", "DS"), (0x0028, 0x1053): ("RescaleSlope", "DS"),
    (0x3002, 0x0011): ("ImagePlanePixelSpacing", "DS"), (0x3002, 0x000D): ("XRayImageReceptorTranslation", "DS"),
    (0x3002, 0x0022): ("RadiationMachineSAD", "DS"), (0x3002, 0x0026): ("RTImageSID", "DS"),
    (0x300A, 0x011E): ("GantryAngle", "DS"), (0x300A, 0x0120): ("BeamLimitingDeviceAngle", "DS"),
    (0x300A, 0x0122): ("PatientSupportAngle", "DS"),
    (0x7FE0, 0x0008): ("FloatPixelData", "OF"), (0x7FE0, 0x0009): ("DoubleFloatPixelData", "OD"), (0x7FE0, 0x0010): ("PixelData", "OW"),
}
_BULK = {"PixelData", "FloatPixelData", "DoubleFloatPixelData"}


class Metadata:
    """The parsed elements by pydicom keyword: attribute access raises ``AttributeError`` for an absent tag and ``get`` returns
    a default, like ``pydicom.Dataset`` -- which is all the reference's image code asks of ``self.metadata``."""

    def __init__(self, values: dict):
        self.__dict__["_values"] = values

    def __getattr__(self, name):
        try:
            return self._values[name]
        except KeyError:
            raise AttributeError(name) from None

    def __contains__(self, name) -> bool:
        return name in self._values

    def get(self, name, default=None):
        return self._values.get(name, default)

    def keys(self):
        return self._values.keys()


def _value(vr: str, raw: bytes, big: bool):
    e = ">" if big else "<"
    if vr in ("US", "SS", "UL", "SL", "FL", "FD"):
        code = {"US": "H", "SS": "h", "UL": "I", "SL": "i", "FL": "f", "FD": "d"}[vr]
        n = len(raw) // struct.calcsize(code)
        vals = struct.unpack(e + code * n, raw[:n * struct.calcsize(code)])
        return vals[0] if n == 1 else list(vals)
    text = raw.decode("latin-1").rstrip(" \x00")
    if vr in ("DS", "IS"):
        parts = [p.strip() for p in text.split("\\")] if text else []
        conv = float if vr == "DS" else (lambda s: int(float(s)))
        vals = [conv(p) for p in parts if p != ""]
        return None if not vals else (vals[0] if len(vals) == 1 else vals)
    if "\\" in text and vr in ("CS", "LO", "SH", "UI"):
        return text.split("\\")
    return text


def _skip_undefined(buf: memoryview, pos: int, explicit: bool, big: bool) -> int:
    """-> the position after the Sequence Delimitation Item that closes an undefined-length value starting at ``pos``
    (PS3.5 section 7.5: items (FFFE,E000) of defined or undefined length, closed by (FFFE,E0DD))."""
    e = ">" if big else "<"
    while True:
        g, el, ln = struct.unpack_from(e + "HHI", buf, pos)
        pos += 8
        if (g, el) == (0xFFFE, 0xE0DD):
            return pos
        if (g, el) != (0xFFFE, 0xE000):
            raise ValueError("malformed sequence: an item tag was expected")
        if ln != 0xFFFFFFFF:
            pos += ln
            continue
        while True:                                        # an item of undefined length: data elements until (FFFE,E00D)
            g, el = struct.unpack_from(e + "HH", buf, pos)
            if (g, el) == (0xFFFE, 0xE00D):
                pos += 8
                break
            pos, _, _, _, _ = _element(buf, pos, explicit, big)


def _element(buf: memoryview, pos: int, explicit: bool, big: bool):
    """One data element at ``pos`` -> (position after it, (group, element), VR or None, value offset, value length)."""
    e = ">" if big else "<"
    g, el = struct.unpack_from(e + "HH", buf, pos)
    pos += 4
    vr = None
    if explicit and g != 0xFFFE:
        vrb = bytes(buf[pos:pos + 2])
        vr = vrb.decode("latin-1")
        if vrb in _LONG_VRS:
            (ln,) = struct.unpack_from(e + "I", buf, pos + 4)
            pos += 8
        else:
            (ln,) = struct.unpack_from(e + "H", buf, pos + 2)
            pos += 4
    else:
        (ln,) = struct.unpack_from(e + "I", buf, pos)
        pos += 4
    start = pos
    if ln == 0xFFFFFFFF:
        if (g, el) == (0x7FE0, 0x0010):
            raise NotImplementedError("encapsulated (compressed) Pixel Data: decoded by pydicom's codec plug-ins, not by this path")
        return _skip_undefined(buf, pos, explicit, big), (g, el), vr, start, -1
    return pos + ln, (g, el), vr, start, ln


def read_part10(source) -> tuple[Metadata, np.ndarray]:
    """``pydicom.dcmread(source, force=True)`` as far as the image classes need it -> (metadata, the file's bytes as a uint8
    array).  ``metadata.PixelData`` (``FloatPixelData`` / ``DoubleFloatPixelData``) is the pair (offset, length) of the value
    inside those bytes -- the samples are never copied on the host."""
    if isinstance(source, (bytes, bytearray, memoryview, np.ndarray)):
        data = np.frombuffer(bytes(source) if not isinstance(source, np.ndarray) else source.tobytes(), dtype=np.uint8)
    elif isinstance(source, (str, Path)):
        data = np.fromfile(str(source), dtype=np.uint8)
    else:
        if isinstance(source, io.IOBase) or hasattr(source, "seek"):
            source.seek(0)
        data = np.frombuffer(source.read(), dtype=np.uint8)
    buf = memoryview(data).cast("B")
    n = len(buf)
    pos = 132 if n >= 132 and bytes(buf[128:132]) == b"DICM" else 0
    values: dict = {}
    ts = IMPLICIT_LE if pos == 0 else EXPLICIT_LE          # (no preamble, force=True: pydicom assumes implicit VR little endian)
    # the File Meta group is always explicit VR little endian (PS3.10 section 7.1)
    while pos + 8 <= n and struct.unpack_from("<H", buf, pos)[0] == 0x0002:
        pos, tag, vr, start, ln = _element(buf, pos, True, False)
        if tag in _TAGS and ln >= 0:
            values[_TAGS[tag][0]] = _value(vr or _TAGS[tag][1], bytes(buf[start:start + ln]), False)
    ts = values.get("TransferSyntaxUID", ts)
    if ts not in (IMPLICIT_LE, EXPLICIT_LE, EXPLICIT_BE):
        if ts == "1.2.840.10008.1.2.1.99":
            raise NotImplementedError("Deflated Explicit VR Little Endian is inflated by pydicom's reader, not by this path")
        explicit, big = True, False                         # every encapsulated syntax is explicit VR little endian
    else:
        explicit, big = ts != IMPLICIT_LE, ts == EXPLICIT_BE
    while pos + 8 <= n:
        pos, tag, vr, start, ln = _element(buf, pos, explicit, big)
        if tag not in _TAGS or ln < 0:
            continue
        key, table_vr = _TAGS[tag]
        if key in _BULK:
            values[key] = (start, ln)
        else:
            values[key] = _value(vr if vr and vr != "UN" else table_vr, bytes(buf[start:start + ln]), big)
    values.setdefault("TransferSyntaxUID", ts)
    return Metadata(values), data


def _layout(meta: Metadata):
    """pydicom.pixel_data_handlers.util.pixel_dtype + get_expected_length for native data -> (torch container dtype, numpy
    dtype string, PL code, bytes per sample, frames, rows, cols, big_endian, (offset, length) of the value)."""
    for need in ("BitsAllocated", "Rows", "Columns", "PixelRepresentation", "SamplesPerPixel"):
        if need not in meta:
            raise AttributeError(f"Unable to convert the pixel data as the following required elements are missing from the dataset: {need}")
    if "PixelData" not in meta:
        raise AttributeError("Unable to convert the pixel data: one of Pixel Data, Float Pixel Data or Double Float Pixel Data "
                             "must be present in the dataset" if "FloatPixelData" not in meta and "DoubleFloatPixelData" not in meta
                             else "Float Pixel Data / Double Float Pixel Data are float32 / float64 arrays as stored: read them with numpy")
    if int(meta.SamplesPerPixel) != 1:
        raise NotImplementedError("SamplesPerPixel = 1 only (the image classes of the hot path are single-channel)")
    bits, rep = int(meta.BitsAllocated), int(meta.PixelRepresentation)
    if bits not in (8, 16, 32):
        raise NotImplementedError(f"BitsAllocated {bits}: 8, 16 and 32 are decoded on the device")
    if rep not in (0, 1):
        raise ValueError(f"Unable to determine the data type to use to contain the Pixel Data as a value of '{rep}' for "
                         "'(0028,0103) Pixel Representation' is invalid")
    tdt = {(8, 0): torch.uint8, (8, 1): torch.int8, (16, 0): torch.uint16, (16, 1): torch.int16, (32, 0): torch.uint32,
           (32, 1): torch.int32}[(bits, rep)]
    code = {8: PL_U8, 16: PL_I16 if rep else PL_U16, 32: PL_I32}[bits]
    frames = int(meta.get("NumberOfFrames") or 1)
    big = meta.get("TransferSyntaxUID") == EXPLICIT_BE
    return tdt, code, bits // 8, frames, int(meta.Rows), int(meta.Columns), big, meta.PixelData


_NP_TO_TORCH = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}


def decode_frames(file_bytes, offsets, rows: int, cols: int, bits_allocated: int, bits_stored: int, pixel_representation: int,
                  big_endian: bool = False, correct_unused_bits: bool = False, out: str = "container", rescale=None,
                  device=None) -> torch.Tensor:
    """``pl_dicom_decode``: frames of one format anywhere inside ``file_bytes`` (uint8 array / tensor; device tensors are used
    in place) -> device tensor [N, rows, cols].  ``out``: "container" | "float32" | "float64"; ``rescale`` = (slope, intercept)
    with "float64".  Raises ``ValueError`` when a frame does not lie inside the buffer (pydicom: "The length of the pixel
    data in the dataset doesn't match the expected length")."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    buf = file_bytes if isinstance(file_bytes, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(file_bytes, dtype=np.uint8))
    buf = buf.to(device=dev, dtype=torch.uint8).contiguous()
    offs = (offsets.to(device=dev, dtype=torch.int64).contiguous() if isinstance(offsets, torch.Tensor)
            else torch.as_tensor(np.asarray(offsets, dtype=np.int64)).to(dev))
    n = int(offs.numel())
    rep = int(pixel_representation)
    cont = {(8, 0): torch.uint8, (8, 1): torch.int8, (16, 0): torch.uint16, (16, 1): torch.int16, (32, 0): torch.uint32,
            (32, 1): torch.int32}[(int(bits_allocated), rep)]
    code = {8: PL_U8, 16: PL_I16 if rep else PL_U16, 32: PL_I32}[int(bits_allocated)]
    odt, ocode = {"container": (cont, code), "float32": (torch.float32, PL_F32), "float64": (torch.float64, PL_F64)}[out]
    res = torch.empty((n, rows, cols), dtype=odt, device=dev)
    status = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    slope, intercept = (float(rescale[0]), float(rescale[1])) if rescale is not None else (1.0, 0.0)
    check(_lib.load().pl_dicom_decode(buf.data_ptr(), buf.numel(), offs.data_ptr(), n, rows, cols, int(bits_allocated),
                                      int(bits_stored), rep, int(bool(big_endian)), int(bool(correct_unused_bits)), res.data_ptr(),
                                      ocode, int(rescale is not None), slope, intercept, status.data_ptr(),
                                      torch.cuda.current_stream(dev).cuda_stream), "pl_dicom_decode")
    res._pl_status = status                                  # checked by the caller that wants the error (one sync)
    return res


def _check_status(frames: torch.Tensor) -> torch.Tensor:
    st = getattr(frames, "_pl_status", None)
    if st is not None and bool(st.any()):
        raise ValueError("The length of the pixel data in the dataset doesn't match the expected length; the dataset may be "
                         "corrupted")
    return frames


def load_frames(sources, dtype=None, raw_pixels: bool = False, invert_pixels: bool | None = None,
                correct_unused_bits: bool = False, device=None, check: bool = True):
    """The batched loader: Part-10 files (paths, bytes or file objects; every file may hold several frames) of ONE pixel
    format and frame size -> (device tensor [N, H, W], list of per-file metadata).  Per file exactly what
    ``DicomImage.__init__`` does (image.py:1431-1444): ``pixel_array`` [``.astype(dtype)``] then ``_rescale_dicom_values``;
    files with both rescale tags take the fused float64 form of the kernel when every file carries the SAME slope and
    intercept (a series), otherwise the frames are decoded once and rescaled file by file."""
    from .image import rescale_dicom_values

    metas, blobs = [], []
    for s in sources:
        m, b = read_part10(s)
        metas.append(m)
        blobs.append(b)
    lay = [_layout(m) for m in metas]
    tdt, code, ib, _, rows, cols, big, _ = lay[0]
    for l in lay[1:]:
        if (l[0], l[4], l[5], l[6]) != (tdt, rows, cols, big):
            raise ValueError("load_frames: the files differ in pixel format or frame size (the reference's stacks refuse that too)")
    # one host buffer, every file at a 4-byte boundary; frame offsets inside it
    starts, offsets, owner, pos = [], [], [], 0
    for k, (b, l) in enumerate(zip(blobs, lay)):
        starts.append(pos)
        off, ln = l[7]
        frames = l[3]
        expected = frames * rows * cols * ib
        if ln < expected:
            raise ValueError(f"The length of the pixel data in the dataset ({ln} bytes) doesn't match the expected length "
                             f"({expected} bytes). The dataset may be corrupted or there may be an issue with the pixel data handler.")
        for f in range(frames):
            offsets.append(pos + off + f * rows * cols * ib)
            owner.append(k)
        pos += (len(b) + 3) & ~3
    host = np.zeros(pos, dtype=np.uint8)
    for st, b in zip(starts, blobs):
        host[st:st + len(b)] = b
    stored = int(metas[0].get("BitsStored") or ib * 8)
    rep = int(metas[0].PixelRepresentation)
    common = dict(rows=rows, cols=cols, bits_allocated=ib * 8, bits_stored=stored, pixel_representation=rep, big_endian=big,
                  correct_unused_bits=correct_unused_bits, device=device)
    has_rescale = [("RescaleSlope" in m and "RescaleIntercept" in m) for m in metas]
    same = all(has_rescale) and len({(m.RescaleSlope, m.RescaleIntercept) for m in metas}) == 1
    inverts = [bool(invert_pixels or (invert_pixels is None and m.get("PixelIntensityRelationshipSign") == -1)) for m in metas]
    np_dt = None if dtype is None else np.dtype(dtype)
    if raw_pixels or not any(has_rescale):
        if np_dt is None:
            x = decode_frames(host, offsets, out="container", **common)
        elif np_dt in _NP_TO_TORCH:
            x = decode_frames(host, offsets, out={4: "float32", 8: "float64"}[np_dt.itemsize], **common)
        else:
            x = _astype(decode_frames(host, offsets, out="container", **common), np_dt)
        if check:
            _check_status(x)
    elif same and np_dt in (None, np.dtype(np.float64)):
        x = decode_frames(host, offsets, out="float64", rescale=(metas[0].RescaleSlope, metas[0].RescaleIntercept), **common)
        if check:
            _check_status(x)
    else:
        if not all(has_rescale):
            raise ValueError("load_frames: some files carry rescale tags and some do not; load them separately")
        base = decode_frames(host, offsets, out="container", **common)
        if check:
            _check_status(base)
        if np_dt is not None:
            base = _astype(base, np_dt)
        own = np.asarray(owner)
        parts = []
        for k, m in enumerate(metas):
            sel = base[torch.from_numpy(np.flatnonzero(own == k)).to(base.device)].contiguous()
            parts.append(rescale_dicom_values(sel, m.RescaleSlope, m.RescaleIntercept, invert_pixels=False))
        x = torch.cat(parts)
    if not raw_pixels and any(inverts):
        from . import ops

        if x.dtype in (torch.int8, torch.uint32):
            raise NotImplementedError(f"inverting {x.dtype} pixel data is not available on the device (no EPID / CT panel stores "
                                      "it); pass dtype= to widen it first")

        own = np.asarray(owner)
        if all(inverts) and len(own) == len(metas):
            x = ops.invert(x)                                 # max - a + min per file (image.py:383-388), one frame per file
        else:                                                 # a multi-frame file inverts about the extrema of ALL its frames
            x = x.clone()
            for k in np.flatnonzero(inverts):
                idx = np.flatnonzero(own == k)
                a, b = int(idx[0]), int(idx[-1]) + 1
                x[a:b] = ops.invert(x[a:b].reshape(1, (b - a) * rows, cols)).reshape(b - a, rows, cols)
    return x, metas


def _astype(x: torch.Tensor, np_dt: np.dtype) -> torch.Tensor:
    """``ndarray.astype`` for the integer targets the kernel does not write itself: same-size targets are a re-view (numpy's
    wrap-around), anything else goes through int64 (every container value is exact there) and wraps like a C cast."""
    name = {"uint8": torch.uint8, "int8": torch.int8, "uint16": torch.uint16, "int16": torch.int16, "uint32": torch.uint32,
            "int32": torch.int32, "int64": torch.int64, "float32": torch.float32, "float64": torch.float64}.get(np_dt.name)
    if name is None:
        raise NotImplementedError(f"dtype {np_dt} is not available on the device")
    if x.dtype == name:
        return x
    if x.element_size() == np_dt.itemsize and not name.is_floating_point:
        return x.view(name)
    if name.is_floating_point:
        return x.to(torch.int64).to(name) if not x.dtype.is_floating_point else x.to(name)
    wide = x.to(torch.int64)
    bits = np_dt.itemsize * 8
    if bits < 64:
        wide = wide & ((1 << bits) - 1)
        if np_dt.kind == "i":
            wide = torch.where(wide >= (1 << (bits - 1)), wide - (1 << bits), wide)
    signed = {1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[np_dt.itemsize]
    return wide.to(signed).view(name) if signed != name else wide.to(signed)


def _image_base():
    from .image import BaseImage

    return BaseImage


class DicomImage(_image_base()):
    """``pylinac.core.image.DicomImage`` (image.py:1356-1578) over the device decode: ``array`` is a numpy array like every
    class-API image; ``metadata`` answers the tags the reference's properties read."""

    def __init__(self, path, *, dtype=None, dpi: float = None, sid: float = None, sad: float = 1000, raw_pixels: bool = False,
                 invert_pixels: bool | None = None):
        self.path = path
        self._sid, self._dpi, self._sad = sid, dpi, sad
        self._raw_pixels, self._invert_pixels = raw_pixels, invert_pixels
        frames, metas = load_frames([path], dtype=dtype, raw_pixels=raw_pixels, invert_pixels=invert_pixels)
        self.metadata = metas[0]
        tdt = _layout(self.metadata)[0]
        self._original_dtype = np.dtype(str(tdt).replace("torch.", ""))
        arr = _to_numpy(frames)
        self.array = arr[0] if arr.shape[0] == 1 else arr      # pydicom: (rows, cols) for one frame, (frames, rows, cols) else
        self.metrics = []
        self.metric_values = {}

    @property
    def z_position(self) -> float:
        """utilities.z_position: ImagePositionPatient[-1], else SliceLocation"""
        ipp = self.metadata.get("ImagePositionPatient")
        if ipp is not None:
            return ipp[-1]
        return self.metadata.SliceLocation

    @property
    def slice_spacing(self) -> float:
        try:
            return abs(self.metadata.SpacingBetweenSlices)
        except AttributeError:
            return self.metadata.SliceThickness

    @property
    def sid(self) -> float:
        try:
            return float(self.metadata.RTImageSID)
        except (AttributeError, ValueError, TypeError):
            return self._sid

    @property
    def sad(self) -> float:
        try:
            return float(self.metadata.RadiationMachineSAD)
        except (AttributeError, ValueError, TypeError):
            return self._sad

    @property
    def dpi(self) -> float:
        try:
            return self.dpmm * MM_PER_INCH
        except Exception:
            return self._dpi

    @property
    def dpmm(self) -> float:
        dpmm = None
        for tag in ("PixelSpacing", "ImagePlanePixelSpacing"):
            mmpd = self.metadata.get(tag)
            if mmpd is not None:
                dpmm = 1 / mmpd[0]
                break
        if dpmm is not None and self.sid is not None:
            dpmm *= self.sid / self.sad
        elif dpmm is None and self._dpi is not None:
            dpmm = self._dpi / MM_PER_INCH
        return dpmm

    @property
    def cax(self) -> Point:
        try:
            mag_factor = self.sid / self.sad
            x = self.center.x - self.metadata.XRayImageReceptorTranslation[0] * self.dpmm / mag_factor
            y = self.center.y + self.metadata.XRayImageReceptorTranslation[1] * self.dpmm / mag_factor
        except (AttributeError, ValueError, TypeError):
            return self.center
        return Point(x, y)


def _to_numpy(x: torch.Tensor) -> np.ndarray:
    t = x.cpu()
    if t.dtype in (torch.uint16, torch.uint32):            # torch -> numpy has no unsigned 16 / 32 bridge on every version
        signed = t.view(torch.int16 if t.dtype == torch.uint16 else torch.int32).numpy()
        return signed.view(np.uint16 if t.dtype == torch.uint16 else np.uint32)
    return t.numpy()
