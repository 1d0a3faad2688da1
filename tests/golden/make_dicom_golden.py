"""Writes tests/golden/dicom.npz: minimal DICOM Part-10 streams (PS3.10 section 7.1, PS3.5 section 7.1) + the pixel arrays
they encode.  pydicom -- the reader the reference uses (pyproject.toml:40) -- exists in no environment this build reaches,
so the streams come from this script's own struct-based writer and the expected arrays are the arrays that were ENCODED:
the reference's own writer states the layout (pylinac/core/array_utils.py:291-297 `array_to_dicom`: `ds.PixelData =
array.tobytes()`, BitsAllocated = itemsize * 8, PixelRepresentation 0, Explicit VR Little Endian), the standard states the
rest (PS3.5 section 8.2 native format, Annex D).  Cases cover what pydicom's native path distinguishes: 8 / 16 / 32 bits
allocated, unsigned / two's complement, BitsStored < BitsAllocated with non-zero unused bits, explicit / implicit VR little
endian, explicit VR big endian, multi-frame, odd-length Pixel Data with its padding byte, a Pixel Data value at each
alignment, undefined-length sequences before the pixels, rescale tags, PixelIntensityRelationshipSign.

    python tests/golden/make_dicom_golden.py
"""
import json
import struct
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
IMPLICIT_LE, EXPLICIT_LE, EXPLICIT_BE = "1.2.840.10008.1.2", "1.2.840.10008.1.2.1", "1.2.840.10008.1.2.2"
LONG = {"OB", "OD", "OF", "OL", "OW", "SQ", "UC", "UN", "UR", "UT"}


def element(tag, vr, value: bytes, explicit=True, big=False, undefined=False):
    e = ">" if big else "<"
    if len(value) % 2:
        value += b"\x00" if vr in ("OB", "UI", "OW") else b" "
    ln = 0xFFFFFFFF if undefined else len(value)
    head = struct.pack(e + "HH", *tag)
    if not explicit:
        return head + struct.pack(e + "I", ln) + value
    if vr in LONG:
        return head + vr.encode() + b"\x00\x00" + struct.pack(e + "I", ln) + value
    return head + vr.encode() + struct.pack(e + "H", ln) + value


def us(v, big=False):
    return struct.pack((">" if big else "<") + "H", v)


def part10(array, ts=EXPLICIT_LE, bits_stored=None, signed=None, frames=None, extra=(), with_sequence=False, preamble=True,
           pad_text=""):
    explicit, big = ts != IMPLICIT_LE, ts == EXPLICIT_BE
    a = np.asarray(array)
    rows, cols = a.shape[-2:]
    rep = int(a.dtype.kind == "i") if signed is None else int(signed)
    meta = element((0x0002, 0x0010), "UI", ts.encode())
    meta = element((0x0002, 0x0000), "UL", struct.pack("<I", len(meta))) + meta
    ds = b""

    def add(tag, vr, value):
        nonlocal ds
        ds += element(tag, vr, value, explicit, big)

    add((0x0008, 0x0060), "CS", b"RTIMAGE")
    add((0x0008, 0x0070), "LO", ("pylinac_amd fixture" + pad_text).encode())
    if with_sequence:
        # an undefined-length sequence holding one undefined-length item with one element, and a defined-length one
        item = element((0x0008, 0x0100), "SH", b"CODE", explicit, big)
        e = ">" if big else "<"
        seq = (struct.pack(e + "HHI", 0xFFFE, 0xE000, 0xFFFFFFFF) + item + struct.pack(e + "HHI", 0xFFFE, 0xE00D, 0)
               + struct.pack(e + "HHI", 0xFFFE, 0xE000, len(item)) + item + struct.pack(e + "HHI", 0xFFFE, 0xE0DD, 0))
        ds += element((0x0008, 0x1140), "SQ", seq, explicit, big, undefined=True)
    add((0x0028, 0x0002), "US", us(1, big))
    add((0x0028, 0x0004), "CS", b"MONOCHROME2")
    if frames:
        add((0x0028, 0x0008), "IS", str(frames).encode())
    add((0x0028, 0x0010), "US", us(rows, big))
    add((0x0028, 0x0011), "US", us(cols, big))
    for tag, vr, val in extra:
        if tag < (0x0028, 0x0100):
            add(tag, vr, val)
    bits = a.dtype.itemsize * 8
    stored = bits_stored or bits
    add((0x0028, 0x0100), "US", us(bits, big))
    add((0x0028, 0x0101), "US", us(stored, big))
    add((0x0028, 0x0102), "US", us(stored - 1, big))
    add((0x0028, 0x0103), "US", us(rep, big))
    for tag, vr, val in extra:
        if tag >= (0x0028, 0x0100):
            add(tag, vr, val)
    payload = a.astype(a.dtype.newbyteorder(">" if big else "<")).tobytes()
    add((0x7FE0, 0x0010), "OW" if bits > 8 else "OB", payload)
    head = (b"\x00" * 128 + b"DICM" + meta) if preamble else b""
    return head + ds


def main():
    rng = np.random.default_rng(20260930)
    cases = {}

    def case(name, array, expect=None, **kw):
        data = part10(array, **kw)
        cases[name] = dict(file=np.frombuffer(data, dtype=np.uint8), expect=np.asarray(array if expect is None else expect))

    u16 = rng.integers(0, 65536, (48, 64), dtype=np.uint16)
    case("u16_explicit", u16)                                                   # array_to_dicom's own layout
    case("u16_explicit_shifted", u16, pad_text="!!")                            # Pixel Data two bytes further: another alignment
    case("u16_implicit", u16, ts=IMPLICIT_LE)
    case("u16_implicit_nopreamble", u16, ts=IMPLICIT_LE, preamble=False)        # dcmread(force=True) on a bare data set
    case("u16_big_endian", u16, ts=EXPLICIT_BE)
    case("u16_sequence", u16, with_sequence=True)
    i16 = rng.integers(-2000, 3000, (40, 56)).astype(np.int16)
    case("i16_ct", i16, extra=[((0x0028, 0x1052), "DS", b"-1024"), ((0x0028, 0x1053), "DS", b"1.5")])
    # 12 bits stored in 16 allocated, unused bits NOT zero (overlay planes of old files): pydicom 2.x returns the container
    raw12 = rng.integers(0, 65536, (32, 48), dtype=np.uint16)
    case("u16_stored12_dirty", raw12, bits_stored=12)
    case("i16_stored12_dirty", raw12.view(np.int16), bits_stored=12, signed=1)
    u8 = rng.integers(0, 256, (3, 20, 28), dtype=np.uint8)
    case("u8_multiframe", u8, frames=3)
    case("u8_odd", rng.integers(0, 256, (5, 3), dtype=np.uint8))               # 15 bytes + the padding byte
    case("i8", rng.integers(-128, 128, (12, 16)).astype(np.int8))
    case("u32", rng.integers(0, 2**32, (16, 24), dtype=np.uint32))
    case("i32_big_endian", rng.integers(-2**31, 2**31, (16, 24)).astype(np.int32), ts=EXPLICIT_BE)
    case("u16_inverted_sign", u16, extra=[((0x0028, 0x1041), "SS", struct.pack("<h", -1)), ((0x0028, 0x1052), "DS", b"0"),
                                          ((0x0028, 0x1053), "DS", b"1")])
    case("u16_epid_tags", u16, extra=[((0x3002, 0x0011), "DS", b"0.336\\0.336"), ((0x3002, 0x0026), "DS", b"1500.0"),
                                      ((0x3002, 0x0022), "DS", b"1000.0"), ((0x3002, 0x000D), "DS", b"1.5\\-2.25\\-500"),
                                      ((0x300A, 0x011E), "DS", b"90.0")])
    out = {}
    for k, c in cases.items():
        out[f"file__{k}"] = c["file"]
        out[f"expect__{k}"] = c["expect"]
    np.savez_compressed(HERE / "dicom.npz", **out)
    print(json.dumps({k: [int(c["file"].size), list(c["expect"].shape), str(c["expect"].dtype)] for k, c in cases.items()}, indent=1))


if __name__ == "__main__":
    main()
