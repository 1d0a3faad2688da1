"""The headline step with the frames arriving from HOST memory ("PCIe-inclusive"): 256 x 1024^2 uint16 frames in
pinned host memory -> the device -> EpidPipeline -> the [N, 9] records back to the host.  Never the bench's `value` (that is
measured with the frames resident in HBM); this is what a caller pays who hands over host buffers.  Three forms:
  serial      one asynchronous copy of the batch, then EpidPipeline.run (round 5's figure)
  overlapped  EpidPipeline.run_from_host: the batch in `chunks` pieces, piece k + 1 copied while piece k is processed
  dicom       the same frames as 256 Part-10 FILES' bytes (pinned): copy of the raw bytes + pl_dicom_decode + the step,
              overlapped the same way (dicom.load_frames's kernel on each piece)
    python scripts/time_pcie_inclusive.py [frames=256] [passes=10]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from pylinac_amd import dicom  # noqa: E402
from pylinac_amd.pipeline import EpidPipeline  # noqa: E402
from pylinac_amd.synthetic import epid_open_field_frames  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
host = epid_open_field_frames(n, 1024, 1024, seed0=1000, device=dev).cpu().pin_memory()
pipe = EpidPipeline(n, 1024, 1024, dev)
stage = torch.empty_like(host, device=dev)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(passes):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / passes


def serial():
    stage.copy_(host, non_blocking=True)
    return pipe.run(stage).record().cpu()


ref = serial()
dt = timed(serial)
dc = timed(lambda: stage.copy_(host, non_blocking=True))
print(f"pcie-inclusive (serial): {dt * 1e3:.3f} ms per {n} frames = {n / dt:.0f} images/s; the copy alone {dc * 1e3:.3f} ms = "
      f"{host.numel() * 2 / dc / 1e9:.1f} GB/s host->device", flush=True)
for chunks in (2, 4, 8, 16):
    got = pipe.run_from_host(host, chunks).record().cpu()
    assert torch.equal(torch.nan_to_num(got), torch.nan_to_num(ref)), "run_from_host changed the records"
    d = timed(lambda: pipe.run_from_host(host, chunks).record().cpu())
    print(f"pcie-inclusive (overlapped, {chunks} chunks): {d * 1e3:.3f} ms per {n} frames = {n / d:.0f} images/s", flush=True)

# ---- the same frames as Part-10 files: a 132-byte preamble + a short header + 2 MiB of Pixel Data each
hdr_len = 132 + 10 * 2 + 330                     # any even header length: the decode takes every alignment
files = torch.zeros((n, hdr_len + 1024 * 1024 * 2), dtype=torch.uint8).pin_memory()
files[:, hdr_len:] = host.view(torch.uint8).reshape(n, -1)
dev_files = torch.empty_like(files, device=dev)
offs = torch.arange(n, dtype=torch.int64, device=dev) * files.shape[1] + hdr_len
copy_stream = torch.cuda.Stream()


def from_files(chunks=8):
    per = -(-n // chunks)
    main = torch.cuda.current_stream()
    copy_stream.wait_stream(main)
    for lo in range(0, n, per):
        m = min(per, n - lo)
        with torch.cuda.stream(copy_stream):
            dev_files[lo:lo + m].copy_(files[lo:lo + m], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        main.wait_event(ev)
        fr = dicom.decode_frames(dev_files.view(-1), offs[lo:lo + m], rows=1024, cols=1024, bits_allocated=16, bits_stored=16,
                                 pixel_representation=0, device=dev)
        stage[lo:lo + m] = fr
    return pipe.run(stage).record().cpu()


got = from_files()
assert torch.equal(torch.nan_to_num(got), torch.nan_to_num(ref)), "the decoded files give other records"
d = timed(from_files)
print(f"pcie-inclusive (Part-10 bytes -> pl_dicom_decode -> step, 8 chunks): {d * 1e3:.3f} ms per {n} files = {n / d:.0f} images/s",
      flush=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
dev_files.copy_(files)
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    fr = dicom.decode_frames(dev_files.view(-1), offs, rows=1024, cols=1024, bits_allocated=16, bits_stored=16,
                             pixel_representation=0, device=dev)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"pl_dicom_decode alone: {ms:.3f} ms per {n} frames = {2 * n * 2 * 1024 * 1024 / ms / 1e6:.0f} GB/s (read + write)", flush=True)
