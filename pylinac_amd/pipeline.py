"""Batched EPID pipeline = BASELINE.json's metric: filter -> threshold -> profile -> peak.

Per frame (SURVEY.md section 8d config #2 composed with rows a7/a8/a10):

    Image.filter(5, "gaussian")      pylinac/core/image.py:695-712 -> array_utils.py:133
    Image.filter(3, "median")        pylinac/core/image.py:695-712 -> array_utils.py:131
    t = threshold_otsu(frame)        skimage semantics (pylinac/ct.py:3323)
    Image.threshold(t, "high")       pylinac/core/image.py:785-800
    profile = np.mean(frame, 0)      pylinac/picketfence.py:747-750
    FWXMProfile(profile) edges/centre/width   pylinac/core/profile.py:578-611, 322-344

All buffers are allocated once (workspace), every stage is a stream-ordered launch through the
C ABI; nothing returns to the host inside ``run``.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import torch

from . import _lib, ops
from ._lib import check

RECORD_FIELDS = ("otsu_threshold", "n_peaks", "peak_idx", "peak_height", "prominence",
                 "left_edge", "right_edge", "center", "width")

# stages of the default path (3x3 median on frames of width % 8 == 0); other geometries run the separate entry points
STAGES = ("gauss2d", "median3_otsu16", "median3_threshold_colsum", "colsum_to_mean", "find_peaks", "fwxm_record")


@dataclass
class EpidResult:
    frames: torch.Tensor     # uint16 [N,H,W]  thresholded frames
    profile: torch.Tensor    # float64 [N,W]   column-mean profile
    threshold: torch.Tensor  # int32 [N]       Otsu threshold
    fwxm: torch.Tensor       # float64 [N,8]   ops.fwxm_record fields
    status: torch.Tensor     # int32 [N]       0 = ok (find_peaks capacity status)

    def record(self) -> torch.Tensor:
        """float64 [N, 9] per-image scalar record (RECORD_FIELDS) -- what is all-gathered."""
        return torch.cat([self.threshold.to(torch.float64)[:, None], self.fwxm], dim=1)


@dataclass
class EpidPipeline:
    n: int
    h: int
    w: int
    device: torch.device
    sigma: float = 5
    median_size: int = 3
    fwxm_height: float = 50
    # True: per-band column sums + ONE launch for profile -> peaks -> record (pl_colparts_profile_fwxm) instead of memset +
    # 64-bit atomics + three small launches.  Same results; measured on 256 x 1024^2 (scripts/time_epid_tail.py): 0.7675 ms
    # per step against 0.760 -- back-to-back launches on one stream cost next to nothing at that size, so the default (None)
    # turns it on for small batches only, where three launch latencies are a tenth of the step
    fused_tail: bool | None = None
    timings: dict = field(default_factory=dict)

    def __post_init__(self):
        dev = self.device
        n, h, w = self.n, self.h, self.w
        if self.fused_tail is None:
            self.fused_tail = n <= 64
        u16 = dict(dtype=torch.uint16, device=dev)
        self.buf_a = torch.empty((n, h, w), **u16)
        self.buf_b = torch.empty((n, h, w), **u16)
        self.out = torch.empty((n, h, w), **u16)
        self.hist = torch.empty((n, 65536), dtype=torch.int32, device=dev)
        self.thr = torch.empty(n, dtype=torch.int32, device=dev)
        self.vmin = torch.empty(n, dtype=torch.int32, device=dev)
        self.vmax = torch.empty(n, dtype=torch.int32, device=dev)
        self.flag = torch.empty(n, dtype=torch.int32, device=dev)
        self.colsum = torch.empty((n, w), dtype=torch.int64, device=dev)
        self.lib = _lib.load()
        self.bands = -(-h // self.lib.pl_colparts_band_rows())
        self.parts = torch.empty((n, self.bands, w), dtype=torch.int32, device=dev)   # uint32 per-band column sums
        self.profile = torch.empty((n, w), dtype=torch.float64, device=dev)
        self.fwxm = torch.empty((n, 8), dtype=torch.float64, device=dev)
        self.peaks = ops.PeakBatch(
            count=torch.empty(n, dtype=torch.int32, device=dev),
            idx=torch.empty((n, 1), dtype=torch.int32, device=dev),
            left_bases=torch.empty((n, 1), dtype=torch.int32, device=dev),
            right_bases=torch.empty((n, 1), dtype=torch.int32, device=dev),
            props=torch.empty((n, 6, 1), dtype=torch.float64, device=dev),
            status=torch.empty(n, dtype=torch.int32, device=dev),
        )
        self.wts, self.host_wts, self.radius = ops._device_weights(self.sigma, dev)
        self.prm = ops.make_peak_params(w, fwxm_height=self.fwxm_height / 100, max_number=1)

    def run(self, frames: torch.Tensor, events: dict | None = None, chunks=None) -> EpidResult:
        """One pass over a resident batch.  ``events``: optional {stage: [(start, stop), ...]} sink;
        when given, every stage is bracketed by HIP events on the launch stream.  ``chunks``: ``run_from_host``'s plan
        [((first frame, count), copy-done event), ...]; None = the whole batch at once."""
        if frames.dtype != torch.uint16 or tuple(frames.shape) != (self.n, self.h, self.w):
            raise ValueError(f"expected uint16 [{self.n},{self.h},{self.w}] frames")
        if not frames.is_cuda:
            raise ValueError("frames must be resident on the GPU")
        lib = self.lib
        h, w = self.h, self.w
        U16 = _lib.PL_U16
        x = frames.contiguous()
        main = torch.cuda.current_stream()

        def stage(name, fn, stream):
            if events is None:
                check(fn(), name)
                return
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            check(fn(), name)
            e1.record(stream)
            events.setdefault(name, []).append((e0, e1))

        fb = h * w * 2                                   # bytes per uint16 frame
        xp, ap, bp, op = x.data_ptr(), self.buf_a.data_ptr(), self.buf_b.data_ptr(), self.out.data_ptr()
        wts, hwts, pk = self.wts.data_ptr(), self.host_wts.ctypes.data, self.peaks
        hist, thr, vmin, vmax = (t.data_ptr() for t in (self.hist, self.thr, self.vmin, self.vmax))
        flag = self.flag.data_ptr()
        colsum, profile, fwxm = self.colsum.data_ptr(), self.profile.data_ptr(), self.fwxm.data_ptr()
        cnt, idx, lb, rb, props, status = (t.data_ptr() for t in (pk.count, pk.idx, pk.left_bases,
                                                                   pk.right_bases, pk.props, pk.status))

        # 3x3 median on frames of width % 8 == 0 (torch allocations are 256-byte aligned): consumed on the fly
        fused_median = self.median_size == 3 and h > 1 and w % 8 == 0 and (h * w) % 8 == 0

        def filters(lo, m, stream):
            """Image.filter(sigma, "gaussian"): axis 0 then axis 1 in ONE launch (the axis-0 plane stays in LDS), frames
            [lo, lo+m); buf_a is the two-pass fallback's scratch."""
            st, o = stream.cuda_stream, lo * fb
            stage("gauss2d", lambda: lib.pl_gaussian2d(xp + o, bp + o, ap + o, U16, m, h, w, wts, hwts, self.radius, st),
                  stream)

        def rest(lo, m, stream):
            """median -> Otsu -> threshold -> column profile -> FWXM record, frames [lo, lo+m)."""
            st, o = stream.cuda_stream, lo * fb

            def separate_tail():
                stage("colsum_to_mean", lambda: lib.pl_colsum_to_mean(colsum + lo * w * 8, m, w, h,
                                                                      profile + lo * w * 8, st), stream)
                stage("find_peaks", lambda: lib.pl_find_peaks(
                    profile + lo * w * 8, m, w, w, C.byref(self.prm), 1, cnt + lo * 4, idx + lo * 4, lb + lo * 4,
                    rb + lo * 4, props + lo * 48, status + lo * 4, st), stream)
                stage("fwxm_record", lambda: lib.pl_fwxm_record(cnt + lo * 4, idx + lo * 4, props + lo * 48, 1, m,
                                                                fwxm + lo * 64, st), stream)

            if fused_median:
                # Image.filter(3, "median") is never materialised: the Otsu histogram and the threshold + column sums each
                # compute the 3x3 medians of the Gaussian plane on the fly (two reads of that plane instead of median write +
                # two reads of the median plane); buf_a is scratch for frames the one-pass Otsu window cannot hold
                stage("median3_otsu16", lambda: lib.pl_median3_otsu16(bp + o, ap + o, U16, m, h, w, None, None, thr + lo * 4,
                                                                      vmin + lo * 4, vmax + lo * 4, flag + lo * 4,
                                                                      hist + lo * 65536 * 4, st), stream)
                if not self.fused_tail:
                    stage("median3_threshold_colsum", lambda: lib.pl_median3_threshold_colsum_u16(
                        bp + o, op + o, m, h, w, thr + lo * 4, colsum + lo * w * 8, st), stream)
                    separate_tail()
                    return
                # threshold + per-band column sums (plain stores), then ONE launch for mean profile -> peaks -> FWXM record
                parts = self.parts.data_ptr() + lo * self.bands * w * 4
                stage("median3_threshold_colsum", lambda: lib.pl_median3_threshold_colparts_u16(
                    bp + o, op + o, m, h, w, thr + lo * 4, parts, st), stream)
                stage("profile_fwxm", lambda: lib.pl_colparts_profile_fwxm(
                    parts, m, self.bands, w, h, C.byref(self.prm), 1, profile + lo * w * 8, cnt + lo * 4, idx + lo * 4,
                    lb + lo * 4, rb + lo * 4, props + lo * 48, status + lo * 4, fwxm + lo * 64, st), stream)
                return
            stage("median3", lambda: lib.pl_median2d(bp + o, ap + o, U16, m, h, w, self.median_size, st), stream)
            med = ap + o
            stage("otsu16", lambda: lib.pl_otsu16(med, U16, m, h * w, None, None, thr + lo * 4,
                                                  vmin + lo * 4, vmax + lo * 4, flag + lo * 4,
                                                  hist + lo * 65536 * 4, st), stream)
            stage("threshold_colsum", lambda: lib.pl_threshold_colsum_u16(med, op + o, m, h, w, thr + lo * 4,
                                                                          colsum + lo * w * 8, st), stream)
            separate_tail()

        if chunks is None:
            filters(0, self.n, main)
            rest(0, self.n, main)
        else:
            # run_from_host: chunk k's kernels wait for chunk k's copy only; the copy engine streams chunk k + 1 meanwhile
            for (lo, m), ready in chunks:
                main.wait_event(ready)
                filters(lo, m, main)
                done = torch.cuda.Event()
                done.record(main)
                self._stage_free.append(done)              # the staging rows may be overwritten once the Gaussian has read them
                rest(lo, m, main)
        return EpidResult(self.out, self.profile, self.thr, self.fwxm, self.peaks.status)

    def run_from_host(self, host_frames: torch.Tensor, chunks: int = 8, events: dict | None = None) -> EpidResult:
        """The step for frames that arrive in (pinned) HOST memory -- what a loader hands over.
        The batch is cut into ``chunks`` pieces; a copy stream moves piece k + 1 over PCIe while the launch stream runs the
        whole pipeline on piece k, so the pass costs the copy plus ONE piece's kernels instead of copy + all kernels
        (scripts/time_pcie_inclusive.py).  Results are those of ``run`` on the same frames (every stage is per frame)."""
        if host_frames.dtype != torch.uint16 or tuple(host_frames.shape) != (self.n, self.h, self.w):
            raise ValueError(f"expected uint16 [{self.n},{self.h},{self.w}] frames")
        if not hasattr(self, "_stage"):
            self._stage = torch.empty((self.n, self.h, self.w), dtype=torch.uint16, device=self.device)
            self._copy_stream = torch.cuda.Stream(device=self.device)
            self._stage_free = []
        chunks = max(1, min(int(chunks), self.n))
        per = -(-self.n // chunks)
        cs = self._copy_stream
        for ev in self._stage_free:                            # the previous pass's Gaussians have read the staging buffer
            cs.wait_event(ev)
        self._stage_free = []
        plan = []
        for lo in range(0, self.n, per):
            m = min(per, self.n - lo)
            with torch.cuda.stream(cs):
                self._stage[lo:lo + m].copy_(host_frames[lo:lo + m], non_blocking=True)
                ready = torch.cuda.Event()
                ready.record(cs)
            plan.append(((lo, m), ready))
        return self.run(self._stage, events=events, chunks=plan)
