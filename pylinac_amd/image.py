"""Mirror of the image classes on the hot path: ``pylinac.core.image.BaseImage`` mutators
(pylinac/core/image.py:695-926) on a numpy-backed :class:`ArrayImage` (drop-in for
``pylinac.core.image.ArrayImage``, image.py:1815-1869) and on a device-resident
:class:`ImageBatch` ``[N,H,W]`` -- the batched fast path the reference does not have.

``ArrayImage.array`` stays a public, mutable numpy attribute that every mutator REBINDS
(``self.array = f(self.array)``), exactly like the reference, so analyzer code that reads
``image.array``, indexes ``image[t:b, l:r]`` or passes the object to numpy keeps working.
"""
from __future__ import annotations

import numpy as np
import torch

from . import array_utils as au
from . import ops

MM_PER_INCH = 25.4


def rescale_dicom_values(frames: torch.Tensor, rescale_slope=None, rescale_intercept=None,
                         pixel_intensity_relationship_sign=None, raw_pixels: bool = False,
                         invert_pixels: bool | None = None) -> torch.Tensor:
    """``_rescale_dicom_values`` (pylinac/core/image.py:363-389) for a device batch [N, H, W] of stored pixel values,
    with the three DICOM tags passed as numbers (None = tag absent).

    ``pixels.apply_rescale`` is pydicom's (``pydicom>=2.0,<3`` in the reference's pyproject; absent from this
    container): when both tags exist, ``arr.astype(float64) * RescaleSlope`` then ``+= RescaleIntercept``; otherwise
    the array is returned as stored.  pydicom cannot be run here; the pin is the set of identities the reference's own
    tests state for this function (tests_basic/core/test_image.py:131-200: raw / no-tag pass-through, ``RescaleSlope *
    pixel_array + RescaleIntercept``, the three inversion cases), checked in tests/next_row_checks.py; the inversion that
    follows is the reference's own expression ``max - a + min`` per frame."""
    x = ops._frames(frames)
    if raw_pixels:
        return x
    if rescale_slope is not None and rescale_intercept is not None:
        f = ops.normalize(x, 1.0) if x.dtype != torch.float64 else x          # exact conversion to float64
        f = ops.scale(f, float(rescale_slope))
        x = ops.ground(f, value=float(rescale_intercept),
                       mn=torch.zeros(f.shape[0], dtype=torch.float64, device=f.device))     # (a - 0) + intercept
    if invert_pixels or (invert_pixels is None and pixel_intensity_relationship_sign == -1):
        x = ops.invert(x)                                                       # -a + max + min == max - a + min
    return x


def _uniquify(seq, value: str) -> str:
    """``pylinac.core.utilities.uniquify`` (utilities.py:368-377): ``value``, else ``value-1``, ``value-2`` ..."""
    if value not in seq:
        return value
    n = 1
    while f"{value}-{n}" in seq:
        n += 1
    return f"{value}-{n}"



def rotate_array(array, angle: float, resize: bool = False, center=None, order=None, mode: str = "edge", cval=0,
                 clip: bool = True, preserve_range: bool = False):
    """``skimage.transform.rotate`` for 2-D arrays (numpy in / numpy out) or device frames ``[N,H,W]`` (tensor out), see
    ``ArrayImage.rotate``.  The warp itself is ``pl_warp_affine``."""
    from . import ops

    if resize or mode != "edge" or order not in (None, 0, 1):
        raise NotImplementedError("rotate: only order 0 / 1, mode='edge', resize=False (what BaseImage.rotate's callers use)")
    s = au._Staged(array)
    if s.ndim == 1:
        raise ValueError("rotate needs a 2-D image")
    t = s.t
    kind = np.dtype(np.asarray(array).dtype if not s.is_tensor else au._NP_OF_TORCH[t.dtype])
    if kind == np.uint64:
        raise TypeError("rotate: uint64 images are not supported")
    if order is None:
        order = 0 if kind.kind == "b" else 1           # skimage's _validate_interpolation_order
    if kind.kind == "b":
        x = t.to(torch.float64)
    elif kind.kind in "ui" and not preserve_range:
        info = np.iinfo(kind)
        x = t.to(torch.float64)
        x = x * (1.0 / info.max) if kind.kind == "u" else (x + 0.5) * (2 / (info.max - info.min))
    elif kind.kind in "ui":
        x = t.to(torch.float64)
    elif kind == np.float32 or kind == np.float64:
        x = t
    else:
        raise TypeError(f"rotate: unsupported dtype {kind}")
    x = x.contiguous()
    frames = x if x.dim() == 3 else x.unsqueeze(0)
    rows, cols = frames.shape[-2:]
    m = ops.rotation_matrix(rows, cols, angle, center)
    inf = float("inf")
    out = ops.warp_affine(frames, m, order=order) if clip else ops.warp_affine(frames, m, -inf, inf, order=order)
    return s.out(out.reshape(x.shape))


class _MutatorMixin:
    """The reference's in-place API; subclasses provide ``array`` (numpy or device tensor)."""

    def filter(self, size=0.05, kind: str = "median") -> None:
        """image.py:695-712."""
        self.array = au.filter(self.array, size=size, kind=kind)

    def invert(self) -> None:
        """image.py:755-757."""
        self.array = au.invert(self.array)

    def normalize(self, norm_val=None) -> None:
        """image.py:855-866 ("max" is the backwards-compatible alias of None)."""
        if isinstance(norm_val, str) and norm_val == "max":
            norm_val = None
        self.array = au.normalize(self.array, value=norm_val)


class BaseImage(_MutatorMixin):
    """The array-level members of ``pylinac.core.image.BaseImage`` (image.py:433-1102) that analyzers call on an image;
    subclasses provide ``array`` / ``dpi`` / ``dpmm`` / ``sid``.  File and plotting members (``date_created``,
    ``truncated_path``, ``as_dicom``, ``plot*``, ``from_multiples``) belong to the loaders and the UI: out of scope."""

    metrics: list
    metric_values: dict

    @property
    def physical_shape(self):
        """image.py:535-538: (rows, columns) in mm"""
        return self.shape[0] / self.dpmm, self.shape[1] / self.dpmm

    def dist2edge_min(self, point) -> float:
        """image.py:817-837: distance from ``point`` (a point or an (x, y) tuple) to the closest image edge"""
        x, y = (point[0], point[1]) if isinstance(point, tuple) else (point.x, point.y)
        rows, cols = self.shape[0], self.shape[1]
        return min(np.array([rows - y, cols - x, y, x], dtype=float))

    @property
    def flat(self):
        return self.array.flat

    @property
    def shape(self):
        return self.array.shape

    @property
    def size(self):
        return self.array.size

    @property
    def ndim(self):
        return self.array.ndim

    @property
    def dtype(self):
        return self.array.dtype

    def sum(self):
        return self.array.sum()

    def ravel(self):
        return self.array.ravel()

    def __len__(self):
        return len(self.array)

    def __getitem__(self, item):
        return self.array[item]

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.array, dtype=dtype)

    # -- mutators with reference-specific return values
    def threshold(self, threshold: float, kind: str = "high") -> None:
        """image.py:785-800: ``np.where(a >= t, a, 0)`` / ``np.where(a <= t, a, 0)``."""
        s = au._Staged(self.array)
        self.array = s.out(ops.threshold(s.t, threshold, kind))

    def as_binary(self, threshold) -> "ArrayImage":
        """image.py:802-815 -> new ArrayImage of int64 0/1."""
        s = au._Staged(self.array)
        return ArrayImage(s.out(ops.as_binary(s.t, threshold)).astype(np.int64))

    def ground(self) -> float:
        """image.py:839-853: returns the amount subtracted."""
        min_val = self.array.min()
        self.array = au.ground(self.array)
        return min_val

    def crop(self, pixels: int = 15, edges=("top", "bottom", "left", "right")) -> None:
        """image.py:714-745 (pure slicing; stays on the host)."""
        if pixels < 0:
            raise ValueError("Pixels to remove must be a positive number")
        if pixels == 0:
            return
        if "top" in edges:
            self.array = self.array[pixels:, :]
        if "bottom" in edges:
            self.array = self.array[:-pixels, :]
        if "left" in edges:
            self.array = self.array[:, pixels:]
        if "right" in edges:
            self.array = self.array[:, :-pixels]
        if self.array.size == 0:
            raise ValueError("Too many pixels removed; array is empty. Pass a smaller crop value.")

    def flipud(self) -> None:
        self.array = np.flipud(self.array)

    def fliplr(self) -> None:
        self.array = np.fliplr(self.array)

    def rot90(self, n: int = 1) -> None:
        self.array = np.rot90(self.array, n)

    def roll(self, direction: str = "x", amount: int = 1) -> None:
        self.array = np.roll(self.array, amount, axis=1 if direction == "x" else 0)

    def rotate(self, angle: float, mode: str = "edge", *args, **kwargs) -> None:
        """image.py:780-783: counter-clockwise rotation by ``angle`` degrees, ``skimage.transform.rotate(array, angle,
        mode=mode, ...)``: integer arrays are first rescaled the way ``img_as_float`` does (unsigned: ``a * (1 / imax)``;
        signed: ``(a + 0.5) * (2 / (imax - imin))``; so a rotated uint16 image is float64 in 0..1, as in the reference),
        then warped bilinearly (bool arrays: nearest neighbour, skimage's default order for them) about ``(cols, rows) / 2 - 0.5`` with edge replication and clipped to the input's range.
        Only what the reference's own callers use is offered (order 1, mode "edge", no resize); anything else raises."""
        self.array = rotate_array(self.array, angle, mode=mode, *args, **kwargs)

    @property
    def center(self):
        """image.py:526-533: the centre of the array as a point (x, y); even lengths give the mid-point between the two
        central indices."""
        from .geometry import Point

        return Point(x=(self.shape[1] / 2) - 0.5, y=(self.shape[0] / 2) - 0.5)

    def as_type(self, dtype) -> np.ndarray:
        return self.array.astype(dtype)

    def bit_invert(self) -> None:
        """image.py:759-761."""
        self.array = au.bit_invert(self.array)

    def check_inversion(self, box_size: int = 20, position=(0.0, 0.0)) -> None:
        """image.py:868-897: invert when the mean of the four corner boxes exceeds the mean of the image (box statistics
        and the frame sum on the device, ``decisions.corners_inverted``)."""
        from . import decisions

        s = au._Staged(self.array)
        if bool(decisions.corners_inverted(s.t, box_size=box_size, position=position)[0]):
            self.invert()

    def compute(self, metrics):
        """image.py:1022-1054, the reference's one extension point: ``metric.inject_image(self)``,
        ``metric.context_calculate()`` (which hashes ``self.array.tobytes()`` before and after ``calculate()`` and raises
        RuntimeError when a metric modified the image, metrics/image.py:61-71), results stored under a unique name in
        ``metric_values`` and the metric objects in ``metrics``.  Accepts any object with the ``MetricBase`` protocol
        (``inject_image``, ``context_calculate``, ``name``) -- the reference's own metric classes included."""
        metric_data = {}
        if not isinstance(metrics, (list, tuple)):
            metrics = [metrics]
        key = None
        for metric in metrics:
            metric.inject_image(self)
            value = metric.context_calculate()
            self.metrics.append(metric)
            key = _uniquify(list(metric_data.keys()) + list(self.metric_values.keys()), metric.name)
            metric_data[key] = value
        self.metric_values |= metric_data
        if len(metrics) == 1:
            return metric_data[key]
        return metric_data

    def check_inversion_by_histogram(self, percentiles=(5, 50, 95)) -> bool:
        """image.py:899-926: invert when |p_mid - p_low| > |p_mid - p_high|.  For 16-bit frames the
        percentiles come from the exact device histogram."""
        a = self.array
        if a.dtype in (np.uint16, np.int16):
            s = au._Staged(a)
            p_low, p_mid, p_high = ops.percentile(s.t, list(percentiles))[0].tolist()
        elif a.dtype in (np.float64, np.float32):
            from .canny import _percentile_f64      # exact float64 order statistics (pl_order_stats_f64)

            # float32 frames: the order statistics are exact in float64; numpy interpolates between them in float32,
            # which can only matter to this comparison when the two distances tie to ~1e-7 relative
            t = au._Staged(a.astype(np.float64, copy=False)).t
            p_low, p_mid, p_high = (float(_percentile_f64(t, q)[0]) for q in percentiles)
        else:
            raise TypeError("check_inversion_by_histogram needs a 16-bit integer or float32 / float64 frame on this backend")
        if abs(p_mid - p_low) > abs(p_mid - p_high):
            self.invert()
            return True
        return False

    def gamma(self, comparison_image: "ArrayImage", doseTA: float = 1, distTA: float = 1, threshold: float = 0.1,
              ground: bool = True, normalize: bool = True) -> np.ndarray:
        """image.py:929-1016: the Bakai gamma map of this image against ``comparison_image`` (same DPI and size).
        Inversion check, ground, normalise, NaN below ``threshold * max``, float32 Sobel gradient, gamma map --
        every array operation on the device, in numpy's precision mix (``pl_bakai_mask`` / ``pl_bakai_gamma``)."""
        from . import _lib
        from ._lib import check

        if not 0 <= threshold <= 1:
            raise ValueError("threshold must be within (0, 1)")
        if abs(self.dpi - comparison_image.dpi) > 0.1:
            raise AttributeError(f"The image DPIs to not match: {self.dpi:.2f} vs. {comparison_image.dpi:.2f}")
        same_x = abs(self.shape[1] - comparison_image.shape[1]) <= 1.1
        same_y = abs(self.shape[0] - comparison_image.shape[0]) <= 1.1
        if not (same_x and same_y):
            raise AttributeError(f"The images are not the same size: {self.shape} vs. {comparison_image.shape}")
        imgs = []
        for src in (self, comparison_image):
            im = ArrayImage(np.array(src.array, copy=True))
            im.check_inversion_by_histogram()
            if ground:
                im.ground()
            if normalize:
                im.normalize()
            imgs.append(np.ascontiguousarray(im.array, dtype=np.float64))
        if imgs[0].shape != imgs[1].shape:
            raise ValueError("operands could not be broadcast together")      # what numpy raises in the reference
        dev = au._device()
        ref, comp = (torch.from_numpy(a).to(dev) for a in imgs)
        h, w = ref.shape
        _, mx = ops.minmax(ref[None])
        cut = (mx * threshold).contiguous()
        masked = torch.empty_like(ref)
        ref32 = torch.empty((1, h, w), dtype=torch.float32, device=dev)
        lib, st = _lib.load(), torch.cuda.current_stream(dev).cuda_stream
        check(lib.pl_bakai_mask(ref.data_ptr(), cut.data_ptr(), 1, h * w, masked.data_ptr(), ref32.data_ptr(), st),
              "pl_bakai_mask")
        gx, gy = ops.sobel(ref32, 1), ops.sobel(ref32, 0)
        dist_px = self.dpmm * distTA
        out = torch.empty_like(ref)
        check(lib.pl_bakai_gamma(masked.data_ptr(), comp.data_ptr(), gx.data_ptr(), gy.data_ptr(),
                                 float(np.float32((doseTA / 100.0) ** 2)), float(np.float32(dist_px**2)), h * w,
                                 out.data_ptr(), st), "pl_bakai_gamma")
        return out.cpu().numpy()

    def profile(self, axis: int = 0, kind: str = "mean") -> np.ndarray:
        """EXTENSION (the reference has no ``Image.profile()``): the axis
        reductions the analyzers write by hand -- ``np.mean(image, axis)`` picketfence.py:747-750,
        ``np.max`` starshot.py:216-217, ``np.sum`` picketfence.py:1513-1514."""
        s = au._Staged(self.array)
        out = ops.reduce_axis(s.t, axis, kind)[0].cpu().numpy()
        if kind in ("max", "min"):
            return out.astype(self.array.dtype)
        if kind == "sum" and self.array.dtype.kind in "iu":
            return out.astype(np.uint64 if self.array.dtype.kind == "u" else np.int64)
        if self.array.dtype == np.float32:
            return out.astype(np.float32)
        return out


class ArrayImage(BaseImage):
    """An image constructed solely from a numpy array (image.py:1815-1869), GPU-computed."""

    def __init__(self, array, *, dpi: float = None, sid: float = None, dtype=None):
        if dtype is not None:
            self.array = np.array(array, dtype=dtype)
        else:
            self.array = np.asarray(array)
        self._dpi = dpi
        self.sid = sid
        self.metrics = []
        self.metric_values = {}

    # image.py:1851-1866
    @property
    def dpi(self):
        dpi = None
        if self._dpi is not None:
            dpi = self._dpi
            if self.sid is not None:
                dpi *= self.sid / 1000
        return dpi

    @property
    def dpmm(self):
        try:
            return self.dpi / MM_PER_INCH
        except Exception:
            return None

    def __sub__(self, other):
        return ArrayImage(self.array - other.array)


class ImageBatch:
    """``[N,H,W]`` frames resident in HBM; the reference's mutator names applied to every frame.
    Per-frame scalars come back as device tensors; nothing leaves the GPU unless asked."""

    def __init__(self, frames: torch.Tensor):
        if not isinstance(frames, torch.Tensor) or not frames.is_cuda or frames.dim() != 3:
            raise ValueError("ImageBatch needs a [N,H,W] tensor on the GPU")
        if frames.numel() == 0:
            raise ValueError("Array must not be empty")
        self.array = frames.contiguous()

    @property
    def shape(self):
        return tuple(self.array.shape)

    def __len__(self):
        return self.array.shape[0]

    def filter(self, size=0.05, kind: str = "median") -> None:
        size = au.resolve_filter_size(self.array.shape[1], size)
        if kind == "median":
            self.array = ops.median_filter(self.array, int(size))
        elif kind == "gaussian":
            self.array = ops.gaussian_filter(self.array, size)
        else:
            raise ValueError(f"Filter type {kind} unsupported. Use one of 'median', 'gaussian'")

    def threshold(self, threshold, kind: str = "high") -> None:
        self.array = ops.threshold(self.array, threshold, kind)

    def as_binary(self, threshold) -> torch.Tensor:
        return ops.as_binary(self.array, threshold)

    def ground(self) -> torch.Tensor:
        mn, _ = ops.minmax(self.array)
        self.array = ops.ground(self.array, mn=mn)
        return mn

    def normalize(self, norm_val=None) -> None:
        self.array = ops.normalize(self.array, None if norm_val in (None, "max") else norm_val)

    def invert(self) -> None:
        self.array = ops.invert(self.array)

    def otsu(self) -> torch.Tensor:
        return ops.threshold_otsu(self.array)

    def percentile(self, q) -> torch.Tensor:
        return ops.percentile(self.array, q)

    def profile(self, axis: int = 0, kind: str = "mean") -> torch.Tensor:
        return ops.reduce_axis(self.array, axis, kind)
