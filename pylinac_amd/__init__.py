"""pylinac_amd -- MI355X-native compute core for pylinac's per-image hot path.

Host code is Python on PyTorch-ROCm tensors; all arithmetic runs in hand-written HIP kernels
behind the C ABI of ``libpylinac_hip.so`` (include/pylinac_hip.h).  See DESIGN.md.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401
