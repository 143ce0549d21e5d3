"""ucnerf_amd -- MI355X-native (gfx950) ray-march for UC-NeRF-style models.

One hot path, built from scratch behind the reference's own interfaces:
  ucnerf_amd.gridencoder   GridEncoder / `_gridencoder` backend  (ref: nerf/gridencoder)
  ucnerf_amd.internal      Model, NerfMLP, PropMLP, render_image (ref: nerf/internal/models.py)
All arithmetic runs in hand-written HIP kernels (ucnerf_amd/csrc, C ABI in include/ucnerf_march.h);
PyTorch only owns device memory, streams and torch.distributed.  There is no CPU fallback.
"""
from . import _lib  # noqa: F401

__all__ = ["gridencoder", "internal"]
