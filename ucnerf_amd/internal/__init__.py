"""Host-side mirror of the reference's `internal` package, restricted to the ray-march path:
models (Model, NerfMLP, PropMLP, render_image), extrinsic_optimizer (colour-correction head),
sky (sky NeRF layer), dist (tile sharding + RCCL all-gather), configs (plain Config object)."""
from . import configs, dist, extrinsic_optimizer, models, sky  # noqa: F401
