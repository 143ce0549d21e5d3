"""`internal.models` of the drop-in overlay: the names the reference's scripts and modules take from it
(`models.Model(config=config)` train.py:69 / render.py:104 / eval.py:82 / extract.py:322 / tsdf.py:255,
`models.render_image(...)` train.py:330 / render.py:146 / eval.py:140 / extract.py:350 / tsdf.py:295), bound to the
MI355X implementation.  The classes are registered with gin when gin is importable (ucnerf_amd/internal/models.py), so
`configs/waymo.gin`'s `Model.* / NerfMLP.* / PropMLP.*` bindings reach them exactly as they reach the reference's."""
from ucnerf_amd.internal.models import (MLP, Model, NerfMLP, PropMLP, bindings, render_image, set_kwargs,  # noqa: F401
                                        unwrap_model)
from ucnerf_amd.internal.sky import NeRF, render_rays  # noqa: F401
from ucnerf_amd.internal.extrinsic_optimizer import BrightnessCorrection  # noqa: F401
from ucnerf_amd.gridencoder import GridEncoder  # noqa: F401
