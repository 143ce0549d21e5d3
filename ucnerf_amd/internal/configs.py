"""Plain-Python stand-in for the gin `Config` dataclass (ref internal/configs.py:22-189).

Only the attributes the ray-march reads are listed (models.py:73,84,94-95,290; render_image
:939-944,1003); names and defaults are the reference's, near/far/chunk from configs/waymo.gin.
Any object with these attributes (e.g. the reference's own Config) works in its place."""
import dataclasses


@dataclasses.dataclass
class Config:
    model_sky: bool = False                 # configs.py:37
    brightness_correction: bool = False     # configs.py:91
    training_views: int = 210               # configs.py:61
    zero_glo: bool = False                  # configs.py:153
    vis_num_rays: int = 16                  # configs.py:58
    render_chunk_size: int = 15000          # waymo.gin:8 (kept for API parity; the HIP path sizes its own passes)
    near: float = 0.                        # waymo.gin:2
    far: float = 8.                         # waymo.gin:3
    render_ray_tile: int = 8                # not in the reference: render_image marches T x T pixel blocks per wave (1 = row order)
