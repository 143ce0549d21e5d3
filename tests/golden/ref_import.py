"""Import the reference's Python (read-only, /root/reference/nerf) on CPU.

Runs ONLY in the authoring container: /root/reference does not exist on the GPU box and
nothing in the gpu tests, smoke() or bench.py uses this module.  It is the harness that
generates and re-checks tests/golden/*.npz (see make_golden.py).

What is stubbed: third-party packages the image lacks (gin, absl,
cv2, lpips, rawpy, skimage, nuscenes, pyquaternion, pycolmap, torch_scatter) -- none of them
does arithmetic on the path except torch_scatter.segment_coo (restated below) -- and the
CUDA-only `_gridencoder` extension, replaced by oracle/grid_cpu.py (C restatement of
gridencoder.cu).  The reference's own grid.py / models.py / render.py / stepfun.py / coord.py
/ math.py / extrinsic_optimizer.py run unmodified.
"""
import importlib.util
import os
import sys
import types

import torch

REF = '/root/reference/nerf'


def available():
    return os.path.isdir(REF)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _segment_coo(src, index, out=None, dim_size=None, reduce='mean'):
    """torch_scatter.segment_coo(reduce='mean') for a sorted 1-D index over dim 0
    (call sites models.py:301-305, :499-504)."""
    n = out.shape[0] if out is not None else (dim_size if dim_size is not None else int(index.max()) + 1)
    acc = torch.zeros((n,) + src.shape[1:], dtype=src.dtype)
    acc.index_add_(0, index, src)
    cnt = torch.zeros(n, dtype=src.dtype).index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
    cnt = cnt.clamp_min(1).reshape((n,) + (1,) * (src.dim() - 1))
    assert reduce == 'mean'
    return acc / cnt


_loaded = None


def load():
    """Returns a namespace with the reference modules: models, render, stepfun, coord, math,
    configs, train_utils, grid (reference grid.py)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError('/root/reference is not present')
    sys.dont_write_bytecode = True
    repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from oracle import grid_cpu

    def configurable(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    gin_cfg = _mod('gin.config', external_configurable=lambda f, module=None: f)
    _mod('gin', configurable=configurable, config=gin_cfg,
         add_config_file_search_path=lambda p: None, REQUIRED=object())
    flags = _mod('absl.flags', DEFINE_multi_string=lambda *a, **k: None, FLAGS=types.SimpleNamespace())
    _mod('absl', flags=flags, app=_mod('absl.app'))
    _mod('cv2')
    _mod('lpips', LPIPS=lambda net=None: None)
    _mod('rawpy')
    _mod('skimage', metrics=_mod('skimage.metrics', structural_similarity=None, peak_signal_noise_ratio=None))
    _mod('nuscenes', nuscenes=_mod('nuscenes.nuscenes', NuScenes=object))
    _mod('pyquaternion', Quaternion=object)
    _mod('pycolmap', SceneManager=object)
    _mod('torch_scatter', segment_coo=_segment_coo)
    # the native op: C restatement with the pybind module's three functions
    _mod('_gridencoder', grid_encode_forward=grid_cpu.grid_encode_forward,
         grid_encode_backward=grid_cpu.grid_encode_backward,
         grad_total_variation=grid_cpu.grad_total_variation)
    # the reference's own grid.py, loaded by path and exposed as `gridencoder`
    spec = importlib.util.spec_from_file_location('ref_gridencoder_grid', os.path.join(REF, 'gridencoder', 'grid.py'))
    grid = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(grid)
    _mod('gridencoder', GridEncoder=grid.GridEncoder, grid=grid)

    cwd = os.getcwd()
    os.chdir(REF)
    sys.path.insert(0, REF)
    try:
        from internal import math as rmath, stepfun, coord, render, configs, train_utils, models
        from internal import extrinsic_optimizer
    finally:
        os.chdir(cwd)
    # waymo.gin:10-20 bindings, applied as class attributes
    models.PropMLP.disable_density_normals = True
    models.PropMLP.disable_rgb = True
    models.NerfMLP.disable_density_normals = True
    _loaded = types.SimpleNamespace(models=models, render=render, stepfun=stepfun, coord=coord,
                                    math=rmath, configs=configs, train_utils=train_utils, grid=grid,
                                    extrinsic_optimizer=extrinsic_optimizer)
    return _loaded


class capture_rng:
    """Record every tensor the reference draws through torch.rand / rand_like / randn_like
    while active (the hot path uses exactly these: stepfun.py:216, render.py:123,124,140)."""

    def __enter__(self):
        self.draws = []
        self._orig = (torch.rand, torch.rand_like, torch.randn_like)

        def wrap(fn, tag):
            def inner(*a, **k):
                out = fn(*a, **k)
                self.draws.append((tag, out.clone()))
                return out
            return inner
        torch.rand = wrap(self._orig[0], 'rand')
        torch.rand_like = wrap(self._orig[1], 'rand_like')
        torch.randn_like = wrap(self._orig[2], 'randn_like')
        return self

    def __exit__(self, *exc):
        torch.rand, torch.rand_like, torch.randn_like = self._orig
        return False


def build_reference_model(ref, spec, sd):
    """Instantiate the reference `Model` for an oracle PathSpec and load the oracle state."""
    m = ref.models
    cfg = ref.configs.Config()
    cfg.model_sky = spec.model_sky
    cfg.brightness_correction = spec.brightness_correction
    cfg.training_views = spec.training_views
    cfg.vis_num_rays = spec.vis_num_rays
    saved = {}

    def setcls(cls, fs):
        for k_ref, k in [('grid_disired_resolution', 'grid_desired_resolution'),
                         ('grid_level_dim', 'grid_level_dim'),
                         ('grid_log2_hashmap_size', 'grid_log2_hashmap_size'),
                         ('bottleneck_width', 'bottleneck_width'),
                         ('net_width_viewdirs', 'net_width_viewdirs')]:
            saved[(cls, k_ref)] = getattr(cls, k_ref)
            setattr(cls, k_ref, getattr(fs, k))
    setcls(m.NerfMLP, spec.nerf)
    setcls(m.PropMLP, spec.props[0])
    try:
        model = m.Model(config=cfg, num_levels=spec.num_levels, num_prop_samples=spec.num_prop_samples,
                        num_nerf_samples=spec.num_nerf_samples, opaque_background=spec.opaque_background,
                        prop_desired_grid_size=list(spec.prop_desired_grid_size),
                        dilation_bias=spec.dilation_bias, dilation_multiplier=spec.dilation_multiplier)
    finally:
        for (cls, k), v in saved.items():
            setattr(cls, k, v)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith('.idx') for k in missing), missing
    return model, cfg
