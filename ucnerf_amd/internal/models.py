"""Model / NerfMLP / PropMLP / render_image -- host side of the fused HIP ray-march.

Interface mirror of /root/reference/nerf/internal/models.py (Model :31-365, MLP :367-695,
render_image :907-1007): same class names, class-attribute knobs, constructor and forward
signatures, sub-module names (=> identical ``state_dict`` keys) and the
same keys in the returned ``renderings`` / ``ray_history``.  What differs is everything below the
signature: a sampling level is five kernel launches on the caller's HIP stream
(resample -> cone basis -> fused cast/contract/hash-grid/erf features -> MFMA MLP -> wave-per-ray
composite) instead of ~300 eager ops, and nothing of size [N*S*6, .] is ever materialised.

Supported configuration = the reference's shipped one (configs/waymo.gin): disable_density_normals,
no GLO, no reflections / diffuse / IDE, raydist_fn=None.  Anything else raises at construction.
In eval mode or with gradients disabled forward is the fused inference march; a model in training mode
with gradients enabled routes to internal/train_graph.py (same kernels for resampling and featurisation, HIP
backward for the tables and the dense layers' dgrad, autograd glue; `Model.march_route`).  Without the HIP library or a GPU every entry point raises.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from ..gridencoder import GridEncoder
from .extrinsic_optimizer import BrightnessCorrection
from .sky import NeRF, render_rays  # noqa: F401  (names kept importable like upstream)


try:                                    # the reference registers its classes with gin (models.py:30,688,693); so do these,
    import gin                          # when gin is importable, so that configs/waymo.gin:10-20 binds to them unchanged
except ImportError:                     # (without gin: `bindings()` below, or constructor keyword arguments)
    gin = None


def _configurable(cls):
    return gin.configurable(cls) if gin is not None else cls


def set_kwargs(self, kwargs):
    """Instance configuration = the class-level knobs AS THEY ARE NOW (gin bindings set class attributes,
    configs/waymo.gin:10-20; `bindings()` below restores them when its block ends) overridden by the constructor's
    keyword arguments.  Frozen on the instance: a field built as 64-wide must still describe itself as 64-wide after
    the class default is back to 256 (the C descriptor is filled from these attributes on every call)."""
    for klass in reversed(type(self).__mro__):
        if klass in (nn.Module, object):
            continue
        for k, v in vars(klass).items():
            if not k.startswith('_') and not callable(v) and not isinstance(v, (property, classmethod, staticmethod)):
                object.__setattr__(self, k, v) if not isinstance(v, (torch.Tensor, nn.Module)) else None
    for k, v in kwargs.items():
        setattr(self, k, v)


def _f32(t, n, c):
    t = t.reshape(n, c)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_U_CACHE = {}


def _u_table(num_samples, train, device):
    """stepfun.py:203-216: the u grid of the inverse-CDF lookup (constant per S)."""
    key = (num_samples, bool(train), str(device))
    hit = _U_CACHE.get(key)
    if hit is None:
        eps = float(torch.finfo(torch.float32).eps)
        if train:
            u_max = eps + (1 - eps) / num_samples
            max_jitter = (1 - u_max) / (num_samples - 1) - eps
            u = torch.linspace(0, 1 - u_max, num_samples)
        else:
            pad = 1 / (2 * num_samples)
            max_jitter = 0.0
            u = torch.linspace(pad, 1. - pad - eps, num_samples)
        hit = (u.to(device), max_jitter)
        _U_CACHE[key] = hit
    return hit


class MLP(nn.Module):
    """One hash-grid field (ref models.py:367-685).  Holds the parameters in the reference's
    layout; evaluation is ucnerf_amd/csrc/{march_features,field_mlp}.hip."""
    bottleneck_width: int = 256
    net_depth_viewdirs: int = 2
    net_width_viewdirs: int = 256
    skip_layer_dir: int = 0
    num_rgb_channels: int = 3
    deg_view: int = 4
    use_reflections: bool = False
    use_directional_enc: bool = False
    enable_pred_roughness: bool = False
    use_diffuse_color: bool = False
    use_specular_tint: bool = False
    use_n_dot_v: bool = False
    bottleneck_noise: float = 0.0
    density_bias: float = -1.
    density_noise: float = 0.
    rgb_premultiplier: float = 1.
    rgb_bias: float = 0.
    rgb_padding: float = 0.001
    enable_pred_normals: bool = False
    disable_density_normals: bool = True     # waymo.gin:15,17 (upstream class default is False)
    disable_rgb: bool = False
    warp_fn = 'contract'
    num_glo_features: int = 0
    num_glo_embeddings: int = 1000
    scale_featurization: bool = False
    grid_num_levels: int = 10
    grid_level_interval: int = 2
    grid_level_dim: int = 4
    grid_base_resolution: int = 16
    grid_disired_resolution: int = 8192
    grid_log2_hashmap_size: int = 21
    net_width_glo: int = 128
    net_depth_glo: int = 2
    # ---- knob of this implementation (not in the reference): arithmetic of the dense layers
    #   0 = fp32-input MFMA (exact fp32 products; 157 TF ceiling)
    #   1 = split-f16 MFMA, fp32 accumulate (hi/lo f16 operands, ~3e-7 relative per product; 5.3x rate)
    mlp_mode: int = 1
    bwd_fixed_point: bool = True      # autocast training step: table-gradient row blocks in guaranteed-range fixed point (UCN_BWD_FIXED_POINT)

    def __init__(self, **kwargs):
        super().__init__()
        set_kwargs(self, kwargs)
        unsupported = dict(use_reflections=False, use_directional_enc=False, enable_pred_roughness=False,
                           use_diffuse_color=False, use_specular_tint=False, use_n_dot_v=False,
                           enable_pred_normals=False, disable_density_normals=True, scale_featurization=False,
                           num_glo_features=0, net_depth_viewdirs=2, skip_layer_dir=0, num_rgb_channels=3,
                           warp_fn='contract', bottleneck_noise=0.0, density_noise=0.0)
        for k, want in unsupported.items():
            if getattr(self, k) != want:
                raise NotImplementedError(
                    f"{type(self).__name__}.{k}={getattr(self, k)!r}: the HIP ray-march implements the shipped "
                    f"waymo.gin configuration only ({k}={want!r})")
        self.grid_num_levels = int(np.log(self.grid_disired_resolution / self.grid_base_resolution)
                                   / np.log(self.grid_level_interval)) + 1
        self.encoder = GridEncoder(input_dim=3, num_levels=self.grid_num_levels, level_dim=self.grid_level_dim,
                                   base_resolution=self.grid_base_resolution,
                                   desired_resolution=self.grid_disired_resolution,
                                   log2_hashmap_size=self.grid_log2_hashmap_size, gridtype='hash', align_corners=False)
        last_dim = self.encoder.output_dim
        self.density_layer = nn.Sequential(nn.Linear(last_dim, 64), nn.ReLU(),
                                           nn.Linear(64, 1 if self.disable_rgb else self.bottleneck_width))
        self.dim_dir_enc = 3 + 6 * self.deg_view
        if not self.disable_rgb:
            last_dim_rgb = self.bottleneck_width + self.dim_dir_enc
            input_dim_rgb = last_dim_rgb
            for i in range(self.net_depth_viewdirs):
                lin = nn.Linear(last_dim_rgb, self.net_width_viewdirs)
                torch.nn.init.kaiming_uniform_(lin.weight)
                self.register_module(f"lin_second_stage_{i}", lin)
                last_dim_rgb = self.net_width_viewdirs
                if i == self.skip_layer_dir:
                    last_dim_rgb += input_dim_rgb
            self.rgb_layer = nn.Linear(last_dim_rgb, self.num_rgb_channels)
        self._fields = {}          # mlp_mode -> (key, ucn_field_t, packed weight stream): one packed buffer per mode

    _UNPICKLED = ('_fields', '_grid_desc', '_grid_desc_key')

    def __getstate__(self):
        """copy.deepcopy / torch.save(model): the cached C descriptors hold raw device pointers (ctypes objects with
        pointers cannot be pickled, and would dangle in a copy anyway) -- they are rebuilt on first use."""
        state = self.__dict__.copy()
        for k in self._UNPICKLED:
            state.pop(k, None)
        state['_fields'] = {}
        return state

    # ---- C-ABI descriptor -------------------------------------------------------------------
    def _weights(self):
        ws = [self.encoder.embeddings, self.density_layer[0].weight, self.density_layer[0].bias,
              self.density_layer[2].weight, self.density_layer[2].bias]
        if not self.disable_rgb:
            ws += [self.lin_second_stage_0.weight, self.lin_second_stage_0.bias, self.lin_second_stage_1.weight,
                   self.lin_second_stage_1.bias, self.rgb_layer.weight, self.rgb_layer.bias]
        return ws

    def field(self, mode=None):
        """ucn_field_t for the current parameters and the given arithmetic mode (default: self.mlp_mode); the
        MFMA-ordered weight copy of that mode is refreshed when any parameter was updated in place or moved
        (render: once; train: once per optimiser step)."""
        mode = int(self.mlp_mode if mode is None else mode)
        ws = self._weights()
        for w in ws:
            _lib.require_device(w, f"{type(self).__name__} parameter")
            if w.dtype != torch.float32 or not w.is_contiguous():
                raise RuntimeError("field parameters must be contiguous float32")
        key = tuple((w.data_ptr(), w._version) for w in ws)
        hit = self._fields.get(mode)
        if hit is not None and hit[0] == key:
            return hit[1]
        lib = _lib.load()
        enc = self.encoder
        d = _lib.UcnField()
        d.embeddings = ws[0].data_ptr()
        d.offsets_host = enc._offsets_np.ctypes.data
        d.grid_sizes_host = enc._sizes_np.ctypes.data
        d.num_levels, d.level_dim, d.base_resolution = enc.num_levels, enc.level_dim, enc.base_resolution
        d.log2_per_level_scale = float(np.log2(enc.per_level_scale))
        d.w_d0, d.b_d0, d.w_d1, d.b_d1 = (w.data_ptr() for w in ws[1:5])
        d.n_bottleneck = 1 if self.disable_rgb else self.bottleneck_width
        if not self.disable_rgb:
            d.w_c0, d.b_c0, d.w_c1, d.b_c1, d.w_rgb, d.b_rgb = (w.data_ptr() for w in ws[5:11])
        d.n_width = self.net_width_viewdirs
        d.n_dir = self.dim_dir_enc
        d.density_bias, d.rgb_premultiplier = float(self.density_bias), float(self.rgb_premultiplier)
        d.rgb_bias, d.rgb_padding = float(self.rgb_bias), float(self.rgb_padding)
        d.mlp_mode = mode
        n = lib.ucn_field_packed_floats(ctypes.byref(d))
        if n == 0:
            raise RuntimeError(lib.ucn_last_error().decode())
        packed = hit[2] if hit is not None else None
        if packed is None or packed.numel() != n or packed.device != ws[0].device:
            packed = torch.empty(n, dtype=torch.float32, device=ws[0].device)
        d.packed = packed.data_ptr()
        _lib.check(lib.ucn_field_pack(ctypes.byref(d), _lib.stream()))
        self._fields[mode] = (key, d, packed)
        return d

    def grid_field(self):
        """ucn_field_t with only the hash-grid part filled in: all the featurisation kernels read.  The training graph
        uses it so that an optimiser step does not trigger a re-pack of the RENDERING engine's weight stream."""
        emb = self.encoder.embeddings
        _lib.require_device(emb, f"{type(self).__name__}.encoder.embeddings")
        key = (emb.data_ptr(),)
        if getattr(self, "_grid_desc_key", None) != key:
            enc = self.encoder
            d = _lib.UcnField()
            d.embeddings = emb.data_ptr()
            d.offsets_host = enc._offsets_np.ctypes.data
            d.grid_sizes_host = enc._sizes_np.ctypes.data
            d.num_levels, d.level_dim, d.base_resolution = enc.num_levels, enc.level_dim, enc.base_resolution
            d.log2_per_level_scale = float(np.log2(enc.per_level_scale))
            self._grid_desc, self._grid_desc_key = d, key
        return self._grid_desc

    # ---- reference API on explicit Gaussians (extract.py:56-57,96) ---------------------------
    @torch.no_grad()
    def predict_density(self, means, stds, rand=False, no_warp=False):
        """ref models.py:485-512.  means [..., G, 3], stds [..., G] -> (raw_density [...],
        x [..., n_out], mean contracted coordinate [..., 3])."""
        raw, x, coord, _, _ = self._evaluate(means, stds, None, no_warp, want_x=True)
        return raw, x, coord

    @torch.no_grad()
    def forward(self, rand, means, stds, viewdirs=None, imageplane=None, glo_vec=None, exposure=None,
                no_warp=False):
        """ref models.py:514-685 (keys of the returned dict identical)."""
        _, _, coord, density, rgb = self._evaluate(means, stds, viewdirs, no_warp, want_x=False)
        if self.disable_rgb or viewdirs is None:
            rgb = torch.zeros(density.shape + (3,), device=density.device)
        return dict(coord=coord, density=density, rgb=rgb, raw_grad_density=None, grad_pred=None, normals=None,
                    normals_pred=None, roughness=None)

    def _evaluate(self, means, stds, viewdirs, no_warp, want_x):
        # the split-f16 kernel composes the bottleneck away; the API that returns it uses the fp32 kernel's packed copy
        mode = 0 if (want_x and not self.disable_rgb) else int(self.mlp_mode)
        lib = _lib.load()
        _lib.require_device(means, "means")
        prefix = means.shape[:-2]
        G = means.shape[-2]
        if not 1 <= G <= 6:
            raise RuntimeError(f"predict_density: 1..6 Gaussians per feature supported, got {G}")
        B = int(np.prod(prefix)) if len(prefix) else 1
        d = self.field(mode)
        dev = means.device
        m = _f32(means, B * G, 3)
        s = _f32(stds, B * G, 1)
        L, C = self.encoder.num_levels, self.encoder.level_dim
        feat = torch.empty(L * B * C, device=dev)
        coord = torch.empty(B, 3, device=dev)
        st = _lib.stream()
        _lib.check(lib.ucn_points_features(ctypes.byref(d), m.data_ptr(), s.data_ptr(), B, G, 0 if no_warp else 1, 1,
                                           feat.data_ptr(), coord.data_ptr(), st))
        density = torch.empty(B, device=dev)
        n_out = 1 if self.disable_rgb else self.bottleneck_width
        x = torch.empty(B, n_out, device=dev) if (want_x and not self.disable_rgb) else None
        rgb = None
        dirb = None
        spr = 1
        if viewdirs is not None and not self.disable_rgb:
            # viewdirs [..., 3] broadcast over the sample axis: samples per ray = B / #rays
            vd = _f32(viewdirs, -1, 3)
            n_rays = vd.shape[0]
            if B % n_rays:
                raise RuntimeError("viewdirs do not divide the sample count")
            spr = B // n_rays
            dirb = torch.empty(lib.ucn_field_dir_floats(ctypes.byref(d), n_rays), device=dev)
            _lib.check(lib.ucn_field_dir_bias(ctypes.byref(d), vd.data_ptr(), n_rays, dirb.data_ptr(), st))
            rgb = torch.empty(B, 3, device=dev)
        _lib.check(lib.ucn_field_mlp(ctypes.byref(d), feat.data_ptr(), B, spr, 0, _lib.ptr(dirb), density.data_ptr(),
                                     _lib.ptr(rgb), _lib.ptr(x), st))
        # raw (pre-activation) density: softplus is inverted only for API parity of predict_density
        raw = None
        if want_x:
            if self.disable_rgb:
                raise NotImplementedError("predict_density on a disable_rgb field: use forward()['density']")
            raw = x[:, 0].reshape(prefix)
            x = x.reshape(prefix + (n_out,))
        return (raw, x, coord.reshape(prefix + (3,)), density.reshape(prefix),
                None if rgb is None else rgb.reshape(prefix + (3,)))


@_configurable
class NerfMLP(MLP):
    pass


@_configurable
class PropMLP(MLP):
    disable_rgb: bool = True      # waymo.gin:16


@_configurable
class Model(nn.Module):
    """ref models.py:31-365."""

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in ('_side_stream', '_sky_stream', '_prof', '_alive_idx', '_alive_cnt', '_alive_stats', '_mixed_cache'):
            state.pop(k, None)
        return state

    num_prop_samples: int = 64
    num_nerf_samples: int = 32
    num_levels: int = 3
    bg_intensity_range = (1., 1.)
    anneal_slope: float = 10
    stop_level_grad: bool = True
    use_viewdirs: bool = True
    raydist_fn = None
    single_jitter: bool = True
    dilation_multiplier: float = 0.5
    dilation_bias: float = 0.0025
    num_glo_features: int = 0
    num_glo_embeddings: int = 1000
    learned_exposure_scaling: bool = False
    near_anneal_rate = None
    near_anneal_init: float = 0.95
    single_mlp: bool = False
    distinct_prop: bool = True
    resample_padding: float = 0.0
    opaque_background: bool = False
    power_lambda: float = -1.5
    std_scale: float = 0.5
    prop_desired_grid_size = [512, 2048]
    # ---- knobs of this implementation (not in the reference) ----
    max_chunk_rays: int = 10240          # rays per internal pass.  At S=128, F=32 the feature workspace is 168 MB: with
    #                                      the tables (64 + 24 MB) it stays inside the 256 MB Infinity Cache between the
    #                                      featurisation that writes it and the MLP that reads it (65 536: 5 % slower)
    levels_per_block: int = 0            # hash-grid levels per thread: 0 = auto (coarse levels together, fine alone)
    overlap_streams: bool = False        # featurisation of pass i+1 beside the MLP of pass i on a second HIP stream:
    #                                      measured +1 % only (both kernels want every CU), so off by default
    rays_fastest: bool = True            # wave lanes = neighbouring rays at one sample index (L1/L2 locality)
    compact_min_weight: float = 0.0      # > 0: early-termination sample compaction at the last level of inference marches
    #                                      that return no per-sample history (render_image): density head first, then the
    #                                      colour layers only for samples whose compositing weight alpha * T reaches this
    #                                      value (pixel error <= num_nerf_samples * compact_min_weight).  The reference
    #                                      evaluates every sample (models.py:221-243).  0 = off: on a field whose samples all
    #                                      carry weight (random initialisation) the extra density pass is pure overhead

    autocast_half_tables: bool = True     # training under autocast gathers a HALF copy of the tables, like the reference's
    #                                      _grid_encode (grid.py:41-44); False keeps the fp32 tables (more exact, 2x the bytes)
    autocast_bf16_features: bool = True   # (with autocast_render) the NeRF level's features leave the gather as bf16 pairs -- what
    #                                      its bf16 MLP would round the floats to anyway: bit-identical pixels, half the
    #                                      workspace traffic
    autocast_render: bool = True          # inference marches called under bf16 autocast (the reference's render_image wraps the
    #                                      model call in accelerator.autocast(), models.py:957) run in the reference's mixed
    #                                      precision: half tables in the gather, dense layers as bf16 MFMAs (the training
    #                                      forward kernels without their stores), fp32 compositing.  False: fp32-class always
    sky_side_stream: bool = True         # (with fused_sky_train) the sky branch of a training step on a second HIP stream
    fused_sky_train: bool = True         # training under bf16 autocast runs the sky NeRF on csrc/sky_train.hip (False: eager torch)
    _warned_eval_route = False
    fused_heads_tail: bool = True        # training: per-ray colour correction + sky blend as one HIP node per level (False: eager torch)
    march_route: str = 'auto'            # which march Model.forward runs: 'auto' = the training graph iff self.training and
    #                                      autograd is enabled, else the fused inference march; 'train' / 'inference' force it
    sky_min_background: float = 0.0      # > 0: inference marches evaluate the sky layer only for rays whose background
    #                                      weight 1 - sum(weights of the last level) reaches this value; the others get
    #                                      sky_rgbs = 0 (their pixel moves by < sky_min_background * |A_sky| through
    #                                      models.py:352-354).  The reference evaluates the sky for every ray
    #                                      (models.py:326-337) and RETURNS sky_rgbs, so 0 = off is the default

    def __init__(self, config=None, **kwargs):
        super().__init__()
        set_kwargs(self, kwargs)
        self.config = config
        for k, want in dict(raydist_fn=None, num_glo_features=0, learned_exposure_scaling=False,
                            near_anneal_rate=None, single_mlp=False, distinct_prop=True, use_viewdirs=True).items():
            if getattr(self, k) != want:
                raise NotImplementedError(f"Model.{k}={getattr(self, k)!r} is outside the shipped waymo.gin path")
        if self.bg_intensity_range[0] != self.bg_intensity_range[1]:
            raise NotImplementedError("random background colours (bg_intensity_range) are not on the shipped path")
        self.nerf_mlp = NerfMLP(num_glo_features=self.num_glo_features, num_glo_embeddings=self.num_glo_embeddings)
        for i in range(self.num_levels - 1):
            self.register_module(f'prop_mlp_{i}', PropMLP(grid_disired_resolution=self.prop_desired_grid_size[i]))
        if getattr(self.config, 'model_sky', False):
            self.skynerf = NeRF(D=8, d_in_view=3, W=256, multires_view=4, output_ch=4, skips=[4])
        if getattr(self.config, 'brightness_correction', False):
            self.brightness_corr = BrightnessCorrection(self.config.training_views,
                                                        model_sky=getattr(self.config, 'model_sky', False))

    # ------------------------------------------------------------------------------------------
    def forward(self, rand, batch, train_frac, compute_extras, zero_glo=True, eval_camidx=None):
        """ref models.py:97-365.  `batch` may carry two optional extra keys that pin the random
        draws (the reference draws them from torch's global RNG, render.py:123-124,140 and
        stepfun.py:216): 'rand_vec' [..., num_levels*3] and 'march_noise' (list of per-level dicts
        with 'jitter', 'flip', 'spin').

        Route (`Model.march_route`): 'auto' builds the differentiable graph of train_graph.py (HIP featurisation
        forward / backward, MFMA train kernels, autograd glue) only for a model in TRAINING mode with autograd enabled
        -- what train.py:160-167 does; a model in eval mode, or any call under torch.no_grad(), runs the fully fused
        inference march, whose outputs carry no graph (an eval-mode call outside no_grad used to allocate the training
        graph's activation buffers, 2.1 GB at 8192 x 128 samples).  'train' / 'inference' force one route."""
        route = self.march_route
        if route not in ('auto', 'train', 'inference'):
            raise ValueError(f"Model.march_route={route!r}: expected 'auto', 'train' or 'inference'")
        if route == 'train' or (route == 'auto' and self.training and torch.is_grad_enabled()):
            from . import train_graph
            return train_graph.march_train(self, rand, batch, train_frac, compute_extras, eval_camidx)
        if route == 'auto' and torch.is_grad_enabled() and not self.training and not Model._warned_eval_route:
            Model._warned_eval_route = True
            import warnings
            warnings.warn("ucnerf_amd Model: eval-mode call with autograd enabled takes the inference route (detached outputs); "
                          "call model.train() before a training step, or set Model.march_route = 'train'", stacklevel=2)
        with torch.no_grad():
            return self._march(rand, batch, train_frac, compute_extras, eval_camidx, want_history=True)

    def _mixed_level(self, mlp, is_prop, F_in):
        """Mixed-precision inference of one level (see `autocast_render`): None = the fp32-class path, else what the bf16
        kernels need -- the half table behind a copy of the grid descriptor and the packed / rounded dense parameters."""
        if not (self.autocast_render and torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16):
            return None
        from . import train_graph as tg
        emb = mlp.encoder.embeddings
        probe = torch.empty(1, F_in, device=emb.device)
        if is_prop:
            if not tg._fusable_prop(mlp, probe):
                return None
        elif not tg._fusable_heads(mlp, probe):
            return None
        # the half table and the packed weights are rebuilt only when a parameter changed (a frame is tens of chunks)
        key = tuple((p.data_ptr(), p._version) for p in mlp.parameters()) + (bool(self.autocast_bf16_features),)
        cache = self.__dict__.setdefault('_mixed_cache', {})
        hit = cache.get(id(mlp))
        if hit is not None and hit[0] == key:
            return hit[1]
        out = {}
        half = mlp.encoder.level_dim % 2 == 0                      # grid.py:41-44: half tables when C is even
        d16 = _lib.UcnField()
        ctypes.memmove(ctypes.byref(d16), ctypes.byref(mlp.grid_field()), ctypes.sizeof(_lib.UcnField))
        if half:
            out['table'] = emb.detach().to(torch.half)
            d16.embeddings = out['table'].data_ptr()
        out['desc'], out['table_flag'] = d16, (_lib.TABLE_F16 if half else 0)
        # the NeRF level's features leave the gather as the bf16 pairs its MLP consumes (half the workspace traffic)
        out['feat_flag'] = _lib.FEATURES_BF16 if (half and mlp.encoder.level_dim == 2 and not is_prop and self.autocast_bf16_features) else 0
        with torch.autocast('cuda', enabled=False):
            if is_prop:
                l0, l1 = mlp.density_layer[0], mlp.density_layer[2]
                out['prop'] = tuple(t.detach().float().contiguous() for t in (l0.weight, l0.bias, l1.weight, l1.bias))
            else:
                d0, d1, c0, c1, lr = mlp.density_layer[0], mlp.density_layer[2], mlp.lin_second_stage_0, mlp.lin_second_stage_1, mlp.rgb_layer
                packed, _, We, be, bias0, bias1, biasr = tg.prepare_heads(d0.weight, d0.bias, d1.weight, d1.bias, c0.weight, c0.bias,
                                                                         c1.weight, c1.bias, lr.weight, lr.bias, dir_in_stream=True)
                out.update(packed=packed, We=We, be=be, bias0=bias0, bias1=bias1, biasr=biasr, NW=c0.weight.shape[0],
                           head=(ctypes.c_float * 4)(float(mlp.density_bias), float(mlp.rgb_premultiplier), float(mlp.rgb_bias),
                                                     float(mlp.rgb_padding)),
                           enc=lambda v: _dir_tiles(tg.view_encoding(v.float(), mlp.deg_view)))
        cache[id(mlp)] = (key, out)
        return out

    def _march(self, rand, batch, train_frac, compute_extras, eval_camidx, want_history):
        lib = _lib.load()
        origins = batch['origins']
        _lib.require_device(origins, "batch['origins']")
        dev = origins.device
        prefix = tuple(origins.shape[:-1])
        N = int(np.prod(prefix))
        o = _f32(origins, N, 3)
        d = _f32(batch['directions'], N, 3)
        vd = _f32(batch['viewdirs'], N, 3)
        cam = _f32(batch['cam_dirs'], N, 3)
        rad = _f32(batch['radii'], N, 1)
        near = _f32(batch['near'], N, 1)
        far = _f32(batch['far'], N, 1)
        pinned_vec = batch.get('rand_vec')
        if pinned_vec is not None:
            pinned_vec = _f32(pinned_vec, N, 3 * self.num_levels)
        pinned = batch.get('march_noise')
        st = _lib.stream()
        cfg = self.config
        n_vis = getattr(cfg, 'vis_num_rays', 16)

        if self.anneal_slope > 0:
            anneal = (self.anneal_slope * train_frac) / ((self.anneal_slope - 1) * train_frac + 1)
        else:
            anneal = 1.
        nerf_desc = self.nerf_mlp.field()
        dirb = torch.empty(lib.ucn_field_dir_floats(ctypes.byref(nerf_desc), N), device=dev)
        _lib.check(lib.ucn_field_dir_bias(ctypes.byref(nerf_desc), vd.data_ptr(), N, dirb.data_ptr(), st))

        renderings, ray_history = [], []
        sdist_prev = weights_prev = None
        n_prev = 0
        prod_num_samples = 1
        nerf_chunk = max(1, int(self.max_chunk_rays))
        nerf_row = self.num_nerf_samples * self.nerf_mlp.encoder.num_levels * self.nerf_mlp.encoder.level_dim
        for i_level in range(self.num_levels):
            is_prop = i_level < self.num_levels - 1
            S = self.num_prop_samples if is_prop else self.num_nerf_samples
            mlp = self.get_submodule(f'prop_mlp_{i_level}') if is_prop else self.nerf_mlp
            desc = mlp.field()
            L, C = mlp.encoder.num_levels, mlp.encoder.level_dim
            # max_chunk_rays is quoted for the NeRF level; a proposal level (fewer samples, narrower features) takes
            # proportionally more rays per pass -- the same workspace bytes, fewer and longer launches
            chunk = nerf_chunk
            if is_prop and S * L * C < nerf_row:
                chunk = max(nerf_chunk, min(nerf_chunk * nerf_row // (S * L * C) // 256 * 256, 1 << 16))
            dilation = self.dilation_bias + self.dilation_multiplier * 1.0 / prod_num_samples
            if not (self.dilation_bias > 0 or self.dilation_multiplier > 0):
                dilation = 0.0                                           # ref :167 use_dilation False: resample undilated
            elif not dilation > 0:
                # use_dilation is on but the value is not positive (a negative dilation_bias): the reference would run
                # max_dilate_weights with it; ucn_resample reads dilation <= 0 as the UNdilated branch -- refuse
                raise NotImplementedError(f"dilation {dilation} <= 0 with use_dilation on: outside the shipped configuration")
            prod_num_samples *= S
            # ---- random draws, in the reference's order (stepfun.py:216, render.py:123,124,140)
            jitter = flip = spin = None
            u_tab, max_jitter = _u_table(S, bool(rand), dev)
            if rand:
                pn = pinned[i_level] if pinned is not None else {}
                jcols = 1 if self.single_jitter else S
                jitter = _f32(pn['jitter'], N, jcols) if 'jitter' in pn else torch.rand(N, jcols, device=dev)
                flip = _f32(pn['flip'], N, S) if 'flip' in pn else torch.rand(N, S, device=dev)
                spin = _f32(pn['spin'], N, S) if 'spin' in pn else torch.rand(N, S, device=dev)
            if pinned_vec is not None:
                rvec = pinned_vec[:, 3 * i_level:3 * i_level + 3].contiguous()
            else:
                rvec = torch.randn(N, 3, device=dev)
            # ---- outputs of the level
            sdist = torch.empty(N, S + 1, device=dev)
            density = torch.empty(N, S, device=dev)
            rgbs = None if is_prop else torch.empty(N, S, 3, device=dev)
            weights = torch.empty(N, S, device=dev)
            main = torch.empty(N, 5, device=dev)
            extras = torch.empty(N, 4, device=dev) if compute_extras else None
            coord = torch.empty(N, S, 3, device=dev) if want_history else None
            basis = torch.empty(N, 6, device=dev)
            nc = min(chunk, N)
            feat = torch.empty(L * nc * S * C, device=dev)
            mixed = self._mixed_level(mlp, is_prop, L * C)
            self._mixed_levels = getattr(self, '_mixed_levels', 0) + (mixed is not None)      # diagnostics / tests
            if mixed is not None and not is_prop:
                vd_enc = mixed['enc'](vd)
            _lib.check(lib.ucn_resample(_lib.ptr(sdist_prev), _lib.ptr(weights_prev), n_prev, dilation, anneal,
                                        float(self.resample_padding), u_tab.data_ptr(), _lib.ptr(jitter),
                                        0 if jitter is None else jitter.shape[1], max_jitter, N, S,
                                        sdist.data_ptr(), st))
            _lib.check(lib.ucn_cone_basis(cam.data_ptr(), rvec.data_ptr(), N, basis.data_ptr(), st))
            prof = getattr(self, '_prof', None)
            prof_every = max(1, int(getattr(self, '_prof_every', 1)))
            # Two HIP streams: featurisation of pass i+1 (L2-request / VALU bound) runs beside the MLP of pass i
            # (MFMA bound) on a second feature buffer; the hardware splits the CUs between the two kernels.
            overlap = bool(self.overlap_streams) and N > chunk
            co = _lib.LAUNCH_CORESIDENT if (overlap and not is_prop and self.overlap_streams != 2) else 0     # launch shapes that share a CU (2: plain shapes, tails only)
            cur = torch.cuda.current_stream()
            feats = [feat]
            if overlap:
                feats.append(torch.empty_like(feat))
                if getattr(self, '_side_stream', None) is None:
                    self._side_stream = torch.cuda.Stream()
                side = self._side_stream
                side.wait_stream(cur)
            mlp_done = [None, None]
            for i_pass, r0 in enumerate(range(0, N, chunk)):
                n = min(chunk, N - r0)
                sl = slice(r0, r0 + n)
                fb = feats[i_pass % len(feats)]
                # (bench.py's live per-kernel times: HIP events around every `_prof_every`-th pass only -- an event record costs the queue
                #  ~10 us of idle time at a kernel boundary; around every pass that was 9 ms of a 470 ms frame, tools/frame_gaps.py)
                timed = prof is not None and i_pass % prof_every == prof_every // 2
                if timed:
                    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                    m0 = torch.cuda.Event(enable_timing=True) if overlap else e1       # one stream: the MLP starts where the gather ends
                fstream = side if overlap else cur
                if overlap and mlp_done[i_pass % 2] is not None:
                    side.wait_event(mlp_done[i_pass % 2])             # the buffer's previous reader
                if timed:
                    e0.record(fstream)
                _lib.check(lib.ucn_march_features(
                    ctypes.byref(desc if mixed is None else mixed['desc']), sdist[sl].data_ptr(), near[sl].data_ptr(), far[sl].data_ptr(),
                    o[sl].data_ptr(), d[sl].data_ptr(), basis[sl].data_ptr(), rad[sl].data_ptr(),
                    None if flip is None else flip[sl].data_ptr(), None if spin is None else spin[sl].data_ptr(),
                    float(self.std_scale), n, S, int(self.levels_per_block),
                    ((2 if self.rays_fastest else 0) | co) if mixed is None else (2 | mixed['table_flag'] | (mixed['feat_flag'] if not is_prop else 0)),
                    fb.data_ptr(), None if coord is None else coord[sl].data_ptr(), None, fstream.cuda_stream))
                if timed:
                    e1.record(fstream)
                if overlap:
                    ready = torch.cuda.Event()
                    ready.record(side)
                    cur.wait_event(ready)
                if timed and overlap:
                    m0.record(cur)
                compact = (not is_prop) and (not want_history) and self.compact_min_weight > 0 and mlp.mlp_mode == 1 and mixed is None
                if mixed is not None and is_prop:
                    w0, b0_, w1, b1_ = mixed['prop']
                    _lib.check(lib.ucn_prop_train_fwd(fb.data_ptr(), L * C, w0.shape[0], w0.data_ptr(), b0_.data_ptr(), w1.data_ptr(),
                                                      b1_.data_ptr(), float(mlp.density_bias), 1, n * S, density[sl].data_ptr(), n, C, st))
                elif mixed is not None:
                    # the ray's direction tile rides in the weight stream's last input tile (no per-ray terms to pre-multiply)
                    _lib.check(lib.ucn_train_fwd(fb.data_ptr(), L * C, mixed['packed'].data_ptr(), mixed['bias0'].data_ptr(),
                                                 mixed['bias1'].data_ptr(), mixed['biasr'].data_ptr(), None, None, n, S,
                                                 None, None, None, None, 0, vd_enc[sl].data_ptr(), None, None, mixed['head'],
                                                 density[sl].data_ptr(), rgbs[sl].data_ptr(), None, None, None,
                                                 C | (_lib.FEAT_BF16 if mixed['feat_flag'] else 0), st))
                elif compact:
                    # density head -> weights of this pass's rays -> alive list -> colour layers of the alive samples
                    rf = int(bool(self.rays_fastest))
                    _lib.check(lib.ucn_field_mlp(ctypes.byref(desc), fb.data_ptr(), n * S, S, rf, None, density[sl].data_ptr(),
                                                 None, None, st))
                    _lib.check(lib.ucn_composite(density[sl].data_ptr(), None, sdist[sl].data_ptr(), near[sl].data_ptr(),
                                                 far[sl].data_ptr(), d[sl].data_ptr(), float(self.bg_intensity_range[0]),
                                                 int(bool(self.opaque_background)), n, S, weights[sl].data_ptr(), main[sl].data_ptr(),
                                                 None, st))
                    if getattr(self, '_alive_idx', None) is None or self._alive_idx.numel() < n * S or self._alive_idx.device != dev:
                        self._alive_idx = torch.empty(max(n, nc) * S, dtype=torch.int32, device=dev)
                        self._alive_cnt = torch.zeros(1, dtype=torch.int32, device=dev)
                    _lib.check(lib.ucn_compact_alive(weights[sl].data_ptr(), n, S, rf, float(self.compact_min_weight),
                                                     self._alive_idx.data_ptr(), self._alive_cnt.data_ptr(), st))
                    rgbs[sl].zero_()
                    _lib.check(lib.ucn_field_rgb_compacted(ctypes.byref(desc), fb.data_ptr(), n * S, S, rf,
                                                           dirb[r0 * (dirb.numel() // N):].data_ptr(), self._alive_idx.data_ptr(),
                                                           self._alive_cnt.data_ptr(), rgbs[sl].data_ptr(), st))
                    stats = getattr(self, '_alive_stats', None)
                    if stats is not None:                              # diagnostics (tools/alive_fraction.py): forces a sync
                        stats.append((int(self._alive_cnt.item()), n * S))
                else:
                    _lib.check(lib.ucn_field_mlp(
                        ctypes.byref(desc), fb.data_ptr(), n * S, S, int(bool(self.rays_fastest)) | co,
                        None if is_prop else dirb[r0 * (dirb.numel() // N):].data_ptr(),
                        density[sl].data_ptr(), None if is_prop else rgbs[sl].data_ptr(), None, st))
                if overlap:
                    mlp_done[i_pass % 2] = torch.cuda.Event()
                    mlp_done[i_pass % 2].record(cur)
                if timed:
                    e2.record(cur)
                    prof.append((i_level, n, e0, e1, m0, e2))      # features: e0..e1 on its stream, MLP: m0..e2
            if overlap:
                cur.wait_stream(side)
            _lib.check(lib.ucn_composite(density.data_ptr(), _lib.ptr(rgbs), sdist.data_ptr(), near.data_ptr(),
                                         far.data_ptr(), d.data_ptr(), float(self.bg_intensity_range[0]),
                                         int(bool(self.opaque_background)), N, S, weights.data_ptr(),
                                         main.data_ptr(), _lib.ptr(extras), st))
            rendering = dict(rgb=main[:, 0:3].reshape(prefix + (3,)), depth=main[:, 3].reshape(prefix),
                             acc=main[:, 4].reshape(prefix))
            if compute_extras:
                rendering['distance_mean'] = extras[:, 0].reshape(prefix)
                rendering['distance_percentile_5'] = extras[:, 1].reshape(prefix)
                rendering['distance_median'] = extras[:, 2].reshape(prefix)
                rendering['distance_percentile_95'] = extras[:, 3].reshape(prefix)
            rendering['weights'] = weights.reshape(prefix + (S,))
            level_rgb = rgbs if rgbs is not None else None
            if compute_extras:
                rendering['ray_sdist'] = sdist[:n_vis]
                rendering['ray_weights'] = weights[:n_vis]
                rendering['ray_rgbs'] = (level_rgb[:n_vis] if level_rgb is not None
                                         else torch.zeros(min(n_vis, N), S, 3, device=dev))
            renderings.append(rendering)
            if want_history:
                hist = dict(coord=coord.reshape(prefix + (S, 3)), density=density.reshape(prefix + (S,)),
                            rgb=(level_rgb if level_rgb is not None
                                 else torch.zeros(N, S, 3, device=dev)).reshape(prefix + (S, 3)),
                            raw_grad_density=None, grad_pred=None, normals=None, normals_pred=None, roughness=None,
                            sdist=sdist.reshape(prefix + (S + 1,)).clone(), weights=weights.reshape(prefix + (S,)).clone())
                ray_history.append(hist)
            sdist_prev, weights_prev, n_prev = sdist, weights, S

        if compute_extras:                                             # ref models.py:313-324
            final = (renderings[-1]['ray_rgbs'] * renderings[-1]['ray_weights'][..., None]).sum(dim=-2)
            for r in renderings[:-1]:
                r['ray_rgbs'] = final[:, None, :].expand(r['ray_rgbs'].shape)

        if getattr(cfg, 'model_sky', False):                           # ref models.py:326-337
            # inference under an active bf16 autocast: the sky NeRF's nn.Linear layers are bf16 in the reference
            sky_mixed = bool(self.autocast_render and torch.is_autocast_enabled()
                             and torch.get_autocast_dtype('cuda') == torch.bfloat16)
            self._sky_mixed = sky_mixed
            if self.sky_min_background > 0 and not rand:
                bgw = 1 - renderings[-1]['weights'].reshape(N, -1).sum(dim=-1)
                keep = torch.nonzero(bgw >= self.sky_min_background).reshape(-1)      # host sync, like far[0] in render()
                sky = torch.zeros(N, 3, device=dev)
                if keep.numel() == N:
                    sky = self.skynerf.render(o, d, cam, far, mixed=sky_mixed)
                elif keep.numel():
                    # far0 = 1.5 * far[0] of the FULL batch (models.py:329), not of the kept rays
                    sky[keep] = self.skynerf.render(o[keep], d[keep], cam[keep], far[keep], far0=far.reshape(-1)[:1], mixed=sky_mixed)
                self._sky_kept = (int(keep.numel()), N)
            else:
                sky = self.skynerf.render(o, d, cam, far, mixed=sky_mixed)
            for r in renderings:
                r['sky_rgbs'] = sky
        if getattr(cfg, 'brightness_correction', False):               # ref models.py:339-363
            with_sky = getattr(cfg, 'model_sky', False)
            if eval_camidx is None:
                cam_idx = batch['cam_idx'].reshape(N, -1)[:, 0]
            else:
                cam_idx = torch.as_tensor(eval_camidx).to(dev).reshape(-1)[:1]
            A, A_sky, row_of = self.brightness_corr.affines(cam_idx)
            last_w = renderings[-1]['weights'].reshape(N, -1)          # loop-leaked `rendering` (Appendix C.3)
            for r in renderings:
                rgb_in = r['rgb'].reshape(N, 3).contiguous()
                rgb_out = torch.empty(N, 3, device=dev)
                _lib.check(lib.ucn_apply_affine(rgb_in.data_ptr(), A.data_ptr(), _lib.ptr(row_of),
                                                last_w.data_ptr() if with_sky else None, last_w.shape[1],
                                                r['sky_rgbs'].data_ptr() if with_sky else None,
                                                _lib.ptr(A_sky), N, rgb_out.data_ptr(), st))
                # ref :356-359: [N,1,1,3] for training batches, [N,3] with eval_camidx
                r['rgb'] = rgb_out.reshape(N, 1, 1, 3) if eval_camidx is None else rgb_out
                full = A if row_of is None else A[row_of]
                r['affine_trans'] = full.reshape(-1, 3, 4).expand(N, 3, 4)
                if with_sky:
                    full_s = A_sky if row_of is None else A_sky[row_of]
                    r['affine_trans_sky'] = full_s.reshape(-1, 3, 4).expand(N, 3, 4)
        return renderings, ray_history


import contextlib


@contextlib.contextmanager
def bindings(**per_class):
    """The gin-binding mechanism of the reference (`NerfMLP.grid_level_dim = 2` in a .gin file sets a
    class attribute, configs/waymo.gin:10-20) as a context manager:
        with bindings(NerfMLP=dict(grid_level_dim=2), PropMLP=dict(grid_level_dim=2)):
            model = Model(config=cfg, num_levels=2, ...)"""
    classes = dict(Model=Model, NerfMLP=NerfMLP, PropMLP=PropMLP, MLP=MLP)
    saved = []
    try:
        for cname, attrs in per_class.items():
            cls = classes[cname]
            for k, v in attrs.items():
                saved.append((cls, k, cls.__dict__.get(k, _MISSING)))
                setattr(cls, k, v)
        yield
    finally:
        for cls, k, old in reversed(saved):
            if old is _MISSING:
                delattr(cls, k)
            else:
                setattr(cls, k, old)


_MISSING = object()


def unwrap_model(model):
    """The bare Model behind DistributedDataParallel / DataParallel / accelerate wrappers (anything whose `.module`
    is the wrapped nn.Module), as `accelerator.unwrap_model` would return it."""
    seen = 0
    while not hasattr(model, '_march') and isinstance(getattr(model, 'module', None), nn.Module) and seen < 8:
        model = model.module
        seen += 1
    if not hasattr(model, '_march'):
        raise TypeError(f"render_image: expected a ucnerf_amd Model (possibly DDP-wrapped), got {type(model).__name__}")
    return model


def _dir_tiles(enc):
    """[N, E <= 31] direction encodings -> [N, 32] bf16 tiles [enc, 1, 0...]: the input tile whose column E meets the bias column
    of the colour layers' weight streams (train_graph._head_gather_index(dir_in_stream=True))."""
    n, e = enc.shape
    t = torch.zeros(n, 32, device=enc.device, dtype=torch.bfloat16)
    t[:, :e] = enc
    t[:, e] = 1.0
    return t


_TILE_ORDER = {}


def _tile_order(height, width, tile, device):
    """(perm, inv) int64 [H*W]: the frame's pixels in tile-major order (tile x tile pixel blocks, row-major inside a block
    and over the blocks; ragged blocks at the right / bottom edge) and the inverse permutation."""
    key = (height, width, tile, str(device))
    hit = _TILE_ORDER.get(key)
    if hit is None:
        r = torch.arange(height).reshape(-1, 1).expand(height, width)
        c = torch.arange(width).reshape(1, -1).expand(height, width)
        block = (r // tile) * ((width + tile - 1) // tile) + c // tile
        inside = (r % tile) * tile + c % tile
        perm = torch.argsort((block * (tile * tile) + inside).reshape(-1), stable=True)
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(perm.numel())
        if len(_TILE_ORDER) > 8:
            _TILE_ORDER.clear()
        hit = _TILE_ORDER[key] = (perm.to(device), inv.to(device))
    return hit


def render_image(model, accelerator, batch, rand, train_frac, config, verbose=True, return_weights=False,
                 eval_camidx=0):
    """ref models.py:907-1007: render every pixel of an [H, W, .] ray batch.

    Same signature and returned dict (2-D buffers of the LAST level reshaped to [H, W, ...], and the
    'ray_*' visualisation bundles of every level).  The reference walks the frame in
    `render_chunk_size` pieces, pads, slices this rank's rows and all-gathers every tensor of every
    level per chunk (O(10^3) small collectives per frame).  Here each rank marches ONE contiguous
    row range of the frame in a single Model call and the finished buffers are exchanged with one
    packed all-gather per frame (ucnerf_amd/internal/dist.py) -- rays are independent, weights are
    replicated, so no other communication exists on the path."""
    from . import dist as udist
    model.eval()
    try:      # the mode is restored on EVERY exit: a failed render must not leave the next training step on the inference route
        # the reference hands over the `accelerator.prepare`d model (train.py:95,330, render.py:119,146, eval.py:103,140):
        # a DistributedDataParallel wrapper when num_processes > 1.  The march itself is a method of the bare Model.
        core = unwrap_model(model)
        height, width = batch['origins'].shape[:2]
        num_rays = height * width
        flat = {k: v.reshape((num_rays, -1)) for k, v in batch.items() if v is not None and torch.is_tensor(v)}
        # Rays are marched in TILE-major order (8 x 8 pixel blocks): a wave's 64 lanes are then the rays of one block, whose
        # samples at one depth index sit within ~8 pixel footprints of each other in BOTH image directions -- on the hash
        # grid's coarse and middle levels they share lattice cells, i.e. cache lines, and the gather's requests coalesce
        # (a 64 x 1 pixel row spreads 8x further).  Pixels do not depend on the order; the outputs are put back below.
        tile = int(getattr(config, 'render_ray_tile', 8))
        perm = inv = None
        if tile > 1 and height > 1 and width > 1 and flat['origins'].is_cuda:
            perm, inv = _tile_order(height, width, tile, flat['origins'].device)
            flat = {k: v.index_select(0, perm) for k, v in flat.items()}
        world = getattr(accelerator, 'num_processes', 1)
        rank = getattr(accelerator, 'process_index', 0)
        lo, hi = udist.shard_bounds(num_rays, world, rank)
        shard = {k: v[lo:hi] for k, v in flat.items()}
        with torch.no_grad():
            renderings, history = core._march(rand, shard, train_frac, True, eval_camidx, want_history=return_weights)
        last = renderings[-1]
        keys = [k for k in last if not k.startswith('ray_')]
        # rendering['weights'] ([H, W, S] of the last level) is 27x the pixels' payload.  The reference gathers it with
        # everything else (models.py:965-968) and so does this function by default -- the returned key set never depends
        # on the world size.  None of the reference's callers reads it unless return_weights=True (where models.py:977
        # overwrites it with the history's copy): `config.render_gather_weights = False` is the explicit opt-out that drops
        # the key at EVERY world size.
        if not return_weights and not getattr(config, 'render_gather_weights', True):
            keys = [k for k in keys if k != 'weights']
        local = {k: last[k].reshape(hi - lo, -1) for k in keys}
        if return_weights:
            local['weights'] = history[-1]['weights'].reshape(hi - lo, -1)
            local['coord'] = history[-1]['coord'].reshape(hi - lo, -1)
        shapes = {k: tuple(last[k].shape[1:]) for k in keys}
        if return_weights:
            shapes['weights'] = tuple(history[-1]['weights'].shape[1:])
            shapes['coord'] = tuple(history[-1]['coord'].shape[1:])
        gathered = udist.all_gather_rows(local, num_rays, world, rank)
        if inv is not None:
            gathered = {k: v.index_select(0, inv) for k, v in gathered.items()}
        rendering = {k: gathered[k].reshape((height, width) + shapes[k]) for k in gathered}
        # 'ray_*' bundles: vis_num_rays rays per level, drawn like the reference's final randperm subset
        bundle_keys = [k for k in last if k.startswith('ray_')]
        if bundle_keys:
            n_vis = getattr(config, 'vis_num_rays', 16)
            per_level = [{k: r[k] for k in bundle_keys} for r in renderings]
            per_level = udist.all_gather_bundles(per_level, world, rank)
            n_have = per_level[0][bundle_keys[0]].shape[0]
            pick = torch.randperm(n_have)[:n_vis].to(per_level[0][bundle_keys[0]].device)
            for k in bundle_keys:
                rendering[k] = [lvl[k][pick] for lvl in per_level]
    finally:
        model.train()                   # ref models.py:1006 (unconditional)
    return rendering
