"""torch-CPU restatement of the UC-NeRF per-ray sampling + integration path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  It is the checker for the HIP
path and the timed CPU baseline of bench.py; the product never imports it.

Every function cites the reference lines it restates (paths relative to
/root/reference/nerf/).  The reference has no tests or golden vectors for this path;
this file is pinned against the reference's own Python imported in the authoring
container (tests/golden/make_golden.py -> tests/golden/*.npz, checked by
tests/test_oracle_golden.py).

State is a flat ``dict`` of float32 CPU tensors keyed exactly like the reference's
``Model.state_dict()``, plus a ``PathSpec`` describing the
architecture.  Randomness is always an explicit input (``LevelNoise``) so that the
same draws can be fed to the HIP path.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import grid_cpu

EPS = float(torch.finfo(torch.float32).eps)
SQRT2 = 2 ** 0.5


class _GridEncodeCPU(torch.autograd.Function):
    """grid.py:24-89 on the CPU: forward and table-gradient through oracle/grid_oracle.c."""

    @staticmethod
    def forward(ctx, pts01, emb, offsets, pls, H):
        out = grid_cpu.encode(pts01.detach(), emb.detach(), offsets, pls, H)
        ctx.save_for_backward(pts01.detach().contiguous().float(), emb.detach(), offsets)
        ctx.meta = (pls, H)
        return out

    @staticmethod
    def backward(ctx, g):
        pts, emb, offsets = ctx.saved_tensors
        pls, H = ctx.meta
        B, D = pts.shape
        L, C = offsets.numel() - 1, emb.shape[1]
        grad = g.reshape(B, L, C).permute(1, 0, 2).contiguous().float()
        gemb = torch.zeros_like(emb)
        grid_cpu.grid_encode_backward(grad, pts, emb.contiguous(), offsets, gemb, B, D, C, L, np.log2(pls), H,
                                      None, None, 0, False, 0)
        return None, gemb, None, None, None


# --------------------------------------------------------------------------- specs
@dataclass
class FieldSpec:
    """One MLP + hash grid (models.py:367-483 `MLP.__init__`)."""
    prefix: str                      # 'nerf_mlp' | 'prop_mlp_0' ...
    grid_base_resolution: int = 16
    grid_desired_resolution: int = 8192
    grid_level_dim: int = 4
    grid_log2_hashmap_size: int = 21
    grid_level_interval: int = 2
    disable_rgb: bool = False        # PropMLP under waymo.gin
    bottleneck_width: int = 256
    net_width_viewdirs: int = 256
    net_depth_viewdirs: int = 2
    skip_layer_dir: int = 0
    deg_view: int = 4
    density_bias: float = -1.0
    rgb_padding: float = 0.001
    rgb_premultiplier: float = 1.0
    rgb_bias: float = 0.0

    @property
    def num_grid_levels(self):       # models.py:425-426
        return int(np.log(self.grid_desired_resolution / self.grid_base_resolution)
                   / np.log(self.grid_level_interval)) + 1

    def layout(self):
        return grid_cpu.table_layout(self.num_grid_levels, self.grid_level_dim,
                                     self.grid_base_resolution, self.grid_desired_resolution,
                                     self.grid_log2_hashmap_size)


@dataclass
class PathSpec:
    """`Model` class attributes that shape the path (models.py:33-55)."""
    num_levels: int = 2
    num_prop_samples: int = 128
    num_nerf_samples: int = 32
    prop_desired_grid_size: List[int] = field(default_factory=lambda: [512, 2048])
    anneal_slope: float = 10.0
    dilation_multiplier: float = 0.5
    dilation_bias: float = 0.0025
    resample_padding: float = 0.0
    single_jitter: bool = True
    opaque_background: bool = False
    bg_intensity: float = 1.0
    std_scale: float = 0.5
    model_sky: bool = False
    brightness_correction: bool = False
    training_views: int = 210
    vis_num_rays: int = 16
    nerf: FieldSpec = None
    props: List[FieldSpec] = None

    def field_for_level(self, i_level):
        return self.props[i_level] if i_level < self.num_levels - 1 else self.nerf


@dataclass
class LevelNoise:
    """The random tensors one sampling level consumes, in the reference's draw order."""
    rand_vec: torch.Tensor                       # render.py:140  randn_like(cam_dirs)  [N,3]
    jitter: Optional[torch.Tensor] = None        # stepfun.py:216 rand(N,1|S)          (train)
    flip: Optional[torch.Tensor] = None          # render.py:123  rand_like(t0[...,0]) [N,S] (train)
    spin: Optional[torch.Tensor] = None          # render.py:124  rand_like(deg[...,0]) [N,S] (train)


def draw_level_noise(spec, n_rays, i_level, train, generator=None):
    """Draw with torch's CPU generator in exactly the order Model.forward does for one level
    (stepfun.py:216 -> render.py:123,124,140)."""
    S = spec.num_prop_samples if i_level < spec.num_levels - 1 else spec.num_nerf_samples
    kw = dict(generator=generator)
    jitter = flip = spin = None
    if train:
        jitter = torch.rand(n_rays, 1 if spec.single_jitter else S, **kw)
        flip = torch.rand(n_rays, S, **kw)
        spin = torch.rand(n_rays, S, **kw)
    rand_vec = torch.randn(n_rays, 3, **kw)
    return LevelNoise(rand_vec=rand_vec, jitter=jitter, flip=flip, spin=spin)


# --------------------------------------------------------------------- step functions
def interp_sorted(x, xp, fp):
    """Piecewise-linear lookup with sorted knots (math.py:88-107 `sorted_interp`).

    The reference builds an O(n*m) mask `x >= xp`; with xp, fp non-decreasing the bracketing
    knots are simply the last knot <= x and the first knot > x (both clamped to the ends),
    which a binary search finds.  The closing arithmetic is the reference's."""
    n = xp.shape[-1]
    cnt = torch.searchsorted(xp.contiguous(), x.contiguous(), right=True)   # #knots <= x
    i0 = (cnt - 1).clamp_min(0)
    i1 = cnt.clamp_max(n - 1)
    xp0, xp1 = xp.gather(-1, i0), xp.gather(-1, i1)
    fp0, fp1 = fp.gather(-1, i0), fp.gather(-1, i1)
    frac = torch.nan_to_num((x - xp0) / (xp1 - xp0), 0).clamp(0, 1)
    return fp0 + frac * (fp1 - fp0)


def cdf_of_weights(w):
    """stepfun.py:108-128 `integrate_weights`: [0, min(cumsum(w[:-1]),1), 1]."""
    body = torch.cumsum(w[..., :-1], dim=-1).clamp_max(1)
    edge = body.new_zeros(body.shape[:-1] + (1,))
    return torch.cat([edge, body, edge + 1], dim=-1)


def dilate_weights(t, w, dilation, lo, hi):
    """stepfun.py:75-105 `max_dilate_weights(renormalize=True)` (+ weight<->pdf :64-72)."""
    pdf = w / (t[..., 1:] - t[..., :-1]).clamp_min(EPS)
    left = t[..., :-1] - dilation
    right = t[..., 1:] + dilation
    knots = torch.sort(torch.cat([t, left, right], dim=-1), dim=-1).values.clamp(lo, hi)
    covers = (left[..., None, :] <= knots[..., :, None]) & (right[..., None, :] > knots[..., :, None])
    env = torch.where(covers, pdf[..., None, :], torch.zeros_like(pdf[..., None, :])).amax(dim=-1)
    wd = env[..., :-1] * (knots[..., 1:] - knots[..., :-1])
    wd = wd / wd.sum(dim=-1, keepdim=True).clamp_min(EPS)
    return knots, wd


def sample_fenceposts(t, logits, num_samples, lo, hi, jitter=None):
    """stepfun.py:251-294 `sample_intervals` -> :175-218 `sample(deterministic_center=True)`
    -> :154-161 `invert_cdf`.  `jitter` None = eval (linspace), else the U[0,1) draw."""
    if jitter is None:
        pad = 1 / (2 * num_samples)
        u = torch.linspace(pad, 1. - pad - EPS, num_samples)
        u = u.expand(t.shape[:-1] + (num_samples,))
    else:
        u_max = EPS + (1 - EPS) / num_samples
        max_jitter = (1 - u_max) / (num_samples - 1) - EPS
        u = torch.linspace(0, 1 - u_max, num_samples) + jitter * max_jitter
    cdf = cdf_of_weights(torch.softmax(logits, dim=-1))
    centers = interp_sorted(u, cdf, t)
    mid = (centers[..., 1:] + centers[..., :-1]) / 2
    first = (2 * centers[..., :1] - mid[..., :1]).clamp_min(lo)
    last = (2 * centers[..., -1:] - mid[..., -1:]).clamp_max(hi)
    return torch.cat([first, mid, last], dim=-1)


def percentiles_of_stepfun(t, w, ps=(5, 50, 95)):
    """stepfun.py:329-339 `weighted_percentile`."""
    cdf = cdf_of_weights(w)
    q = (torch.tensor(ps) / 100).expand(t.shape[:-1] + (len(ps),))
    return interp_sorted(q, cdf, t)


# ------------------------------------------------------------------------ ray geometry
_HEX_ORDER = (0, 2, 4, 3, 5, 1)


def cone_multisamples(tdist, origins, directions, cam_dirs, radii, rand_vec, std_scale=0.5,
                      flip=None, spin=None):
    """render.py:94-152 `cast_rays`: six points per interval on a spiral inside the cone.

    tdist [N,S+1]; origins/directions/cam_dirs [N,3]; radii [N,1]; rand_vec [N,3].
    flip/spin None = the deterministic (rand=False) pattern; otherwise the two U[0,1) draws."""
    t0 = tdist[..., :-1, None]
    t1 = tdist[..., 1:, None]
    r = radii[..., None]
    t_m = (t0 + t1) / 2
    t_d = (t1 - t0) / 2
    j = torch.arange(6)
    t = t0 + t_d / (t_d ** 2 + 3 * t_m ** 2) * (
        t1 ** 2 + 2 * t_m ** 2 + 3 / 7 ** 0.5 * (2 * j / 5 - 1) *
        ((t_d ** 2 - t_m ** 2) ** 2 + 4 * t_m ** 4).sqrt())
    ang = (torch.pi / 3 * torch.tensor(_HEX_ORDER, dtype=torch.float)).expand(t.shape)
    if flip is not None:
        keep = flip > 0.5
        ang = ang + 2 * torch.pi * spin[..., None]
        ang = torch.where(keep[..., None], ang, torch.pi * 5 / 3 - ang)
    else:
        even = (torch.arange(t.shape[-2]) % 2 == 0).expand(t.shape[:-1])
        ang = torch.where(even[..., None], ang, ang + torch.pi / 6)
        ang = torch.where(even[..., None], ang, torch.pi * 5 / 3 - ang)
    local = torch.stack([r * t * torch.cos(ang) / SQRT2, r * t * torch.sin(ang) / SQRT2, t], dim=-1)
    stds = std_scale * r * t / SQRT2
    e1 = F.normalize(torch.cross(cam_dirs, rand_vec, dim=-1), dim=-1)
    e2 = F.normalize(torch.cross(cam_dirs, e1, dim=-1), dim=-1)
    # math.py:10-11 elementwise "matmul" with basis^T: sum_k local_k * axis_k
    axes = torch.stack([e1, e2, directions], dim=-2)            # [N, k, xyz]
    world = (local[..., :, None] * axes[..., None, None, :, :]).sum(dim=-2)
    return world + origins[..., None, None, :], stds, t


def contract_points(mean, std):
    """coord.py:60-72 `contract_mean_std` (reached via track_linearize, coord.py:75-116)."""
    m = (mean ** 2).sum(dim=-1, keepdim=True).clamp_min(EPS)
    root = torch.sqrt(m)
    inside = m <= 1
    z = torch.where(inside, mean, ((2 * torch.sqrt(m) - 1) / m) * mean)
    shrink = (torch.pow(2 * root - 1, 1 / 3) / root) ** 2
    s = torch.where(inside[..., 0], std, shrink[..., 0] * std)
    return z, s


def view_encoding(d, deg=4):
    """coord.py:214-225 `pos_enc(min_deg=0, max_deg=deg, append_identity=True)` -> 3+6*deg."""
    scales = 2 ** torch.arange(0, deg)
    scaled = (d[..., None, :] * scales[:, None]).reshape(d.shape[:-1] + (-1,))
    return torch.cat([d, torch.sin(torch.cat([scaled, scaled + 0.5 * torch.pi], dim=-1))], dim=-1)


# ------------------------------------------------------------------------------ fields
def _lin(x, sd, name):
    return F.linear(x, sd[name + '.weight'], sd[name + '.bias'])


def level_damping(stds, grid_sizes):
    """models.py:495.  NOTE the reference squares the *int32* `grid_sizes` buffer, which wraps
    for sizes >= 46341 (e.g. 65537**2 -> 131073); torch reproduces the wrap, and so must any
    re-implementation."""
    return torch.erf(1 / torch.sqrt(8 * stds[..., None] ** 2 * grid_sizes ** 2))


def field_density_features(fs: FieldSpec, sd, means, stds, no_warp=False):
    """models.py:485-512 `predict_density`: contract, /2, hash-grid, per-level erf damping,
    mean over the multisample axis (-2 of means' leading dims), density MLP.
    means [...,G,3], stds [...,G] -> raw_density [...], bottleneck [...,NB], coord [...,3]."""
    pls, offsets, grid_sizes, _ = fs.layout()
    if not no_warp:
        flat_m, flat_s = contract_points(means.reshape(-1, 3), stds.reshape(-1))
        means = flat_m.reshape(means.shape) / 2
        stds = flat_s.reshape(stds.shape) / 2
    emb = sd[fs.prefix + '.encoder.embeddings']
    pts01 = ((means + 1) / 2).reshape(-1, 3)                         # grid.py:162
    feat = _GridEncodeCPU.apply(pts01, emb, offsets, pls, fs.grid_base_resolution)
    feat = feat.reshape(means.shape[:-1] + (fs.num_grid_levels, fs.grid_level_dim))
    damp = level_damping(stds, grid_sizes)
    feat = (feat * damp[..., None]).mean(dim=-3).flatten(-2, -1)
    h = F.relu(_lin(feat, sd, fs.prefix + '.density_layer.0'))
    x = _lin(h, sd, fs.prefix + '.density_layer.2')
    return x[..., 0], x, means.mean(dim=-2), feat


def field_forward(fs: FieldSpec, sd, means, stds, viewdirs, no_warp=False):
    """models.py:514-685 `MLP.forward` under waymo.gin (no normals, no GLO, no reflections)."""
    raw, x, coord, _ = field_density_features(fs, sd, means, stds, no_warp)
    density = F.softplus(raw + fs.density_bias)
    if fs.disable_rgb:
        rgb = torch.zeros(density.shape + (3,))
    else:
        enc = view_encoding(viewdirs, fs.deg_view)
        enc = enc[..., None, :].expand(x.shape[:-1] + (enc.shape[-1],))
        h = torch.cat([x, enc], dim=-1)
        skip = h
        for i in range(fs.net_depth_viewdirs):
            h = F.relu(_lin(h, sd, f'{fs.prefix}.lin_second_stage_{i}'))
            if i == fs.skip_layer_dir:
                h = torch.cat([h, skip], dim=-1)
        rgb = torch.sigmoid(fs.rgb_premultiplier * _lin(h, sd, fs.prefix + '.rgb_layer') + fs.rgb_bias)
        rgb = rgb * (1 + 2 * fs.rgb_padding) - fs.rgb_padding
    return dict(coord=coord, density=density, rgb=rgb)


# --------------------------------------------------------------------------- rendering
def alpha_weights(density, tdist, dirs, opaque_background=False):
    """render.py:155-174 `compute_alpha_weights`."""
    delta = (tdist[..., 1:] - tdist[..., :-1]) * torch.norm(dirs[..., None, :], dim=-1)
    tau = density * delta
    if opaque_background:
        tau = torch.cat([tau[..., :-1], torch.full_like(tau[..., -1:], torch.inf)], dim=-1)
    alpha = 1 - torch.exp(-tau)
    trans = torch.exp(-torch.cat([torch.zeros_like(tau[..., :1]),
                                  torch.cumsum(tau[..., :-1], dim=-1)], dim=-1))
    return alpha * trans


def composite(rgbs, weights, tdist, bg, t_far, extras=True):
    """render.py:177-244 `volumetric_rendering` incl. the acc<0.6 -> depth 300 sentinel."""
    out = {}
    acc = weights.sum(dim=-1)
    bg_w = (1 - acc[..., None]).clamp_min(0.)
    out['rgb'] = (weights[..., None] * rgbs).sum(dim=-2) + bg_w * bg
    t_mid = 0.5 * (tdist[..., :-1] + tdist[..., 1:])
    depth = torch.nan_to_num((weights * t_mid).sum(dim=-1) / acc.clamp_min(EPS), torch.inf)
    depth = torch.clip(depth, tdist[..., 0], tdist[..., -1]).clone()
    depth[acc < 0.6] = 300
    out['depth'] = depth
    out['acc'] = acc
    if extras:
        logmean = (weights * torch.log(t_mid)).sum(dim=-1) / acc.clamp_min(EPS)
        out['distance_mean'] = torch.clip(torch.nan_to_num(torch.exp(logmean), torch.inf),
                                          tdist[..., 0], tdist[..., -1])
        pct = percentiles_of_stepfun(torch.cat([tdist, t_far], dim=-1),
                                     torch.cat([weights, bg_w], dim=-1))
        out['distance_percentile_5'] = pct[..., 0]
        out['distance_median'] = pct[..., 1]
        out['distance_percentile_95'] = pct[..., 2]
    return out


# ------------------------------------------------------------------------- sky + colour
def sky_layer(sd, origins, directions, cam_dirs, far, n_samples=120):
    """models.py:326-337 call + :852-904 `render_rays` + :822-850 `raw2outputs` +
    :743-820 `NeRF(D=8,W=256,skips=[4],multires_view=4)`.  Quirks kept: z runs from
    batch.far DOWN to 1/(1.5*far[0]) (models.py:872), the view branch is fed cam_dirs."""
    near = far.reshape(-1, 1)
    sky_far = torch.full_like(near, float(near[0]) * 1.5)
    tv = torch.linspace(0., 1., steps=n_samples)
    z = (near * (1. - tv) + 1. / sky_far * tv).expand(near.shape[0], n_samples)
    pts = origins[:, None, :] + directions[:, None, :] * z[:, :, None]
    views = cam_dirs[:, None, :].expand(-1, n_samples, -1)
    freqs = 2. ** torch.linspace(0., 3., 4)                                   # models.py:714
    venc = torch.cat([views] + [fn(views * f) for f in freqs for fn in (torch.sin, torch.cos)], -1)
    h = pts
    for i in range(8):
        h = F.relu(_lin(h, sd, f'skynerf.pts_linears.{i}'))
        if i == 4:
            h = torch.cat([pts, h], dim=-1)
    sigma = _lin(h, sd, 'skynerf.alpha_linear')
    h = torch.cat([_lin(h, sd, 'skynerf.feature_linear'), venc], dim=-1)
    h = F.relu(_lin(h, sd, 'skynerf.views_linears.0'))
    rgb = torch.sigmoid(_lin(h, sd, 'skynerf.rgb_linear'))
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)], dim=-1)
    dists = dists * torch.norm(directions[:, None, :], dim=-1)
    alpha = 1. - torch.exp(-F.relu(sigma[..., 0]) * dists)
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1. - alpha + 1e-10], -1), -1)[:, :-1]
    w = alpha * trans
    return (w[..., None] * rgb).sum(dim=-2)


def brightness_affine(sd, cam_idx, which='latent_code'):
    """extrinsic_optimizer.py:4-48: latent[idx] -> 4->256->256->256->12 -> [.,3,4]."""
    x = sd['brightness_corr.' + which][cam_idx.reshape(-1).long()]
    for i in range(3):
        x = F.relu(_lin(x, sd, f'brightness_corr.brightness_MLP.pts_linears.{i}'))
    return _lin(x, sd, 'brightness_corr.brightness_MLP.output_linear').view(-1, 3, 4)


# -------------------------------------------------------------------------- whole path
def model_forward(spec: PathSpec, sd, batch, noise: List[LevelNoise], train_frac=1.0,
                  compute_extras=True, eval_camidx=None, training=False):
    """models.py:97-365 `Model.forward` for flat [N,.] batches (eval layout).

    Returns (renderings, ray_history) with the reference's keys.  `noise[i]` carries the
    random tensors of level i; jitter/flip/spin None selects the rand=False branches."""
    near, far = batch['near'], batch['far']
    sdist = torch.cat([torch.zeros_like(near), torch.ones_like(far)], dim=-1)
    weights = torch.ones_like(near)
    prod = 1
    renderings, history = [], []
    for lvl in range(spec.num_levels):
        fs = spec.field_for_level(lvl)
        is_prop = lvl < spec.num_levels - 1
        S = spec.num_prop_samples if is_prop else spec.num_nerf_samples
        dilation = spec.dilation_bias + spec.dilation_multiplier * 1.0 / prod       # :158-159
        prod *= S
        if lvl > 0 and (spec.dilation_bias > 0 or spec.dilation_multiplier > 0):               # :167-168 use_dilation
            sdist, weights = dilate_weights(sdist, weights, dilation, 0., 1.)
            sdist, weights = sdist[..., 1:-1], weights[..., 1:-1]
        anneal = (spec.anneal_slope * train_frac) / ((spec.anneal_slope - 1) * train_frac + 1)
        logits = torch.where(sdist[..., 1:] > sdist[..., :-1],
                             anneal * torch.log(weights + spec.resample_padding),
                             torch.full_like(sdist[..., :-1], -torch.inf))
        nz = noise[lvl]
        sdist = sample_fenceposts(sdist, logits, S, 0., 1., nz.jitter).detach()
        tdist = sdist * far + (1 - sdist) * near                                     # coord.py:176
        means, stds, ts = cone_multisamples(tdist, batch['origins'], batch['directions'],
                                            batch['cam_dirs'], batch['radii'], nz.rand_vec,
                                            spec.std_scale, nz.flip, nz.spin)
        res = field_forward(fs, sd, means, stds, batch['viewdirs'])
        if spec.brightness_correction:                                               # models.py:232-235
            res['rgb'], res['density'] = _GradientScaler.apply(res['rgb'], res['density'], ts.mean(dim=-1))
        weights = alpha_weights(res['density'], tdist, batch['directions'], spec.opaque_background)
        rendering = composite(res['rgb'], weights, tdist, spec.bg_intensity, far, compute_extras)
        rendering['weights'] = weights
        if compute_extras:
            n = spec.vis_num_rays
            rendering['ray_sdist'] = sdist.reshape(-1, sdist.shape[-1])[:n]
            rendering['ray_weights'] = weights.reshape(-1, weights.shape[-1])[:n]
            rendering['ray_rgbs'] = res['rgb'].reshape((-1,) + res['rgb'].shape[-2:])[:n]
        if training:                                                                  # :297-306
            emb = sd[fs.prefix + '.encoder.embeddings']
            _, offsets, _, _ = fs.layout()
            per_level = torch.stack([(emb[offsets[i]:offsets[i + 1]] ** 2).mean(dim=0)
                                     for i in range(fs.num_grid_levels)])
            res['loss_hash_decay'] = per_level.mean()
        res['sdist'] = sdist.clone()
        res['weights'] = weights.clone()
        res['ts_mean'] = ts.mean(dim=-1)
        renderings.append(rendering)
        history.append(res)
    if compute_extras:                                                               # :313-324
        final = (renderings[-1]['ray_rgbs'] * renderings[-1]['ray_weights'][..., None]).sum(dim=-2)
        for r in renderings[:-1]:
            r['ray_rgbs'] = final[:, None, :].expand(r['ray_rgbs'].shape)
    if spec.model_sky:
        sky = sky_layer(sd, batch['origins'], batch['directions'], batch['cam_dirs'], far)
        for r in renderings:
            r['sky_rgbs'] = sky
    if spec.brightness_correction:                                                   # :339-363
        n = renderings[0]['rgb'].shape[0]
        idx = batch['cam_idx'][..., 0] if eval_camidx is None else eval_camidx.repeat(n)
        A = brightness_affine(sd, idx)
        A_sky = brightness_affine(sd, idx, 'sky_latent_code') if spec.model_sky else None
        last_w = renderings[-1]['weights']          # the loop-leaked `rendering` (Appendix C.3)
        for r in renderings:
            rgb = torch.bmm(A[:, :3, :3], r['rgb'].unsqueeze(-1)) + A[:, :3, 3:]
            if spec.model_sky:
                opac = 1 - last_w.sum(dim=-1, keepdim=True)
                rgb = rgb + opac.unsqueeze(-1).repeat(1, 3, 1) * (
                    torch.bmm(A_sky[:, :3, :3], r['sky_rgbs'].unsqueeze(-1)) + A_sky[:, :3, 3:])
            r['rgb'] = rgb.squeeze(-1)
            r['affine_trans'] = A
            if spec.model_sky:
                r['affine_trans_sky'] = A_sky
    return renderings, history


class _GradientScaler(torch.autograd.Function):
    """train_utils.py:101-111 `GradientScaler`: identity forward; the gradients of colours and densities are scaled by
    clamp(mean t of the sample's multisamples ^ 2, 0, 1).  The reference applies it at every level whenever
    `config.brightness_correction` is on (models.py:232-235) -- found missing here in round 3, when the sky + colour-head
    training fixture was first compared with this restatement's autograd (field gradients were 7-23 % too large)."""

    @staticmethod
    def forward(ctx, colors, sigmas, ray_dist):
        ctx.save_for_backward(ray_dist)
        return colors.view_as(colors), sigmas.view_as(sigmas)

    @staticmethod
    def backward(ctx, g_colors, g_sigmas):
        (ray_dist,) = ctx.saved_tensors
        scaling = torch.square(ray_dist).clamp(0, 1)
        return g_colors * scaling[..., None], g_sigmas * scaling, None


# ------------------------------------------------------------------- parameter factory
def init_state(spec: PathSpec, seed=0, table_range=1.0):
    """A state_dict with the reference's keys/shapes and init laws (default nn.Linear init,
    kaiming_uniform for lin_second_stage_*, models.py:478) but tables ~U(-range, range)
    (the reference's +-1e-4 init makes the grid a no-op)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def linear(name, n_in, n_out, kaiming=False):
        bound_w = (6.0 / n_in) ** 0.5 if kaiming else (1.0 / n_in) ** 0.5
        sd[name + '.weight'] = (torch.rand(n_out, n_in, generator=g) * 2 - 1) * bound_w
        sd[name + '.bias'] = (torch.rand(n_out, generator=g) * 2 - 1) * (1.0 / n_in) ** 0.5

    for fs in list(spec.props[:spec.num_levels - 1]) + [spec.nerf]:
        _, offsets, grid_sizes, idx = fs.layout()
        rows = int(offsets[-1])
        sd[fs.prefix + '.encoder.embeddings'] = (torch.rand(rows, fs.grid_level_dim, generator=g) * 2 - 1) * table_range
        sd[fs.prefix + '.encoder.offsets'] = offsets
        sd[fs.prefix + '.encoder.grid_sizes'] = grid_sizes
        n_feat = fs.num_grid_levels * fs.grid_level_dim
        linear(fs.prefix + '.density_layer.0', n_feat, 64)
        linear(fs.prefix + '.density_layer.2', 64, 1 if fs.disable_rgb else fs.bottleneck_width)
        if not fs.disable_rgb:
            n_dir = 3 + 6 * fs.deg_view
            n_in = fs.bottleneck_width + n_dir
            last = n_in
            for i in range(fs.net_depth_viewdirs):
                linear(f'{fs.prefix}.lin_second_stage_{i}', last, fs.net_width_viewdirs, kaiming=True)
                last = fs.net_width_viewdirs + (n_in if i == fs.skip_layer_dir else 0)
            linear(fs.prefix + '.rgb_layer', last, 3)
    if spec.model_sky:
        for i in range(8):
            linear(f'skynerf.pts_linears.{i}', 3 if i == 0 else (259 if i == 5 else 256), 256)
        linear('skynerf.views_linears.0', 283, 128)
        linear('skynerf.feature_linear', 256, 256)
        linear('skynerf.alpha_linear', 256, 1)
        linear('skynerf.rgb_linear', 128, 3)
    if spec.brightness_correction:
        sd['brightness_corr.latent_code'] = torch.randn(spec.training_views, 4, generator=g) * 0.1
        if spec.model_sky:
            sd['brightness_corr.sky_latent_code'] = torch.randn(spec.training_views, 4, generator=g) * 0.1
        for i in range(3):
            linear(f'brightness_corr.brightness_MLP.pts_linears.{i}', 4 if i == 0 else 256, 256)
        linear('brightness_corr.brightness_MLP.output_linear', 256, 12)
    return sd


def make_spec(kind='B', **over):
    """Named architectures: 'R' = reference waymo.gin, 'B' = BASELINE
    configs 2-4, 'tiny' = small tables for fast parity tests."""
    if kind == 'R':
        nerf = FieldSpec('nerf_mlp')
        prop = FieldSpec('prop_mlp_0', grid_desired_resolution=512, disable_rgb=True)
        spec = PathSpec(num_levels=2, num_prop_samples=128, num_nerf_samples=32, nerf=nerf, props=[prop])
    elif kind == 'B':
        nerf = FieldSpec('nerf_mlp', grid_desired_resolution=524288, grid_level_dim=2, grid_log2_hashmap_size=19)
        prop = FieldSpec('prop_mlp_0', grid_desired_resolution=512, grid_level_dim=2,
                         grid_log2_hashmap_size=19, disable_rgb=True)
        spec = PathSpec(num_levels=2, num_prop_samples=64, num_nerf_samples=128, nerf=nerf, props=[prop])
    elif kind == 'tiny':
        nerf = FieldSpec('nerf_mlp', grid_desired_resolution=524288, grid_level_dim=2, grid_log2_hashmap_size=12)
        prop = FieldSpec('prop_mlp_0', grid_desired_resolution=512, grid_level_dim=2,
                         grid_log2_hashmap_size=12, disable_rgb=True)
        spec = PathSpec(num_levels=2, num_prop_samples=64, num_nerf_samples=128, nerf=nerf, props=[prop])
    elif kind in ('cfg1', 'tiny64'):
        # BASELINE.json configs[0]: 64 + 64 samples, "2x64" colour MLP = bottleneck_width 64,
        # net_width_viewdirs 64, L = 16 / C = 2 / T = 2^19 tables ('tiny64': T = 2^12 for small fixtures)
        T = 19 if kind == 'cfg1' else 12
        nerf = FieldSpec('nerf_mlp', grid_desired_resolution=524288, grid_level_dim=2, grid_log2_hashmap_size=T,
                         bottleneck_width=64, net_width_viewdirs=64)
        prop = FieldSpec('prop_mlp_0', grid_desired_resolution=512, grid_level_dim=2, grid_log2_hashmap_size=T,
                         disable_rgb=True)
        spec = PathSpec(num_levels=2, num_prop_samples=64, num_nerf_samples=64, nerf=nerf, props=[prop])
    elif kind == 'tinyR':
        nerf = FieldSpec('nerf_mlp', grid_log2_hashmap_size=12)
        prop = FieldSpec('prop_mlp_0', grid_desired_resolution=512, grid_log2_hashmap_size=12, disable_rgb=True)
        spec = PathSpec(num_levels=2, num_prop_samples=128, num_nerf_samples=32, nerf=nerf, props=[prop])
    else:
        raise ValueError(kind)
    for k, v in over.items():
        setattr(spec, k, v)
    return spec


def synthetic_rays(n, seed=0, near=0.0, far=8.0, width=1920, height=1280, focal=2000.0):
    """Waymo-like pinhole rays (formulas of camera_utils.py:482-557,
    datasets.py:446): random pixels of one camera at a random pose near the origin."""
    g = torch.Generator().manual_seed(seed)
    px = torch.rand(n, generator=g) * width
    py = torch.rand(n, generator=g) * height
    def cam_ray(x, y):
        return torch.stack([(x - width / 2) / focal, -(y - height / 2) / focal, -torch.ones_like(x)], -1)
    yaw = 0.3
    R = torch.tensor([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]], dtype=torch.float32)
    d = cam_ray(px, py) @ R.T
    dx = cam_ray(px + 1, py) @ R.T
    dy = cam_ray(px, py + 1) @ R.T
    v = d / d.norm(dim=-1, keepdim=True)
    vx = dx / dx.norm(dim=-1, keepdim=True)
    vy = dy / dy.norm(dim=-1, keepdim=True)
    radii = (0.5 * ((v - vx).norm(dim=-1) + (v - vy).norm(dim=-1)))[:, None] * 2 / np.sqrt(12)
    o = torch.tensor([0.1, -0.05, 0.2]).expand(n, 3).contiguous()
    cam_dir = (-R[:, 2]).expand(n, 3).contiguous()
    return dict(origins=o, directions=d.contiguous(), viewdirs=v.contiguous(), cam_dirs=cam_dir,
                radii=radii.float().contiguous(), near=torch.full((n, 1), near), far=torch.full((n, 1), far),
                cam_idx=torch.zeros(n, 1, dtype=torch.long), lossmult=torch.ones(n, 1))
