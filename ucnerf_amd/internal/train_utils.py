"""Training-only reductions over the ray-march outputs, torch device ops.

Interface mirror of the loss functions the reference's train.py:173-216 calls from
/root/reference/nerf/internal/train_utils.py (:149-305) -- same names, arguments and `config` fields --
so the reference's training loop can take them from here.  They are O(N*S) elementwise / scan work on
tensors already resident in HBM; formulations differ where the reference is quadratic:

* distortion loss: the reference builds the [N,S,S] |u_i-u_j| matrix (stepfun.py:297-307); midpoints of
  sorted fenceposts are sorted, so sum_ij w_i w_j |u_i-u_j| = 2 sum_i w_i (u_i W_i - M_i) with exclusive
  prefix sums W, M  -> O(S) per ray, same value.
* anti-interlevel loss: `sorted_interp_quad` (math.py:110-133) uses O(n*m) masks; here a binary search.
"""
import collections

import numpy as np
import torch

from .train_graph import GradientScaler  # noqa: F401  (ref train_utils.py:101-111)


# ------------------------------------------------------------------ step-function helpers
def lossfun_distortion(t, w):
    """ref stepfun.py:297-307, O(S)."""
    u = (t[..., 1:] + t[..., :-1]) / 2
    zero = torch.zeros_like(w[..., :1])
    W = torch.cat([zero, torch.cumsum(w[..., :-1], dim=-1)], dim=-1)
    M = torch.cat([zero, torch.cumsum((w * u)[..., :-1], dim=-1)], dim=-1)
    inter = 2 * torch.sum(w * (u * W - M), dim=-1)
    intra = torch.sum(w ** 2 * (t[..., 1:] - t[..., :-1]), dim=-1) / 3
    return inter + intra


def blur_stepfun(x, y, r):
    """ref stepfun.py:395-403: convolve the step function (x, y) with a box of half-width r; returns
    the knots and values of the resulting piecewise-linear function."""
    xr, order = torch.sort(torch.cat([x - r, x + r], dim=-1))
    pad = torch.zeros_like(y[..., :1])
    slope = (torch.cat([y, pad], dim=-1) - torch.cat([pad, y], dim=-1)) / (2 * r)
    dslope = torch.cat([slope, -slope], dim=-1).take_along_dim(order[..., :-1], dim=-1)
    yr = torch.cumsum((xr[..., 1:] - xr[..., :-1]) * torch.cumsum(dslope, dim=-1), dim=-1).clamp_min(0)
    return xr, torch.cat([torch.zeros_like(yr[..., :1]), yr], dim=-1)


def interp_quad(x, xp, fpdf, fcdf):
    """ref math.py:110-133 `sorted_interp_quad`: integrate a piecewise-linear pdf up to x."""
    n = xp.shape[-1]
    cnt = torch.searchsorted(xp.contiguous(), x.contiguous(), right=True)
    i0 = (cnt - 1).clamp_min(0)
    i1 = cnt.clamp_max(n - 1)
    xp0, xp1 = xp.gather(-1, i0), xp.gather(-1, i1)
    p0, p1 = fpdf.gather(-1, i0), fpdf.gather(-1, i1)
    c0 = fcdf.gather(-1, i0)
    off = torch.nan_to_num((x - xp0) / (xp1 - xp0), 0).clamp(0, 1)
    return c0 + (x - xp0) * (p0 + p1 * off + p0 * (1 - off)) / 2


# ------------------------------------------------------------------ losses (train.py:173-216)
def compute_data_loss(batch, renderings, config):
    """ref train_utils.py:171-230 ('mse' and 'charb')."""
    stats = collections.defaultdict(list)
    target = batch['rgb'][..., :3]
    lossmult = torch.broadcast_to(batch['lossmult'], target.shape)
    if getattr(config, 'disable_multiscale_loss', False):
        lossmult = torch.ones_like(lossmult)
    denom = lossmult.sum()
    per_level = []
    for r in renderings:
        resid_sq = (r['rgb'] - target) ** 2
        stats['mses'].append(((lossmult * resid_sq).sum() / denom).item())
        kind = getattr(config, 'data_loss_type', 'charb')
        if kind == 'mse':
            term = resid_sq
        elif kind == 'charb':
            term = torch.sqrt(resid_sq + getattr(config, 'charb_padding', 0.001) ** 2)
        else:
            raise NotImplementedError(f"data_loss_type={kind!r}")
        per_level.append((lossmult * term).sum() / denom)
    loss = (getattr(config, 'data_coarse_loss_mult', 0.) * sum(per_level[:-1])
            + getattr(config, 'data_loss_mult', 1.0) * per_level[-1])
    return loss, {k: np.array(v) for k, v in stats.items()}


def anti_interlevel_loss(ray_history, config):
    """ref train_utils.py:247-270."""
    c = ray_history[-1]['sdist'].detach()
    w = ray_history[-1]['weights'].detach()
    pdf = w / (c[..., 1:] - c[..., :-1])
    total = 0.
    widths = getattr(config, 'pulse_width', [0.03, 0.003])
    for i, level in enumerate(ray_history[:-1]):
        cp, wp = level['sdist'], level['weights']
        knots, vals = blur_stepfun(c, pdf, widths[i])
        area = 0.5 * (vals[..., 1:] + vals[..., :-1]) * (knots[..., 1:] - knots[..., :-1])
        cdf = torch.cat([torch.zeros_like(area[..., :1]), torch.cumsum(area, dim=-1)], dim=-1)
        w_s = torch.diff(interp_quad(cp, knots, vals, cdf), dim=-1)
        total = total + ((w_s - wp).clamp_min(0) ** 2 / (wp + 1e-5)).mean()
    return getattr(config, 'anti_interlevel_loss_mult', 0.01) * total


def distortion_loss(ray_history, config):
    """ref train_utils.py:273-279."""
    last = ray_history[-1]
    return getattr(config, 'distortion_loss_mult', 0.005) * lossfun_distortion(last['sdist'], last['weights']).mean()


def hash_decay_loss(ray_history, config):
    """ref train_utils.py:301-305."""
    return sum(getattr(config, 'hash_decay_mults', 0.1) * h['loss_hash_decay'] for h in ray_history)


def sky_loss(batch, renderings):
    """ref train_utils.py:149-157."""
    loss = 0
    target = 1 - batch['sky_segs']
    for r in renderings:
        acc = torch.sum(r['weights'], dim=-1)
        loss = loss + torch.nn.functional.binary_cross_entropy(acc.clip(1e-3, 1.0 - 1e-3), target)
    return loss


def transformIdentityLoss(renderings):
    """ref train_utils.py:159-169."""
    A = renderings[0]['affine_trans']
    eye = torch.eye(4, dtype=torch.float64, device=A.device)[:3].unsqueeze(0).expand(A.shape[0], 3, 4)
    loss = torch.abs(eye - A)
    if 'affine_trans_sky' in renderings[0]:
        loss = loss + torch.abs(eye - renderings[0]['affine_trans_sky'])
    return loss.mean()
