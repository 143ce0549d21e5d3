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
import collections.abc
import ctypes

import numpy as np
import torch

from .. import _lib
from .train_graph import GradientScaler  # noqa: F401  (ref train_utils.py:101-111)


# ------------------------------------------------------------------ step-function helpers
class _Distortion(torch.autograd.Function):
    """lossfun_distortion per ray as two HIP launches (`ucn_distortion_loss`): forward [N], backward d/dw [N, S]."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, t, w):
        lib = _lib.load()
        S = w.shape[-1]
        t2, w2 = t.reshape(-1, S + 1).contiguous(), w.reshape(-1, S).contiguous()
        out = torch.empty(w2.shape[0], device=w.device)
        _lib.check(lib.ucn_distortion_loss(t2.data_ptr(), w2.data_ptr(), w2.shape[0], S, None, out.data_ptr(), _lib.stream()))
        ctx.save_for_backward(t2, w2)
        ctx.shape = w.shape
        return out.reshape(w.shape[:-1])

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        lib = _lib.load()
        t2, w2 = ctx.saved_tensors
        g = g.reshape(-1).float().contiguous()
        gw = torch.empty_like(w2)
        _lib.check(lib.ucn_distortion_loss(t2.data_ptr(), w2.data_ptr(), w2.shape[0], w2.shape[1], g.data_ptr(), gw.data_ptr(),
                                           _lib.stream()))
        return None, gw.reshape(ctx.shape)


def lossfun_distortion(t, w):
    """ref stepfun.py:297-307, O(S).  Device tensors whose fenceposts carry no gradient (the training graph: sample
    positions are detached, models.py:204) take the HIP kernel; the torch form below is the general one."""
    if w.is_cuda and not t.requires_grad and w.shape[-1] <= 512:
        return _Distortion.apply(t, w)
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


class _InterLevel(torch.autograd.Function):
    """One proposal level of anti_interlevel_loss as the HIP kernel `ucn_interlevel_loss`: mean over rays and proposal
    intervals of max(w_s - wp, 0)^2 / (wp + 1e-5); the NeRF level (c, w) and the fenceposts cp are constants."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, c, w, cp, wp, r):
        lib = _lib.load()
        S1, Sp = w.shape[-1], wp.shape[-1]
        c2, w2 = c.reshape(-1, S1 + 1).contiguous(), w.reshape(-1, S1).contiguous()
        cp2, wp2 = cp.reshape(-1, Sp + 1).contiguous(), wp.reshape(-1, Sp).contiguous()
        N = w2.shape[0]
        loss_ray = torch.empty(N, device=w.device)
        dterm = torch.empty(N, Sp, device=w.device)
        _lib.check(lib.ucn_interlevel_loss(c2.data_ptr(), w2.data_ptr(), S1, cp2.data_ptr(), wp2.data_ptr(), Sp, float(r), N,
                                           loss_ray.data_ptr(), dterm.data_ptr(), _lib.stream()))
        ctx.save_for_backward(dterm)
        ctx.shape = wp.shape
        return loss_ray.sum() / (N * Sp)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        (dterm,) = ctx.saved_tensors
        return None, None, None, (dterm * (g / dterm.numel())).reshape(ctx.shape), None


# ------------------------------------------------------------------ losses (train.py:173-216)
class LazyStats(collections.abc.MutableMapping):
    """The stats dict of compute_data_loss with its device values fetched on first READ.  The reference converts every
    level's mse with `.item()` inside the loss function (train_utils.py:187) -- a host sync in the middle of the step,
    before the remaining losses and the whole backward are even queued; its loop first reads the stats after the
    optimiser step (train.py:226).  Same keys, same numpy values, one transfer at that point instead.  A mutable mapping
    (train.py adds 'loss', 'psnrs', ... to it); `dict(stats)`, iteration and `.items()` all go through the fetch."""

    def __init__(self, **device_values):
        self._d = dict(device_values)

    def __getitem__(self, k):
        v = self._d[k]
        if isinstance(v, torch.Tensor):                # host tensors too: the reference's stats are numpy on any device
            v = self._d[k] = v.detach().float().cpu().numpy()
        return v

    def __setitem__(self, k, v):
        self._d[k] = v

    def __delitem__(self, k):
        del self._d[k]

    def __iter__(self):
        return iter(self._d)

    def __len__(self):
        return len(self._d)

    def pending(self, k):
        """True while the value of `k` has not been fetched from the device."""
        v = self._d[k]
        return isinstance(v, torch.Tensor) and v.is_cuda

    def __repr__(self):
        return repr(dict(self.items()))


class _DataLoss(torch.autograd.Function):
    """train_utils.py:171-230 for all levels in ONE launch forward + one backward (`ucn_data_loss`): the weighted loss and the
    per-level mse statistics; ~10 + ~15 eager launches otherwise."""

    @staticmethod
    def forward(ctx, target, mult, pad, w_mse, w_charb, *levels):
        lib = _lib.load()
        N = target.shape[0]
        L = len(levels)
        levels = [l.contiguous() for l in levels]
        out = torch.empty(2 * L + 2, device=target.device)
        wm, wc = (ctypes.c_float * L)(*w_mse), (ctypes.c_float * L)(*w_charb)
        ptrs = (ctypes.c_void_p * L)(*[l.data_ptr() for l in levels])
        _lib.check(lib.ucn_data_loss(ptrs, L, wm, wc, target.data_ptr(), _lib.ptr(mult), N, float(pad), out.data_ptr(), None, None, _lib.stream()))
        ctx.save_for_backward(target, out, *levels)
        ctx.mult, ctx.pad, ctx.w = mult, float(pad), (list(w_mse), list(w_charb))
        ctx.mark_non_differentiable(out)
        return out[2 * L + 1], out

    @staticmethod
    def backward(ctx, g_loss, _g_out):
        lib = _lib.load()
        target, out, *levels = ctx.saved_tensors
        N, L = target.shape[0], len(levels)
        g = g_loss.reshape(1).float().contiguous()
        grads = [torch.empty_like(l) for l in levels]
        wm, wc = (ctypes.c_float * L)(*ctx.w[0]), (ctypes.c_float * L)(*ctx.w[1])
        ptrs = (ctypes.c_void_p * L)(*[l.data_ptr() for l in levels])
        gptrs = (ctypes.c_void_p * L)(*[x.data_ptr() for x in grads])
        _lib.check(lib.ucn_data_loss(ptrs, L, wm, wc, target.data_ptr(), _lib.ptr(ctx.mult), N, ctx.pad, out.data_ptr(), g.data_ptr(), gptrs,
                                     _lib.stream()))
        return (None, None, None, None, None, *grads)


class _SkyLoss(torch.autograd.Function):
    """train_utils.py:149-157 over all levels: one launch forward, one backward (`ucn_sky_loss`)."""

    @staticmethod
    def forward(ctx, sky_segs, *accs):
        lib = _lib.load()
        accs = [a.contiguous() for a in accs]
        N, L = accs[0].shape[0], len(accs)
        out = torch.empty(1, device=sky_segs.device)
        ptrs = (ctypes.c_void_p * L)(*[a.data_ptr() for a in accs])
        _lib.check(lib.ucn_sky_loss(ptrs, L, sky_segs.data_ptr(), N, out.data_ptr(), None, None, _lib.stream()))
        ctx.save_for_backward(sky_segs, *accs)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        sky_segs, *accs = ctx.saved_tensors
        N, L = accs[0].shape[0], len(accs)
        g = g.reshape(1).float().contiguous()
        grads = [torch.empty_like(a) for a in accs]
        ptrs = (ctypes.c_void_p * L)(*[a.data_ptr() for a in accs])
        gptrs = (ctypes.c_void_p * L)(*[x.data_ptr() for x in grads])
        _lib.check(lib.ucn_sky_loss(ptrs, L, sky_segs.data_ptr(), N, None, g.data_ptr(), gptrs, _lib.stream()))
        return (None, *grads)


class _IdentityLoss(torch.autograd.Function):
    """train_utils.py:159-169: mean |eye - A| (+ |eye - A_sky|) in float64 like the reference: one launch each way (`ucn_identity_loss`)."""

    @staticmethod
    def forward(ctx, A, A_sky):
        lib = _lib.load()
        A = A.contiguous()
        A_sky = A_sky.contiguous() if A_sky is not None else None
        N = A.numel() // 12
        out = torch.empty(1, device=A.device, dtype=torch.float64)
        _lib.check(lib.ucn_identity_loss(A.data_ptr(), _lib.ptr(A_sky), N, out.data_ptr(), None, None, None, _lib.stream()))
        ctx.save_for_backward(A, *([A_sky] if A_sky is not None else []))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        A, *rest = ctx.saved_tensors
        A_sky = rest[0] if rest else None
        N = A.numel() // 12
        g = g.reshape(1).double().contiguous()
        gA = torch.empty_like(A)
        gB = torch.empty_like(A_sky) if A_sky is not None else None
        _lib.check(lib.ucn_identity_loss(A.data_ptr(), _lib.ptr(A_sky), N, None, g.data_ptr(), gA.data_ptr(), _lib.ptr(gB), _lib.stream()))
        return gA, gB


def _f32_cuda(*tensors, half_ok=False):
    ok = (torch.float32, torch.bfloat16, torch.float16) if half_ok else (torch.float32,)
    return all(t.is_cuda and t.dtype in ok for t in tensors)


def compute_data_loss(batch, renderings, config):
    """ref train_utils.py:171-230 ('mse' and 'charb').  All levels in one set of elementwise / reduce launches (the
    levels' rgb stacked), the per-level mse statistics handed back as LazyStats."""
    target = batch['rgb'][..., :3]
    kind = getattr(config, 'data_loss_type', 'charb')
    if kind not in ('mse', 'charb'):
        raise NotImplementedError(f"data_loss_type={kind!r}")
    n_lvl = len(renderings)
    coarse = getattr(config, 'data_coarse_loss_mult', 0.)
    levels = [r['rgb'] for r in renderings]
    if (n_lvl <= 4 and target.shape[-1] == 3 and _f32_cuda(target, *levels) and all(l.numel() == target.numel() for l in levels)
            and batch['lossmult'].numel() * 3 == target.numel()):
        # the whole function as ONE HIP node (csrc/heads_train.hip): weighted loss + the mse statistics
        N = target.numel() // 3
        mult = None if getattr(config, 'disable_multiscale_loss', False) else batch['lossmult'].reshape(N).float().contiguous()
        w_level = [coarse if (coarse != 0 and n_lvl > 1) else 0.0] * (n_lvl - 1) + [getattr(config, 'data_loss_mult', 1.0)]
        w_mse = w_level if kind == 'mse' else [0.0] * n_lvl
        w_charb = [0.0] * n_lvl if kind == 'mse' else w_level
        loss, out = _DataLoss.apply(target.reshape(N, 3).contiguous(), mult, getattr(config, 'charb_padding', 0.001), w_mse, w_charb,
                                    *[l.reshape(N, 3) for l in levels])
        return loss, LazyStats(mses=out[0:2 * n_lvl:2])
    lossmult = torch.broadcast_to(batch['lossmult'], target.shape)
    if getattr(config, 'disable_multiscale_loss', False):
        lossmult = torch.ones_like(lossmult)
    denom = lossmult.sum()
    dims = tuple(range(1, target.dim() + 1))
    resid_sq = (torch.stack([r['rgb'] for r in renderings]) - target) ** 2                # [levels, ...]
    mses = (lossmult * resid_sq).sum(dim=dims) / denom
    if kind == 'mse':
        per_level = mses
    else:
        per_level = (lossmult * torch.sqrt(resid_sq + getattr(config, 'charb_padding', 0.001) ** 2)).sum(dim=dims) / denom
    loss = getattr(config, 'data_loss_mult', 1.0) * per_level[-1]
    if coarse != 0 and n_lvl > 1:
        loss = loss + coarse * per_level[:-1].sum()
    return loss, LazyStats(mses=mses.detach())


def anti_interlevel_loss(ray_history, config):
    """ref train_utils.py:247-270."""
    c = ray_history[-1]['sdist'].detach()
    w = ray_history[-1]['weights'].detach()
    pdf = w / (c[..., 1:] - c[..., :-1])
    total = 0.
    widths = getattr(config, 'pulse_width', [0.03, 0.003])
    for i, level in enumerate(ray_history[:-1]):
        cp, wp = level['sdist'], level['weights']
        if wp.is_cuda and not cp.requires_grad and w.shape[-1] <= 512:
            total = total + _InterLevel.apply(c, w, cp, wp, widths[i])     # the same arithmetic as one HIP launch
            continue
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
    if len(renderings) <= 4 and all('acc' in r for r in renderings) and _f32_cuda(batch['sky_segs'], *[r['acc'] for r in renderings]):
        # one HIP node for all levels (csrc/heads_train.hip); `acc` IS the sum of the level's weights (render.py:199)
        N = batch['sky_segs'].numel()
        return _SkyLoss.apply(batch['sky_segs'].reshape(N).contiguous(), *[r['acc'].reshape(N) for r in renderings])
    loss = 0
    target = 1 - batch['sky_segs']
    for r in renderings:
        acc = torch.sum(r['weights'], dim=-1)
        loss = loss + torch.nn.functional.binary_cross_entropy(acc.clip(1e-3, 1.0 - 1e-3), target)
    return loss


def transformIdentityLoss(renderings):
    """ref train_utils.py:159-169."""
    A = renderings[0]['affine_trans']
    A_sky = renderings[0].get('affine_trans_sky')
    maps = [A] + ([A_sky] if A_sky is not None else [])
    if _f32_cuda(*maps, half_ok=True) and A.shape[-2:] == (3, 4):
        # bf16 maps (the heads ran under train.py:165's autocast) are upcast first: exact, and what `eye (float64) - A` promotes through
        return _IdentityLoss.apply(A.float(), A_sky.float() if A_sky is not None else None)     # one HIP node (csrc/heads_train.hip)
    eye = torch.eye(4, dtype=torch.float64, device=A.device)[:3].unsqueeze(0).expand(A.shape[0], 3, 4)
    loss = torch.abs(eye - A)
    if 'affine_trans_sky' in renderings[0]:
        loss = loss + torch.abs(eye - renderings[0]['affine_trans_sky'])
    return loss.mean()


# ------------------------------------------------------------------ optimiser (train_utils.py:335-366)
class FusedAdam(torch.optim.Adam):
    """torch.optim.Adam -- what `create_optimizer` builds (train_utils.py:347-366) -- whose fp32 device tensors are stepped
    by this repo's kernels: the hash tables (7.1 M x 2 and 1.9 M x 2 parameters in config B) by ONE pass of
    `ucn_adam_step` each instead of ~12 elementwise passes, the small dense-layer parameters of a group together by ONE
    launch of `ucn_adam_step_many` instead of torch's seven foreach launches; the `grad.nan_to_num_()` of `clip_gradients`
    is folded in.  State keys and layout are torch's (`step`, `exp_avg`, `exp_avg_sq`), so state_dicts interchange with
    torch.optim.Adam.  Everything else (other dtypes / devices, groups using amsgrad / weight_decay / maximize) goes
    through the parent class unchanged."""
    MIN_NUMEL = 1 << 20                                          # from here on a tensor gets its own launch

    def _fusable(self, group, p):
        return (p.grad is not None and p.is_cuda and p.dtype == torch.float32 and p.grad.dtype == torch.float32
                and p.is_contiguous() and p.grad.is_contiguous() and not p.grad.is_sparse and not group.get('amsgrad', False)
                and group.get('weight_decay', 0) == 0 and not group.get('maximize', False)
                and not group.get('capturable', False) and not group.get('differentiable', False)
                and not group.get('fused', False) and p.numel() < (1 << 31) * (4 if p.numel() >= self.MIN_NUMEL else 1))

    @torch.no_grad()
    def step(self, closure=None):
        import ctypes
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        held = []
        for group in self.param_groups:
            for p in group['params']:
                if self._fusable(group, p):
                    held.append((group, p, p.grad))
                    p.grad = None                                # the parent skips parameters without a gradient
        try:
            super().step()
        finally:
            for _, p, g in held:
                p.grad = g
        batches = {}                                             # (group, step count) -> small tensors stepped together
        for group, p, g in held:
            state = self.state[p]
            if len(state) == 0:                                  # torch/optim/adam.py _init_group
                state['step'] = torch.tensor(0.0, dtype=torch.float32)
                state['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            state['step'] += 1
            step = int(state['step'].item())
            beta1, beta2 = group['betas']
            if p.numel() >= self.MIN_NUMEL and all(t.data_ptr() % 16 == 0 for t in (p, g, state['exp_avg'], state['exp_avg_sq'])):
                _lib.check(lib.ucn_adam_step(p.data_ptr(), g.data_ptr(), state['exp_avg'].data_ptr(), state['exp_avg_sq'].data_ptr(),
                                             p.numel(), float(group['lr']), float(beta1), float(beta2), float(group['eps']),
                                             step, 1, _lib.stream()))
            else:
                batches.setdefault((id(group), step), (group, []))[1].append((p, g, state))
        for (_, step), (group, items) in batches.items():
            n = len(items)
            arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
            beta1, beta2 = group['betas']
            _lib.check(lib.ucn_adam_step_many(arr([p for p, _, _ in items]), arr([g for _, g, _ in items]),
                                              arr([s['exp_avg'] for _, _, s in items]), arr([s['exp_avg_sq'] for _, _, s in items]),
                                              (ctypes.c_uint64 * n)(*[p.numel() for p, _, _ in items]), n, float(group['lr']),
                                              float(beta1), float(beta2), float(group['eps']), step, 1, _lib.stream()))
        # the kernels wrote through raw pointers: tell autograd's version counters, which key the inference caches of the packed MLP
        # weights and of the half-precision table copies (models.py `field()`, sky.py) -- a render between two training steps
        # (train.py:330's periodic test render) must not reuse the operands packed before the step
        for _, p, _ in held:
            torch.autograd.graph.increment_version(p)
        return loss


class ShardedFusedAdam(FusedAdam):
    """FusedAdam for `dist.wrap_ddp(model, grad_exchange="reduce_scatter")`: every parameter
    marked `_ucn_sharded` (the hash tables) is represented in the optimiser by THIS rank's 1 / N of its elements -- a
    Parameter that is a view of the table's own storage, so the update lands in place.  step():
      1. per table one `reduce_scatter_tensor(AVG)` of the local full-size gradient -> the mean gradient of this rank's rows
         (DDP's all-reduce = reduce-scatter + all-gather of GRADIENTS; the second half is not needed before the step);
      2. nan_to_num on the reduced shard (what clip_gradients does to the reduced gradient in the reference,
         train_utils.py:342-344) and the parent's step: `ucn_adam_step` over 1 / N of the rows, moments held for those only;
      3. per table one `all_gather_into_tensor` of the updated rows into the table.
    Every rank ends the step with identical tables (the same bits: each element is computed on exactly one rank).  With two
    ranks the result is bit-identical to the all-reduce route (a + b is commutative); with more, equal up to the
    summation order of the collective.  zero_grad() also clears the full-size table gradients, which are not in
    param_groups.  `state_dict()` (= `gathered_state_dict()`, collective) is torch.optim.Adam-compatible over the FULL parameters
    (moments all-gathered) and `load_state_dict()` slices such a state back to this rank's rows: checkpoints move between the
    sharded and the unsharded optimiser and between world sizes (r06, ADVICE r05: the inherited rank-local state_dict loaded
    rank 0's rows into every rank without an error)."""

    def __init__(self, params, process_group=None, **kw):
        import torch.distributed as dist
        params = list(params)
        if params and isinstance(params[0], dict):
            raise ValueError("ShardedFusedAdam takes a flat parameter list (create_optimizer's model.parameters())")
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("ShardedFusedAdam needs an initialised torch.distributed process group")
        self._group = process_group
        self._world, self._rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        self._tables = []                                      # (full parameter, shard parameter)
        mine = []
        for p in params:
            if getattr(p, "_ucn_sharded", False):
                n = p.numel() // self._world
                assert p.numel() % self._world == 0 and p.is_contiguous()
                shard = torch.nn.Parameter(p.data.view(-1)[self._rank * n:(self._rank + 1) * n], requires_grad=True)
                self._tables.append((p, shard))
                mine.append(shard)
            else:
                mine.append(p)
        super().__init__(mine, **kw)
        self.exchange_bytes = dict(mode="reduce_scatter", reduce_scatter_in=sum(p.numel() * 4 for p, _ in self._tables),
                                   all_gather_out=sum(p.numel() * 4 for p, _ in self._tables), tables=len(self._tables))

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none=set_to_none)
        for p, _ in self._tables:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        import torch.distributed as dist
        if closure is not None:
            raise NotImplementedError("ShardedFusedAdam.step(closure): re-evaluating the model between the exchange steps is not supported")
        live = []
        for p, shard in self._tables:
            # a table without a gradient on THIS rank (none of its rays touched the field) still takes part in the collectives -- the
            # other ranks enter them -- with a zero contribution (skipping would hang the job)
            g = torch.zeros(p.numel(), device=p.device, dtype=p.dtype) if p.grad is None else p.grad.contiguous().view(-1)
            out = torch.empty(shard.numel(), device=g.device, dtype=g.dtype)
            dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.AVG, group=self._group)
            if not (out.is_cuda and out.numel() >= self.MIN_NUMEL):
                out.nan_to_num_()                              # (the table kernel sanitises its gradient itself)
            shard.grad = out
            live.append((p, shard))
        loss = super().step()
        for p, shard in live:
            # out-of-place send buffer: the shard is a slice of the receive buffer (see _exchange in dist.py for the in-place form)
            dist.all_gather_into_tensor(p.data.view(-1), shard.data.clone(), group=self._group)
            torch.autograd.graph.increment_version(p)           # (p.data is a detached alias: its writes do not count for p)
            shard.grad = None
        return loss

    def state_dict(self):
        """torch.optim.Adam's layout over the FULL parameters: the moments of the sharded tables all-gathered.  COLLECTIVE -- every
        rank must call it, which is what the reference's checkpoint flow does (checkpoints.py:24 -> accelerator.save_state calls
        optimizer.state_dict() on every process and writes on the main one).  The rank-local form (1 / N of each table's moments:
        shapes that load without an error into the wrong rows of another rank) is `local_state_dict()`."""
        return self.gathered_state_dict()

    def local_state_dict(self):
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """Accepts the full-size state `state_dict()` returns (or one saved by FusedAdam / torch.optim.Adam over model.parameters()):
        every rank keeps rows [rank n, (rank + 1) n) of each sharded table's moments, whatever world size wrote the checkpoint."""
        import copy
        sd = {"state": dict(state_dict["state"]), "param_groups": copy.deepcopy(state_dict["param_groups"])}
        order = [p for g in self.param_groups for p in g['params']]
        shard_ids = {id(shard): p for p, shard in self._tables}
        for i, q in enumerate(order):
            if id(q) in shard_ids and i in sd["state"]:
                full, n = shard_ids[id(q)].numel(), q.numel()
                st = dict(sd["state"][i])
                for k in ('exp_avg', 'exp_avg_sq'):
                    v = st[k]
                    if v.numel() != full:
                        raise ValueError(f"ShardedFusedAdam.load_state_dict: parameter {i} carries {v.numel()} moment elements, expected the "
                                         f"full table's {full} (a rank-local state of another run? save with state_dict(), not local_state_dict())")
                    st[k] = v.reshape(-1)[self._rank * n:(self._rank + 1) * n].clone().view_as(q)
                sd["state"][i] = st
        super().load_state_dict(sd)

    def gathered_state_dict(self):
        """The optimiser state in torch.optim.Adam's layout over the FULL parameters (moments all-gathered): loads into
        FusedAdam / torch.optim.Adam built on model.parameters().  Collective: every rank must call it."""
        import torch.distributed as dist
        sd = super().state_dict()
        order = [p for g in self.param_groups for p in g['params']]
        full_of = {id(shard): p for p, shard in self._tables}
        sd = dict(sd, state=dict(sd['state']))                 # (torch hands out the optimiser's OWN per-parameter dicts: copy before editing)
        for i, q in enumerate(order):
            if id(q) in full_of and i in sd['state']:
                p = full_of[id(q)]
                st = dict(sd['state'][i])
                for k in ('exp_avg', 'exp_avg_sq'):
                    full = torch.empty(p.numel(), device=p.device, dtype=p.dtype)
                    dist.all_gather_into_tensor(full, st[k].contiguous().view(-1), group=self._group)
                    st[k] = full.view_as(p)
                sd['state'][i] = st
        return sd


def clip_gradients(model, accelerator, config):
    """ref train_utils.py:335-344: norm / value clipping, then nan_to_num on every gradient.  (The tables stepped by
    FusedAdam are sanitised again inside its kernel -- idempotent.)"""
    if (getattr(config, 'grad_max_norm', 0) > 0 or getattr(config, 'grad_max_val', 0) > 0) and \
            any(getattr(p, "_ucn_sharded", False) for p in model.parameters()):
        raise NotImplementedError("gradient clipping acts on the REDUCED gradient; the reduce-scatter exchange reduces the table "
                                  "gradients inside the optimiser step -- use dist.wrap_ddp(grad_exchange='all_reduce') with clipping")
    if getattr(config, 'grad_max_norm', 0) > 0 and accelerator.sync_gradients:
        accelerator.clip_grad_norm_(model.parameters(), config.grad_max_norm)
    if getattr(config, 'grad_max_val', 0) > 0 and accelerator.sync_gradients:
        accelerator.clip_grad_value_(model.parameters(), config.grad_max_val)
    sanitize_gradients(model.parameters())


def sanitize_gradients(params):
    """`param.grad.nan_to_num_()` for every parameter (train_utils.py:342-344): the contiguous fp32 device gradients in ONE
    launch (`ucn_nan_to_num_many`, a read-only pass unless something is not finite) instead of one per parameter.
    Anything else takes torch's op."""
    import ctypes
    small = []
    for p in params:
        g = p.grad
        if g is None or getattr(p, "_ucn_sharded", False):        # sharded tables: sanitised AFTER their reduce-scatter (ShardedFusedAdam)
            continue
        if g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and not g.is_sparse and g.numel() < (1 << 31):
            small.append(g)
        else:
            g.nan_to_num_()
    if small:
        lib = _lib.load()
        ptrs = (ctypes.c_void_p * len(small))(*[g.data_ptr() for g in small])
        cnts = (ctypes.c_uint64 * len(small))(*[g.numel() for g in small])
        _lib.check(lib.ucn_nan_to_num_many(ptrs, cnts, len(small), _lib.stream()))


def learning_rate_decay(step, lr_init, lr_final, max_steps, lr_delay_steps=0, lr_delay_mult=1):
    """ref math.py:53-85: log-linear interpolation with an optional sine warm-up."""
    if lr_delay_steps > 0:
        delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
    else:
        delay_rate = 1.
    if lr_init <= 0 or lr_final <= 0:                                         # math.py:44-50 log_lerp
        raise ValueError(f'Interpolants {lr_init} and {lr_final} must be positive.')
    lv0, lv1 = np.log(lr_init), np.log(lr_final)
    return delay_rate * np.exp(np.clip(step / max_steps, 0, 1) * (lv1 - lv0) + lv0)


def create_optimizer(config, model):
    """ref train_utils.py:347-366: Adam over all parameters + the learning-rate schedule."""
    lr_fn_main = lambda step: learning_rate_decay(step, lr_init=config.lr_init, lr_final=config.lr_final,
                                                  max_steps=config.max_steps, lr_delay_steps=config.lr_delay_steps,
                                                  lr_delay_mult=config.lr_delay_mult)
    params = list(model.parameters())
    # tables taken out of DDP by dist.wrap_ddp(model, grad_exchange="reduce_scatter") need the optimiser that exchanges them
    cls = ShardedFusedAdam if any(getattr(p, "_ucn_sharded", False) for p in params) else FusedAdam
    optimizer = cls(params, lr=config.lr_init, betas=[config.adam_beta1, config.adam_beta2], eps=config.adam_eps)
    return optimizer, lr_fn_main


# ------------------------------------------------------------------ virtual-pose depth warp (train_utils.py:19-98)
def _warp_inputs(ref_pose, src_pose, depth, intrinsic):
    import ctypes
    as_t = lambda a: a if torch.is_tensor(a) else torch.from_numpy(np.asarray(a))
    d = as_t(depth).to(torch.float32)
    if not d.is_cuda:
        d = d.cuda()                                               # the depth map is the only O(H*W) input
    d = d.contiguous()
    ref, src = as_t(ref_pose).cpu().to(torch.float32), as_t(src_pose).cpu().to(torch.float32)
    rel = (src.inverse() @ ref)[:3].contiguous()                   # :39, float32 on the host like the reference
    K = as_t(intrinsic).cpu().to(torch.float32).contiguous()
    return d, rel, K, ctypes


def img_warping(ref_pose, src_pose, virtual_pose_ref_depth, virtual_intrinsic, return_depth=False):
    """ref train_utils.py:19-55: where every pixel of the reference frame lands in the (virtual) source camera, and
    which pixels are usable (valid depth, inside the source frame).  Same arguments (numpy arrays or tensors); the
    O(H*W) work is the HIP kernel `ucn_img_warping`, results stay on the device: pts_in_tgt [H,W,2] float32, mask
    [H,W] bool."""
    lib = _lib.load()
    d, rel, K, ctypes = _warp_inputs(ref_pose, src_pose, virtual_pose_ref_depth, virtual_intrinsic)
    H, W = d.shape
    pts = torch.empty(H, W, 2, device=d.device)
    mask = torch.empty(H, W, dtype=torch.uint8, device=d.device)
    z = torch.empty(H, W, device=d.device) if return_depth else None
    _lib.check(lib.ucn_img_warping(d.data_ptr(), rel.data_ptr(), K.data_ptr(), H, W, pts.data_ptr(), mask.data_ptr(),
                                   _lib.ptr(z), _lib.stream()))
    return (pts, mask.bool(), z) if return_depth else (pts, mask.bool())


def img_warping_for_depth(ref_pose, src_pose, virtual_pose_ref_depth, virtual_intrinsic):
    """ref train_utils.py:58-98: the reference depth splatted into the source frame (later pixels win)."""
    lib = _lib.load()
    pts, mask, z = img_warping(ref_pose, src_pose, virtual_pose_ref_depth, virtual_intrinsic, return_depth=True)
    H, W = z.shape
    out = torch.empty(H, W, device=z.device)
    owner = torch.empty(H, W, dtype=torch.int32, device=z.device)
    m8 = mask.to(torch.uint8)
    _lib.check(lib.ucn_warp_scatter_depth(pts.data_ptr(), m8.data_ptr(), z.data_ptr(), H, W, owner.data_ptr(), out.data_ptr(),
                                          _lib.stream()))
    return out


def sample_virtual_pixels(pts_in_src, mask, num, generator=None):
    """datasets.py:531-545: `num` random valid reference pixels and the source pixels they warp to
    (torch.round = half-to-even, like the reference).  Returns int32 device tensors (ref_x, ref_y, src_x, src_y)."""
    valid = torch.nonzero(mask)                                    # [n_valid, 2] = (y, x), row-major like pixel_coords[mask]
    if valid.shape[0] == 0:
        raise RuntimeError("sample_virtual_pixels: the warp leaves no valid pixel (datasets.py:528 retries another pose)")
    pick = torch.randint(0, valid.shape[0], (num,), device=mask.device, generator=generator)
    ry, rx = valid[pick, 0], valid[pick, 1]
    src = torch.round(pts_in_src[ry, rx])
    return rx.int(), ry.int(), src[:, 0].int(), src[:, 1].int()
