"""Generate tests/golden/*.npz by running the REFERENCE's own Python on CPU.

    python tests/golden/make_golden.py          (authoring container only)

Every fixture stores the inputs (rays, random draws, seeds) and the outputs the reference
produced for them.  Model weights are not stored: they are regenerated from a seed by
oracle.raymarch.init_state (deterministic torch CPU generator) and guarded by a checksum.
The CUDA-only hash-grid kernel is replaced by oracle/grid_oracle.c (see ref_import.py);
all other arithmetic is the reference's unmodified code.

Fixture index:
  stepfun.npz       G1 sample_intervals (eval+train), G2 max_dilate_weights, weighted_percentile
  cast.npz          G3 cast_rays (eval+train, drawn rand_vec stored), G4 track_linearize
  field.npz         G5 MLP.forward / predict_density on a small grid (incl. int32-wrap levels)
  composite.npz     G6 compute_alpha_weights + volumetric_rendering (acc<0.6 sentinel, extras)
  model_tiny.npz    G7 Model.forward eval, BASELINE-like (64+128, L16 C2)  small tables
  model_tinyR.npz   G7 Model.forward eval, waymo.gin-like (128+32, L10 C4) small tables
  model_sky.npz     G8 + sky NeRF + brightness correction (eval_camidx)
  model_train.npz   G7' Model.forward with rand=True (all draws captured), train_frac<1
  model_tiny64.npz  G7 Model.forward eval, BASELINE configs[0] architecture (64+64, 64-wide colour MLP)  [`cfg1` mode]
  model_nodilate.npz G7 Model.forward eval with dilation_bias = dilation_multiplier = 0 (use_dilation False)  [`nodilate` mode]
  render_image.npz  G9 render_image on a 16x24 frame incl. a ragged last chunk
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import ref_import  # noqa: E402
from oracle import raymarch as rm  # noqa: E402


def state_checksum(sd):
    return float(sum(v.double().abs().sum() for k, v in sd.items() if v.is_floating_point()))


def npify(d, prefix=''):
    out = {}
    for k, v in d.items():
        if v is None:
            continue
        if isinstance(v, (list, tuple)):
            for i, vi in enumerate(v):
                out[f'{prefix}{k}.{i}'] = vi.detach().cpu().numpy()
        else:
            out[prefix + k] = torch.as_tensor(v).detach().cpu().numpy()
    return out


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays')


def gen_stepfun(ref):
    g = torch.Generator().manual_seed(11)
    N, n = 24, 64
    # a sorted fencepost set with clustered, tied and zero-width bins
    t = torch.sort(torch.rand(N, n + 1, generator=g), dim=-1).values
    t[:, 0] = 0
    t[:, -1] = 1
    t[3, 10:14] = t[3, 10:11]                     # zero-width intervals
    t[4] = torch.linspace(0, 1, n + 1) ** 3        # strongly clustered near 0
    w = torch.rand(N, n, generator=g) ** 4
    w[5, :20] = 0
    w = w / w.sum(-1, keepdim=True)
    out = dict(t=t, w=w)
    dil = 0.0025 + 0.5 / 64
    td, wd = ref.stepfun.max_dilate_weights(t, w, dil, domain=(0., 1.), renormalize=True)
    out.update(dilation=torch.tensor(dil), t_dilate=td, w_dilate=wd)
    sd_, wd_ = td[..., 1:-1], wd[..., 1:-1]
    for frac in (1.0, 0.25):
        anneal = 10 * frac / (9 * frac + 1)
        logits = torch.where(sd_[..., 1:] > sd_[..., :-1], anneal * torch.log(wd_),
                             torch.full_like(sd_[..., :-1], -torch.inf))
        out[f'logits_{frac}'] = logits
        out[f'sample_eval_{frac}'] = ref.stepfun.sample_intervals(False, sd_, logits, 128, single_jitter=True, domain=(0., 1.))
    torch.manual_seed(12)      # the reference draws its training jitter from torch's GLOBAL generator (stepfun.py:216): seeded, so that
    #                            this fixture re-generates bit for bit (r03's stored an unseeded draw)
    with ref_import.capture_rng() as cap:
        out['sample_train'] = ref.stepfun.sample_intervals(True, sd_, out['logits_0.25'], 32, single_jitter=True, domain=(0., 1.))
    out['sample_train_jitter'] = cap.draws[0][1]
    # level-0 case: one interval [0,1] of weight 1
    t0 = torch.tensor([[0., 1.]]).expand(N, 2).contiguous()
    out['sample_level0'] = ref.stepfun.sample_intervals(False, t0, torch.zeros(N, 1), 64, single_jitter=True, domain=(0., 1.))
    wa = torch.cat([w * 0.7, 1 - (w * 0.7).sum(-1, keepdim=True)], dim=-1)
    ta = torch.cat([t * 8, torch.full((N, 1), 8.5)], dim=-1)
    out['pct_t'] = ta
    out['pct_w'] = wa
    out['pct'] = ref.stepfun.weighted_percentile(ta, wa, [5, 50, 95])
    save('stepfun.npz', **npify(out))


def gen_cast(ref):
    rays = rm.synthetic_rays(20, seed=21)
    g = torch.Generator().manual_seed(22)
    S = 32
    s = torch.sort(torch.rand(20, S + 1, generator=g), dim=-1).values
    tdist = s * 8.0
    tdist[0, :3] = 0.0                           # degenerate leading intervals (near = 0)
    out = dict(tdist=tdist, **{k: rays[k] for k in ('origins', 'directions', 'cam_dirs', 'radii')})
    torch.manual_seed(23)
    with ref_import.capture_rng() as cap:
        m, sdev, t = ref.render.cast_rays(tdist, rays['origins'], rays['directions'], rays['cam_dirs'],
                                          rays['radii'], False, std_scale=0.5)
    out.update(eval_rand_vec=cap.draws[0][1], eval_means=m, eval_stds=sdev, eval_t=t)
    with ref_import.capture_rng() as cap:
        m2, s2, t2 = ref.render.cast_rays(tdist, rays['origins'], rays['directions'], rays['cam_dirs'],
                                          rays['radii'], True, std_scale=0.5)
    tags = [d[0] for d in cap.draws]
    assert tags == ['rand_like', 'rand_like', 'randn_like'], tags
    out.update(train_flip=cap.draws[0][1], train_spin=cap.draws[1][1], train_rand_vec=cap.draws[2][1],
               train_means=m2, train_stds=s2, train_t=t2)
    # contraction: points inside, on and far outside the unit ball, plus the origin
    pts = torch.randn(200, 3, generator=g) * torch.logspace(-2, 1.5, 200)[:, None]
    pts[0] = 0
    pts[1] = torch.tensor([1.0, 0, 0])
    sig = torch.rand(200, generator=g) * 0.1
    cm, cs = ref.coord.track_linearize('contract', pts, sig)
    out.update(contract_in_mean=pts, contract_in_std=sig, contract_mean=cm, contract_std=cs)
    save('cast.npz', **npify(out))


def gen_field(ref):
    spec = rm.make_spec('tiny')
    sd = rm.init_state(spec, seed=31)
    model, _ = ref_import.build_reference_model(ref, spec, sd)
    model.eval()
    g = torch.Generator().manual_seed(32)
    N, S = 6, 16
    # world-space multisample means spanning inside/outside the unit ball, tiny..large stds
    means = torch.randn(N, S, 6, 3, generator=g) * torch.logspace(-1, 1, S)[None, :, None, None]
    stds = torch.rand(N, S, 6, generator=g) * torch.logspace(-5, -1, S)[None, :, None]
    vd = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    out = dict(seed=torch.tensor(31), checksum=torch.tensor(state_checksum(sd), dtype=torch.float64),
               means=means, stds=stds, viewdirs=vd)
    with torch.no_grad():
        for name, mlp in (('nerf', model.nerf_mlp), ('prop', model.prop_mlp_0)):
            raw, x, coord_ = mlp.predict_density(means, stds)
            res = mlp(False, means, stds, viewdirs=vd)
            out.update({f'{name}_raw_density': raw, f'{name}_bottleneck': x, f'{name}_coord': res['coord'],
                        f'{name}_density': res['density'], f'{name}_rgb': res['rgb']})
            # the encoder + damping + mean-of-6 alone (input of density_layer)
            cm, cs = ref.coord.track_linearize('contract', means, stds)
            feat = mlp.encoder(cm / 2, bound=1).unflatten(-1, (mlp.encoder.num_levels, -1))
            wgt = torch.erf(1 / torch.sqrt(8 * (cs / 2)[..., None] ** 2 * mlp.encoder.grid_sizes ** 2))
            out[f'{name}_features'] = (feat * wgt[..., None]).mean(dim=-3).flatten(-2, -1)
        # no_warp API used by extract.py:56-57
        pm = torch.rand(50, 1, 3, generator=g) * 2 - 1
        ps = torch.full((50, 1), 1e-3)
        raw_nw, _, _ = model.nerf_mlp.predict_density(pm, ps, no_warp=True)
        out.update(nowarp_means=pm, nowarp_stds=ps, nowarp_raw_density=raw_nw)
    save('field.npz', **npify(out))


def gen_composite(ref):
    g = torch.Generator().manual_seed(41)
    N, S = 40, 48
    tdist = torch.sort(torch.rand(N, S + 1, generator=g), dim=-1).values * 8
    dens = torch.rand(N, S, generator=g) ** 3 * torch.logspace(-2, 2, N)[:, None]   # thin..opaque rays
    dens[7] = 0
    dirs = torch.randn(N, 3, generator=g)
    rgbs = torch.rand(N, S, 3, generator=g)
    far = torch.full((N, 1), 8.0)
    w, alpha, trans = ref.render.compute_alpha_weights(dens, tdist, dirs, opaque_background=False)
    r = ref.render.volumetric_rendering(rgbs, w, tdist, 1.0, far, True, extras={})
    out = dict(tdist=tdist, density=dens, dirs=dirs, rgbs=rgbs, far=far, weights=w)
    out.update({'out_' + k: v for k, v in r.items()})
    wo, _, _ = ref.render.compute_alpha_weights(dens, tdist, dirs, opaque_background=True)
    out['weights_opaque'] = wo
    save('composite.npz', **npify(out))


def run_model(ref, spec, seed, n_rays, ray_seed, rand, train_frac=1.0, compute_extras=True,
              eval_camidx=None, far_jitter=False):
    sd = rm.init_state(spec, seed=seed)
    model, cfg = ref_import.build_reference_model(ref, spec, sd)
    batch = rm.synthetic_rays(n_rays, seed=ray_seed)
    if far_jitter:
        g = torch.Generator().manual_seed(ray_seed + 1)
        batch['far'] = batch['far'] * (1 + 0.1 * torch.rand(n_rays, 1, generator=g))
        batch['cam_idx'] = torch.randint(0, spec.training_views, (n_rays, 1), generator=g)
    model.train(rand)
    torch.manual_seed(seed + 100)
    with ref_import.capture_rng() as cap, torch.no_grad():
        rend, hist = model(rand, dict(batch), train_frac=train_frac, compute_extras=compute_extras,
                           zero_glo=not rand, eval_camidx=eval_camidx)
    out = dict(seed=torch.tensor(seed), checksum=torch.tensor(state_checksum(sd), dtype=torch.float64),
               train_frac=torch.tensor(train_frac))
    out.update({'ray_' + k: v for k, v in batch.items()})
    per = 4 if rand else 1
    assert len(cap.draws) == per * spec.num_levels, [d[0] for d in cap.draws]
    for lvl in range(spec.num_levels):
        d = cap.draws[per * lvl: per * (lvl + 1)]
        if rand:
            out[f'noise{lvl}_jitter'], out[f'noise{lvl}_flip'], out[f'noise{lvl}_spin'], out[f'noise{lvl}_rand_vec'] = [x[1] for x in d]
        else:
            out[f'noise{lvl}_rand_vec'] = d[0][1]
        for k, v in rend[lvl].items():
            out[f'L{lvl}_{k}'] = v
        for k in ('sdist', 'weights', 'density', 'rgb', 'coord'):
            out[f'L{lvl}_hist_{k}'] = hist[lvl][k]
        if 'loss_hash_decay' in hist[lvl]:
            out[f'L{lvl}_hist_loss_hash_decay'] = hist[lvl]['loss_hash_decay']
    if eval_camidx is not None:
        out['eval_camidx'] = eval_camidx
    return out


def gen_models(ref):
    save('model_tiny.npz', **npify(run_model(ref, rm.make_spec('tiny'), 51, 48, 52, False)))
    save('model_tinyR.npz', **npify(run_model(ref, rm.make_spec('tinyR'), 53, 48, 54, False)))
    save('model_sky.npz', **npify(run_model(ref, rm.make_spec('tiny', model_sky=True, brightness_correction=True),
                                            55, 40, 56, False, eval_camidx=torch.tensor([7]), far_jitter=True)))
    save('model_train.npz', **npify(run_model(ref, rm.make_spec('tiny'), 57, 40, 58, True, train_frac=0.3,
                                              compute_extras=False)))


def gen_render_image(ref):
    import accelerate
    spec = rm.make_spec('tiny')
    sd = rm.init_state(spec, seed=61)
    model, cfg = ref_import.build_reference_model(ref, spec, sd)
    cfg.render_chunk_size = 100
    H, W = 16, 24
    rays = rm.synthetic_rays(H * W, seed=62)
    batch = {k: v.reshape(H, W, -1) for k, v in rays.items()}
    acc = accelerate.Accelerator(cpu=True)
    torch.manual_seed(63)
    with ref_import.capture_rng() as cap:
        out_r = ref.models.render_image(model, acc, dict(batch), False, 1.0, cfg, verbose=False)
    out = dict(seed=torch.tensor(61), checksum=torch.tensor(state_checksum(sd), dtype=torch.float64),
               chunk=torch.tensor(100), H=torch.tensor(H), W=torch.tensor(W))
    out.update({'ray_' + k: v for k, v in rays.items()})
    # draws: per chunk, per level one randn_like of that chunk's rays -> stitch into [levels, N, 3]
    n_chunks = (H * W + 99) // 100
    assert len(cap.draws) == n_chunks * spec.num_levels
    for lvl in range(spec.num_levels):
        out[f'noise{lvl}_rand_vec'] = torch.cat([cap.draws[c * spec.num_levels + lvl][1] for c in range(n_chunks)])
    for k, v in out_r.items():
        if k.startswith('ray_'):
            continue                       # random 16-ray visualisation subset (torch.randperm)
        out['out_' + k] = v
    save('render_image.npz', **npify(out))


def grad_digest(name, g, out, gen):
    """Compact record of one gradient tensor: all of it when small, else sums + a row sample."""
    g = g.detach()
    out[f'{name}.sum'] = g.double().sum()
    out[f'{name}.abs'] = g.double().abs().sum()
    if g.numel() <= 4096:
        out[f'{name}.full'] = g
    else:
        rows = torch.randperm(g.shape[0], generator=gen)[:64].sort().values
        out[f'{name}.rows'] = rows
        out[f'{name}.sample'] = g[rows]


def gen_train_step(ref, name, spec, seed, sky_alpha_bias=None):
    """G10: one training step of the REFERENCE -- Model.forward(rand=True) with autograd, the
    reference's own loss functions (train.py:173-216 / train_utils.py), backward -- with every random
    draw captured.  Stores loss terms and gradient digests.
    sky_alpha_bias: added to skynerf.alpha_linear.bias before the step (stored in the fixture, helpers.state_for applies it):
    with the default initialisation the sky NeRF's density head is negative for every sample of these rays, relu(sigma) = 0,
    and the reference's own step hands the whole sky network EXACTLY ZERO gradients -- a fixture that pins nothing about it."""
    sd = rm.init_state(spec, seed=seed)
    if sky_alpha_bias is not None:
        sd['skynerf.alpha_linear.bias'] = sd['skynerf.alpha_linear.bias'] + sky_alpha_bias
    model, cfg = ref_import.build_reference_model(ref, spec, sd)
    model.train()
    n = 96
    rays = rm.synthetic_rays(n, seed=seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    rays['rgb'] = torch.rand(n, 3, generator=g)
    rays['cam_idx'] = torch.randint(0, spec.training_views, (n, 1), generator=g)
    rays['sky_segs'] = (torch.rand(n, generator=g) > 0.7).float()
    batch = {k: (v[:, None, None, :] if v.dim() == 2 else v[:, None, None]) for k, v in rays.items()}
    train_frac = 0.4
    torch.manual_seed(seed + 3)
    with ref_import.capture_rng() as cap:
        rend, hist = model(True, dict(batch), train_frac=train_frac, compute_extras=False, zero_glo=False)
    tu = ref.train_utils
    losses = {}
    losses['data'], stats = tu.compute_data_loss(batch, rend, cfg)
    losses['anti_interlevel'] = tu.anti_interlevel_loss(hist, cfg)
    losses['distortion'] = tu.distortion_loss(hist, cfg)
    losses['hash_decay'] = tu.hash_decay_loss(hist, cfg)
    if spec.model_sky:
        losses['sky'] = 0.002 * tu.sky_loss(batch, rend)            # train.py:181 sky_loss_mult
    if spec.brightness_correction:
        losses['identity'] = 0.002 * tu.transformIdentityLoss(rend)  # train.py:185
    total = sum(losses.values())
    total.backward()
    out = dict(seed=torch.tensor(seed), checksum=torch.tensor(state_checksum(sd), dtype=torch.float64),
               train_frac=torch.tensor(train_frac), mse=torch.tensor(stats['mses']))
    if sky_alpha_bias is not None:
        out['sky_alpha_bias'] = torch.tensor(float(sky_alpha_bias))
        alive = {n_: float(p.grad.abs().sum()) for n_, p in model.named_parameters() if n_.startswith('skynerf') and p.grad is not None}
        assert alive and min(alive.values()) > 0, alive           # the sky network does receive gradients in this fixture
    out.update({'ray_' + k: v for k, v in rays.items()})
    assert len(cap.draws) == 4 * spec.num_levels
    for lvl in range(spec.num_levels):
        d = cap.draws[4 * lvl: 4 * lvl + 4]
        out[f'noise{lvl}_jitter'], out[f'noise{lvl}_flip'], out[f'noise{lvl}_spin'], out[f'noise{lvl}_rand_vec'] = [x[1] for x in d]
        out[f'L{lvl}_sdist'] = hist[lvl]['sdist']
        out[f'L{lvl}_weights'] = hist[lvl]['weights']
        out[f'L{lvl}_rgb'] = rend[lvl]['rgb']
    for k, v in losses.items():
        out['loss_' + k] = v.detach().double()
    out['loss_total'] = total.detach().double()
    gg = torch.Generator().manual_seed(seed + 4)
    for pname, p in model.named_parameters():
        if p.grad is not None:
            grad_digest('grad_' + pname, p.grad, out, gg)
    save(name, **npify(out))


if __name__ == '__main__':
    ref = ref_import.load()
    torch.set_num_threads(1)              # fixed reduction order for the generating run
    if len(sys.argv) > 1 and sys.argv[1] == 'cfg1':
        # G7 for BASELINE configs[0]'s architecture (64 + 64 samples, 64-wide colour MLP), small tables
        save('model_tiny64.npz', **npify(run_model(ref, rm.make_spec('tiny64'), 61, 48, 62, False)))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'nodilate':
        # G7 with both dilation knobs 0: the reference's use_dilation == False branch (models.py:167-168)
        save('model_nodilate.npz', **npify(run_model(ref, rm.make_spec('tiny', dilation_bias=0., dilation_multiplier=0.),
                                                     63, 48, 64, False)))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'train':
        gen_train_step(ref, 'train_step.npz', rm.make_spec('tiny'), 71)
        gen_train_step(ref, 'train_step_sky.npz', rm.make_spec('tiny', model_sky=True, brightness_correction=True), 81, sky_alpha_bias=0.5)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'stepfun':
        gen_stepfun(ref)
        sys.exit(0)
    gen_stepfun(ref)
    gen_cast(ref)
    gen_field(ref)
    gen_composite(ref)
    gen_models(ref)
    gen_render_image(ref)
