"""Locate a NaN in a training loop of the small heads model of tests/test_dropin_device.py (T = 2^15, 64 + 32 samples, sky + colour head,
reference table initialisation).  python tools/nan_hunt2.py [--autocast] [--steps 60]   (UCN_F32_LIBRARY=1: library GEMMs)"""
import argparse, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from ucnerf_amd.internal import models, configs, train_utils as tu
ap = argparse.ArgumentParser()
ap.add_argument("--autocast", action="store_true")
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--no-sky", action="store_true")
ap.add_argument("--no-bc", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0)
config = configs.Config(model_sky=not a.no_sky, brightness_correction=not a.no_bc, training_views=210)
for k, v in dict(data_loss_type='charb', charb_padding=0.001, data_loss_mult=1.0, data_coarse_loss_mult=0., anti_interlevel_loss_mult=0.01,
                 pulse_width=[0.03, 0.003], distortion_loss_mult=0.005, hash_decay_mults=0.1, disable_multiscale_loss=False,
                 sky_weight=0.002, idt_weight=0.002).items():
    setattr(config, k, v)
torch.manual_seed(0)
kw = dict(grid_level_dim=2, grid_log2_hashmap_size=15)
with models.bindings(NerfMLP=dict(grid_disired_resolution=8192, **kw), PropMLP=dict(**kw)):
    model = models.Model(config=config, num_levels=2, num_prop_samples=64, num_nerf_samples=32).to(dev)
if not a.no_sky:
    model.skynerf.alpha_linear.bias.data.fill_(0.05)
opt = tu.FusedAdam(model.parameters(), lr=0.01, betas=(0.9, 0.99), eps=1e-8)
rays = bench.frame_rays(dev)
n_total = bench.H_IMG * bench.W_IMG
flat = {k: v.reshape(n_total, -1) for k, v in rays.items()}
g = torch.Generator(device=dev).manual_seed(1)
target = torch.rand(n_total, 3, device=dev, generator=g)
n = 2048
model.train()


def bad(t):
    return t is not None and torch.is_tensor(t) and t.is_floating_point() and not bool(torch.isfinite(t).all())


for step in range(1, a.steps + 1):
    idx = torch.randint(0, n_total, (n,), device=dev, generator=g)
    batch = {k: v[idx][:, None, None, :] for k, v in flat.items()}
    batch['rgb'] = target[idx][:, None, None, :]
    batch['cam_idx'] = torch.randint(0, 210, (n, 1, 1, 1), device=dev, generator=g)
    batch['sky_segs'] = (torch.rand(n, 1, 1, device=dev, generator=g) > 0.7).float()
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=a.autocast):
        rend, hist = model(True, batch, train_frac=0.1, compute_extras=False, zero_glo=False)
    terms = dict(data=tu.compute_data_loss(batch, rend, config)[0], inter=tu.anti_interlevel_loss(hist, config), dist=tu.distortion_loss(hist, config),
                 decay=tu.hash_decay_loss(hist, config))
    if not a.no_sky:
        terms['sky'] = config.sky_weight * tu.sky_loss(batch, rend)
    if not a.no_bc:
        terms['idt'] = config.idt_weight * tu.transformIdentityLoss(rend)
    loss = sum(terms.values())
    loss.backward()
    rep = [f"rend[{i}].{k}" for i, r in enumerate(rend) for k, v in r.items() if bad(v)]
    rep += [f"hist[{i}].{k}" for i, h in enumerate(hist) for k, v in h.items() if bad(v)]
    rep += [f"term {k}={float(v)}" for k, v in terms.items() if bad(v.detach())]
    gbad = [(k, int((~torch.isfinite(p.grad)).sum())) for k, p in model.named_parameters() if p.grad is not None and bad(p.grad)]
    gmax = {k: float(p.grad.abs().max()) for k, p in model.named_parameters() if p.grad is not None}
    tu.clip_gradients(model, None, config)
    opt.step()
    pbad = [k for k, p in model.named_parameters() if bad(p.data)]
    big = sorted(gmax.items(), key=lambda kv: -kv[1] if kv[1] == kv[1] else -1e30)[:3]
    print(f"step {step}: loss {float(loss):.5f} { {k: round(float(v), 6) for k, v in terms.items()} } max|grad| {big}", flush=True)
    if rep or gbad or pbad:
        print("  non-finite forward/loss:", rep[:10]); print("  non-finite grads:", gbad[:10]); print("  non-finite params:", pbad[:10])
        if rep or pbad:
            break
