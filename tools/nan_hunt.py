"""Reproduce / locate the NaN of profiles/r02c/bench_cfg5_fit.json (`bench.py --cfg5 --fit-steps 600`: tools/fit_scene.py's fit on the
heads model went NaN before step 100).  Runs the same fit loop step by step and reports the first non-finite tensor among the
forward's outputs, the loss terms, the gradients and the parameters.   python tools/nan_hunt.py [--steps 150] [--loss-outside]"""
import argparse, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, bench, fit_scene
from ucnerf_amd.internal import train_utils as tu

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=150)
ap.add_argument("--loss-outside", action="store_true", help="losses outside the autocast region (what train.py does)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
model, _, _ = bench.build_model(dev, heads=True)
for mlp in (model.nerf_mlp, model.prop_mlp_0):
    mlp.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
pool = fit_scene.training_rays(dev)
rgb, _ = fit_scene.scene_colour(pool["origins"], pool["directions"])
cfg = types.SimpleNamespace(data_loss_type='charb', charb_padding=0.001, data_loss_mult=1.0, data_coarse_loss_mult=0., anti_interlevel_loss_mult=0.01,
                            pulse_width=[0.03, 0.003], distortion_loss_mult=0.005, hash_decay_mults=0.1, disable_multiscale_loss=False)
opt = tu.FusedAdam(model.parameters(), lr=0.01, betas=(0.9, 0.99), eps=1e-8)
g = torch.Generator(device=dev).manual_seed(5)
model.train()
n_total = pool["origins"].shape[0]


def bad(t):
    return t is not None and torch.is_tensor(t) and t.is_floating_point() and not bool(torch.isfinite(t).all())


for it in range(a.steps):
    idx = torch.randint(0, n_total, (8192,), device=dev, generator=g)
    batch = {k: v[idx][:, None, None, :] for k, v in pool.items()}
    batch['rgb'] = rgb[idx][:, None, None, :]
    terms = {}
    with torch.autocast('cuda', dtype=torch.bfloat16):
        rend, hist = model(True, batch, min(1.0, it / 600 + 0.1), False, zero_glo=False)
        if not a.loss_outside:
            terms = dict(data=tu.compute_data_loss(batch, rend, cfg)[0], inter=tu.anti_interlevel_loss(hist, cfg),
                         dist=tu.distortion_loss(hist, cfg), decay=tu.hash_decay_loss(hist, cfg))
    if a.loss_outside:
        terms = dict(data=tu.compute_data_loss(batch, rend, cfg)[0], inter=tu.anti_interlevel_loss(hist, cfg),
                     dist=tu.distortion_loss(hist, cfg), decay=tu.hash_decay_loss(hist, cfg))
    loss = sum(terms.values())
    opt.zero_grad(set_to_none=True)
    loss.backward()
    report = []
    for li, r in enumerate(rend):
        report += [f"rend[{li}].{k} ({v.dtype})" for k, v in r.items() if bad(v)]
    for li, h in enumerate(hist):
        report += [f"hist[{li}].{k}" for k, v in h.items() if bad(v)]
    report += [f"loss term {k} = {float(v)}" for k, v in terms.items() if bad(v.detach())]
    gbad = [k for k, p in model.named_parameters() if p.grad is not None and bad(p.grad)]
    for p in model.parameters():
        if p.grad is not None:
            p.grad.nan_to_num_()
    opt.step()
    pbad = [k for k, p in model.named_parameters() if bad(p.data)]
    if it % 10 == 0 or report or gbad or pbad:
        A = rend[0].get('affine_trans')
        print(f"step {it}: loss {float(loss):.5f} terms { {k: round(float(v), 6) for k, v in terms.items()} } "
              f"|A|max {float(A.float().abs().max()) if A is not None else None} sky max {float(rend[0]['sky_rgbs'].float().abs().max()):.3g}", flush=True)
    if report or gbad or pbad:
        print("  non-finite forward/loss:", report[:12])
        print("  non-finite grads (before nan_to_num):", gbad[:12], "... total", len(gbad))
        print("  non-finite params after the step:", pbad[:12])
        if report or pbad:
            break
print("done")
