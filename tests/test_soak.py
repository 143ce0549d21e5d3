"""Training soak inside -m gpu: many optimiser steps of bench.py's step on each training route; the loss must
fall and every parameter and optimiser state stay finite.  The same loops, longer, are tools/train_soak.py (log kept under
profiles/r04/train_soak.txt).  Includes the exact flow that went NaN in profiles/r02c/bench_cfg5_fit.json (`bench.py --cfg5
--fit-steps 600`: tools/fit_scene.py's fit on the sky + colour-head model, NaN before step 100 with r02's eager sky layer): at HEAD it
does not reproduce (tools/nan_hunt.py, profiles/r04/nan_hunt.txt: 150 steps, every forward / loss / gradient / parameter finite) and
this test keeps it that way."""
import os
import sys
import types

import numpy as np
import pytest
import torch

import bench
from ucnerf_amd.internal import train_utils as tu

pytestmark = pytest.mark.gpu


def _cfg():
    return types.SimpleNamespace(data_loss_type='charb', charb_padding=0.001, data_loss_mult=1.0, data_coarse_loss_mult=0.,
                                 anti_interlevel_loss_mult=0.01, pulse_width=[0.03, 0.003], distortion_loss_mult=0.005,
                                 hash_decay_mults=0.1, disable_multiscale_loss=False, sky_weight=0.002, idt_weight=0.002)


def _soak(heads, grid, autocast, steps, n=4096):
    dev = torch.device("cuda", 0)
    model, _, _ = bench.build_model(dev, heads=heads, grid=grid)
    rays = bench.frame_rays(dev)
    n_total = bench.H_IMG * bench.W_IMG
    flat = {k: v.reshape(n_total, -1) for k, v in rays.items()}
    cfg = _cfg()
    g = torch.Generator(device=dev).manual_seed(3)
    opt = tu.FusedAdam(model.parameters(), lr=0.01, betas=(0.9, 0.99), eps=1e-8)
    model.train()
    target = torch.rand(n_total, 3, device=dev, generator=g)
    losses = []
    for it in range(steps):
        idx = torch.randint(0, n_total, (n,), device=dev, generator=g)
        batch = {k: v[idx][:, None, None, :] for k, v in flat.items()}
        batch['rgb'] = target[idx][:, None, None, :]
        if heads:
            batch['cam_idx'] = torch.randint(0, 210, (n, 1, 1, 1), device=dev, generator=g)
            batch['sky_segs'] = (torch.rand(n, 1, 1, device=dev, generator=g) > 0.7).float()
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=autocast):
            rend, hist = model(True, batch, 0.5, False, zero_glo=False)
        loss = (tu.compute_data_loss(batch, rend, cfg)[0] + tu.anti_interlevel_loss(hist, cfg) + tu.distortion_loss(hist, cfg)
                + tu.hash_decay_loss(hist, cfg))
        if heads:
            loss = loss + cfg.sky_weight * tu.sky_loss(batch, rend) + cfg.idt_weight * tu.transformIdentityLoss(rend)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        tu.clip_gradients(model, None, cfg)
        opt.step()
        losses.append(loss.detach())
    losses = torch.stack(losses).float().cpu().numpy()
    bad = [k for k, p in model.named_parameters() if not torch.isfinite(p).all()]
    bad += [f"opt:{i}:{k}" for i, st in enumerate(opt.state.values()) for k, v in st.items() if torch.is_tensor(v) and not torch.isfinite(v).all()]
    return losses, bad


@pytest.mark.parametrize("heads,grid,autocast,steps", [(True, "B", True, 80), (True, "R", True, 60), (False, "R", True, 60), (False, "B", False, 30),
                                                       (True, "B", False, 20)])
def test_training_soak_loss_falls_and_everything_stays_finite(heads, grid, autocast, steps):
    losses, bad = _soak(heads, grid, autocast, steps)
    print(f"soak heads={heads} grid={grid} autocast={autocast}: loss {losses[0]:.5f} -> {losses[-1]:.5f} over {steps} steps (min {losses.min():.5f})")
    assert np.isfinite(losses).all(), losses
    assert not bad, bad
    assert losses[-5:].mean() < 0.9 * losses[:3].mean(), (losses[:3], losses[-5:])


def test_fit_on_the_heads_model_stays_finite():
    """`bench.py --cfg5 --fit-steps N`'s flow (tools/fit_scene.fit on the sky + colour-head model, reference table initialisation): the
    run of profiles/r02c/bench_cfg5_fit.json printed `loss nan` from step 100 on."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fit_scene
    dev = torch.device("cuda", 0)
    model, _, _ = bench.build_model(dev, heads=True)
    for mlp in (model.nerf_mlp, model.prop_mlp_0):
        mlp.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
    fit_scene.fit(model, dev, 130)
    bad = [k for k, p in model.named_parameters() if not torch.isfinite(p).all()]
    assert not bad, bad
    batch = bench.frame_rays(dev)
    flat = {k: v.reshape(-1, v.shape[-1])[::2400][:1024].contiguous() for k, v in batch.items()}
    flat["rand_vec"] = torch.randn(1024, 6, device=dev)
    with torch.no_grad():
        rend, _ = model(False, flat, 1.0, True, eval_camidx=torch.tensor([7]))
    assert torch.isfinite(rend[-1]["rgb"]).all()
