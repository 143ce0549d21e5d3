"""Training soak (the long form of tests/test_soak.py; its log: profiles/r04/train_soak.txt): 400 optimiser steps of bench.py's step on (heads, config-B grid), (heads, waymo.gin grid), (no heads, waymo.gin grid) against a fixed random target image; the loss must fall and every parameter stay finite.  GPU box:  python tools/train_soak.py"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from ucnerf_amd.internal import train_utils as tu
dev = torch.device("cuda", 0)
for heads, grid in ((True, "B"), (True, "R"), (False, "R")):
    model, _, _ = bench.build_model(dev, heads=heads, grid=grid)
    rays = bench.frame_rays(dev)
    n_total = bench.H_IMG * bench.W_IMG
    flat = {k: v.reshape(n_total, -1) for k, v in rays.items()}
    cfg = types.SimpleNamespace(data_loss_type='charb', charb_padding=0.001, data_loss_mult=1.0, data_coarse_loss_mult=0., anti_interlevel_loss_mult=0.01, pulse_width=[0.03, 0.003],
                                distortion_loss_mult=0.005, hash_decay_mults=0.1, disable_multiscale_loss=False, sky_weight=0.002, idt_weight=0.002)
    g = torch.Generator(device=dev).manual_seed(3)
    opt = tu.FusedAdam(model.parameters(), lr=0.01, betas=(0.9, 0.99), eps=1e-8)
    model.train()
    target_img = torch.rand(n_total, 3, device=dev, generator=g)
    n = 8192
    first = last = None
    for it in range(400):
        idx = torch.randint(0, n_total, (n,), device=dev, generator=g)
        batch = {k: v[idx][:, None, None, :] for k, v in flat.items()}
        batch['rgb'] = target_img[idx][:, None, None, :]
        if heads:
            batch['cam_idx'] = torch.randint(0, 210, (n, 1, 1, 1), device=dev, generator=g)
            batch['sky_segs'] = (torch.rand(n, 1, 1, device=dev, generator=g) > 0.7).float()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            rend, hist = model(True, batch, 0.5, False, zero_glo=False)
        loss = (tu.compute_data_loss(batch, rend, cfg)[0] + tu.anti_interlevel_loss(hist, cfg) + tu.distortion_loss(hist, cfg) + tu.hash_decay_loss(hist, cfg))
        if heads:
            loss = loss + cfg.sky_weight * tu.sky_loss(batch, rend) + cfg.idt_weight * tu.transformIdentityLoss(rend)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        tu.clip_gradients(model, None, cfg); opt.step()
        if it % 100 == 0 or it == 399:
            v = float(loss)
            assert v == v and abs(v) < 1e6, v
            first = v if first is None else first
            last = v
            print(f"heads {heads} grid {grid} step {it}: loss {v:.5f}")
    bad = [k for k, p in model.named_parameters() if not torch.isfinite(p).all()]
    assert not bad, bad
    print("  finite parameters; loss", first, "->", last)
    del model, opt
    torch.cuda.empty_cache()
print("soak ok")
