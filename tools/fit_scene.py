"""A TRAINED-LIKE field for the questions a random-initialised one cannot answer (early-termination
sample compaction).  The config-B model (both fields, reference initialisation: tables +-1e-4) is fitted with THIS
repo's training graph to an analytic scene with opaque surfaces and empty space -- a ground plane, spheres at several
depths, a constant far background -- seen from 24 translated / rotated poses (parallax pins the depths), then the
benchmark frame is rendered and the compositing weights of its NeRF-level samples are tallied.

    python tools/fit_scene.py [--steps 600] [--out gpurun_out/fit_scene.pt]      (GPU box)

Prints the fraction of samples an alive rule `weight >= w_min` keeps, and the frame time with and without
Model.compact_min_weight (bench.py --fit-steps N --compact W runs the same fit in front of its timed region).
"""
import argparse
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench


def scene_colour(o, d):
    """Analytic opaque scene, rays o + t d (d unnormalised like the model's `directions`): returns rgb [N,3], depth t."""
    n = o.shape[0]
    dev = o.device
    t_best = torch.full((n,), 1e9, device=dev)
    rgb = torch.tensor([0.55, 0.7, 0.9], device=dev).expand(n, 3).clone()          # far background
    # ground plane y = -0.6, checkerboard
    ty = (-0.6 - o[:, 1]) / d[:, 1].clamp(max=-1e-6)
    hit = (d[:, 1] < -1e-6) & (ty > 0.05) & (ty < 7.5)
    p = o + ty[:, None] * d
    chk = ((torch.floor(p[:, 0] * 1.5) + torch.floor(p[:, 2] * 1.5)) % 2 == 0).float()
    col = chk[:, None] * torch.tensor([0.8, 0.8, 0.75], device=dev) + (1 - chk[:, None]) * torch.tensor([0.25, 0.3, 0.2], device=dev)
    rgb = torch.where(hit[:, None], col, rgb)
    t_best = torch.where(hit, ty, t_best)
    spheres = [((-0.8, -0.2, -2.0), 0.4, (0.9, 0.2, 0.2)), ((0.5, -0.3, -3.0), 0.3, (0.2, 0.8, 0.3)),
               ((1.6, 0.1, -4.5), 0.7, (0.2, 0.3, 0.9)), ((-1.8, 0.3, -5.5), 0.9, (0.9, 0.8, 0.2)),
               ((0.0, 0.0, -1.3), 0.15, (0.9, 0.5, 0.9)), ((2.5, -0.2, -2.2), 0.4, (0.4, 0.9, 0.9))]
    dd = (d * d).sum(-1)
    for c, r, colr in spheres:
        c = torch.tensor(c, device=dev)
        oc = o - c
        b = (oc * d).sum(-1)
        disc = b * b - dd * ((oc * oc).sum(-1) - r * r)
        t = (-b - torch.sqrt(disc.clamp(min=0))) / dd
        hit = (disc > 0) & (t > 0.05) & (t < t_best)
        nrm = torch.nn.functional.normalize(oc + t[:, None] * d, dim=-1)
        shade = (0.35 + 0.65 * (nrm * torch.tensor([0.4, 0.8, 0.45], device=dev)).sum(-1).clamp(min=0))[:, None]
        rgb = torch.where(hit[:, None], shade * torch.tensor(colr, device=dev), rgb)
        t_best = torch.where(hit, t, t_best)
    return rgb.clamp(0, 1), t_best


def training_rays(device, n_poses=24, down=8):
    from ucnerf_amd.internal import camera_utils
    W, H = bench.W_IMG // down, bench.H_IMG // down
    K = np.array([[bench.FOCAL / down, 0.0, W / 2], [0.0, bench.FOCAL / down, H / 2], [0.0, 0.0, 1.0]])
    rng = np.random.default_rng(11)
    c2ws = []
    for i in range(n_poses):
        yaw = 0.3 + rng.uniform(-0.35, 0.35)
        R = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
        t = np.array([0.1, -0.05, 0.2]) + rng.uniform(-0.5, 0.5, 3) * np.array([1.0, 0.25, 1.0])
        c2ws.append(np.concatenate([R, t[:, None]], axis=1))
    cams = (np.repeat(np.linalg.inv(K)[None], n_poses, 0), np.stack(c2ws), None, None)
    keys = ("origins", "directions", "viewdirs", "cam_dirs", "radii", "near", "far", "cam_idx", "lossmult")
    per = [camera_utils.generate_ray_batch(cams, i, W, H, 0.0, 8.0, device=device) for i in range(n_poses)]
    return {k: torch.cat([b[k].reshape(-1, b[k].shape[-1]) for b in per], dim=0) for k in keys}


def fit(model, device, steps):
    from ucnerf_amd.internal import train_utils as tu
    pool = training_rays(device)
    rgb, _ = scene_colour(pool["origins"], pool["directions"])
    cfg = types.SimpleNamespace(data_loss_type='charb', charb_padding=0.001, data_loss_mult=1.0, data_coarse_loss_mult=0.,
                                anti_interlevel_loss_mult=0.01, pulse_width=[0.03, 0.003], distortion_loss_mult=0.005,
                                hash_decay_mults=0.1, disable_multiscale_loss=False)
    opt = tu.FusedAdam(model.parameters(), lr=0.01, betas=(0.9, 0.99), eps=1e-8)
    g = torch.Generator(device=device).manual_seed(5)
    model.train()
    n_total = pool["origins"].shape[0]
    for it in range(steps):
        idx = torch.randint(0, n_total, (8192,), device=device, generator=g)
        batch = {k: v[idx][:, None, None, :] for k, v in pool.items()}
        batch['rgb'] = rgb[idx][:, None, None, :]
        with torch.autocast('cuda', dtype=torch.bfloat16):
            rend, hist = model(True, batch, min(1.0, it / steps + 0.1), False, zero_glo=False)
            loss = (tu.compute_data_loss(batch, rend, cfg)[0] + tu.anti_interlevel_loss(hist, cfg)
                    + tu.distortion_loss(hist, cfg) + tu.hash_decay_loss(hist, cfg))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        for p in model.parameters():
            if p.grad is not None:
                p.grad.nan_to_num_()
        opt.step()
        if it % 100 == 0 or it == steps - 1:
            mse = float(((rend[-1]['rgb'].float().reshape(-1, 3) - batch['rgb'].reshape(-1, 3)) ** 2).mean())
            print(f"  step {it:4d}: loss {float(loss):.5f}  psnr {-10 * np.log10(max(mse, 1e-12)):.2f} dB", flush=True)
    model.eval()


def frame_ms(model, flat, reps=2):
    with torch.no_grad():
        model._march(False, flat, 1.0, True, None, want_history=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r, _ = model._march(False, flat, 1.0, True, None, want_history=False)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / reps, r[-1]["rgb"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--out", default="", help="write the fitted state dict here (57 MB; not under gpurun_out/: its merge-back is capped at 64 MiB)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    model, cfg, _ = bench.build_model(dev)
    for mlp in (model.nerf_mlp, model.prop_mlp_0):               # the reference's own initialisation (grid.py:151-153)
        mlp.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
    t0 = time.time()
    fit(model, dev, a.steps)
    print(f"fitted {a.steps} steps in {time.time() - t0:.1f} s")
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        torch.save({k: v.cpu() for k, v in model.state_dict().items()}, a.out)
    batch = bench.frame_rays(dev)
    flat = {k: v.reshape(-1, v.shape[-1]) for k, v in batch.items()}
    n = flat["origins"].shape[0]
    flat["rand_vec"] = torch.randn(n, 6, generator=torch.Generator().manual_seed(1)).to(dev)
    idx = torch.arange(0, n, 37, device=dev)[:65536]
    sub = {k: v[idx] for k, v in flat.items()}
    with torch.no_grad():
        rend, hist = model._march(False, sub, 1.0, True, None, want_history=True)
    w = hist[-1]["weights"].reshape(idx.numel(), -1).double()
    true_rgb, t_hit = scene_colour(sub["origins"], sub["directions"])
    hit = t_hit < 1e8                                       # rays that meet the plane or a sphere (the rest see the far background)
    acc = w.sum(-1)
    mse = float(((rend[-1]["rgb"].reshape(-1, 3).double() - true_rgb.double()) ** 2).mean())
    print(f"benchmark frame, 65536 strided rays: psnr vs the analytic scene {-10 * np.log10(max(mse, 1e-12)):.2f} dB; acc mean {float(acc.mean()):.4f}; "
          f"on the {int(hit.sum())} surface-hit rays: acc mean {float(acc[hit].mean()):.4f}, 1 % quantile {float(torch.quantile(acc[hit], 0.01)):.4f}")
    for w_min in (1e-9, 4e-8, 1e-6, 1e-5, 1e-4):
        keep = w >= w_min
        print(f"  weight >= {w_min:g}: keep {float(keep.double().mean()):.4f} of the samples; worst lost weight per ray "
              f"{float((w * (~keep)).sum(-1).max()):.3e}")
    q = torch.tensor([0.01, 0.1, 0.5, 0.9, 0.99], device=dev, dtype=torch.double)
    print("  weight quantiles (1, 10, 50, 90, 99 %):", [f"{v:.3e}" for v in torch.quantile(w.flatten()[:4000000], q).tolist()])
    # frame time against the alive fraction: thresholds from "everything alive" to "nothing alive" -> the fixed cost of the
    # compacted route, its slope, and the alive fraction below which it beats the plain route
    model.compact_min_weight = 0.0
    ms0, rgb0 = frame_ms(model, flat)
    pts = []
    for thr in (1e-45, 4e-8, 1e-6, 1e-4, 1e-2, 2.0):
        model.compact_min_weight = thr
        model._alive_stats = []
        frame_ms(model, flat, reps=1)                        # the statistics run (its host reads are not timed)
        alive = sum(a for a, _ in model._alive_stats) / max(1, sum(b for _, b in model._alive_stats))
        model._alive_stats = None
        ms1, rgb1 = frame_ms(model, flat)
        pts.append((alive, ms1))
        print(f"  compact_min_weight {thr:g}: alive {alive:.4f}, frame {ms1:.1f} ms (plain route {ms0:.1f} ms), rgb L-inf vs plain {float((rgb0 - rgb1).abs().max()):.2e}")
    a = np.array([p[0] for p in pts]); t = np.array([p[1] for p in pts])
    slope, fixed = np.polyfit(a, t, 1)
    print(f"compacted route: {fixed:.1f} ms at 0 % alive + {slope:.1f} ms x alive fraction; plain route {ms0:.1f} ms; "
          f"break-even alive fraction {(ms0 - fixed) / slope:.3f}")
    model.compact_min_weight = 0.0


if __name__ == "__main__":
    main()
