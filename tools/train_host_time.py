"""Is the training step host-bound?  Per step: time until Python has ISSUED everything (no sync) against time until the GPU is done.
argv[1]: '' | heads | fp32 | heads_fp32 | R"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from ucnerf_amd.internal import train_utils as tu
dev = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else ""
heads = mode in ("heads", "heads_fp32")
fp32 = mode in ("fp32", "heads_fp32")
model, cfg0, sd = bench.build_model(dev, heads=heads, grid="R" if mode == "R" else "B")
fr = bench.frame_rays(dev)
n = bench.H_IMG * bench.W_IMG
flat = {k: v.reshape(n, -1) for k, v in fr.items()}
cfg = types.SimpleNamespace(data_loss_type='charb', charb_padding=0.001, data_loss_mult=1.0, data_coarse_loss_mult=0.,
                            anti_interlevel_loss_mult=0.01, pulse_width=[0.03, 0.003], distortion_loss_mult=0.005,
                            hash_decay_mults=0.1, disable_multiscale_loss=False, sky_weight=0.002, idt_weight=0.002)
g = torch.Generator(device=dev).manual_seed(2)
opt = tu.FusedAdam(model.parameters(), lr=0.01, betas=(0.9, 0.99), eps=1e-8)
model.train()
R = 8192
host, total, fwd, bwd = [], [], [], []
for it in range(14):
    idx = torch.randint(0, n, (R,), device=dev, generator=g)
    b = {k: v[idx][:, None, None, :] for k, v in flat.items()}
    b['rgb'] = torch.rand(R, 1, 1, 3, device=dev, generator=g)
    if heads:
        b['cam_idx'] = torch.randint(0, 210, (R, 1, 1, 1), device=dev, generator=g)
        b['sky_segs'] = (torch.rand(R, 1, 1, device=dev, generator=g) > 0.7).float()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=not fp32):
        rend, hist = model(True, b, 0.5, False, zero_glo=False)
    loss = (tu.compute_data_loss(b, rend, cfg)[0] + tu.anti_interlevel_loss(hist, cfg) + tu.distortion_loss(hist, cfg) + tu.hash_decay_loss(hist, cfg))
    if heads:
        loss = loss + cfg.sky_weight * tu.sky_loss(b, rend) + cfg.idt_weight * tu.transformIdentityLoss(rend)
    t1 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    t2 = time.perf_counter()
    tu.clip_gradients(model, None, cfg)
    opt.step()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    if it >= 2:
        host.append((t3 - t0) * 1e3); total.append((t4 - t0) * 1e3); fwd.append((t1 - t0) * 1e3); bwd.append((t2 - t1) * 1e3)
m = lambda v: float(np.median(v))
print(f"mode {mode!r}: host has issued the step after {m(host):.2f} ms (forward + losses {m(fwd):.2f}, backward {m(bwd):.2f}, clip + Adam {m(host) - m(fwd) - m(bwd):.2f}); GPU done after {m(total):.2f} ms")
