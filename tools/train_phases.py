"""Wall time of the phases of bench.py's training step (synchronised between phases, so the sum exceeds the step)."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from ucnerf_amd.internal import train_utils as tu
dev = torch.device("cuda", 0)
model, cfg0, sd = bench.build_model(dev)
rays = bench.frame_rays(dev)
n_total = bench.H_IMG * bench.W_IMG
flat = {k: v.reshape(n_total, -1) for k, v in rays.items()}
cfg = types.SimpleNamespace(data_loss_type='charb', charb_padding=0.001, data_loss_mult=1.0, data_coarse_loss_mult=0.,
                            anti_interlevel_loss_mult=0.01, pulse_width=[0.03, 0.003], distortion_loss_mult=0.005,
                            hash_decay_mults=0.1, disable_multiscale_loss=False)
g = torch.Generator(device=dev).manual_seed(2)
opt = tu.FusedAdam(model.parameters(), lr=0.01, betas=(0.9, 0.99), eps=1e-8)
model.train()
n = 8192
acc = {}


def tick(name, t0):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    acc.setdefault(name, []).append((t1 - t0) * 1e3)
    return t1


for it in range(10):
    idx = torch.randint(0, n_total, (n,), device=dev, generator=g)
    batch = {k: v[idx][:, None, None, :] for k, v in flat.items()}
    batch['rgb'] = torch.rand(n, 1, 1, 3, device=dev, generator=g)
    torch.cuda.synchronize()
    t = time.perf_counter()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        rend, hist = model(True, batch, 0.5, False, zero_glo=False)
        t = tick("forward", t)
        l_data = tu.compute_data_loss(batch, rend, cfg)[0]
        t = tick("loss_data", t)
        l_inter = tu.anti_interlevel_loss(hist, cfg)
        t = tick("loss_interlevel", t)
        l_dist = tu.distortion_loss(hist, cfg)
        t = tick("loss_distortion", t)
        l_hash = tu.hash_decay_loss(hist, cfg)
        t = tick("loss_hash_decay", t)
        loss = l_data + l_inter + l_dist + l_hash
    opt.zero_grad(set_to_none=True)
    loss.backward()
    t = tick("backward", t)
    for p in model.parameters():
        if p.grad is not None:
            p.grad.nan_to_num_()
    t = tick("nan_to_num", t)
    opt.step()
    t = tick("adam", t)
for k, v in acc.items():
    print(f"{k:18s} {np.median(v[3:]):7.3f} ms")
print("sum", sum(np.median(v[3:]) for v in acc.values()))
