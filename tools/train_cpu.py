"""torch.profiler view of bench.py's training step: the host-side cost per operator (what keeps the GPU waiting).
    python tools/train_cpu.py [heads]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
from ucnerf_amd.internal import train_utils as tu
dev = torch.device("cuda", 0)
heads = len(sys.argv) > 1 and sys.argv[1] == "heads"
model, cfg0, sd = bench.build_model(dev, heads=heads)
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


def step():
    idx = torch.randint(0, n_total, (n,), device=dev, generator=g)
    batch = {k: v[idx][:, None, None, :] for k, v in flat.items()}
    batch['rgb'] = torch.rand(n, 1, 1, 3, device=dev, generator=g)
    if heads:
        batch['cam_idx'] = torch.randint(0, 210, (n, 1, 1, 1), device=dev, generator=g)
        batch['sky_segs'] = (torch.rand(n, 1, 1, device=dev, generator=g) > 0.7).float()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        rend, hist = model(True, batch, 0.5, False, zero_glo=False)
    loss = (tu.compute_data_loss(batch, rend, cfg)[0] + tu.anti_interlevel_loss(hist, cfg) + tu.distortion_loss(hist, cfg)
            + tu.hash_decay_loss(hist, cfg))
    if heads:
        loss = loss + 0.002 * tu.sky_loss(batch, rend) + 0.002 * tu.transformIdentityLoss(rend)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    tu.clip_gradients(model, None, cfg)
    opt.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=60))
