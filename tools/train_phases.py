"""Wall time of the phases of bench.py's training step (synchronised between phases, so the sum exceeds the step).
    python tools/train_phases.py [heads]      heads: sky NeRF + colour-correction head on (scripts/train_waymo.sh:11-12)"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from ucnerf_amd.internal import train_utils as tu
from ucnerf_amd.internal import train_graph as tg
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
acc = {}


def tick(name, t0):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    acc.setdefault(name, []).append((t1 - t0) * 1e3)
    return t1


# sub-phases of the sky layer, wrapped around the functions the model calls
if heads:
    for name in ("sky_forward_fused", "brightness_forward"):
        fn = getattr(tg, name)

        def timed(*a, _fn=fn, _name=name, **k):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = _fn(*a, **k)
            tick("  fwd:" + _name, t0)
            return out
        setattr(tg, name, timed)

for it in range(10):
    idx = torch.randint(0, n_total, (n,), device=dev, generator=g)
    batch = {k: v[idx][:, None, None, :] for k, v in flat.items()}
    batch['rgb'] = torch.rand(n, 1, 1, 3, device=dev, generator=g)
    if heads:
        batch['cam_idx'] = torch.randint(0, 210, (n, 1, 1, 1), device=dev, generator=g)
        batch['sky_segs'] = (torch.rand(n, 1, 1, device=dev, generator=g) > 0.7).float()
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
    if heads:
        loss = loss + 0.002 * tu.sky_loss(batch, rend) + 0.002 * tu.transformIdentityLoss(rend)
        t = tick("loss_sky_identity", t)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    t = tick("backward", t)
    tu.clip_gradients(model, None, cfg)
    t = tick("nan_to_num", t)
    opt.step()
    t = tick("adam", t)
for k, v in acc.items():
    print(f"{k:28s} {np.median(v[3:]):7.3f} ms")
print("sum", sum(np.median(v[3:]) for k, v in acc.items() if not k.startswith("  ")))
print("peak memory GB", torch.cuda.max_memory_allocated() / 2**30, "reserved", torch.cuda.memory_reserved() / 2**30,
      "alloc retries", torch.cuda.memory_stats().get("num_alloc_retries"), "device mallocs", torch.cuda.memory_stats().get("num_device_alloc"))
