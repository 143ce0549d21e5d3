"""Fraction of NeRF-level samples of the benchmark frame that an early-termination rule would keep
(transmittance before the sample >= t_min and weight >= w_min).  Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
model, cfg, sd = bench.build_model(dev)
batch = bench.frame_rays(dev)
flat = {k: v.reshape(-1, v.shape[-1]) for k, v in batch.items()}
n = flat["origins"].shape[0]
idx = torch.arange(0, n, 37, device=dev)[:65536]
sub = {k: v[idx] for k, v in flat.items()}
g = torch.Generator().manual_seed(1)
sub["rand_vec"] = torch.randn(idx.numel(), 6, generator=g).to(dev)
with torch.no_grad():
    rend, hist = model(None, sub, 1.0, True)
w = hist[-1]["weights"].reshape(idx.numel(), -1).double()
T = 1.0 - torch.cumsum(w, -1) + w            # transmittance in front of each sample (approx., from the weights)
print("samples", tuple(w.shape), "acc mean", float(w.sum(-1).mean()))
for t_min, w_min in ((1e-5, 2e-8), (1e-4, 1e-7), (1e-3, 1e-6), (1e-5, 0.0), (0.0, 2e-8), (0.0, 1e-6), (0.0, 1e-5)):
    keep = (T >= t_min) & (w >= w_min)
    lost = (w * (~keep)).sum(-1).max()
    print(f"T>={t_min:g} w>={w_min:g}: keep {float(keep.double().mean()):.4f}  worst lost weight/ray {float(lost):.3e}")
q = torch.tensor([0.01, 0.1, 0.5, 0.9, 0.99], device=dev, dtype=torch.double)
print("weight quantiles", torch.quantile(w.flatten()[:4000000], q).tolist())
