"""Compare the two MLP arithmetic modes on the benchmark model (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from ucnerf_amd.internal import models
dev = torch.device("cuda", 0)
model, cfg, sd = bench.build_model(dev)
batch = bench.frame_rays(dev)
n = 65536
flat = {k: v.reshape(-1, v.shape[-1])[::37][:n].contiguous() for k, v in batch.items()}
flat["rand_vec"] = torch.randn(n, 6, device=dev)
outs = {}
for mode in (0, 1):
    for m in (model.nerf_mlp, model.prop_mlp_0):
        m.mlp_mode = mode
    with torch.no_grad():
        r, h = model(False, flat, 1.0, True)
    torch.cuda.synchronize()
    outs[mode] = (r[-1]["rgb"].clone(), h[-1]["density"].clone(), h[-1]["rgb"].clone(), r[-1]["acc"].clone())
for i, name in enumerate(("pixel rgb", "sample density", "sample rgb", "acc")):
    d = (outs[0][i] - outs[1][i]).abs()
    print(f"{name}: max |fp32 - split| = {float(d.max()):.3e}  mean = {float(d.mean()):.3e}  (max |value| {float(outs[0][i].abs().max()):.3f})")
