"""One 1280x1920 frame on the reference's own waymo.gin grid (L 10, C 4, T 2^21; 128 + 32 samples): rays/s and the kernel split.
GPU box:  python tools/render_R.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from ucnerf_amd.internal import models
grid = sys.argv[1] if len(sys.argv) > 1 else "R"
dev = torch.device("cuda", 0)
model, cfg, sd = bench.build_model(dev, grid=grid)
cfg.render_gather_weights = False
batch = bench.frame_rays(dev)
g = torch.Generator().manual_seed(1)
batch["rand_vec"] = torch.randn(bench.H_IMG * bench.W_IMG, 6, generator=g).reshape(bench.H_IMG, bench.W_IMG, 6).to(dev)
acc = bench.Ranks(1, 0)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = models.render_image(model, acc, batch, False, 1.0, cfg, verbose=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"grid {grid}: frame {dt * 1e3:.1f} ms = {bench.H_IMG * bench.W_IMG / dt / 1e6:.2f} M rays/s")
