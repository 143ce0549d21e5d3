"""Run only bench.py's training step (for rocprofv3 --stats).  `heads` as argv[1]: the sky + colour-head step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else ""
heads = mode in ("heads", "heads_fp32", "launch_fp32")
fp32 = mode in ("fp32", "heads_fp32", "launch_fp32")      # the non-autocast step (hand-written MFMA dense layers)
grid = "R" if mode in ("R", "launch_fp32") else "B"       # R: the reference's waymo.gin grid (L 10, C 4, T 2^21; 128 + 32 samples)
# launch_fp32: the reference's literal launch (scripts/train_waymo.sh: fp32, 15 000 rays, waymo.gin grid, sky + colour head)
model, cfg, sd = bench.build_model(dev, heads=heads, grid=grid)
batch = bench.frame_rays(dev)
n = bench.H_IMG * bench.W_IMG
flat = {k: v.reshape(n, -1) for k, v in batch.items()}
print(bench.train_step_ms(model, flat, dev, n_rays=15000 if mode == "launch_fp32" else 8192, steps=6, heads=heads, autocast=not fp32))
