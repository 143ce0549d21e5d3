"""Run only bench.py's training step (for rocprofv3 --stats).  `heads` as argv[1]: the sky + colour-head step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
heads = len(sys.argv) > 1 and sys.argv[1] in ("heads", "heads_fp32")
fp32 = len(sys.argv) > 1 and sys.argv[1] in ("fp32", "heads_fp32")      # the non-autocast step (hand-written fp32 MFMA dense layers)
grid = "R" if len(sys.argv) > 1 and sys.argv[1] == "R" else "B"       # R: the reference's waymo.gin grid (L 10, C 4, T 2^21; 128 + 32 samples)
model, cfg, sd = bench.build_model(dev, heads=heads, grid=grid)
batch = bench.frame_rays(dev)
n = bench.H_IMG * bench.W_IMG
flat = {k: v.reshape(n, -1) for k, v in batch.items()}
print(bench.train_step_ms(model, flat, dev, steps=6, heads=heads, autocast=not fp32))
