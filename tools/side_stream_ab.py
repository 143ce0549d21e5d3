"""bf16 training step with heads: the sky NeRF on its side stream against everything on one stream (ms per step, medians of 3 x 8 steps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
model, cfg, sd = bench.build_model(dev, heads=True)
batch = bench.frame_rays(dev)
n = bench.H_IMG * bench.W_IMG
flat = {k: v.reshape(n, -1) for k, v in batch.items()}
for rep in range(3):
    for side in (True, False):
        model.sky_side_stream = side
        print("side stream" if side else "one stream ", round(bench.train_step_ms(model, flat, dev, steps=8, heads=True)["ms"], 3))
