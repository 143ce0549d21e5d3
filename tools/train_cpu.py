"""Host-side view of bench.py's training step: CPU self time per op (torch.profiler), 4 steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
model, cfg, sd = bench.build_model(dev)
batch = bench.frame_rays(dev)
n = bench.H_IMG * bench.W_IMG
flat = {k: v.reshape(n, -1) for k, v in batch.items()}
bench.train_step_ms(model, flat, dev, steps=2)
with profile(activities=[ProfilerActivity.CPU], record_shapes=False) as prof:
    bench.train_step_ms(model, flat, dev, steps=2)
rows = sorted(prof.key_averages(), key=lambda e: -e.self_cpu_time_total)
tot = sum(e.self_cpu_time_total for e in rows) / 1e3 / 4
print(f"CPU self time, all ops: {tot:.2f} ms/step")
for e in rows[:28]:
    print(f"{e.key[:46]:46s} {e.count / 4:8.1f} calls/step {e.self_cpu_time_total / 1e3 / 4:8.3f} ms/step")
