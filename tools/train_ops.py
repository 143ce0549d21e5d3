"""Op-level view of bench.py's training step (torch.profiler): which aten ops own the elementwise / reduce / cat time."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    bench.train_step_ms(model, flat, dev, steps=2)          # 4 steps inside (2 warm-up + 2)
rows = sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.self_device_time_total)
print(f"{'op':42s} {'calls/step':>10s} {'ms/step':>9s}  shapes")
tot = 0.0
for e in rows[:60]:
    ms = e.self_device_time_total / 1e3 / 4
    tot += ms
    print(f"{e.key[:42]:42s} {e.count / 4:10.1f} {ms:9.3f}  {str(e.input_shapes)[:110]}")
print("listed", tot, "all", sum(e.self_device_time_total for e in rows) / 1e3 / 4)
