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
# small-tensor glue: ops whose largest input has <= 8192 * 130 * 3 elements
import math
small_ms, small_n, by = 0.0, 0, {}
for e in rows:
    if e.key.startswith(("void ", "Cijk", "Custom", "Memcpy", "Memset")) or "(anonymous" in e.key or "k_" in e.key[:3]:
        continue
    sizes = [math.prod(s) for s in e.input_shapes if isinstance(s, (list, tuple)) and len(s) > 0 and all(isinstance(x, int) for x in s)]
    if not sizes or max(sizes) > 8192 * 130 * 3:
        continue
    ms = e.self_device_time_total / 1e3 / 4
    small_ms += ms; small_n += e.count / 4
    by[e.key] = (by.get(e.key, (0, 0))[0] + ms, by.get(e.key, (0, 0))[1] + e.count / 4)
print(f"small-tensor ops: {small_ms:.3f} ms/step in {small_n:.0f} calls/step")
for k, (ms, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"   {k[:40]:40s} {ms:7.3f} ms {n:6.1f} calls")
