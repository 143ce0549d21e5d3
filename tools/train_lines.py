"""Source-line view of bench.py's training step: device time of every aten / custom op by the repo line that issued it
(torch.profiler with_stack), forward lines only -- autograd's backward of a line is attributed to the engine thread and
listed under its node name."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
model, cfg, sd = bench.build_model(dev)
batch = bench.frame_rays(dev)
n = bench.H_IMG * bench.W_IMG
flat = {k: v.reshape(n, -1) for k, v in batch.items()}
bench.train_step_ms(model, flat, dev, steps=2)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    bench.train_step_ms(model, flat, dev, steps=2)
STEPS = 4
by = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    if e.self_device_time_total <= 0:
        continue
    where = "(no python frame: autograd engine / optimizer C++)"
    for fr in e.stack:
        if "/repo/" in fr and "tools/train_lines" not in fr:
            where = fr.split("/repo/")[-1]
            break
    k = (where, e.name[:44])
    by[k][0] += e.self_device_time_total / 1e3 / STEPS
    by[k][1] += 1 / STEPS
rows = sorted(by.items(), key=lambda kv: -kv[1][0])
tot = sum(v[0] for v in by.values())
print(f"total device time {tot:.3f} ms/step")
for (where, name), (ms, cnt) in rows[:int(os.environ.get("TOP", "90"))]:
    print(f"{ms:7.3f} ms {cnt:5.1f}x  {name:44s} {where[:90]}")
