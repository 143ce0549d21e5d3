"""The non-autocast training steps of bench.py on both engines of internal/dense_f32.py ("exact": csrc/gemm_f32.hip, "split":
csrc/gemm_h3.hip): ms per step.   python tools/train_fp32_ab.py [plain] [heads] [launch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from ucnerf_amd.internal import dense_f32 as D
dev = torch.device("cuda", 0)
which = sys.argv[1:] or ["plain", "heads", "launch"]
batch = bench.frame_rays(dev)
n = bench.H_IMG * bench.W_IMG
flat = {k: v.reshape(n, -1) for k, v in batch.items()}
for w in which:
    heads = w != "plain"
    model, cfg, sd = bench.build_model(dev, heads=heads, grid="R" if w == "launch" else "B")
    for eng in ("exact", "split"):
        prev = D.set_engine(eng)
        r = bench.train_step_ms(model, flat, dev, n_rays=15000 if w == "launch" else 8192, steps=6, heads=heads, autocast=False)
        D.set_engine(prev)
        print(f"{w:7s} {eng:6s} {r['ms']:8.3f} ms / {r['rays']} rays", flush=True)
    del model
    torch.cuda.empty_cache()
