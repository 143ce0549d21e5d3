import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from ucnerf_amd import _lib
if os.environ.get("UCN_TOOL_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["UCN_TOOL_LIB"])
dev = torch.device("cuda", 0)
model, cfg, sd = bench.build_model(dev, heads=True)
model.sky_side_stream = False
batch = bench.frame_rays(dev)
n = bench.H_IMG * bench.W_IMG
flat = {k: v.reshape(n, -1) for k, v in batch.items()}
print(bench.train_step_ms(model, flat, dev, steps=6, heads=True)["ms"])
