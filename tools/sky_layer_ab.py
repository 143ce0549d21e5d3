"""bench.py's sky_layer figure (65 536 rays through the inference sky kernels) for an experiment build: UCN_TOOL_LIB=<lib.so> python tools/sky_layer_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ucnerf_amd import _lib
if os.environ.get("UCN_TOOL_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["UCN_TOOL_LIB"])
import torch, bench
dev = torch.device("cuda", 0)
batch = bench.frame_rays(dev)
n = bench.H_IMG * bench.W_IMG
flat = {k: v.reshape(n, -1) for k, v in batch.items()}
for _ in range(2):
    print(bench.sky_layer_ms(flat, dev))
