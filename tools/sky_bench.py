"""Time the sky layer (a12): NeRF.render on 65536 rays x 120 samples (562,688 MAC per sample)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucnerf_amd import _lib
if os.environ.get("UCN_TOOL_LIB"):
    _lib.LIB_PATH = os.environ["UCN_TOOL_LIB"]
from ucnerf_amd.internal.sky import NeRF
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = NeRF(D=8, d_in_view=3, W=256, multires_view=4, output_ch=4, skips=[4]).to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
mixed = len(sys.argv) > 2 and sys.argv[2] == 'mixed'
o = torch.randn(n, 3, device=dev) * 0.1
d = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1)
cam = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1)
far = torch.full((n,), 8.0, device=dev)
for _ in range(2):
    out = net.render(o, d, cam, far, mixed=mixed)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    out = net.render(o, d, cam, far, mixed=mixed)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
mac = 562688
print(f"sky render (mixed={mixed}): {ms:.2f} ms / {n} rays = {n / ms / 1e3:.3f} M rays/s, {2 * mac * 120 * n / ms / 1e9:.1f} TFLOP/s algorithmic "
      f"(fp32 MFMA peak 157.3); one 1280x1920 frame = {ms * 2457600 / n:.0f} ms; rgb mean {out.mean().item():.5f}")
