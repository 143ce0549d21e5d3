"""Time ucn_field_mlp alone on the benchmark NeRF field (random features, 65536 rays x 128 samples).

usage: python tools/mlp_bench.py [--lib path/to/libucnerf_march.so] [--mode 0|1] [--rays-fastest] [--iters 5]
The --lib override is for experiment builds (tools/build_exp.sh); the product always loads the in-tree library.
"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--mode", type=int, default=1)
ap.add_argument("--rays-fastest", action="store_true")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--rays", type=int, default=65536)
ap.add_argument("--zero", action="store_true")
a = ap.parse_args()
import torch
from ucnerf_amd import _lib
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
import bench
lib = _lib.load()
dev = torch.device("cuda", 0)
from ucnerf_amd.internal import models
models.MLP.mlp_mode = a.mode
model, cfg, sd = bench.build_model(dev)
mlp = model.nerf_mlp
if a.zero:                      # DVFS probe: same instruction stream, all-zero operands
    with torch.no_grad():
        for prm in mlp.parameters():
            prm.zero_()
d = mlp.field()
n, S = a.rays, 128
B = n * S
F = mlp.encoder.num_levels * mlp.encoder.level_dim
feat = torch.randn(F // 2, B, 2, device=dev) * (0.0 if a.zero else 0.1)
vd = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1)
dirb = torch.empty(lib.ucn_field_dir_floats(ctypes.byref(d), n), device=dev)
st = _lib.stream()
_lib.check(lib.ucn_field_dir_bias(ctypes.byref(d), vd.data_ptr(), n, dirb.data_ptr(), st))
dens, rgb = torch.empty(B, device=dev), torch.empty(B, 3, device=dev)
args = (ctypes.byref(d), feat.data_ptr(), B, S, int(a.rays_fastest), dirb.data_ptr(), dens.data_ptr(), rgb.data_ptr(), None, st)
for _ in range(2):
    _lib.check(lib.ucn_field_mlp(*args))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    _lib.check(lib.ucn_field_mlp(*args))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
macs = F * 64 + 64 * 256 + 283 * 256 + (256 + 283) * 256
print(f"lib={a.lib or 'in-tree'} mode={a.mode} rays_fastest={a.rays_fastest}: {ms:.3f} ms / {B} samples  "
      f"= {2 * macs * B / ms / 1e9:.1f} algorithmic TFLOP/s, frame-equivalent {ms * 2457600 / n:.1f} ms; "
      f"rgb mean {rgb.mean().item():.6f}")
