"""Shader clock and power while a kernel variant runs back to back: python tools/clock_probe.py [--lib so] [--zero]"""
import argparse, ctypes, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--zero", action="store_true")
ap.add_argument("--seconds", type=float, default=4.0)
a = ap.parse_args()
import torch
from ucnerf_amd import _lib
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
import bench
lib = _lib.load()
dev = torch.device("cuda", 0)
model, cfg, sd = bench.build_model(dev)
mlp = model.nerf_mlp
if a.zero:
    with torch.no_grad():
        for prm in mlp.parameters():
            prm.zero_()
d = mlp.field()
n, S = 10240, 128
B = n * S
feat = torch.randn(16, B, 2, device=dev) * (0.0 if a.zero else 0.1)
vd = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1)
dirb = torch.empty(lib.ucn_field_dir_floats(ctypes.byref(d), n), device=dev)
st = _lib.stream()
_lib.check(lib.ucn_field_dir_bias(ctypes.byref(d), vd.data_ptr(), n, dirb.data_ptr(), st))
dens, rgb = torch.empty(B, device=dev), torch.empty(B, 3, device=dev)
args = (ctypes.byref(d), feat.data_ptr(), B, S, 1, dirb.data_ptr(), dens.data_ptr(), rgb.data_ptr(), None, st)
samples = []
stop = False
def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            keep = [l.strip() for l in out.splitlines() if "sclk" in l or "ower" in l]
            samples.append(" | ".join(keep))
        except Exception as e:
            samples.append(repr(e))
        time.sleep(0.3)
t = threading.Thread(target=poll); t.start()
t0 = time.time(); k = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < a.seconds:
    for _ in range(200):
        _lib.check(lib.ucn_field_mlp(*args))
    k += 200
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop = True; t.join()
print(f"lib={a.lib or 'in-tree'} zero={a.zero}: {e0.elapsed_time(e1) / k:.3f} ms per launch over {k} launches")
for s in samples[2:8]:
    print("   ", s)
