"""k_composite timing on one frame of rays (S = 64 proposal level without colours, S = 128 NeRF level with colours + extras)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucnerf_amd import _lib
if os.environ.get("UCN_TOOL_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["UCN_TOOL_LIB"])
lib = _lib.load()
dev = torch.device("cuda", 0)
N = 1280 * 1920
g = torch.Generator(device=dev).manual_seed(0)
for S, colours in ((64, False), (128, True)):
    sd = torch.sort(torch.rand(N, S + 1, device=dev, generator=g), dim=-1).values
    dens = torch.rand(N, S, device=dev, generator=g) * 3
    rgb = torch.rand(N, S, 3, device=dev, generator=g) if colours else None
    near, far = torch.full((N,), 0.2, device=dev), torch.full((N,), 1e3, device=dev)
    d = torch.nn.functional.normalize(torch.randn(N, 3, device=dev, generator=g), dim=-1)
    w, main, ex = torch.empty(N, S, device=dev), torch.empty(N, 5, device=dev), torch.empty(N, 4, device=dev)
    args = (dens.data_ptr(), _lib.ptr(rgb), sd.data_ptr(), near.data_ptr(), far.data_ptr(), d.data_ptr(), 1.0, 0, N, S,
            w.data_ptr(), main.data_ptr(), ex.data_ptr(), _lib.stream())
    for _ in range(2):
        _lib.check(lib.ucn_composite(*args))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        _lib.check(lib.ucn_composite(*args))
    e1.record(); torch.cuda.synchronize()
    print(f"composite S {S:3d} colours {colours}: {e0.elapsed_time(e1) / 5:6.3f} ms   checksum {float(main.double().sum()):.6f} {float(ex.double().sum()):.3f}")
