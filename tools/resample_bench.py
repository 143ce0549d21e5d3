"""k_resample timing on one frame of rays: first level (n_prev = 0, S = 64) and NeRF level (n_prev = 64, S = 128)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucnerf_amd import _lib
from ucnerf_amd.internal.models import _u_table
if os.environ.get("UCN_TOOL_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["UCN_TOOL_LIB"])
lib = _lib.load()
dev = torch.device("cuda", 0)
N = 1280 * 1920
g = torch.Generator(device=dev).manual_seed(0)


def run(sd_prev, w_prev, n_prev, S, dil):
    u, mj = _u_table(S, False, dev)
    out = torch.empty(N, S + 1, device=dev)
    args = (_lib.ptr(sd_prev), _lib.ptr(w_prev), n_prev, dil, 1.0, 0.0, u.data_ptr(), None, 0, mj, N, S, out.data_ptr(), _lib.stream())
    for _ in range(2):
        _lib.check(lib.ucn_resample(*args))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        _lib.check(lib.ucn_resample(*args))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    byts = N * 4 * ((n_prev + 1) + n_prev + S + 1)
    print(f"n_prev {n_prev:3d} S {S:3d}: {ms:6.3f} ms  ({byts / ms / 1e6:7.1f} GB/s of {byts / 1e6:.0f} MB)")
    return out


sd1 = run(None, None, 0, 64, 0.0)
w1 = torch.rand(N, 64, device=dev, generator=g) ** 4
w1 = w1 / w1.sum(-1, keepdim=True)
run(sd1, w1, 64, 128, 0.0025 + 0.5 / 64)
