"""csrc/gemm_f32.hip at the training step's shapes (M = 8192 rays x 128 samples): ms and TFLOP/s per call, with torch's library
GEMM beside it.   python tools/gemm_f32_bench.py   (UCN_TOOL_LIB=<variant .so> for experiment builds)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucnerf_amd import _lib
if os.environ.get("UCN_TOOL_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["UCN_TOOL_LIB"])
from ucnerf_amd.internal import dense_f32 as D
dev = torch.device("cuda", 0)
M = 8192 * 128


def timed(fn, reps=5):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def both(fn):
    out = {}
    for eng in ("exact", "split"):
        prev = D.set_engine(eng)
        out[eng] = timed(fn)
        D.set_engine(prev)
    return out


for N, K in ((256, 256), (256, 64), (64, 256), (256, 544), (4, 256), (128, 256), (256, 4)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    D.amax_of(x)                                      # (the producing kernel records it in the training step)
    t = both(lambda: D.gemm(x, w, b, out=y))
    tl = timed(lambda: torch.addmm(b, x, w.t(), out=y))
    fl, by = 2.0 * M * N * K, M * (N + K) * 4.0
    print(f"gemm  M {M} N {N:3d} K {K:3d}: exact {t['exact']:6.3f} ms {fl / t['exact'] / 1e9:6.1f} TF | split {t['split']:6.3f} ms {fl / t['split'] / 1e9:6.1f} TF-equivalent "
          f"{by / t['split'] / 1e9:5.2f} TB/s | library {tl:6.3f} ms {fl / tl / 1e9:6.1f} TF")
for N, K in ((256, 256), (256, 64), (64, 32), (4, 256), (256, 544), (128, 256)):
    gy = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev)
    D.amax_of(gy); D.amax_of(x)
    t = both(lambda: D.wgrad(gy, x, True))
    tl = timed(lambda: gy.t() @ x)
    fl, by = 2.0 * M * N * K, M * (N + K) * 4.0
    print(f"wgrad M {M} N {N:3d} K {K:3d}: exact {t['exact']:6.3f} ms {fl / t['exact'] / 1e9:6.1f} TF | split {t['split']:6.3f} ms {fl / t['split'] / 1e9:6.1f} TF-equivalent "
          f"{by / t['split'] / 1e9:5.2f} TB/s | library {tl:6.3f} ms {fl / tl / 1e9:6.1f} TF")
