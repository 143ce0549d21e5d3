"""csrc/gemm_f32.hip against the library over M at fixed (N, K): is the kernel bound by where X comes from (HBM / Infinity Cache / L2)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucnerf_amd.internal import dense_f32 as D
dev = torch.device("cuda", 0)
def timed(fn, reps=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for N, K in ((256, 256), (256, 544), (256, 64), (64, 256)):
    for M in (8192, 32768, 131072, 524288, 1048576, 4194304):
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
        t = timed(lambda: D.gemm(x, w, b, out=y)); tl = timed(lambda: torch.addmm(b, x, w.t(), out=y))
        fl = 2.0 * M * N * K
        print(f"N {N:3d} K {K:3d} M {M:8d}: hip {t*1e3:8.1f} us {fl/t/1e9:6.1f} TF | library {tl*1e3:8.1f} us {fl/tl/1e9:6.1f} TF")
        del x, y
