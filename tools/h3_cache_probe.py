"""Does the 256 MB memory-side cache serve the split engine's GEMMs when the step is cut into row chunks?  A ping-pong chain of 256 x 256
layers (each product's input is the previous product's output) and the weight gradient behind it, ns per row and call, over chunk sizes.
   python tools/h3_cache_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucnerf_amd.internal import dense_f32 as D
dev = torch.device("cuda", 0)
D.set_engine("split")
w = torch.randn(256, 256, device=dev) / 16
b = torch.zeros(256, device=dev)
for M in (1 << 20, 1 << 19, 1 << 18, 1 << 17, 1 << 16, 1 << 15):
    a, c = torch.randn(M, 256, device=dev), torch.empty(M, 256, device=dev)
    D.amax_of(a)
    reps = max(4, (1 << 22) // M)

    def chain():
        x, y = a, c
        for _ in range(4):
            D.gemm(x, w, b, D.RELU, out=y)
            x, y = y, x

    def grads():
        D.wgrad(a, c, True)
    for name, fn, calls in (("gemm", chain, 4), ("wgrad", grads, 1)):
        fn(); fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps / calls
        print(f"{name:5s} M {M:8d} ({M * 1024 / 2**20:6.0f} MB per operand): {ms:7.4f} ms per call, {ms * 1e6 / M:6.3f} ns per row, {2 * M * 1024 / ms / 1e9:5.2f} TB/s", flush=True)
