"""The split engine at M = 2^20, N = 256, K = 64 / 256 / 320 with each epilogue piece on and off (bias, ReLU + bit mask, per-ray row-group\nbias, accumulation onto a stored output, the concatenated K = 320 form): ms per call.  profiles/r06/gemm_h3_notes.txt item 7."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucnerf_amd.internal import dense_f32 as D
dev = torch.device("cuda", 0)
D.set_engine("split")
M, S = 8192 * 128, 128
def timed(fn, reps=8):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
h0 = torch.randn(M, 64, device=dev); h1 = torch.randn(M, 256, device=dev); h2 = torch.empty(M, 256, device=dev)
A = torch.randn(256, 64, device=dev); W = torch.randn(256, 256, device=dev) / 16; pr = torch.randn(M // S, 256, device=dev)
cat = torch.randn(M, 320, device=dev); Wc = torch.randn(256, 320, device=dev) / 16
D.amax_of(h0); D.amax_of(h1); D.amax_of(cat)
print("K=64 relu rowbias      ", timed(lambda: D.gemm(h0, A, None, D.RELU, out=h2, rowbias=pr, rgroup=S)))
print("K=256 plain            ", timed(lambda: D.gemm(h1, W, out=h2)))
print("K=64 accumulate relu rb", timed(lambda: D.gemm(h0, A, None, D.ACCUMULATE | D.RELU, out=h2, rowbias=pr, rgroup=S)))
print("K=320 relu rowbias     ", timed(lambda: D.gemm(cat, Wc, None, D.RELU, out=h2, rowbias=pr, rgroup=S)))
g64 = torch.empty(M, 64, device=dev); d = torch.randn(M, 256, device=dev); D.amax_of(d)
At = torch.randn(64, 256, device=dev)
print("N=64 K=256             ", timed(lambda: D.gemm(d, At, out=g64)))
print("N=64 K=256 accumulate  ", timed(lambda: D.gemm(d, At, None, D.ACCUMULATE, out=g64)))
b = torch.randn(256, device=dev)
print("K=64 bias only         ", timed(lambda: D.gemm(h0, A, b, 0, out=h2)))
print("K=64 relu only         ", timed(lambda: D.gemm(h0, A, b, D.RELU, out=h2)))
print("K=64 rowbias only      ", timed(lambda: D.gemm(h0, A, None, 0, out=h2, rowbias=pr, rgroup=S)))
D.RELU_BITS = False
print("K=64 relu, no bits     ", timed(lambda: D.gemm(h0, A, b, D.RELU, out=h2)))
D.RELU_BITS = True
print("again: K=64 relu rowbias      ", timed(lambda: D.gemm(h0, A, None, D.RELU, out=h2, rowbias=pr, rgroup=S)))
print("again: K=64 relu rowbias + b  ", timed(lambda: D.gemm(h0, A, b, D.RELU, out=h2, rowbias=pr, rgroup=S)))
print("again: K=64 bias only         ", timed(lambda: D.gemm(h0, A, b, 0, out=h2)))
D.RELU_BITS = False
print("again: K=64 relu rowbias nobit", timed(lambda: D.gemm(h0, A, None, D.RELU, out=h2, rowbias=pr, rgroup=S)))
