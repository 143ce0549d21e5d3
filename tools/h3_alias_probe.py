"""k_gemm_h3 at M = 2^20, N = K = 256 with operand row strides that are / are not powers of two (channel aliasing probe)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucnerf_amd.internal import dense_f32 as D
dev = torch.device("cuda", 0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192 * 128


def timed(fn, reps=8):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


D.set_engine("split")
for N, K in ((256, 256), (4, 256), (256, 64)):
    for ldx, ldy in ((K, N), (K + 8, N), (K, N + 8), (K + 8, N + 8), (K + 32, N + 32), (K + 72, N + 40)):
        if N == 4 and ldy != N and ldy != N + 8:
            continue
        xw = torch.randn(M, ldx, device=dev); x = xw[:, :K]
        yw = torch.empty(M, ldy, device=dev); y = yw[:, :N]
        w = torch.randn(N, K, device=dev)
        D.amax_of(x)
        t = timed(lambda: D.gemm(x, w, None, out=y))
        print(f"M {M} N {N} K {K} ldx {ldx} ldy {ldy}: {t:.3f} ms  {(M * (N + K) * 4.0) / t / 1e9:.2f} TB/s", flush=True)
