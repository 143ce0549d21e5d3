"""Per-phase cycles of k_gemm_h3 (experiment build -DUCN_H3_CLOCK, UCN_LIB_PATH=tools/_ab/h3_clock/lib.so): prologue / k loop / epilogue / amax."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucnerf_amd import _lib
from ucnerf_amd.internal import dense_f32 as D
lib = _lib.load()
lib.ucn_h3_clock_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
D.set_engine("split")
M = 983040
buf = (ctypes.c_ulonglong * 8)()
for N, K in ((256, 256), (256, 64), (4, 256), (64, 256), (128, 256)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); y = torch.empty(M, N, device="cuda")
    D.amax_of(x)
    for _ in range(2):
        D.gemm(x, w, None, out=y)
    torch.cuda.synchronize(); lib.ucn_h3_clock_read(buf, 1)
    reps = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        D.gemm(x, w, None, out=y)
    e1.record(); torch.cuda.synchronize(); lib.ucn_h3_clock_read(buf, 1)
    wgs = (M + 255) // 256 * reps
    ph = [buf[i] / wgs for i in range(4)]
    print(f"N {N} K {K}: {e0.elapsed_time(e1) / reps:.3f} ms/call; per workgroup (s_memtime ticks): prologue {ph[0]:.0f}  k loop {ph[1]:.0f}  epilogue {ph[2]:.0f}  amax {ph[3]:.0f}  "
          f"sum {sum(ph):.0f} x {wgs // reps / 256:.1f} rounds", flush=True)
