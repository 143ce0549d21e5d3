"""One csrc/gemm_f32.hip shape a few times (for rocprofv3 passes):  python tools/gemm_one.py N K [reps] [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucnerf_amd.internal import dense_f32 as D
N, K = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
M = int(sys.argv[4]) if len(sys.argv) > 4 else 8192 * 128
dev = torch.device("cuda", 0)
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
y = torch.empty(M, N, device=dev)
for _ in range(reps):
    D.gemm(x, w, b, out=y)
if os.environ.get("GEMM_ONE_LIBRARY"):
    for _ in range(reps):
        torch.addmm(b, x, w.t(), out=y)
torch.cuda.synchronize()
