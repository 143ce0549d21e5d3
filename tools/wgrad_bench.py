"""ucn_wgrad_bf16 at the shapes of the training steps: us per call and TB/s of operand reads.   python tools/wgrad_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucnerf_amd import _lib
if os.environ.get("UCN_TOOL_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["UCN_TOOL_LIB"])
lib = _lib.load()
dev = torch.device("cuda", 0)


def run(M, KA, kb1, kb2, lda, ldb1, ldb2=32, reps=10):
    A = torch.randn(M, lda, device=dev).bfloat16()
    B1 = torch.randn(M, ldb1, device=dev).bfloat16()
    B2 = torch.randn(M, max(ldb2, 32), device=dev).bfloat16()
    KB = kb1 + kb2
    ws = torch.empty(lib.ucn_wgrad_ws_floats(KA, KB, M), device=dev)
    out = torch.empty(KA, KB, device=dev)
    call = lambda: _lib.check(lib.ucn_wgrad_bf16(A.data_ptr(), lda, KA, B1.data_ptr(), ldb1, kb1, B2.data_ptr() if kb2 else None, ldb2, kb2, M,
                                                 ws.data_ptr(), out.data_ptr(), _lib.stream()))
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    gb = M * (KA + KB) * 2 / 1e9
    print(f"M {M:8d} KA {KA:3d} KB {kb1:3d}+{kb2:2d} lda {lda:5d} ldb {ldb1:5d}: {us:7.1f} us  {gb / us * 1e3:5.2f} TB/s")


for M in (524288, 520000, 983040, 1048576):
    run(M, 256, 256, 0, 256, 1024)          # d1^T h1 of the field MLP (activation row stride 1024)
    run(M, 256, 256, 32, 2240, 2240, 2240)  # a sky layer
    run(M, 256, 64, 32, 256, 1024, 1024)    # <3>
