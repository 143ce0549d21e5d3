"""HBM write-only / read-only / copy rates with plain torch kernels (context for store-bound kernels)."""
import torch
dev = torch.device("cuda", 0)
def timed(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
n = 1 << 30                                  # 4 GiB of float32
x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
t = timed(lambda: x.fill_(1.0)); print(f"fill  4 GiB: {t*1e3:.3f} ms  {4*n/t/1e12:.2f} TB/s written")
t = timed(lambda: x.sum());      print(f"sum   4 GiB: {t*1e3:.3f} ms  {4*n/t/1e12:.2f} TB/s read")
t = timed(lambda: y.copy_(x));   print(f"copy  4 GiB: {t*1e3:.3f} ms  {8*n/t/1e12:.2f} TB/s read + written")
xb = torch.empty(n, device=dev, dtype=torch.bfloat16)
t = timed(lambda: xb.copy_(x));  print(f"f32 -> bf16 4 + 2 GiB: {t*1e3:.3f} ms  {6*n/t/1e12:.2f} TB/s")
