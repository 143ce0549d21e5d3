"""What one rank of an 8-rank render_image does on its own (no collective): the full-frame ray permutation, the march of
its 1/8 shard, against 1/8 of the single-rank frame time.  Estimates the strong-scaling loss outside the all-gather."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from ucnerf_amd.internal import models, dist as udist
dev = torch.device("cuda", 0)
model, cfg, sd = bench.build_model(dev)
batch = bench.frame_rays(dev)
n = bench.H_IMG * bench.W_IMG
batch["rand_vec"] = torch.randn(bench.H_IMG, bench.W_IMG, 6, device=dev)
flat = {k: v.reshape(n, -1) for k, v in batch.items()}
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3 / reps
perm, inv = models._tile_order(bench.H_IMG, bench.W_IMG, 8, dev)
t_perm = timed(lambda: {k: v.index_select(0, perm) for k, v in flat.items()})
pf = {k: v.index_select(0, perm) for k, v in flat.items()}
with torch.no_grad():
    t_full = timed(lambda: model._march(False, pf, 1.0, True, 0, want_history=False), reps=2)
    for world in (2, 4, 8):
        lo, hi = udist.shard_bounds(n, world, world - 1)
        shard = {k: v[lo:hi] for k, v in pf.items()}
        t_sh = timed(lambda: model._march(False, shard, 1.0, True, 0, want_history=False), reps=3)
        outs = 12                                              # floats per ray gathered (rgb, depth, acc, 4 distances, ...)
        t_unperm = timed(lambda: [torch.empty(n, 3, device=dev).index_select(0, inv) for _ in range(4)])
        print(f"world {world}: shard march {t_sh:.1f} ms vs full/{world} = {t_full / world:.1f} ms; + full-frame ray permutation {t_perm:.2f} ms "
              f"+ output un-permutation ~{t_unperm:.2f} ms -> efficiency before the all-gather {t_full / world / (t_sh + t_perm + t_unperm):.3f}")
