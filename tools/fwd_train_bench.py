"""Forward featurisation (k_march_features) of the 8192-ray training batch on the benchmark NeRF grid: which of the
differences to the rendering call (random rays instead of a pixel tile, half table, sample-major feature rows) costs
what.  GPU box:  python tools/fwd_train_bench.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from ucnerf_amd import _lib
if os.environ.get("UCN_TOOL_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["UCN_TOOL_LIB"])
lib = _lib.load()
dev = torch.device("cuda", 0)
model, cfg, sd = bench.build_model(dev)
batch = bench.frame_rays(dev)
n, S = 8192, 128
mlp = model.nerf_mlp
enc = mlp.encoder
L, C = enc.num_levels, 2
emb16 = enc.embeddings.detach().to(torch.half)


def rays(kind):
    tot = bench.H_IMG * bench.W_IMG
    if kind == "strided":
        f = {k: v.reshape(-1, v.shape[-1])[::tot // n][:n].contiguous() for k, v in batch.items()}
    elif kind == "random":
        idx = torch.randperm(tot, device=dev)[:n]
        f = {k: v.reshape(-1, v.shape[-1])[idx].contiguous() for k, v in batch.items()}
    else:                                                     # a contiguous run of pixels of one image row block
        f = {k: v.reshape(-1, v.shape[-1])[:n].contiguous() for k, v in batch.items()}
    f["rand_vec"] = torch.randn(n, 6, device=dev)
    return f


for kind in ("tile", "strided", "random"):
    flat = rays(kind)
    with torch.no_grad():
        r, h = model(False, flat, 1.0, True)
    sdist = h[-1]["sdist"].contiguous()
    basis = torch.empty(n, 6, device=dev)
    _lib.check(lib.ucn_cone_basis(flat["cam_dirs"].data_ptr(), flat["rand_vec"][:, 3:6].contiguous().data_ptr(), n, basis.data_ptr(), _lib.stream()))
    near, far = flat["near"].reshape(-1).contiguous(), flat["far"].reshape(-1).contiguous()
    rad = flat["radii"].reshape(-1).contiguous()
    feat = torch.empty(n * S * L * C, device=dev)
    coord = torch.empty(n * S * 3, device=dev)
    tmean = torch.empty(n * S, device=dev)
    for half in (0, 1):
        d = _lib.UcnField()
        ctypes.memmove(ctypes.byref(d), ctypes.byref(mlp.grid_field()), ctypes.sizeof(_lib.UcnField))
        if half:
            d.embeddings = emb16.data_ptr()
        for layout in (0, 1, 2):
            for lpb in (0, 1, 16):
                args = (ctypes.byref(d), sdist.data_ptr(), near.data_ptr(), far.data_ptr(), flat["origins"].data_ptr(),
                        flat["directions"].data_ptr(), basis.data_ptr(), rad.data_ptr(), None, None, 0.5, n, S, lpb,
                        layout | (_lib.TABLE_F16 if half else 0), feat.data_ptr(), coord.data_ptr(), tmean.data_ptr(), _lib.stream())
                for _ in range(2):
                    _lib.check(lib.ucn_march_features(*args))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    _lib.check(lib.ucn_march_features(*args))
                e1.record(); torch.cuda.synchronize()
                print(f"rays {kind:8s} table {'f16' if half else 'f32'} layout {layout} levels/block {lpb:2d}: {e0.elapsed_time(e1) / 5:6.3f} ms")
