"""Per-level time of k_march_features_bwd on the benchmark NeRF grid (one fake 1-level field per level), 8192-ray train batch."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from ucnerf_amd import _lib
if os.environ.get("UCN_TOOL_LIB"):               # experiment builds; the product loads the in-tree library
    _lib.LIB_PATH = os.path.abspath(os.environ["UCN_TOOL_LIB"])
lib = _lib.load()
dev = torch.device("cuda", 0)
model, cfg, sd = bench.build_model(dev)
batch = bench.frame_rays(dev)
n, S = 8192, 128
flat = {k: v.reshape(-1, v.shape[-1])[::(bench.H_IMG * bench.W_IMG) // n][:n].contiguous() for k, v in batch.items()}
flat["rand_vec"] = torch.randn(n, 6, device=dev)
with torch.no_grad():
    r, h = model(False, flat, 1.0, True)
sdist = h[-1]["sdist"].contiguous()
mlp = model.nerf_mlp
enc = mlp.encoder
basis = torch.empty(n, 6, device=dev)
_lib.check(lib.ucn_cone_basis(flat["cam_dirs"].data_ptr(), flat["rand_vec"][:, 3:6].contiguous().data_ptr(), n, basis.data_ptr(), _lib.stream()))
near, far = flat["near"].reshape(-1).contiguous(), flat["far"].reshape(-1).contiguous()
rad = flat["radii"].reshape(-1).contiguous()
feat = torch.randn(n * S * 2, device=dev)
grad = torch.zeros_like(enc.embeddings)
ws = torch.empty(32 * n * S, device=dev)     # >= ucn_march_features_backward_ws_floats of any 1-level field
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
tot = 0.0
levels = [int(a) for a in sys.argv[2:]] or range(enc.num_levels)
for l in levels:
    d = _lib.UcnField()
    off = np.array([0, int(enc._offsets_np[l + 1] - enc._offsets_np[l])], dtype=np.int32)
    gs = np.array([int(enc._sizes_np[l])], dtype=np.int32)
    d.embeddings = enc.embeddings.data_ptr() + int(enc._offsets_np[l]) * 2 * 4
    d.offsets_host, d.grid_sizes_host = off.ctypes.data, gs.ctypes.data
    d.num_levels, d.level_dim, d.base_resolution, d.log2_per_level_scale = 1, 2, 16 * 2 ** l, 1.0
    args = (ctypes.byref(d), sdist.data_ptr(), near.data_ptr(), far.data_ptr(), flat["origins"].data_ptr(), flat["directions"].data_ptr(),
            basis.data_ptr(), rad.data_ptr(), None, None, 0.5, n, S, mode, 0, feat.data_ptr(), grad.data_ptr() + int(enc._offsets_np[l]) * 2 * 4, ws.data_ptr(), _lib.stream())
    for _ in range(2):
        _lib.check(lib.ucn_march_features_backward(*args))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        _lib.check(lib.ucn_march_features_backward(*args))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    tot += ms
    print(f"level {l:2d} res {16 * 2 ** l:7d} rows {off[1]:7d}: {ms:6.3f} ms   ({n * S * 48 / ms / 1e6:7.1f} G scatter-adds/s)")
print("sum", tot)
