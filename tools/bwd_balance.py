"""Load balance of k_march_features_bwd_cmp on the benchmark NeRF grid (8192-ray train batch, all 16 levels in one call).
Needs an experiment build with the workgroup clocks:  tools/build_variant.sh march_features wgclock -DUCN_WG_CLOCK
then  UCN_TOOL_LIB=tools/_exp/wgclock/lib.so python tools/bwd_balance.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from ucnerf_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ["UCN_TOOL_LIB"])
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
dev = torch.device("cuda", 0)
GRID = os.environ.get("UCN_TOOL_GRID", "B")          # R: the reference's waymo.gin grid (L 10, C 4, T 2^21)
FIELD = os.environ.get("UCN_TOOL_FIELD", "nerf")      # or prop
model, cfg, sd = bench.build_model(dev, grid=GRID)
batch = bench.frame_rays(dev)
n = 8192
flat = {k: v.reshape(-1, v.shape[-1])[::(bench.H_IMG * bench.W_IMG) // n][:n].contiguous() for k, v in batch.items()}
flat["rand_vec"] = torch.randn(n, 6, device=dev)
with torch.no_grad():
    r, h = model(False, flat, 1.0, True)
sdist = h[-1 if FIELD == "nerf" else 0]["sdist"].contiguous()
S = sdist.shape[-1] - 1
mlp = model.nerf_mlp if FIELD == "nerf" else model.prop_mlp_0
enc = mlp.encoder
basis = torch.empty(n, 6, device=dev)
_lib.check(lib.ucn_cone_basis(flat["cam_dirs"].data_ptr(), flat["rand_vec"][:, 3:6].contiguous().data_ptr(), n, basis.data_ptr(), _lib.stream()))
near, far = flat["near"].reshape(-1).contiguous(), flat["far"].reshape(-1).contiguous()
rad = flat["radii"].reshape(-1).contiguous()
L = enc.num_levels
feat = torch.randn(L * n * S * enc.level_dim, device=dev)
grad = torch.zeros_like(enc.embeddings)
fld = mlp.grid_field()
ws = torch.empty(lib.ucn_march_features_backward_ws_floats(ctypes.byref(fld), n, S), device=dev)
args = (ctypes.byref(fld), sdist.data_ptr(), near.data_ptr(), far.data_ptr(), flat["origins"].data_ptr(), flat["directions"].data_ptr(),
        basis.data_ptr(), rad.data_ptr(), None, None, 0.5, n, S, 0, (_lib.BWD_FIXED_POINT if os.environ.get('UCN_TOOL_FX') == '1' else 0), feat.data_ptr(), grad.data_ptr(), ws.data_ptr(), _lib.stream())
for _ in range(3):
    _lib.check(lib.ucn_march_features_backward(*args))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    _lib.check(lib.ucn_march_features_backward(*args))
e1.record(); torch.cuda.synchronize()
print(f"whole call (masks + compacted kernel): {e0.elapsed_time(e1) / 5:.3f} ms")
clk = np.zeros((8192, 3), dtype=np.uint64)                     # one row per task of the compacted kernel; unused rows stay 0
assert raw.ucn_debug_wg_clock(clk.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(8192)) == 0
clk = clk[clk[:, 1] > 0]
T = clk.shape[0]
t0 = clk[:, 0].min()
start = (clk[:, 0] - t0).astype(np.float64) / 100.0          # us (100 MHz constant clock)
end = (clk[:, 1] - t0).astype(np.float64) / 100.0
dur = end - start
print(f"{T} workgroups; kernel span {end.max():.0f} us; sum of workgroup times {dur.sum() / 1000:.1f} ms = {dur.sum() / 256:.0f} us per CU at 256 CUs")
lv = clk[:, 2].astype(np.int64)
for l in range(L):
    d = dur[lv == l]
    if d.size == 0:
        continue
    print(f"level {l:2d}: {d.size:4d} wgs  mean {d.mean():7.1f} us  max {d.max():7.1f}  min {d.min():7.1f}  sum {d.sum() / 1000:6.2f} ms  first start {start[lv == l].min():7.0f}  last end {end[lv == l].max():7.0f}")
# how busy the chip is over time: workgroups in flight at 20 sample times
for q in np.linspace(0, end.max(), 21)[:-1]:
    print(f"t = {q:7.0f} us: {int(((start <= q) & (end > q)).sum()):4d} workgroups in flight")
