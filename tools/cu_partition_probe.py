"""What would a DISJOINT split of the CUs between the NeRF-level gather and the NeRF-level MLP buy?  CU masks
are not honoured on this stack, so k CUs are taken away with hog workgroups (tools/hog.hip: each reserves 150 KiB of LDS and sleeps), and
one pass (10,240 rays x 128 samples) of each kernel is timed on the remaining 256 - k CUs.  The gather is launched with 16 KiB of dummy
LDS per workgroup so that it cannot sit beside a hog; the MLP's 64 KiB rings cannot either.
    python tools/cu_partition_probe.py      (GPU box; UCN_FEAT_DUMMY_LDS is set by the script)"""
import ctypes, os, sys, time
os.environ["UCN_FEAT_DUMMY_LDS"] = "16384"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from ucnerf_amd import _lib
lib = _lib.load()
_hog = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_exp", "libhog.so")
if not os.path.exists(_hog):                    # tools/_exp/ is not tracked: build the hog on first use
    import subprocess
    os.makedirs(os.path.dirname(_hog), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-w",
                           os.path.join(os.path.dirname(os.path.abspath(__file__)), "hog.hip"), "-o", _hog])
hog = ctypes.CDLL(_hog)
hog.hog_launch.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda", 0)
model, cfg, sd = bench.build_model(dev)
batch = bench.frame_rays(dev)
n, S = 10240, 128
idx = bench_idx = None
from ucnerf_amd.internal import models
perm, inv = models._tile_order(bench.H_IMG, bench.W_IMG, 8, dev)
flat = {k: v.reshape(-1, v.shape[-1]).index_select(0, perm)[:n].contiguous() for k, v in batch.items()}
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
L, C = enc.num_levels, enc.level_dim
feat = torch.empty(L * n * S * C, device=dev)
desc = mlp.field()
from ucnerf_amd.internal.models import _dir_tiles
vd = torch.nn.functional.normalize(flat["viewdirs"], dim=-1)
density = torch.empty(n, S, device=dev)
rgbs = torch.empty(n, S, 3, device=dev)
dirb = torch.empty(lib.ucn_field_dir_floats(ctypes.byref(desc), n), device=dev)
_lib.check(lib.ucn_field_dir_bias(ctypes.byref(desc), vd.contiguous().data_ptr(), n, dirb.data_ptr(), _lib.stream()))
side = torch.cuda.Stream()
hogs = torch.cuda.Stream()
where = torch.zeros(512, dtype=torch.int32, device=dev)


def gather(st):
    _lib.check(lib.ucn_march_features(ctypes.byref(desc), sdist.data_ptr(), near.data_ptr(), far.data_ptr(), flat["origins"].data_ptr(),
                                      flat["directions"].data_ptr(), basis.data_ptr(), rad.data_ptr(), None, None, float(model.std_scale), n, S, 0, 2,
                                      feat.data_ptr(), None, None, st.cuda_stream))


def mlp_pass(st):
    _lib.check(lib.ucn_field_mlp(ctypes.byref(desc), feat.data_ptr(), n * S, S, 1, dirb.data_ptr(), density.data_ptr(), rgbs.data_ptr(), None, st.cuda_stream))


def timed(fn, k, reps=4):
    torch.cuda.synchronize()
    if k:
        assert hog.hog_launch(k, 150 * 1024, 40.0, where.data_ptr(), ctypes.c_void_p(hogs.cuda_stream)) == 0
        time.sleep(0.005)                       # the hogs are resident before the timed kernels are enqueued
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        fn(side)
        e0.record(side)
        for _ in range(reps):
            fn(side)
        e1.record(side)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def both(k, reps=4):
    """gather on `side`, MLP on the current stream, concurrently, with k hogs: the time until both are done"""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cur = torch.cuda.current_stream()
    for _ in range(reps):
        gather(side)
        mlp_pass(cur)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / reps


for _ in range(2):
    gather(side); mlp_pass(side)
torch.cuda.synchronize()
print("hogged CUs | gather pass ms | MLP pass ms   (one pass = 10,240 rays x 128 samples; 240 passes per frame)")
for k in (0, 32, 64, 96, 128, 160, 192):
    g, m = timed(gather, k), timed(mlp_pass, k)
    xcc = torch.bincount(where[:k].long(), minlength=8).tolist() if k else []
    print(f"{k:10d} | {g:14.3f} | {m:11.3f}   hogs per XCD {xcc}")
print("both kernels of a pass launched together on two streams, no hogs (the dispatcher's own interleaving):", f"{both(0):.3f} ms per pass; sequential sum above")
