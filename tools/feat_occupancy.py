"""Featurisation throughput vs resident waves per SIMD (one process per setting: the knob is read once).
UCN_FEAT_DUMMY_LDS = bytes of unused dynamic LDS per 256-thread workgroup: 160 KiB / that = workgroups per CU
= waves per SIMD (a workgroup is one wave per SIMD).  Usage: python tools/feat_occupancy.py [n_rays]"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def worker(n):
    import torch, bench
    from ucnerf_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    model, cfg, sd = bench.build_model(dev)
    batch = bench.frame_rays(dev)
    flat = {k: v.reshape(-1, v.shape[-1])[:n].contiguous() for k, v in batch.items()}
    flat["rand_vec"] = torch.randn(n, 6, device=dev)
    with torch.no_grad():
        r, h = model(False, flat, 1.0, True)
    S = 128
    sdist = h[-1]["sdist"].contiguous()
    mlp = model.nerf_mlp
    d = mlp.field()
    basis = torch.empty(n, 6, device=dev)
    _lib.check(lib.ucn_cone_basis(flat["cam_dirs"].data_ptr(), flat["rand_vec"][:, 3:6].contiguous().data_ptr(), n, basis.data_ptr(), _lib.stream()))
    near, far = flat["near"].reshape(-1).contiguous(), flat["far"].reshape(-1).contiguous()
    rad = flat["radii"].reshape(-1).contiguous()
    feat = torch.empty(n * S * 32, device=dev)
    args = (ctypes.byref(d), sdist.data_ptr(), near.data_ptr(), far.data_ptr(), flat["origins"].data_ptr(), flat["directions"].data_ptr(),
            basis.data_ptr(), rad.data_ptr(), None, None, 0.5, n, S, 0, 2, feat.data_ptr(), None, None, _lib.stream())
    for _ in range(3):
        _lib.check(lib.ucn_march_features(*args))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        _lib.check(lib.ucn_march_features(*args))
    e1.record(); torch.cuda.synchronize()
    print(f"dummy_lds {os.environ.get('UCN_FEAT_DUMMY_LDS', '0'):>7}  {e0.elapsed_time(e1) / 10:7.3f} ms per {n} rays")

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "worker":
        worker(int(sys.argv[2]))
    else:
        n = sys.argv[1] if len(sys.argv) > 1 else "10240"
        for lds, label in ((0, "4 waves/SIMD (register-limited)"), (53248, "3"), (81920, "2"), (163840, "1")):
            env = dict(os.environ, UCN_FEAT_DUMMY_LDS=str(lds))
            print(label, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "worker", n], env=env)
