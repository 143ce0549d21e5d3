"""Fuzz over GRID SHAPES: the row-block table gradient (levels_per_block = 0: persistent workgroups, point / sample items, the
> 32-block path) against the global-atomic kernel (levels_per_block = 1) for random (L, C, T, resolutions, N, S) -- few and many
row blocks per level, dense / hashed / strided-wrap levels, ragged sample counts, zero-gradient samples, rays that leave the
unit cube.  GPU box:  python tools/fuzz_backward_grids.py [n_cases]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ucnerf_amd import _lib
from ucnerf_amd.gridencoder import GridEncoder
lib = _lib.load()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(os.environ.get("UCN_FUZZ_SEED", "0")))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for case in range(n_cases):
    L = int(rng.integers(1, 11))
    C = int(rng.choice([1, 2, 4, 8]))
    T = int(rng.choice([12, 15, 17, 19, 20, 21] if C <= 4 else [12, 15, 17, 19]))
    base = int(rng.choice([4, 16, 32]))
    desired = int(base * 2 ** rng.integers(1, 14)) if L > 1 else None
    enc = GridEncoder(num_levels=L, level_dim=C, base_resolution=base, log2_hashmap_size=T, desired_resolution=desired).to(dev)
    N = int(rng.choice([1, 3, 65, 700, 2048, 5000]))
    S = int(rng.choice([1, 7, 32, 64, 128]))
    g = torch.Generator(device=dev).manual_seed(case)
    d = _lib.UcnField()
    d.embeddings = enc.embeddings.data_ptr()
    d.offsets_host, d.grid_sizes_host = enc._offsets_np.ctypes.data, enc._sizes_np.ctypes.data
    d.num_levels, d.level_dim, d.base_resolution = L, C, base
    d.log2_per_level_scale = float(np.log2(enc.per_level_scale))
    # rays: origins near the centre, some far outside (their contracted samples stay inside; a few leave [0, 1] by rounding)
    origins = (torch.rand(N, 3, device=dev, generator=g) - 0.5) * float(rng.choice([0.5, 4.0]))
    dirs = torch.nn.functional.normalize(torch.randn(N, 3, device=dev, generator=g), dim=-1) * (0.5 + torch.rand(N, 1, device=dev, generator=g))
    cam = torch.nn.functional.normalize(torch.randn(N, 3, device=dev, generator=g), dim=-1)
    rvec = torch.randn(N, 3, device=dev, generator=g)
    basis = torch.empty(N, 6, device=dev)
    _lib.check(lib.ucn_cone_basis(cam.contiguous().data_ptr(), rvec.contiguous().data_ptr(), N, basis.data_ptr(), _lib.stream()))
    near = torch.full((N,), 0.0, device=dev); far = torch.full((N,), float(rng.choice([2.0, 8.0, 60.0])), device=dev)
    radii = torch.rand(N, device=dev, generator=g) * 2e-3 + 1e-4
    sdist = torch.sort(torch.rand(N, S + 1, device=dev, generator=g), dim=-1).values.contiguous()
    grad = torch.randn(L, N * S, C, device=dev, generator=g)
    grad[:, torch.rand(N * S, device=dev, generator=g) < 0.2] = 0.0
    ws = torch.empty(max(1, lib.ucn_march_features_backward_ws_floats(ctypes.byref(d), N, S)), device=dev)

    def run(lpb, work, flags=0):
        out = torch.zeros_like(enc.embeddings)
        _lib.check(lib.ucn_march_features_backward(ctypes.byref(d), sdist.data_ptr(), near.data_ptr(), far.data_ptr(), origins.contiguous().data_ptr(),
                                                   dirs.contiguous().data_ptr(), basis.data_ptr(), radii.data_ptr(), None, None, 0.5, N, S, lpb, flags,
                                                   grad.data_ptr(), out.data_ptr(), _lib.ptr(work), _lib.stream()))
        torch.cuda.synchronize()
        return out
    want = run(1, None)
    got = run(0, ws)
    # a level whose resolution^2 wraps negative in int32 has a NaN damping factor in the reference too (models.py:495: sqrt of the wrapped
    # int32 square; grid_op.hip keeps that): such rows must be non-finite on BOTH routes, the rest is compared
    fin = torch.ones_like(want, dtype=torch.bool)
    nan_levels = []
    for l in range(L):
        r = np.int64(enc._sizes_np[l])
        if np.int32(np.uint32((r * r) & 0xFFFFFFFF)) < 0:                       # what torch's int32 square holds (models.py:495)
            fin[int(enc._offsets_np[l]):int(enc._offsets_np[l + 1])] = False
            nan_levels.append(l)
    same_pattern = bool(torch.isfinite(got)[fin].all()) and bool(torch.isfinite(want)[fin].all())
    big = max(float(want[fin].abs().max()) if bool(fin.any()) else 0.0, 1e-30)
    tol = 3e-5 * big
    err = (float((got - want)[fin].abs().max()) if bool(fin.any()) else 0.0) if same_pattern else float("inf")
    rows = [int(enc._offsets_np[l + 1] - enc._offsets_np[l]) for l in range(L)]
    rpb = 128 * 1024 // (C * 4)
    tag = f"case {case:3d}: L {L:2d} C {C} T 2^{T} base {base} desired {desired} N {N} S {S}  max blocks/level {max((r + rpb - 1) // rpb for r in rows):4d}"
    # r06: the FIXED-POINT row blocks (the autocast step's route: its own cut between sample and point items, byte planes everywhere).
    # Resolution 2^-30 ... 2^-29 of a task's summed |g| per addend: held to 1e-3 of the largest entry
    err_fx = 0.0
    if C % 2 == 0:
        got_fx = run(0, ws, _lib.BWD_FIXED_POINT)
        err_fx = (float((got_fx - want)[fin].abs().max()) if bool(fin.any()) else 0.0) if bool(torch.isfinite(got_fx)[fin].all()) else float("inf")
    tol_fx = 1e-3 * big
    if not err <= tol or not err_fx <= tol_fx:
        bad += 1
        print("MISMATCH", tag, err, tol, err_fx, tol_fx, "non-finite entries: atomic kernel", int((~torch.isfinite(want)).sum()),
              "row blocks", int((~torch.isfinite(got)).sum()), "resolutions", [int(r) for r in enc._sizes_np[:L]] if hasattr(enc, "_sizes_np") else "")
    else:
        print("ok      ", tag, f"err {err:.2e} (fixed-point rows {err_fx:.2e}) of {big:.2e}" + ("" if not nan_levels else f"  [levels {nan_levels}: NaN damping in the reference too, not compared]"))
print("fuzz done, mismatches:", bad)
sys.exit(1 if bad else 0)
