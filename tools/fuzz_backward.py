"""Fuzz: row-block backward algorithms vs the atomic scatter over ragged batch sizes, both fields, two table layouts (run on the GPU box)."""
import sys, ctypes, itertools
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import helpers as H
from oracle import raymarch as rm
from ucnerf_amd import _lib
lib = _lib.load()
bad = 0
for n, kind, seed in itertools.product((1, 3, 255, 257, 1023, 4097), ("tiny", "tinyR"), (0, 1)):
    spec = rm.make_spec(kind)
    sd = rm.init_state(spec, seed=seed)
    rays = rm.synthetic_rays(n, seed=seed + 7)
    noise = [rm.draw_level_noise(spec, n, l, False, torch.Generator().manual_seed(5 + l)) for l in range(2)]
    model, _ = H.hip_model(spec, sd)
    with torch.no_grad():
        _, hist = model(False, H.pin_noise(H.to_dev(rays), noise), 1.0, True)
    for mlp, h in ((model.prop_mlp_0, hist[0]), (model.nerf_mlp, hist[1])):
        enc = mlp.encoder
        sdist = h["sdist"].reshape(n, -1).contiguous()
        S = sdist.shape[-1] - 1
        b = H.to_dev(rays)
        flat = {k: b[k].reshape(n, -1).contiguous() for k in ("origins", "directions", "cam_dirs", "radii", "near", "far")}
        basis = torch.empty(n, 6, device="cuda")
        lvl = 0 if mlp is model.prop_mlp_0 else 1
        _lib.check(lib.ucn_cone_basis(flat["cam_dirs"].data_ptr(), noise[lvl].rand_vec.cuda().contiguous().data_ptr(), n, basis.data_ptr(), _lib.stream()))
        L, C = enc.num_levels, enc.level_dim
        g0 = torch.randn(L, n * S, C, device="cuda")
        g1 = g0.permute(1, 0, 2).contiguous()
        ws = torch.empty(lib.ucn_march_features_backward_ws_floats(ctypes.byref(mlp.field()), n, S), device="cuda")
        def run(lpb, layout, g, work=None):
            out = torch.zeros_like(enc.embeddings)
            _lib.check(lib.ucn_march_features_backward(ctypes.byref(mlp.field()), sdist.data_ptr(), flat["near"].data_ptr(), flat["far"].data_ptr(),
                flat["origins"].data_ptr(), flat["directions"].data_ptr(), basis.data_ptr(), flat["radii"].data_ptr(), None, None,
                float(model.std_scale), n, S, lpb, layout, g.data_ptr(), out.data_ptr(), _lib.ptr(work), _lib.stream()))
            return out
        want = run(1, 0, g0)
        tol = 3e-5 * float(want.abs().max())
        for name, got in (("cmp0", run(0, 0, g0, ws)), ("cmp1", run(0, 1, g1, ws)), ("blk", run(0, 0, g0))):
            d = float((got - want).abs().max())
            if not d <= tol:
                bad += 1
                print("MISMATCH", n, kind, seed, lvl, name, d, tol)
print("fuzz done, mismatches:", bad)
