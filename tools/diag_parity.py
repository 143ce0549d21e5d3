"""Print HIP-vs-golden max differences per quantity (debug aid; not a test)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import helpers as H
from oracle import raymarch as rm
from ucnerf_amd import _lib

lib = _lib.load()
dev = lambda t: t.cuda().contiguous()

fx = H.load("field.npz")
spec = rm.make_spec("tiny")
sd = H.state_for(fx, spec)
model, _ = H.hip_model(spec, sd)
means, stds, vd = dev(fx["means"]), dev(fx["stds"]), dev(fx["viewdirs"])
for name, mlp in (("nerf", model.nerf_mlp), ("prop", model.prop_mlp_0)):
    # features alone
    d = mlp.field()
    B = means.shape[0] * means.shape[1]
    L, C = mlp.encoder.num_levels, mlp.encoder.level_dim
    feat = torch.empty(L * B * C, device="cuda")
    _lib.check(lib.ucn_points_features(ctypes.byref(d), means.reshape(-1, 3).contiguous().data_ptr(), stds.reshape(-1).contiguous().data_ptr(), B, 6, 1, 1, feat.data_ptr(), None, _lib.stream()))
    f = feat.reshape(L, B, C).permute(1, 0, 2).reshape(B, L * C).cpu()
    want = fx[f"{name}_features"].reshape(B, L * C)
    dd = (f - want).abs()
    print(name, "features maxdiff", float(dd.max()), "per-level", [float(x) for x in dd.reshape(B, L, C).amax(dim=(0, 2))])
    res = mlp(False, means, stds, viewdirs=vd)
    for k in ("density", "coord", "rgb"):
        print(name, k, H.maxdiff(res[k].cpu(), fx[f"{name}_{k}"]))
raw, x, coord = model.nerf_mlp.predict_density(means, stds)
print("nerf raw", H.maxdiff(raw.cpu(), fx["nerf_raw_density"]), "bottleneck", H.maxdiff(x.cpu(), fx["nerf_bottleneck"]))

for name, kind, over in [("model_tiny.npz", "tiny", {}), ("model_tinyR.npz", "tinyR", {}),
                         ("model_sky.npz", "tiny", dict(model_sky=True, brightness_correction=True)),
                         ("model_train.npz", "tiny", {})]:
    fx = H.load(name)
    spec = rm.make_spec(kind, **over)
    sd = H.state_for(fx, spec)
    model, cfg = H.hip_model(spec, sd)
    train = "noise0_jitter" in fx
    noise = H.noise_of(fx, spec.num_levels)
    batch = H.pin_noise(H.to_dev(H.batch_of(fx)), noise)
    cam = fx.get("eval_camidx")
    rend, hist = model(train, batch, float(fx["train_frac"]), not train, zero_glo=not train, eval_camidx=None if cam is None else cam.cuda())
    for lvl in range(spec.num_levels):
        out = []
        for k in ("sdist", "density", "rgb", "coord", "weights"):
            out.append(f"h.{k}={H.maxdiff(hist[lvl][k].cpu().reshape(-1), fx[f'L{lvl}_hist_{k}'].reshape(-1)):.2e}")
        for k in rend[lvl]:
            if f"L{lvl}_{k}" in fx and torch.is_tensor(rend[lvl][k]):
                try:
                    out.append(f"{k}={H.maxdiff(rend[lvl][k].cpu().reshape(-1), fx[f'L{lvl}_{k}'].reshape(-1)):.2e}")
                except AssertionError as e:
                    out.append(f"{k}=ERR({e})")
        print(name, lvl, " ".join(out))
