"""oracle/raymarch.py against the fixtures the REFERENCE's own Python produced
(tests/golden/make_golden.py).  This is what pins the oracle: the reference has no tests
of its own for this path.  Tolerances: the generating run is torch-CPU fp32; a different host
may take a different SIMD path inside exp/log/erf/sin, so equality is held to 2e-6 (abs, on
O(1) quantities) rather than bit-for-bit.  In the authoring container the differences are 0.
"""
import pytest
import torch

import helpers as H
from oracle import raymarch as rm

TOL = 2e-6


def test_g1_g2_step_functions():
    fx = H.load("stepfun.npz")
    td, wd = rm.dilate_weights(fx["t"], fx["w"], float(fx["dilation"]), 0.0, 1.0)
    assert H.maxdiff(td, fx["t_dilate"]) <= TOL
    assert H.maxdiff(wd, fx["w_dilate"]) <= TOL
    t_in = fx["t_dilate"][..., 1:-1]
    for frac in ("1.0", "0.25"):
        got = rm.sample_fenceposts(t_in, fx[f"logits_{frac}"], 128, 0.0, 1.0)
        assert H.maxdiff(got, fx[f"sample_eval_{frac}"]) <= TOL
        assert (got[..., 1:] >= got[..., :-1]).all()
    got = rm.sample_fenceposts(t_in, fx["logits_0.25"], 32, 0.0, 1.0, jitter=fx["sample_train_jitter"])
    assert H.maxdiff(got, fx["sample_train"]) <= TOL
    t0 = torch.tensor([[0.0, 1.0]]).expand(fx["t"].shape[0], 2)
    got = rm.sample_fenceposts(t0, torch.zeros(fx["t"].shape[0], 1), 64, 0.0, 1.0)
    assert H.maxdiff(got, fx["sample_level0"]) <= TOL
    assert H.maxdiff(rm.percentiles_of_stepfun(fx["pct_t"], fx["pct_w"]), fx["pct"]) <= 8 * TOL


def test_g3_g4_cone_cast_and_contraction():
    fx = H.load("cast.npz")
    args = [fx[k] for k in ("tdist", "origins", "directions", "cam_dirs", "radii")]
    m, s, t = rm.cone_multisamples(*args, fx["eval_rand_vec"], 0.5)
    assert H.maxdiff(m, fx["eval_means"]) <= 8 * TOL      # |means| up to ~8
    assert H.maxdiff(s, fx["eval_stds"]) <= TOL
    assert H.maxdiff(t, fx["eval_t"]) <= 8 * TOL
    m, s, t = rm.cone_multisamples(*args, fx["train_rand_vec"], 0.5, flip=fx["train_flip"], spin=fx["train_spin"])
    assert H.maxdiff(m, fx["train_means"]) <= 8 * TOL
    assert H.maxdiff(s, fx["train_stds"]) <= TOL
    cm, cs = rm.contract_points(fx["contract_in_mean"], fx["contract_in_std"])
    assert H.maxdiff(cm, fx["contract_mean"]) <= TOL
    assert H.maxdiff(cs, fx["contract_std"]) <= TOL
    assert (cm.norm(dim=-1) <= 2 + 1e-6).all()


def test_g5_field_on_small_grid():
    fx = H.load("field.npz")
    spec = rm.make_spec("tiny")
    sd = H.state_for(fx, spec)
    for name, fs in (("nerf", spec.nerf), ("prop", spec.props[0])):
        raw, x, coord, feat = rm.field_density_features(fs, sd, fx["means"], fx["stds"])
        assert H.maxdiff(feat, fx[f"{name}_features"]) <= TOL
        assert H.maxdiff(raw, fx[f"{name}_raw_density"]) <= 4 * TOL
        assert H.maxdiff(x, fx[f"{name}_bottleneck"]) <= 4 * TOL
        res = rm.field_forward(fs, sd, fx["means"], fx["stds"], fx["viewdirs"])
        assert H.maxdiff(res["density"], fx[f"{name}_density"]) <= 4 * TOL
        assert H.maxdiff(res["rgb"], fx[f"{name}_rgb"]) <= 4 * TOL
        assert H.maxdiff(res["coord"], fx[f"{name}_coord"]) <= TOL
    raw, _, _, _ = rm.field_density_features(spec.nerf, sd, fx["nowarp_means"], fx["nowarp_stds"], no_warp=True)
    assert H.maxdiff(raw, fx["nowarp_raw_density"]) <= 4 * TOL
    # the int32 wrap of grid_sizes**2 (models.py:495) is part of the contract
    _, _, sizes, _ = spec.nerf.layout()
    assert int((sizes ** 2)[12]) == 131073 and int((sizes ** 2)[15]) == 1048577


def test_g6_alpha_and_composite():
    fx = H.load("composite.npz")
    w = rm.alpha_weights(fx["density"], fx["tdist"], fx["dirs"])
    assert H.maxdiff(w, fx["weights"]) <= TOL
    assert H.maxdiff(rm.alpha_weights(fx["density"], fx["tdist"], fx["dirs"], True), fx["weights_opaque"]) <= TOL
    out = rm.composite(fx["rgbs"], w, fx["tdist"], 1.0, fx["far"], extras=True)
    for k, v in out.items():
        assert H.maxdiff(v, fx["out_" + k]) <= (300 * TOL if k == "depth" else 8 * TOL), k
    assert (out["depth"][out["acc"] < 0.6] == 300).all() and (out["acc"] < 0.6).any() and (out["acc"] >= 0.6).any()


@pytest.mark.parametrize("name,kind,over", [
    ("model_tiny.npz", "tiny", {}),
    ("model_tinyR.npz", "tinyR", {}),
    ("model_tiny64.npz", "tiny64", {}),
    ("model_sky.npz", "tiny", dict(model_sky=True, brightness_correction=True)),
    ("model_train.npz", "tiny", {}),
    ("model_nodilate.npz", "tiny", dict(dilation_bias=0., dilation_multiplier=0.)),
])
def test_g7_g8_model_forward(name, kind, over):
    fx = H.load(name)
    spec = rm.make_spec(kind, **over)
    sd = H.state_for(fx, spec)
    train = "noise0_jitter" in fx
    batch = H.batch_of(fx)
    with torch.no_grad():
        rend, hist = rm.model_forward(spec, sd, batch, H.noise_of(fx, spec.num_levels),
                                      train_frac=float(fx["train_frac"]), compute_extras=not train,
                                      eval_camidx=fx.get("eval_camidx"), training=train)
    for lvl in range(spec.num_levels):
        for k, v in rend[lvl].items():
            want = fx[f"L{lvl}_{k}"]
            assert H.maxdiff(v.reshape(want.shape), want) <= (300 * TOL if k == "depth" else 8 * TOL), (lvl, k)
        for k in ("sdist", "weights", "density", "rgb", "coord"):
            want = fx[f"L{lvl}_hist_{k}"]
            assert H.maxdiff(hist[lvl][k].reshape(want.shape), want) <= 8 * TOL, (lvl, k)
        if train:
            assert H.maxdiff(hist[lvl]["loss_hash_decay"], fx[f"L{lvl}_hist_loss_hash_decay"]) <= TOL


def test_g9_render_image_chunks():
    fx = H.load("render_image.npz")
    spec = rm.make_spec("tiny")
    sd = H.state_for(fx, spec)
    batch = H.batch_of(fx)
    n = batch["origins"].shape[0]
    chunk = int(fx["chunk"])
    outs = []
    with torch.no_grad():
        for i0 in range(0, n, chunk):
            sub = {k: v[i0:i0 + chunk] for k, v in batch.items()}
            noise = [rm.LevelNoise(rand_vec=fx[f"noise{l}_rand_vec"][i0:i0 + chunk]) for l in range(spec.num_levels)]
            rend, _ = rm.model_forward(spec, sd, sub, noise)
            outs.append(rend[-1])
    Hh, Ww = int(fx["H"]), int(fx["W"])
    for k in [k[4:] for k in fx if k.startswith("out_")]:
        got = torch.cat([o[k] for o in outs]).reshape((Hh, Ww) + outs[0][k].shape[1:])
        assert H.maxdiff(got, fx["out_" + k]) <= (300 * TOL if k == "depth" else 8 * TOL), k


def test_rays_oracle_vs_reference():
    """oracle/rays.py (numpy float64 restatement of camera_utils.pixels_to_rays + datasets._make_ray_batch) against the
    reference's own output (tests/golden/rays.npz, make_rays_golden.py): bit-exact after the float32 cast."""
    import numpy as np
    from oracle import rays
    fx = np.load(H.GOLDEN + "/rays.npz")
    for tag in ("frame", "batch"):
        b = rays.make_ray_batch(fx[f"{tag}.pix_x"], fx[f"{tag}.pix_y"], fx[f"{tag}.cam_idx"], fx["pixtocams"],
                                fx["camtoworlds"], 0.0, 8.0)
        for k in ("origins", "directions", "viewdirs", "radii", "imageplane", "cam_dirs"):
            assert b[k].dtype == np.float32 and np.array_equal(b[k], fx[f"{tag}.{k}"]), (tag, k)
        assert np.array_equal(b["cam_idx"][..., 0], fx[f"{tag}.cam_idx"].astype(np.float32))
        assert float(b["near"].max()) == 0.0 and float(b["far"].min()) == 8.0 and float(b["lossmult"].min()) == 1.0


WARP_CASES = ("up", "stereo", "forward", "rot_right")


def check_warp(fx, name, uv, mask, depth_tgt=None):
    """Shared by the oracle (here) and the HIP kernels (tests/test_warp.py): projected coordinates to a few float32 ulp
    of a pixel coordinate where the depth is valid; the mask up to pixels that sit ON a frame boundary (a pure shift
    keeps one coordinate exactly integral, so `u >= 0` at column 0 flips with the rounding of a BLAS call); the
    splatted depth on the non-degenerate poses, up to the same truncation ambiguity."""
    import numpy as np
    want, wm = fx[name + ".pts"], fx[name + ".mask"].astype(bool)
    H, W = wm.shape
    ok = fx["depth"] > 0
    assert float(np.abs(uv - want)[ok].max()) <= 2e-3, name
    edge = np.minimum.reduce([np.abs(want[..., 0]), np.abs(want[..., 0] - (W - 0.5)), np.abs(want[..., 1]),
                              np.abs(want[..., 1] - (H - 0.5))]) < 1e-2
    assert not ((np.asarray(mask).astype(bool) != wm) & ~edge).any(), name
    if depth_tgt is not None and name in ("forward", "rot_right"):
        wd = fx[name + ".depth_tgt"]
        assert int((np.abs(depth_tgt - wd) > 1e-4).sum()) <= 0.01 * int((wd != 0).sum()), name


def test_warp_oracle_vs_reference():
    """oracle/warp.py against the reference's own img_warping / img_warping_for_depth (tests/golden/warp.npz)."""
    import numpy as np
    from oracle import warp
    fx = np.load(H.GOLDEN + "/warp.npz")
    flip = np.diag([1., -1., -1., 1.])                                   # datasets.py:523-524
    for name in WARP_CASES:
        ref, src = fx["ref_pose"] @ flip, fx[name + ".src_pose"] @ flip
        uv, mask = warp.img_warping(ref, src, fx["depth"], fx["intrinsic"])
        check_warp(fx, name, uv, mask, warp.img_warping_for_depth(ref, src, fx["depth"], fx["intrinsic"]))


def test_live_sky_layer_overflows_by_the_references_own_formula():
    """The NaN of profiles/r02c/bench_cfg5_fit.json explained on the CPU, with the oracle that reproduces the
    reference's sky layer bit for bit (model_sky.npz): the sky samples run from z = batch.far DOWN to 1 / far (models.py:872, SURVEY.md
    Appendix C.2), so every spacing but the last is NEGATIVE, alpha = 1 - exp(+relu(sigma) |dz|) <= 0 and the transmittance
    prod(1 - alpha) GROWS like exp(sum sigma |dz|).  A sky density head that training has pushed to sigma ~ 12 over the ray
    (|dz| 0.066 x 119 samples x 12 = 95 > log(float32 max) = 88.7) makes rgb_map inf / NaN -- what the reference reports itself as
    `[Numerical Error] rgb_map contains nan or inf` (models.py:899-901).  Not an artefact of this implementation: a dead head
    (sigma <= 0, the default initialisation on these rays) renders exactly 0, a mildly live one stays finite."""
    spec = rm.make_spec("tiny", model_sky=True, brightness_correction=True)
    sd = rm.init_state(spec, seed=3)
    rays = rm.synthetic_rays(64, seed=4)
    w, b = sd["skynerf.alpha_linear.weight"], sd["skynerf.alpha_linear.bias"]
    out = {}
    for name, bias in (("dead", -5.0), ("mild", 0.5), ("live", 2.0), ("pushed", 12.0)):
        sd2 = dict(sd)
        sd2["skynerf.alpha_linear.weight"] = torch.zeros_like(w)           # sigma = bias everywhere
        sd2["skynerf.alpha_linear.bias"] = torch.full_like(b, bias)
        with torch.no_grad():
            out[name] = rm.sky_layer(sd2, rays["origins"], rays["directions"], rays["cam_dirs"], rays["far"])
    assert float(out["dead"].abs().max()) == 0.0
    assert bool(torch.isfinite(out["mild"]).all()) and float(out["mild"].abs().max()) > 0
    assert not bool(torch.isfinite(out["pushed"]).all())
    # well before the overflow the layer's "colour" has left [0, 1]: at sigma = 2 it is already in the thousands, of either sign
    # (119 negative weights alpha_i T_i against the one positive weight of the last sample, whose spacing is +1e10)
    assert bool(torch.isfinite(out["live"]).all()) and float(out["live"].abs().max()) > 100.0


@pytest.mark.parametrize("name", ["train_step.npz", "train_step_sky.npz"])
def test_oracle_losses_match_reference_values(name):
    """oracle/losses.py (the oracle side's own restatement of train_utils.py:149-305 / stepfun.py:297-307, 395-403 / math.py:110-133)
    on the reference's own per-level outputs stored in the G10 fixtures, against the loss values the imported reference produced for
    them (tests/golden/make_golden.py).  The product's O(S) forms are held to the same values by tests/test_train_step.py."""
    from oracle import losses as ol
    fx = H.load(name)
    hist = [dict(sdist=fx[f"L{l}_sdist"], weights=fx[f"L{l}_weights"]) for l in range(2)]
    rend = [dict(rgb=fx[f"L{l}_rgb"], weights=fx[f"L{l}_weights"]) for l in range(2)]
    batch = {k[4:]: v for k, v in fx.items() if k.startswith("ray_")}
    batch = {k: (v[:, None, None, :] if v.dim() == 2 else v[:, None, None]) for k, v in batch.items()}
    assert abs(float(ol.data_loss(batch, rend)) - float(fx["loss_data"])) <= 1e-6
    a = float(ol.anti_interlevel_loss(hist))
    assert abs(a - float(fx["loss_anti_interlevel"])) <= 2e-6 * max(1.0, float(fx["loss_anti_interlevel"])), (a, float(fx["loss_anti_interlevel"]))
    d = float(ol.distortion_loss(hist))
    assert abs(d - float(fx["loss_distortion"])) <= 1e-6 * max(1.0, float(fx["loss_distortion"])), (d, float(fx["loss_distortion"]))
    if "loss_sky" in fx:
        assert abs(0.002 * float(ol.sky_loss(batch, rend)) - float(fx["loss_sky"])) <= 1e-7
