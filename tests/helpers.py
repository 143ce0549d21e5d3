"""Shared helpers: fixture loading and oracle drivers used by both the CPU and the GPU tests."""
import os

import numpy as np
import torch

from oracle import raymarch as rm

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# fp32 parity bar of BASELINE.json.north_star: 1e-4 L-inf on RGB.  Intermediate quantities are
# held to tighter, per-quantity tolerances stated next to each assert.
RGB_TOL = 1e-4


def load(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def state_checksum(sd):
    return float(sum(v.double().abs().sum() for k, v in sd.items() if v.is_floating_point()))


def state_for(fx, spec):
    """Regenerate the fixture's weights from its seed and verify the checksum it recorded."""
    sd = rm.init_state(spec, seed=int(fx["seed"]))
    if "sky_alpha_bias" in fx:               # fixtures whose sky density head was lifted so that the sky network is alive (make_golden.py)
        sd["skynerf.alpha_linear.bias"] = sd["skynerf.alpha_linear.bias"] + float(fx["sky_alpha_bias"])
    got = state_checksum(sd)
    want = float(fx["checksum"])
    assert abs(got - want) <= 1e-9 * abs(want), "torch RNG drifted: fixture weights not reproducible"
    return sd


def batch_of(fx, flat=True):
    keys = [k[4:] for k in fx if k.startswith("ray_")]
    b = {k: fx["ray_" + k] for k in keys}
    return b


def noise_of(fx, num_levels):
    out = []
    for lvl in range(num_levels):
        g = lambda k: fx.get(f"noise{lvl}_{k}")
        sq = lambda t: None if t is None else t.reshape(t.shape[0], -1)
        out.append(rm.LevelNoise(rand_vec=sq(g("rand_vec")), jitter=sq(g("jitter")), flip=sq(g("flip")),
                                 spin=sq(g("spin"))))
    return out


def maxdiff(a, b):
    a = torch.as_tensor(a).double().reshape(-1)
    b = torch.as_tensor(b).double().reshape(-1)
    assert a.shape == b.shape, (a.shape, b.shape)
    # NaN / inf must appear in the same places (the reference yields NaN for a zero-width
    # interval at t = 0, render.py:116); everywhere else compare values.
    assert torch.equal(torch.isnan(a), torch.isnan(b)), "NaN pattern differs"
    same = (torch.isinf(a) & torch.isinf(b) & (torch.sign(a) == torch.sign(b))) | torch.isnan(a)
    d = (a - b).abs()
    d[same] = 0
    return float(d.max()) if d.numel() else 0.0


# ------------------------------------------------------------------ HIP-side helpers (gpu tests)
def hip_model(spec, sd, device="cuda", **model_kw):
    """ucnerf_amd Model configured like an oracle PathSpec and loaded with an oracle state dict."""
    from ucnerf_amd.internal import configs, models

    def fkw(fs):
        return dict(grid_disired_resolution=fs.grid_desired_resolution, grid_level_dim=fs.grid_level_dim,
                    grid_log2_hashmap_size=fs.grid_log2_hashmap_size, bottleneck_width=fs.bottleneck_width,
                    net_width_viewdirs=fs.net_width_viewdirs)
    cfg = configs.Config(model_sky=spec.model_sky, brightness_correction=spec.brightness_correction,
                         training_views=spec.training_views, vis_num_rays=spec.vis_num_rays)
    with models.bindings(NerfMLP=fkw(spec.nerf), PropMLP=fkw(spec.props[0])):
        model = models.Model(config=cfg, num_levels=spec.num_levels, num_prop_samples=spec.num_prop_samples,
                             num_nerf_samples=spec.num_nerf_samples, opaque_background=spec.opaque_background,
                             prop_desired_grid_size=list(spec.prop_desired_grid_size),
                             dilation_bias=spec.dilation_bias, dilation_multiplier=spec.dilation_multiplier, **model_kw)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith(".idx") for k in missing), missing
    return model.to(device).eval(), cfg


def to_dev(batch, device="cuda"):
    return {k: v.to(device) for k, v in batch.items()}


def pin_noise(batch, noise, device="cuda"):
    """Feed the oracle's random draws to the HIP Model through its optional batch keys."""
    b = dict(batch)
    b["rand_vec"] = torch.cat([nz.rand_vec for nz in noise], dim=-1).to(device)
    if noise[0].jitter is not None:
        b["march_noise"] = [dict(jitter=nz.jitter.to(device), flip=nz.flip.to(device), spin=nz.spin.to(device))
                            for nz in noise]
    return b
