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
