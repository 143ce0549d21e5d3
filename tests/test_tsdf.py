"""TSDF fusion (reference tsdf.py:31-219) -- numpy oracle and HIP kernel against the golden of the
reference's own TSDF class (tests/golden/tsdf.npz, made by tests/golden/make_tsdf_golden.py).

Parity bar.  The update is a running mean of clamped signed distances; a voxel's sample pixel is found by
round-to-nearest of a projected coordinate, so a 1-ulp difference of the projection (the reference's two batched
GEMMs run through BLAS, fused or not per host) can move a voxel ON a pixel boundary to the neighbouring pixel.
Those voxels are identified (projection within 1e-4 px of a boundary in any view) and excluded; all others must agree
to 2e-6 (values), exactly (weights) and 2e-6 (colours)."""
import types

import numpy as np
import pytest
import torch

import helpers as H
from oracle import tsdf as otsdf


def _boundary_voxels(fx):
    world, c2w, K = fx["voxel_world_coords"].numpy()[0].astype(np.float64), fx["in_c2w"].numpy().astype(np.float64), fx["in_K"].numpy().astype(np.float64)
    Hh, Ww = fx["in_depth"].shape[-2:]
    risky = np.zeros(world.shape[1], bool)
    for i in range(c2w.shape[0]):
        cam = np.linalg.inv(c2w[i]) @ world
        cam[1], cam[2] = -cam[1], -cam[2]
        with np.errstate(divide="ignore", invalid="ignore"):
            pix = K @ (cam[:3] / cam[2])
        for p, n in ((pix[0], Ww), (pix[1], Hh)):
            u = p - 0.5                                   # unnormalised sample coordinate
            frac = np.abs(u - np.floor(u) - 0.5)
            risky |= (frac < 1e-4) | ~np.isfinite(p)
        trunc = float(fx["truncation"])
        risky |= np.abs(cam[2]) < 1e-6
    return risky


def test_tsdf_oracle_matches_the_reference_class():
    fx = H.load("tsdf.npz")
    coords, world, vs = otsdf.volume(2.0, 24)
    assert np.array_equal(coords, fx["voxel_coords"].numpy()) and float(vs) == float(fx["voxel_size"])
    ref_world = fx["voxel_world_coords"].numpy()          # corners of the radius-2 cube sit at the contraction's pole: 1.7e7
    assert (np.abs(world - ref_world) <= 2e-6 * np.maximum(1.0, np.abs(ref_world))).all()
    world = fx["voxel_world_coords"].numpy()
    n = world.shape[2]
    values, weights, colors = np.ones(n, np.float32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
    c2w, K, depth, color = (fx["in_" + k].numpy() for k in ("c2w", "K", "depth", "color"))
    ok = ~_boundary_voxels(fx)
    assert ok.mean() > 0.99
    otsdf.integrate(world, c2w[:3], K, depth[:3], color[:3], float(fx["truncation"]), values, weights, colors)
    assert np.array_equal(weights[ok], fx["weights_after3"].numpy()[ok])
    assert np.abs(values[ok] - fx["values_after3"].numpy()[ok]).max() <= 2e-6
    otsdf.integrate(world, c2w[3:], K, depth[3:], color[3:], float(fx["truncation"]), values, weights, colors)
    assert np.array_equal(weights[ok], fx["weights"].numpy()[ok]) and (weights > 0).sum() > 1000
    assert np.abs(values[ok] - fx["values"].numpy()[ok]).max() <= 2e-6
    assert np.abs(colors[ok] - fx["colors"].numpy()[ok]).max() <= 2e-6


@pytest.mark.gpu
def test_tsdf_kernel_matches_the_reference_class():
    from ucnerf_amd.internal.tsdf import TSDF
    fx = H.load("tsdf.npz")
    cfg = types.SimpleNamespace(tsdf_radius=2.0, tsdf_resolution=24, truncation_margin=5.0, tsdf_max_radius=10.0)
    acc = types.SimpleNamespace(device=torch.device("cuda", 0), num_processes=1, process_index=0, is_main_process=True)
    vol = TSDF(cfg, acc)
    assert torch.equal(vol.voxel_coords.cpu(), fx["voxel_coords"])
    refw = fx["voxel_world_coords"]
    assert bool(((vol.voxel_world_coords.cpu() - refw).abs() <= 2e-6 * refw.abs().clamp_min(1.0)).all())
    vol.voxel_world_coords = fx["voxel_world_coords"].cuda().contiguous()       # identical geometry for the comparison
    ok = torch.from_numpy(~_boundary_voxels(fx))
    c2w, K, depth, color = (fx["in_" + k].cuda() for k in ("c2w", "K", "depth", "color"))
    vol.integrate_tsdf(c2w[:3], K, depth[:3], color[:3])
    assert torch.equal(vol.weights.cpu()[ok], fx["weights_after3"][ok])
    assert float((vol.values.cpu() - fx["values_after3"])[ok].abs().max()) <= 2e-6
    vol.integrate_tsdf(c2w[3:], K, depth[3:], color[3:])                       # the running mean continues across calls
    assert torch.equal(vol.weights.cpu()[ok], fx["weights"][ok])
    assert float((vol.values.cpu() - fx["values"])[ok].abs().max()) <= 2e-6
    assert float((vol.colors.cpu() - fx["colors"])[ok].abs().max()) <= 2e-6
    # depth only (color_images=None) leaves the colour volume alone
    vol2 = TSDF(cfg, acc)
    vol2.integrate_tsdf(c2w, K, depth)
    assert float(vol2.colors.abs().max()) == 0.0 and float((vol2.values.cpu() - fx["values"])[ok].abs().max()) <= 2e-6
    # a 256^3 volume, 8 views of 1280 x 1920: size-independent properties -- weights are integer counts <= views, values in [-1, 1]
    big = TSDF(types.SimpleNamespace(tsdf_radius=2.0, tsdf_resolution=256, truncation_margin=5.0, tsdf_max_radius=10.0), acc)
    g = torch.Generator(device="cuda").manual_seed(0)
    d8 = torch.rand(8, 1, 320, 480, device="cuda", generator=g) * 3 + 0.5
    c8 = torch.eye(4, device="cuda")[None].repeat(8, 1, 1)
    c8[:, 2, 3] = torch.linspace(1.5, 3.0, 8, device="cuda")
    Kb = torch.tensor([[400.0, 0, 240], [0, 400.0, 160], [0, 0, 1]], device="cuda")
    big.integrate_tsdf(c8, Kb, d8)
    torch.cuda.synchronize()
    assert float(big.weights.max()) <= 8 and torch.equal(big.weights, big.weights.round())
    assert float(big.values.min()) >= -1.0 and float(big.values.max()) <= 1.0 and int((big.weights > 0).sum()) > 10000
