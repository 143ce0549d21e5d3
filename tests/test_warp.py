"""Virtual-pose depth warp on the GPU vs the reference's golden vectors and the numpy oracle (-m gpu)."""
import os

import numpy as np
import pytest
import torch

from oracle import warp as oracle_warp
from test_oracle_golden import WARP_CASES, check_warp

pytestmark = pytest.mark.gpu
FLIP = np.diag([1., -1., -1., 1.])


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "warp.npz"))


@pytest.mark.parametrize("name", WARP_CASES)
def test_img_warping_vs_reference(fx, name):
    from ucnerf_amd.internal import train_utils as tu
    ref, src = fx["ref_pose"] @ FLIP, fx[name + ".src_pose"] @ FLIP
    pts, mask = tu.img_warping(ref, src, fx["depth"], fx["intrinsic"])
    assert pts.is_cuda and mask.dtype == torch.bool
    dt = tu.img_warping_for_depth(ref, src, fx["depth"], fx["intrinsic"])
    check_warp(fx, name, pts.cpu().numpy(), mask.cpu().numpy(), dt.cpu().numpy())


def test_full_frame_vs_oracle_and_sampler(fx):
    """BASELINE frame size (1280 x 1920 depth map, what datasets.py:527 warps every training step)."""
    from ucnerf_amd.internal import train_utils as tu
    H, W = 1280, 1920
    rng = np.random.default_rng(3)
    K = np.array([[2055.556149, 0.0, 939.657470], [0.0, 2055.556149, 641.072182], [0.0, 0.0, 1.0]])
    yy, xx = np.mgrid[0:H, 0:W]
    depth = (8.0 + 5.0 * np.sin(xx / 170.0) * np.cos(yy / 110.0) + rng.random((H, W))).astype(np.float32)
    depth[rng.random((H, W)) < 0.4] = 0.0
    ref = fx["ref_pose"] @ FLIP
    a = np.radians(-12.0)
    src = fx["ref_pose"].copy()
    src[:3, :3] = src[:3, :3] @ np.array([[np.cos(a), 0, -np.sin(a)], [0, 1, 0], [np.sin(a), 0, np.cos(a)]])
    src[:3, 3] += fx["ref_pose"][:3, :3] @ np.array([0.2, 0.3, 0.1])
    src = src @ FLIP
    pts, mask = tu.img_warping(ref, src, depth, K)
    want, wm = oracle_warp.img_warping(ref, src, depth, K)
    ok = depth > 0
    assert float(np.abs(pts.cpu().numpy() - want)[ok].max()) <= 2e-3
    edge = np.minimum.reduce([np.abs(want[..., 0]), np.abs(want[..., 0] - (W - 0.5)), np.abs(want[..., 1]),
                              np.abs(want[..., 1] - (H - 0.5))]) < 1e-2
    assert not ((mask.cpu().numpy() != wm) & ~edge).any() and int(wm.sum()) > 100000
    g = torch.Generator(device="cuda").manual_seed(0)
    rx, ry, sx, sy = tu.sample_virtual_pixels(pts, mask, 4096, generator=g)
    assert bool(mask[ry.long(), rx.long()].all())                                     # only valid reference pixels
    assert int(sx.min()) >= 0 and int(sx.max()) <= W - 1 and int(sy.min()) >= 0 and int(sy.max()) <= H - 1
    assert torch.equal(torch.stack([sx, sy], -1), torch.round(pts[ry.long(), rx.long()]).int())
    dt = tu.img_warping_for_depth(ref, src, depth, K).cpu().numpy()
    wd = oracle_warp.img_warping_for_depth(ref, src, depth, K)
    assert int((np.abs(dt - wd) > 1e-4).sum()) <= 0.01 * int((wd != 0).sum())
