"""On-GPU ray generation vs the reference's golden vectors and the numpy oracle (-m gpu)."""
import os

import numpy as np
import pytest
import torch

from oracle import rays as oracle_rays

pytestmark = pytest.mark.gpu
KEYS = ("origins", "directions", "viewdirs", "radii", "imageplane", "cam_dirs")


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "rays.npz"))


def _cameras(fx):
    return (torch.from_numpy(fx["pixtocams"]), torch.from_numpy(fx["camtoworlds"]), None, None)


@pytest.mark.parametrize("tag", ["frame", "batch"])
def test_cast_ray_batch_bit_exact_vs_reference(fx, tag):
    from ucnerf_amd.internal import camera_utils as cu
    px, py = torch.from_numpy(fx[f"{tag}.pix_x"]).cuda(), torch.from_numpy(fx[f"{tag}.pix_y"]).cuda()
    ci = torch.from_numpy(fx[f"{tag}.cam_idx"]).cuda()[..., None]
    pixels = dict(pix_x_int=px, pix_y_int=py, cam_idx=ci, near=None, far=None, lossmult=None)
    b = cu.cast_ray_batch(_cameras(fx), pixels, cu.ProjectionType.PERSPECTIVE)
    for k in KEYS:
        got = b[k].cpu().numpy()
        assert got.shape == fx[f"{tag}.{k}"].shape and np.array_equal(got, fx[f"{tag}.{k}"]), (tag, k)
    # the per-pixel-matrices form of pixels_to_rays (what the reference's batch_index hands over)
    idx = torch.from_numpy(fx[f"{tag}.cam_idx"]).long()
    o, d, v, r, ip = cu.pixels_to_rays(px, py, torch.from_numpy(fx["pixtocams"])[idx], torch.from_numpy(fx["camtoworlds"])[idx])
    for k, got in zip(KEYS[:5], (o, d, v, r, ip)):
        assert np.array_equal(got.cpu().numpy(), fx[f"{tag}.{k}"]), (tag, k)


def test_full_frame_bit_exact_vs_oracle_and_consistent_with_pixel_batches(fx):
    """BASELINE size: every pixel of a 1920x1280 frame (2 457 600 rays) against the numpy oracle; a pixel batch drawn
    from the frame equals the same pixels of the full frame."""
    from ucnerf_amd.internal import camera_utils as cu
    W, H, cam = int(fx["width"]), int(fx["height"]), 3
    b = cu.generate_ray_batch(_cameras(fx), cam, W, H, 0.25, 8.0)
    xs, ys = np.meshgrid(np.arange(W), np.arange(H), indexing="xy")
    want = oracle_rays.make_ray_batch(xs, ys, cam, fx["pixtocams"], fx["camtoworlds"], 0.25, 8.0)
    for k in KEYS + ("near", "far", "lossmult", "cam_idx"):
        assert tuple(b[k].shape) == want[k].shape and np.array_equal(b[k].cpu().numpy(), want[k]), k
    assert torch.equal(b["camera_id"], b["cam_idx"][..., 0])
    g = torch.Generator().manual_seed(0)
    px, py = torch.randint(0, W, (5000,), generator=g).cuda(), torch.randint(0, H, (5000,), generator=g).cuda()
    pixels = dict(pix_x_int=px, pix_y_int=py, cam_idx=torch.full((5000, 1), cam, device="cuda"))
    sub = cu.cast_ray_batch(_cameras(fx), pixels)
    for k in KEYS:
        assert torch.equal(sub[k], b[k][py.long(), px.long()]), k
    xs_t, ys_t = cu.pixel_coordinates(W, H)
    assert np.array_equal(xs_t.cpu().numpy(), xs) and np.array_equal(ys_t.cpu().numpy(), ys)


def test_unsupported_configurations_and_host_tensors_raise(fx):
    from ucnerf_amd.internal import camera_utils as cu
    px = torch.zeros(4, dtype=torch.int32, device="cuda")
    cams = _cameras(fx)
    with pytest.raises(NotImplementedError, match="distortion"):
        cu.pixels_to_rays(px, px, cams[0][0], cams[1][0], distortion_params=dict(k1=0.1))
    with pytest.raises(NotImplementedError, match="NDC"):
        cu.pixels_to_rays(px, px, cams[0][0], cams[1][0], pixtocam_ndc=np.eye(3))
    with pytest.raises(NotImplementedError, match="perspective"):
        cu.pixels_to_rays(px, px, cams[0][0], cams[1][0], camtype=cu.ProjectionType.FISHEYE)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        cu.pixels_to_rays(px.cpu(), px.cpu(), cams[0][0], cams[1][0])
    with pytest.raises(RuntimeError, match="out of range"):
        cu.cast_ray_batch(cams, dict(pix_x_int=px, pix_y_int=px, cam_idx=torch.full((4, 1), 99, device="cuda")))
    empty = torch.zeros(0, dtype=torch.int32, device="cuda")
    o, d, v, r, ip = cu.pixels_to_rays(empty, empty, cams[0][0], cams[1][0])
    assert o.shape == (0, 3) and r.shape == (0, 1)
