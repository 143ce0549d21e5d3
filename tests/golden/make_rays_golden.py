"""Golden vectors for on-GPU ray generation from the reference's own camera_utils.py.

Runs ONLY in the authoring container (imports /root/reference/nerf read-only through ref_import's stubs):
    python tests/golden/make_rays_golden.py          -> tests/golden/rays.npz
Inputs are Waymo-like: 1920x1280 frames, inverse intrinsics = np.linalg.inv(K) in float64 (datasets.py:855),
float64 camera-to-world poses; outputs are what the DataLoader hands to the model, i.e. the float64 results of
camera_utils.pixels_to_rays / cast_ray_batch + the cam_dirs of datasets.py:446, cast with `.float()` (datasets.py:476).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402


def rotation(rng):
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def main():
    ref_import.load()
    cwd = os.getcwd()
    os.chdir(ref_import.REF)
    try:
        from internal import camera_utils
    finally:
        os.chdir(cwd)
    rng = np.random.default_rng(11)
    W, H, n_cam = 1920, 1280, 5
    K = np.array([[2055.556149, 0.0, 939.657470], [0.0, 2055.556149, 641.072182], [0.0, 0.0, 1.0]])
    intr = np.stack([K + np.diag([i * 3.25, i * 3.25, 0.0]) for i in range(n_cam)])
    pixtocams = np.array([np.linalg.inv(k) for k in intr])                       # datasets.py:855
    camtoworlds = np.concatenate([np.stack([rotation(rng) for _ in range(n_cam)]),
                                  rng.standard_normal((n_cam, 3, 1)) * 0.5], axis=-1)
    cameras = (pixtocams, camtoworlds, None, None)
    out = dict(pixtocams=pixtocams, camtoworlds=camtoworlds, width=W, height=H)

    def cast(px, py, ci, tag):
        b = lambda v: np.broadcast_to(v, px.shape)[..., None]
        pixels = dict(pix_x_int=px, pix_y_int=py, lossmult=b(1.), near=b(0.), far=b(8.), cam_idx=b(ci))
        batch = camera_utils.cast_ray_batch(cameras, pixels, camera_utils.ProjectionType.PERSPECTIVE)
        batch['cam_dirs'] = -camtoworlds[pixels['cam_idx'][..., 0]][..., :3, 2]   # datasets.py:446
        out[f'{tag}.pix_x'], out[f'{tag}.pix_y'] = px.astype(np.int32), py.astype(np.int32)
        out[f'{tag}.cam_idx'] = np.broadcast_to(ci, px.shape).astype(np.int32)
        for k in ('origins', 'directions', 'viewdirs', 'radii', 'imageplane', 'cam_dirs'):
            out[f'{tag}.{k}'] = np.asarray(batch[k]).astype(np.float32)           # datasets.py:476 `.float()`

    # (a) a full small frame of one camera, pixel_coordinates order (datasets.py:572-587 with a 48x32 crop)
    px, py = camera_utils.pixel_coordinates(48, 32)
    cast(px + 900, py + 620, 2, 'frame')
    # (b) a training batch: random pixels of random cameras, incl. the four frame corners
    n = 4096
    px = rng.integers(0, W, size=n)
    py = rng.integers(0, H, size=n)
    px[:4], py[:4] = [0, W - 1, 0, W - 1], [0, 0, H - 1, H - 1]
    cast(px, py, rng.integers(0, n_cam, size=n), 'batch')
    path = os.path.join(HERE, 'rays.npz')
    np.savez_compressed(path, **out)
    print(f'rays.npz: {os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays')


if __name__ == '__main__':
    main()
