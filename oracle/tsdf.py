"""numpy restatement of the reference's TSDF fusion (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows /root/reference/nerf/tsdf.py:115-219 (`TSDF.integrate_tsdf`) and :31-62 (volume set-up), pinned by
tests/golden/tsdf.npz (the reference's own class run on the CPU, tests/golden/make_tsdf_golden.py)."""
import numpy as np


def inv_contract(z):
    """coord.py:28-35."""
    eps = np.finfo(np.float32).eps
    m = np.maximum(np.sum(z ** 2, axis=-1, keepdims=True), eps)
    return np.where(m <= 1, z, z / np.maximum(2 * np.sqrt(m) - m, eps)).astype(np.float32)


def volume(radius, resolution):
    """tsdf.py:36-54 for one process: voxel_coords [3,N], voxel_world_coords [1,4,N], voxel_size."""
    voxel_size = np.float32(2 * radius / (resolution - 1))
    dim = np.arange(resolution)
    grid = np.stack(np.meshgrid(dim, dim, dim, indexing="ij"), axis=0).reshape(3, -1)
    coords = (np.float32(-radius) + grid.astype(np.float32) * voxel_size).astype(np.float32)
    world = inv_contract(coords.T).T
    world = np.concatenate([world, np.ones((1, world.shape[1]), np.float32)], axis=0)[None]
    return coords, world, voxel_size


def integrate(world, c2w, K, depth, color, truncation, values, weights, colors):
    """tsdf.py:131-219; world [1,4,N], c2w [B,4,4], K [3,3], depth [B,1,H,W], color [B,3,H,W] | None.  In place."""
    f32 = np.float32
    B, _, H, W = depth.shape
    w2c = np.linalg.inv(c2w.astype(np.float64)).astype(f32)
    for i in range(B):
        cam = (w2c[i] @ world[0]).astype(f32)                        # [4,N]
        cam[2] = -cam[2]
        cam[1] = -cam[1]
        zc = cam[2]
        with np.errstate(divide="ignore", invalid="ignore"):
            pix = (K.astype(f32) @ (cam[0:3] / zc)).astype(f32)     # [3,N]
            gx = (f32(2.0) * pix[0] / f32(W) - f32(1.0)).astype(f32)
            gy = (f32(2.0) * pix[1] / f32(H) - f32(1.0)).astype(f32)
            fx = np.rint(((gx + f32(1)) * f32(W) - f32(1)) / f32(2))  # grid_sample nearest, align_corners=False
            fy = np.rint(((gy + f32(1)) * f32(H) - f32(1)) / f32(2))
        inb = (fx >= 0) & (fx <= W - 1) & (fy >= 0) & (fy <= H - 1)
        ix = np.where(inb, fx, 0).astype(np.int64)
        iy = np.where(inb, fy, 0).astype(np.int64)
        sd = np.where(inb, depth[i, 0][iy, ix], f32(0)).astype(f32)
        dist = sd - zc
        tsdf = np.clip(dist / f32(truncation), -1.0, 1.0).astype(f32)
        valid = (zc > 0) & (sd > 0) & (dist > -f32(truncation))
        old_w = weights[valid]
        total = old_w + f32(1)
        values[valid] = (values[valid] * old_w + tsdf[valid] * f32(1)) / total
        if color is not None:
            sc = np.where(inb[None], color[i][:, iy, ix], f32(0)).astype(f32)        # [3,N]
            colors[valid] = (colors[valid] * old_w[:, None] + sc[:, valid].T * f32(1)) / total[:, None]
        weights[valid] = total
