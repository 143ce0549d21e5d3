"""CPU restatement (numpy float32) of the reference's virtual-pose depth warp -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import oracle/.  Follows
train_utils.py:19-55 (`img_warping`) and :58-98 (`img_warping_for_depth`); pinned against the reference's own functions
through tests/golden/warp.npz (tests/golden/make_warp_golden.py).  The reference's two [3,3] @ [3,HW] products go
through BLAS, whose k = 3 reductions may or may not be fused: agreement with the golden is to a few float32 ulp of a
pixel coordinate (tests/test_oracle_golden.py::test_warp_oracle_vs_reference), not bit-for-bit.
"""
import numpy as np


def relative_pose(ref_pose, src_pose):
    """:39 `src_pose.inverse() @ ref_pose` in float32 (both poses camera-to-world, OpenCV axes)."""
    ref = np.asarray(ref_pose, dtype=np.float32)
    src = np.asarray(src_pose, dtype=np.float32)
    return (np.linalg.inv(src) @ ref).astype(np.float32)


def img_warping(ref_pose, src_pose, depth, intrinsic, with_z=False):
    """Projected source-frame coordinates [H,W,2] of every reference pixel and the validity mask [H,W]."""
    f32 = np.float32
    K = np.asarray(intrinsic).astype(f32)
    d = np.asarray(depth).astype(f32)
    H, W = d.shape
    rel = relative_pose(ref_pose, src_pose)
    cols, rows = np.meshgrid(np.arange(W, dtype=f32), np.arange(H, dtype=f32))
    X = (cols - K[0, 2]) / K[0, 0]                                          # :34
    Y = (rows - K[1, 2]) / K[1, 1]                                          # :35
    P = np.stack([X * d, Y * d, f32(1) * d], axis=0).reshape(3, -1)         # :36-37
    Q = rel[:3, :3] @ P + rel[:3, 3:4]                                      # :41
    with np.errstate(divide="ignore", invalid="ignore"):
        Qn = Q / Q[2:3]                                                     # :43
        uv = (K @ Qn)[:2].T.reshape(H, W, 2)                                # :45-46
        inside = (uv[..., 0] >= 0) & (uv[..., 1] >= 0) & (uv[..., 0] < W - 0.5) & (uv[..., 1] < H - 0.5)   # :48-50
    mask = (d > 0) & inside                                                 # :28, :51
    if with_z:
        return uv.astype(f32), mask, Q[2].reshape(H, W).astype(f32)
    return uv.astype(f32), mask


def img_warping_for_depth(ref_pose, src_pose, depth, intrinsic):
    """:58-98: splat the warped depth into the source frame, later (row-major) pixels overwriting earlier ones."""
    uv, mask, z = img_warping(ref_pose, src_pose, depth, intrinsic, with_z=True)
    out = np.zeros_like(z)
    xy = uv[mask].astype(np.int64)                                          # :94 `.to(torch.long)`: truncation
    out[xy[:, 1], xy[:, 0]] = z[mask]                                       # numpy: the last duplicate wins, like torch-CPU
    return out
