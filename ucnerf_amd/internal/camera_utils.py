"""On-GPU ray generation behind the reference's camera_utils / datasets interface.

ref /root/reference/nerf/internal/camera_utils.py:368-370 (`pixel_coordinates`), :442-445 (`ProjectionType`),
:448-557 (`pixels_to_rays`), :560-608 (`cast_ray_batch`), datasets.py:386-476 (`_make_ray_batch`) and :572-587
(`generate_ray_batch`).  Same names, argument meaning and result keys; tensors live on the device and the arithmetic
is the HIP kernel `ucn_generate_rays` (float64 with the reference's operation order, rounded once to float32 -- the
batch is bit-identical to the numpy one after its `.float()`).  Supported configuration = the one UC-NeRF trains and
renders Waymo with: perspective pinhole, no lens distortion, no NDC; anything else raises NotImplementedError instead
of silently diverging.  No CPU path: host tensors raise like every other entry point.
"""
import enum

import torch

from .. import _lib


class ProjectionType(enum.Enum):
    """camera_utils.py:442-445."""
    PERSPECTIVE = 'perspective'
    FISHEYE = 'fisheye'


def pixel_coordinates(width, height, device="cuda"):
    """camera_utils.py:368-370: x and y integer coordinates of a pixel grid ('xy' meshgrid -> [height, width])."""
    ys, xs = torch.meshgrid(torch.arange(height, device=device, dtype=torch.int32),
                            torch.arange(width, device=device, dtype=torch.int32), indexing="ij")
    return xs, ys


def _check_supported(distortion_params, pixtocam_ndc, camtype):
    if distortion_params is not None:
        raise NotImplementedError("lens distortion (camera_utils.py:503-509) is outside the UC-NeRF / Waymo configuration")
    if pixtocam_ndc is not None:
        raise NotImplementedError("NDC rays (camera_utils.py:551-560) are outside the UC-NeRF / Waymo configuration")
    if camtype not in (ProjectionType.PERSPECTIVE, 'perspective'):
        raise NotImplementedError(f"camera type {camtype!r}: only the perspective pinhole model is implemented")


def _cam_table(a, cols, device, what):
    """[..., 3, cols] inverse intrinsics / poses as a float64 device table [n, 3, cols]."""
    t = torch.as_tensor(a)
    if t.shape[-2:] != (3, cols) and not (cols == 4 and t.shape[-2:] == (4, 4)):
        raise RuntimeError(f"{what} must end in [3, {cols}], got {tuple(t.shape)}")
    t = t[..., :3, :].to(device=device, dtype=torch.float64)
    return t.reshape(-1, 3, cols).contiguous()


def _launch(pix_x, pix_y, cam_idx, cam_scalar, p2c, c2w, width, height, n, near, far, shape, with_columns):
    lib = _lib.load()
    dev = p2c.device
    f32 = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
    out = dict(origins=f32(n, 3), directions=f32(n, 3), viewdirs=f32(n, 3), radii=f32(n, 1), imageplane=f32(n, 2),
               cam_dirs=f32(n, 3))
    cols = dict(lossmult=f32(n, 1), near=f32(n, 1), far=f32(n, 1), cam_idx=f32(n, 1)) if with_columns else {}
    _lib.check(lib.ucn_generate_rays(
        _lib.ptr(pix_x), _lib.ptr(pix_y), _lib.ptr(cam_idx), int(cam_scalar), p2c.data_ptr(), c2w.data_ptr(), p2c.shape[0],
        int(width), int(height), n, float(near), float(far), out['origins'].data_ptr(), out['directions'].data_ptr(),
        out['viewdirs'].data_ptr(), out['radii'].data_ptr(), out['imageplane'].data_ptr(), out['cam_dirs'].data_ptr(),
        _lib.ptr(cols.get('near')), _lib.ptr(cols.get('far')), _lib.ptr(cols.get('lossmult')), _lib.ptr(cols.get('cam_idx')),
        _lib.stream()))
    out.update(cols)
    return {k: v.reshape(tuple(shape) + (v.shape[-1],)) for k, v in out.items()}


def _pixels(pix_x_int, pix_y_int):
    _lib.require_device(pix_x_int, "pix_x_int")
    _lib.require_device(pix_y_int, "pix_y_int")
    shape = torch.broadcast_shapes(pix_x_int.shape, pix_y_int.shape)
    px = pix_x_int.expand(shape).to(torch.int32).reshape(-1).contiguous()
    py = pix_y_int.expand(shape).to(torch.int32).reshape(-1).contiguous()
    return px, py, shape


def pixels_to_rays(pix_x_int, pix_y_int, pixtocams, camtoworlds, distortion_params=None, pixtocam_ndc=None,
                   camtype=ProjectionType.PERSPECTIVE):
    """camera_utils.py:448-557.  pixtocams [3,3] / camtoworlds [3,4] for one camera, or per-pixel stacks
    (pixel shape + [3,3] / [3,4], what `batch_index` produces at :585).  Returns origins, directions, viewdirs, radii,
    imageplane on the pixels' device."""
    _check_supported(distortion_params, pixtocam_ndc, camtype)
    px, py, shape = _pixels(pix_x_int, pix_y_int)
    n = px.numel()
    p2c, c2w = _cam_table(pixtocams, 3, px.device, "pixtocams"), _cam_table(camtoworlds, 4, px.device, "camtoworlds")
    if p2c.shape[0] != c2w.shape[0]:
        if 1 not in (p2c.shape[0], c2w.shape[0]):
            raise RuntimeError("pixtocams and camtoworlds must be broadcastable")
        m = max(p2c.shape[0], c2w.shape[0])
        p2c, c2w = p2c.expand(m, 3, 3).contiguous(), c2w.expand(m, 3, 4).contiguous()
    if p2c.shape[0] == 1:
        idx = None
    elif p2c.shape[0] == n:
        idx = torch.arange(n, device=px.device, dtype=torch.int32)          # one matrix pair per pixel
    else:
        raise RuntimeError(f"{p2c.shape[0]} cameras for {n} pixels: pass one camera or one per pixel")
    b = _launch(px, py, idx, 0, p2c, c2w, 0, 0, n, 0.0, 0.0, shape, False)
    return b['origins'], b['directions'], b['viewdirs'], b['radii'], b['imageplane']


def cast_ray_batch(cameras, pixels, camtype=ProjectionType.PERSPECTIVE):
    """camera_utils.py:560-608: `cameras` = (pixtocams, camtoworlds, distortion_params, pixtocam_ndc) with 1 or N
    stacked matrices, `pixels` = dict(pix_x_int, pix_y_int, cam_idx [..., 1], lossmult, near, far).  The camera tables
    stay whole on the device; the kernel indexes them per ray (the reference gathers [.., 3, 3] / [.., 3, 4] copies)."""
    pixtocams, camtoworlds, distortion_params, pixtocam_ndc = cameras
    _check_supported(distortion_params, pixtocam_ndc, camtype)
    px, py, shape = _pixels(pixels['pix_x_int'], pixels['pix_y_int'])
    n = px.numel()
    p2c, c2w = _cam_table(pixtocams, 3, px.device, "pixtocams"), _cam_table(camtoworlds, 4, px.device, "camtoworlds")
    m = max(p2c.shape[0], c2w.shape[0])
    p2c, c2w = p2c.expand(m, 3, 3).contiguous(), c2w.expand(m, 3, 4).contiguous()
    ci = pixels['cam_idx']
    ci = torch.as_tensor(ci, device=px.device)[..., 0].expand(shape).to(torch.int32).reshape(-1).contiguous()
    if m == 1:
        ci = torch.zeros_like(ci)                                           # :583 `arr if arr.ndim == 2`
    elif n and (int(ci.min()) < 0 or int(ci.max()) >= m):
        raise RuntimeError(f"cam_idx out of range [0, {m})")
    b = _launch(px, py, ci, 0, p2c, c2w, 0, 0, n, 0.0, 0.0, shape, False)
    b.update(lossmult=pixels.get('lossmult'), near=pixels.get('near'), far=pixels.get('far'), cam_idx=pixels.get('cam_idx'),
             exposure_idx=pixels.get('exposure_idx'), exposure_values=pixels.get('exposure_values'))
    return b


def generate_ray_batch(cameras, cam_idx, width, height, near, far, device="cuda"):
    """datasets.py:572-587 `generate_ray_batch` -> :386-476 `_make_ray_batch` for a real (non-virtual, non-spherical)
    camera: every pixel of frame `cam_idx`, [height, width, k] float32 tensors with the model's batch keys.  The pixel
    grid is never materialised (the kernel derives x, y from the ray index)."""
    pixtocams, camtoworlds, distortion_params, pixtocam_ndc = cameras
    _check_supported(distortion_params, pixtocam_ndc, ProjectionType.PERSPECTIVE)
    p2c, c2w = _cam_table(pixtocams, 3, device, "pixtocams"), _cam_table(camtoworlds, 4, device, "camtoworlds")
    if not p2c.is_cuda:
        raise RuntimeError("generate_ray_batch: device must be a CUDA device")
    m = max(p2c.shape[0], c2w.shape[0])
    p2c, c2w = p2c.expand(m, 3, 3).contiguous(), c2w.expand(m, 3, 4).contiguous()
    b = _launch(None, None, None, int(cam_idx) if m > 1 else 0, p2c, c2w, width, height, int(width) * int(height), near, far,
                (int(height), int(width)), True)
    if m == 1:
        b['cam_idx'].fill_(float(cam_idx))
    b['camera_id'] = b['cam_idx'][..., 0]
    return b
