"""Iso-surface extraction on the device -- host side of `ucn_marching_cubes_*`.

`marching_cubes(volume, level, spacing)` has the call shape of `skimage.measure.marching_cubes` as the reference uses it
(extract.py:379-383 / :452, tsdf.py:98-102): a dense [X, Y, Z] lattice in, `(verts, faces, normals, values)` out -- but the
lattice is a DEVICE tensor (what `nerf_mlp.predict_density` / `TSDF.integrate_tsdf` left in HBM) and so are the outputs; a
512^3 volume does not cross PCIe to be meshed.  Conventions: inside = value < level; vertices in lattice coordinates x spacing
(skimage's convention), shared between triangles; normals = unit gradient of the volume (pointing from inside to outside, the
side the triangles' winding faces); `values` is returned as None (the reference discards it).  The triangulation inside
ambiguous cells comes from this repo's own derived table (tools/gen_mc_table.py), not from Lewiner's."""
import torch

from .. import _lib


@torch.no_grad()
def marching_cubes(volume, level=0.0, spacing=(1.0, 1.0, 1.0), allow_degenerate=True, with_normals=True):
    lib = _lib.load()
    _lib.require_device(volume, "volume")
    if volume.dim() != 3:
        raise RuntimeError(f"marching_cubes: expected a [X, Y, Z] volume, got {tuple(volume.shape)}")
    vol = volume.float().contiguous()
    X, Y, Z = (int(v) for v in vol.shape)
    dev = vol.device
    ws = torch.empty(lib.ucn_marching_cubes_ws_bytes(X, Y, Z), dtype=torch.uint8, device=dev)
    counts = torch.zeros(2, dtype=torch.int32, device=dev)
    st = _lib.stream()
    _lib.check(lib.ucn_marching_cubes_count(vol.data_ptr(), X, Y, Z, float(level), ws.data_ptr(), counts.data_ptr(), st))
    nv, nt = (int(v) for v in counts.cpu())                    # the one host read: the outputs have to be allocated
    if nv >= 1 << 29:         # csrc/mesh.hip packs a lattice point's first vertex id into 29 bits beside 3 flag bits
        raise RuntimeError(f"marching_cubes: {nv} vertices do not fit the kernel's 29-bit vertex ids; extract the volume in blocks")
    verts = torch.empty(nv, 3, device=dev)
    normals = torch.empty(nv, 3, device=dev) if with_normals else None
    faces = torch.empty(nt, 3, dtype=torch.int32, device=dev)
    if nv:
        sx, sy, sz = (float(s) for s in spacing)
        _lib.check(lib.ucn_marching_cubes_emit(vol.data_ptr(), X, Y, Z, float(level), sx, sy, sz, ws.data_ptr(), verts.data_ptr(),
                                               _lib.ptr(normals), faces.data_ptr() if nt else None, st))
    if not allow_degenerate and nt:
        # skimage's allow_degenerate=False: drop triangles of zero area (a vertex exactly on a lattice point collapses an edge)
        a, b, c = (verts[faces[:, k].long()] for k in range(3))
        keep = torch.linalg.cross(b - a, c - a).abs().sum(dim=-1) > 0
        faces = faces[keep]
    return verts, faces, normals, None
