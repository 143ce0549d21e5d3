"""Marching cubes on the host, TEST INFRASTRUCTURE: the numpy restatement of ucnerf_amd/csrc/mesh.hip (same derived table,
oracle/mc_table.py; same conventions and output order), against which the device kernels are compared bit for bit.

There is nothing of the reference's own to restate here: it calls skimage.measure.marching_cubes (extract.py:379-383,
tsdf.py:98-102), a third-party library absent from this image -- parity with the reference is UNPINNED for this row; what is
pinned is device == this restatement, and the properties any marching-cubes output must have (tests/test_mesh.py: closed
2-manifold away from the volume border, vertices on the linearly interpolated crossing, outward orientation, convergence of
area / volume on an analytic surface).
  volume [X, Y, Z] float32, inside = value < level
  vertices: one per cut lattice edge, owned by the edge's lower end point p, ordered by p (x slowest, z fastest) then axis;
            position = (p + t e_axis) * spacing, t = (level - v0) / (v1 - v0) in float32
  faces:    by cell (same order) then table order; indices into the vertices
  normals:  unit gradient (central differences, one-sided at the border) interpolated along the edge, divided by the spacing
"""
import numpy as np

from . import mc_table

_TABLE = None


def tables():
    global _TABLE
    if _TABLE is None:
        _TABLE = mc_table.build()
    return _TABLE


def marching_cubes(volume, level=0.0, spacing=(1.0, 1.0, 1.0)):
    vol = np.ascontiguousarray(volume, np.float32)
    X, Y, Z = vol.shape
    tri_table, tri_count, edges = tables()
    level = np.float32(level)
    inside = vol < level
    sp = np.asarray(spacing, np.float32)
    # ---- vertices: cut flags per point and axis
    cut = np.zeros((X, Y, Z, 3), bool)
    cut[:-1, :, :, 0] = inside[:-1] != inside[1:]
    cut[:, :-1, :, 1] = inside[:, :-1] != inside[:, 1:]
    cut[:, :, :-1, 2] = inside[:, :, :-1] != inside[:, :, 1:]
    flat = cut.reshape(-1, 3)
    nper = flat.sum(axis=1)
    vbase = np.concatenate([[0], np.cumsum(nper)[:-1]]).astype(np.int64)
    pidx, axis = np.nonzero(flat)                                   # row-major: by point, then axis
    px, py, pz = pidx // (Y * Z), (pidx // Z) % Y, pidx % Z
    d = np.eye(3, dtype=np.int64)[axis]
    v0 = vol[px, py, pz]
    v1 = vol[px + d[:, 0], py + d[:, 1], pz + d[:, 2]]
    t = ((level - v0) / (v1 - v0)).astype(np.float32)
    pos = np.stack([px, py, pz], 1).astype(np.float32)
    pos[np.arange(len(axis)), axis] += t
    verts = (pos * sp[None, :]).astype(np.float32)
    # ---- normals
    def grad(x, y, z):
        g = np.zeros((len(x), 3), np.float32)
        for a, (c, n) in enumerate(((x, X), (y, Y), (z, Z))):
            lo, hi = np.maximum(c - 1, 0), np.minimum(c + 1, n - 1)
            idx_lo, idx_hi = [x, y, z], [x, y, z]
            idx_lo[a], idx_hi[a] = lo, hi
            den = np.maximum(hi - lo, 1).astype(np.float32)
            g[:, a] = (vol[tuple(idx_hi)] - vol[tuple(idx_lo)]) / den
        return g
    g0, g1 = grad(px, py, pz), grad(px + d[:, 0], py + d[:, 1], pz + d[:, 2])
    n = ((g0 + t[:, None] * (g1 - g0)) / sp[None, :]).astype(np.float32)
    ln = np.sqrt((n[:, 0] * n[:, 0] + n[:, 1] * n[:, 1]) + n[:, 2] * n[:, 2]).astype(np.float32)
    inv = np.where(ln > 0, np.float32(1.0) / np.where(ln > 0, ln, 1), 0).astype(np.float32)
    normals = (n * inv[:, None]).astype(np.float32)
    # ---- faces
    cube = np.zeros((X - 1, Y - 1, Z - 1), np.int64)
    for c in range(8):
        cx, cy, cz = c & 1, (c >> 1) & 1, (c >> 2) & 1
        cube |= inside[cx:X - 1 + cx, cy:Y - 1 + cy, cz:Z - 1 + cz].astype(np.int64) << c
    faces = []
    cells = np.nonzero(tri_count[cube] > 0)
    cell_lin = (cells[0] * Y + cells[1]) * Z + cells[2]             # point index of the cell's corner 0
    order = np.argsort(cell_lin, kind="stable")
    first = np.array([mc_table.EDGES[e][0] for e in range(12)])
    for i in order:
        x, y, z = cells[0][i], cells[1][i], cells[2][i]
        cs = cube[x, y, z]
        for k in range(tri_count[cs]):
            tri = []
            for e in tri_table[cs, 3 * k:3 * k + 3]:
                c0, a = first[e], e >> 2
                q = ((x + (c0 & 1)) * Y + (y + ((c0 >> 1) & 1))) * Z + (z + ((c0 >> 2) & 1))
                tri.append(vbase[q] + int(flat[q, :a].sum()))
            faces.append(tri)
    return verts, np.asarray(faces, np.int32).reshape(-1, 3), normals
