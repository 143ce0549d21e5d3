"""SURVEY 8 row f4 (remainder): iso-surface extraction.  CPU: the derived case table and the numpy restatement have the
properties every marching-cubes output must have.  GPU: ucn_marching_cubes_* == the restatement, bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import marching, mc_table  # noqa: E402


def sphere(n, r=0.7, centre=(0.0, 0.0, 0.0)):
    lin = np.linspace(-1, 1, n, dtype=np.float32)
    X, Y, Z = np.meshgrid(lin, lin, lin, indexing="ij")
    return (np.sqrt((X - centre[0]) ** 2 + (Y - centre[1]) ** 2 + (Z - centre[2]) ** 2) - r).astype(np.float32)


def noise_volume(n, seed):
    """Smooth random field: many ambiguous faces / cells, several components, surfaces that leave the volume."""
    g = np.random.default_rng(seed)
    k = g.normal(size=(6, 3))
    ph = g.uniform(0, 6.28, 6)
    lin = np.linspace(0, 1, n, dtype=np.float32)
    X, Y, Z = np.meshgrid(lin, lin, lin, indexing="ij")
    v = sum(np.sin(6 * (k[i, 0] * X + k[i, 1] * Y + k[i, 2] * Z) + ph[i]) for i in range(6))
    return (v + 0.3 * g.normal(size=v.shape)).astype(np.float32)


def edge_stats(faces):
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    _, cd = np.unique(e, axis=0, return_counts=True)
    und, cu = np.unique(np.sort(e, 1), axis=0, return_counts=True)
    return und, cu, int(cd.max())


def test_committed_header_is_the_generators_output():
    assert open(mc_table.HEADER).read() == mc_table.header_text()


def test_case_table_is_closed_consistent_and_complementary():
    table, count, edges = mc_table.build()
    assert count[0] == 0 and count[255] == 0 and int(count.max()) == 5
    for case in range(256):
        tris = mc_table.case_triangles(case)
        used = {e for t in tris for e in t}
        cut = {e for e, (c0, c1) in enumerate(mc_table.EDGES) if ((case >> c0) & 1) != ((case >> c1) & 1)}
        assert used == cut, case                                    # exactly the cut edges carry vertices
        comp = {e for t in mc_table.case_triangles(255 - case) for e in t}
        assert comp == cut, case                                    # inside / outside swapped: the same edges
        # inside a cell every triangle edge is either shared by two triangles (interior of a loop's fan) or lies on a face
        e = [(t[i], t[(i + 1) % 3]) for t in tris for i in range(3)]
        assert len(set(e)) == len(e), case                          # no directed edge twice: consistent winding


@pytest.mark.parametrize("n", [12, 24])
def test_restatement_on_a_sphere_is_a_closed_oriented_manifold_that_converges(n):
    r = 0.7
    v, f, nr = marching.marching_cubes(sphere(n, r), 0.0, (2 / (n - 1),) * 3)
    und, cu, dmax = edge_stats(f)
    assert (cu == 2).all() and dmax == 1                            # closed 2-manifold, consistently oriented
    assert len(v) - len(und) + len(f) == 2                          # Euler characteristic of a sphere
    a, b, c = v[f[:, 0]] - 1, v[f[:, 1]] - 1, v[f[:, 2]] - 1        # centre of the lattice is at (1, 1, 1)
    nn = np.cross(b - a, c - a)
    assert (np.einsum("ij,ij->i", nn, (a + b + c) / 3) > 0).all()   # winding faces outward (towards larger values)
    area = 0.5 * np.linalg.norm(nn, axis=1).sum()
    vol = np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6
    h = 2 / (n - 1)
    assert abs(area - 4 * np.pi * r * r) <= 3 * h * h * 4 * np.pi and abs(vol - 4 / 3 * np.pi * r ** 3) <= 3 * h * h * 4
    rad = np.linalg.norm(v - 1, axis=1)
    assert np.abs(rad - r).max() <= h * h                           # vertices on the (linearly interpolated) crossing
    assert (np.einsum("ij,ij->i", nr, (v - 1) / rad[:, None]) > 0.99).all()      # normals = outward unit gradient


def test_restatement_on_a_noise_field_has_no_cracks():
    vol = noise_volume(20, 3)
    v, f, _ = marching.marching_cubes(vol, 0.1, (1, 1, 1))
    assert len(f) > 2000
    und, cu, dmax = edge_stats(f)
    assert dmax == 1 and cu.max() == 2
    # an edge used once must lie on the border of the volume (the surface leaves the lattice there), never inside
    open_e = und[cu == 1]
    pts = v[open_e.reshape(-1)]
    on_border = ((pts <= 1e-6) | (pts >= vol.shape[0] - 1 - 1e-6)).any(axis=1)
    assert on_border.all()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sphere", "noise", "ragged", "empty"])
def test_device_marching_cubes_is_the_restatement(kind):
    from ucnerf_amd.internal import mesh
    if kind == "sphere":
        vol, level, sp = sphere(33, 0.6, (0.1, -0.05, 0.2)), 0.0, (0.5, 0.25, 1.0)
    elif kind == "noise":
        vol, level, sp = noise_volume(40, 5), 0.1, (1.0, 1.0, 1.0)
    elif kind == "ragged":
        vol, level, sp = noise_volume(37, 6)[:19, :37, :5].copy(), -0.2, (1.0, 2.0, 3.0)      # a block boundary inside, thin axis
    else:
        vol, level, sp = np.ones((9, 9, 9), np.float32), 0.0, (1.0, 1.0, 1.0)
    want_v, want_f, want_n = marching.marching_cubes(vol, level, sp)
    v, f, n, vals = mesh.marching_cubes(torch.from_numpy(vol).cuda(), level, sp)
    assert vals is None and v.shape == (len(want_v), 3) and f.shape == (len(want_f), 3)
    assert np.array_equal(f.cpu().numpy(), want_f)
    assert np.array_equal(v.cpu().numpy(), want_v)                  # same float32 operations in the same order
    if len(want_v):
        assert float(np.abs(n.cpu().numpy() - want_n).max()) <= 2e-6
    v2, f2, _, _ = mesh.marching_cubes(torch.from_numpy(vol).cuda(), level, sp, allow_degenerate=False, with_normals=False)
    assert torch.equal(v2, v) and f2.shape[0] <= f.shape[0]


@pytest.mark.gpu
def test_tsdf_export_mesh_runs_on_the_device_mesher(tmp_path):
    """tsdf.py:73-113 through the mirror: a fused volume -> marching cubes on the device -> a PLY on disk."""
    import types
    from ucnerf_amd.internal.tsdf import TSDF
    cfg = types.SimpleNamespace(tsdf_radius=1.0, tsdf_resolution=48, truncation_margin=5.0, tsdf_max_radius=10.0)
    acc = types.SimpleNamespace(device=torch.device("cuda"), num_processes=1, process_index=0, is_main_process=True,
                                gather=lambda t: t)
    vol = TSDF(cfg, acc)
    vol.values = torch.from_numpy(sphere(48, 0.5).reshape(-1)).cuda() * 4          # a sphere of radius 0.5 as the "fused" TSDF
    vol.colors = torch.rand(48 ** 3, 3, device="cuda")
    path = str(tmp_path / "mesh.ply")
    stats = vol.export_mesh(path)
    assert os.path.getsize(path) > 1000 and stats["faces"] > 1000
    head = open(path, "rb").read(400).decode("latin1")
    assert head.startswith("ply") and f"element vertex {stats['vertices']}" in head and f"element face {stats['faces']}" in head
