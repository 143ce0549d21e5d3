"""Golden for SURVEY 8 row f4 (TSDF fusion): the REFERENCE's own TSDF class (/root/reference/nerf/tsdf.py:31-219,
imported unmodified; third-party mesh / image packages stubbed -- none of them takes part in `integrate_tsdf`) run on
the CPU on a small volume and synthetic depth / colour images of four poses.

    python tests/golden/make_tsdf_golden.py            (authoring container only)  ->  tests/golden/tsdf.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402


def scene_inputs(res=24, H=40, W=56, n_views=4, seed=0):
    g = torch.Generator().manual_seed(seed)
    K = torch.tensor([[45.0, 0.0, W / 2], [0.0, 45.0, H / 2], [0.0, 0.0, 1.0]])
    c2w = []
    for i in range(n_views):
        yaw = 0.4 * i - 0.5
        R = torch.tensor([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]], dtype=torch.float32)
        t = torch.tensor([0.3 * i - 0.4, 0.1 * i, 2.5 - 0.2 * i])
        m = torch.eye(4)
        m[:3, :3], m[:3, 3] = R, t
        c2w.append(m)
    c2w = torch.stack(c2w)
    depth = torch.rand(n_views, 1, H, W, generator=g) * 3 + 1.0
    depth[torch.rand(n_views, 1, H, W, generator=g) < 0.15] = 0.0           # holes
    color = torch.rand(n_views, 3, H, W, generator=g)
    return dict(c2w=c2w, K=K, depth=depth, color=color)


def main():
    ref = ref_import.load()
    for name in ("trimesh", "pymeshlab", "tqdm"):
        if name not in sys.modules:
            ref_import._mod(name)
    sys.modules["tqdm"].tqdm = lambda x, **k: x
    sys.modules["skimage"].measure = ref_import._mod("skimage.measure", marching_cubes=None)
    sys.modules["absl"].app = ref_import._mod("absl.app", run=lambda f: None)
    ref.configs.define_common_flags = lambda: None
    for name in ("datasets", "utils", "checkpoints"):                     # imported by tsdf.py, unused by the TSDF class
        sys.modules.setdefault("internal." + name, types.ModuleType("internal." + name))
        setattr(sys.modules["internal"], name, sys.modules["internal." + name])
    cwd = os.getcwd()
    os.chdir(ref_import.REF)
    try:
        spec = importlib.util.spec_from_file_location("ref_tsdf", os.path.join(ref_import.REF, "tsdf.py"))
        tsdf = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(tsdf)
    finally:
        os.chdir(cwd)
    cfg = types.SimpleNamespace(tsdf_radius=2.0, tsdf_resolution=24, truncation_margin=5.0, tsdf_max_radius=10.0)
    acc = types.SimpleNamespace(device=torch.device("cpu"), num_processes=1, process_index=0, is_main_process=True)
    vol = tsdf.TSDF(cfg, acc)
    inp = scene_inputs(res=cfg.tsdf_resolution)
    out = {"in_" + k: v.numpy() for k, v in inp.items()}
    out["voxel_coords"] = vol.voxel_coords.numpy()
    out["voxel_world_coords"] = vol.voxel_world_coords.numpy()
    # two calls: the running average continues across calls
    vol.integrate_tsdf(inp["c2w"][:3], inp["K"], inp["depth"][:3], inp["color"][:3])
    out["values_after3"], out["weights_after3"], out["colors_after3"] = vol.values.numpy().copy(), vol.weights.numpy().copy(), vol.colors.numpy().copy()
    vol.integrate_tsdf(inp["c2w"][3:], inp["K"], inp["depth"][3:], inp["color"][3:])
    out["values"], out["weights"], out["colors"] = vol.values.numpy(), vol.weights.numpy(), vol.colors.numpy()
    out["truncation"] = np.float32(vol.truncation)
    out["voxel_size"] = np.float32(vol.voxel_size)
    np.savez_compressed(os.path.join(HERE, "tsdf.npz"), **out)
    print("tsdf.npz:", {k: v.shape for k, v in out.items() if hasattr(v, "shape")}, "updated voxels", int((vol.weights > 0).sum()))


if __name__ == "__main__":
    main()
