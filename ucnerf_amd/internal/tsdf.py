"""TSDF fusion volume -- host side of `ucn_tsdf_integrate`.

Interface mirror of /root/reference/nerf/tsdf.py:31-219 (`TSDF`): same constructor (config, accelerator), attributes
(`origin`, `voxel_size`, `resolution`, `voxel_coords`, `voxel_world_coords`, `values`, `weights`, `colors`,
`truncation`) and `integrate_tsdf(c2w, K, depth_images, color_images=None)`, so tsdf.py's main loop runs on it
unchanged.  The volume lives on the device; integration is one kernel launch per call; `export_mesh` meshes it on the
device too (internal/mesh.py, csrc/mesh.hip)."""
import torch

from .. import _lib


def inv_contract(z):
    """coord.py:18-25."""
    eps = torch.finfo(z.dtype).eps
    z_mag_sq = torch.sum(z ** 2, dim=-1, keepdim=True).clamp_min(eps)
    return torch.where(z_mag_sq <= 1, z, z / (2 * torch.sqrt(z_mag_sq) - z_mag_sq).clamp_min(eps))


class TSDF:
    def __init__(self, config, accelerator):
        self.config = config
        self.device = accelerator.device
        self.accelerator = accelerator
        self.origin = torch.tensor([-config.tsdf_radius] * 3, dtype=torch.float32, device=self.device)
        self.voxel_size = 2 * config.tsdf_radius / (config.tsdf_resolution - 1)
        self.resolution = config.tsdf_resolution
        dim = torch.arange(self.resolution)
        grid = torch.stack(torch.meshgrid(dim, dim, dim, indexing="ij"), dim=0).reshape(3, -1)
        period = int(grid.shape[1] / accelerator.num_processes + 0.5)             # this rank's slab (tsdf.py:44-45)
        grid = grid[:, period * accelerator.process_index: period * (accelerator.process_index + 1)]
        self.voxel_coords = self.origin.view(3, 1) + grid.to(self.device) * self.voxel_size
        n = self.voxel_coords.shape[1]
        world = inv_contract(self.voxel_coords.permute(1, 0)).permute(1, 0).reshape(3, -1)
        world = torch.cat([world, torch.ones(1, n, device=self.device)], dim=0)
        self.voxel_world_coords = world.unsqueeze(0).contiguous()                 # [1, 4, N]
        self.values = torch.ones(n, dtype=torch.float32, device=self.device)
        self.weights = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.colors = torch.zeros(n, 3, dtype=torch.float32, device=self.device)

    @property
    def truncation(self):
        return self.voxel_size * self.config.truncation_margin

    @torch.no_grad()
    def integrate_tsdf(self, c2w, K, depth_images, color_images=None):
        """tsdf.py:115-219.  c2w [B,4,4], K [3,3], depth_images [B,1,H,W], color_images [B,3,H,W] or None."""
        lib = _lib.load()
        _lib.require_device(depth_images, "depth_images")
        B = int(c2w.shape[0])
        H, W = int(depth_images.shape[-2]), int(depth_images.shape[-1])
        w2c = torch.inverse(c2w.to(self.device).float())[:, :3, :].contiguous()   # tsdf.py:147 (tiny: stays a torch op)
        Kd = K.to(self.device).float().contiguous()
        depth = depth_images.to(self.device).float().reshape(B, H, W).contiguous()
        color = None if color_images is None else color_images.to(self.device).float().reshape(B, 3, H, W).contiguous()
        n = self.values.shape[0]
        _lib.check(lib.ucn_tsdf_integrate(self.voxel_world_coords.data_ptr(), n, w2c.data_ptr(), Kd.data_ptr(), depth.data_ptr(),
                                          _lib.ptr(color), B, H, W, float(self.truncation), self.values.data_ptr(),
                                          self.weights.data_ptr(), None if color is None else self.colors.data_ptr(), _lib.stream()))

    def export_mesh(self, path):
        """tsdf.py:73-113: marching cubes on the gathered volume, vertex colours from the nearest voxel, vertices back through
        the inverse contraction, a mesh file.  The reference copies the volume to the host for skimage; here the mesher is
        `internal.mesh.marching_cubes` (csrc/mesh.hip) on the device, and only the finished mesh crosses PCIe.  Written with
        trimesh when it is importable (as upstream), else as a binary PLY by this module.  Returns the mesh's sizes."""
        from . import mesh
        tsdf_values = self.values.clamp(-1, 1)
        mask = self.voxel_world_coords[:, :3].permute(0, 2, 1).norm(p=2, dim=-1) > self.config.tsdf_max_radius
        tsdf_values[mask.reshape(self.values.shape)] = 1.
        r = self.resolution
        vol = self.accelerator.gather(tsdf_values).reshape(r, r, r)
        cols = self.accelerator.gather(self.colors).reshape(r, r, r, 3)
        if not self.accelerator.is_main_process:
            return None
        vertices, faces, normals, _ = mesh.marching_cubes(vol, level=0.0, allow_degenerate=False)
        vi = torch.round(vertices).long().clamp_(0, r - 1)
        colors = cols[vi[:, 0], vi[:, 1], vi[:, 2]]
        vertices = inv_contract(self.origin + vertices * self.voxel_size)
        v, f, n, c = (t.cpu().numpy() for t in (vertices, faces, normals, colors))
        try:
            import trimesh
            trimesh.Trimesh(vertices=v, faces=f, normals=n, vertex_colors=c).export(path)
        except ImportError:
            write_ply(path, v, f, n, c)
        return dict(vertices=int(v.shape[0]), faces=int(f.shape[0]))


def write_ply(path, vertices, faces, normals=None, colors=None):
    """Binary little-endian PLY: float32 positions (+ normals), uint8 colours, int32 triangles."""
    import numpy as np
    fields = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
    if normals is not None:
        fields += [("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4")]
    if colors is not None:
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
    vert = np.zeros(len(vertices), dtype=fields)
    vert["x"], vert["y"], vert["z"] = vertices[:, 0], vertices[:, 1], vertices[:, 2]
    if normals is not None:
        vert["nx"], vert["ny"], vert["nz"] = normals[:, 0], normals[:, 1], normals[:, 2]
    if colors is not None:
        c8 = np.clip(np.round(np.asarray(colors, np.float64) * 255), 0, 255).astype(np.uint8)
        vert["red"], vert["green"], vert["blue"] = c8[:, 0], c8[:, 1], c8[:, 2]
    face = np.zeros(len(faces), dtype=[("n", "u1"), ("v", "<i4", (3,))])
    face["n"], face["v"] = 3, faces
    names = {"<f4": "float", "u1": "uchar"}
    head = ["ply", "format binary_little_endian 1.0", f"element vertex {len(vert)}"]
    head += [f"property {names[t]} {n}" for n, t in fields]
    head += [f"element face {len(face)}", "property list uchar int vertex_indices", "end_header"]
    with open(path, "wb") as fh:
        fh.write(("\n".join(head) + "\n").encode("ascii"))
        fh.write(vert.tobytes())
        fh.write(face.tobytes())
