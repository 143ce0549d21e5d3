"""TSDF fusion volume -- host side of `ucn_tsdf_integrate`.

Interface mirror of /root/reference/nerf/tsdf.py:31-219 (`TSDF`): same constructor (config, accelerator), attributes
(`origin`, `voxel_size`, `resolution`, `voxel_coords`, `voxel_world_coords`, `values`, `weights`, `colors`,
`truncation`) and `integrate_tsdf(c2w, K, depth_images, color_images=None)`, so tsdf.py's main loop runs on it
unchanged.  The volume lives on the device; integration is one kernel launch per call.  `export_mesh` needs the same
third-party marching cubes as upstream (skimage + trimesh) and raises if they are absent."""
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
        """tsdf.py:73-113: marching cubes on the gathered volume (third-party, host side, as upstream)."""
        try:
            from skimage import measure
            import trimesh
        except ImportError as e:                                                   # pragma: no cover
            raise NotImplementedError("TSDF.export_mesh needs skimage.measure.marching_cubes and trimesh, like the "
                                      "reference; the fused volume is in .values / .colors") from e
        import numpy as np
        tsdf_values = self.values.clamp(-1, 1)
        mask = self.voxel_world_coords[:, :3].permute(0, 2, 1).norm(p=2, dim=-1) > self.config.tsdf_max_radius
        tsdf_values[mask.reshape(self.values.shape)] = 1.
        r = self.resolution
        vol = self.accelerator.gather(tsdf_values).cpu().reshape((r, r, r)).numpy()
        cols = self.accelerator.gather(self.colors).cpu().reshape((r, r, r, 3)).numpy()
        if self.accelerator.is_main_process:
            vertices, faces, normals, _ = measure.marching_cubes(vol, level=0, allow_degenerate=False)
            vi = np.round(vertices).astype(int)
            colors = cols[vi[:, 0], vi[:, 1], vi[:, 2]]
            vertices = self.origin.cpu().numpy() + vertices * self.voxel_size
            vertices = inv_contract(torch.from_numpy(vertices)).numpy()
            trimesh.Trimesh(vertices=vertices, faces=faces, normals=normals, vertex_colors=colors).export(path)
