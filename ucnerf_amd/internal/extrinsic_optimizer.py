"""Layer-based colour correction head (host side).

ref /root/reference/nerf/internal/extrinsic_optimizer.py:4-48 -- same module / parameter names
(latent_code, sky_latent_code, brightness_MLP.pts_linears.{0,1,2}, brightness_MLP.output_linear).
The reference pushes one latent per RAY through the 4->256->256->256->12 MLP; all rays of a camera
share the latent, so here the MLP runs once per distinct camera index (ucn_dense) and rays index
the resulting 3x4 affines (ucn_apply_affine).
"""
import torch

from .. import _lib


class BrightnessMLP(torch.nn.Module):
    def __init__(self, D=3, W=256, input_ch=4, output_ch=12, use_viewdirs=False):
        super().__init__()
        self.D, self.W, self.input_ch = D, W, input_ch
        self.pts_linears = torch.nn.ModuleList(
            [torch.nn.Linear(input_ch, W)] + [torch.nn.Linear(W, W) for _ in range(D - 1)])
        self.output_linear = torch.nn.Linear(W, output_ch)

    @torch.no_grad()
    def forward(self, x):
        lib = _lib.load()
        _lib.require_device(x, "latent code")
        x = x.float().contiguous()
        st = _lib.stream()
        for lin in list(self.pts_linears) + [self.output_linear]:
            y = torch.empty(x.shape[0], lin.out_features, device=x.device)
            _lib.check(lib.ucn_dense(x.data_ptr(), lin.weight.data_ptr(), lin.bias.data_ptr(), x.shape[0],
                                     lin.in_features, lin.out_features, int(lin is not self.output_linear),
                                     y.data_ptr(), st))
            x = y
        return x


class BrightnessCorrection(torch.nn.Module):
    def __init__(self, n_views, model_sky=False, n_dim=4):
        super().__init__()
        self.latent_code = torch.nn.Parameter(torch.zeros(size=(n_views, n_dim), dtype=torch.float32))
        self.model_sky = model_sky
        if model_sky:
            self.sky_latent_code = torch.nn.Parameter(torch.zeros(size=(n_views, 4), dtype=torch.float32))
        self.brightness_MLP = BrightnessMLP()

    @torch.no_grad()
    def affines(self, cam_idx):
        """(A [M,12], A_sky [M,12] | None, row_of int64 [N] | None) for the distinct cameras of cam_idx."""
        idx = cam_idx.reshape(-1).long()
        if idx.numel() == 1:
            uniq, row_of = idx, None
        else:
            uniq, row_of = torch.unique(idx, return_inverse=True)
            row_of = row_of.contiguous()
        A = self.brightness_MLP(self.latent_code[uniq])
        A_sky = self.brightness_MLP(self.sky_latent_code[uniq]) if self.model_sky else None
        return A, A_sky, row_of

    def forward(self, indices=None):
        """ref extrinsic_optimizer.py:15-25: [n,3,4] (and the sky affine when model_sky)."""
        A, A_sky, row_of = self.affines(indices.squeeze())
        n = indices.reshape(-1).shape[0]
        full = (A if row_of is None else A[row_of]).view(-1, 3, 4).expand(n, 3, 4)
        if self.model_sky:
            return full, (A_sky if row_of is None else A_sky[row_of]).view(-1, 3, 4).expand(n, 3, 4)
        return full
