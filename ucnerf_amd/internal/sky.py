"""Sky NeRF layer (host side).  ref /root/reference/nerf/internal/models.py:743-904.

`NeRF` holds the parameters under the reference's names (pts_linears.{0..7}, views_linears.0,
feature_linear, alpha_linear, rgb_linear => same state_dict keys); evaluation is
ucnerf_amd/csrc/sky.hip (register-chained fp32 MFMA MLP + per-ray compositing).
"""
import ctypes

import torch
import torch.nn as nn

from .. import _lib

N_SKY_SAMPLES = 120      # render_rays(N_samples=120), models.py:854

_T_CACHE = {}


def _t_vals(device):
    key = str(device)
    if key not in _T_CACHE:
        _T_CACHE[key] = torch.linspace(0., 1., steps=N_SKY_SAMPLES).to(device)    # models.py:870
    return _T_CACHE[key]


class NeRF(nn.Module):
    def __init__(self, D=8, W=256, d_in=3, d_in_view=3, multires=0, multires_view=0, output_ch=4, skips=[4],
                 in_channel_a=48, in_channels_dir=27, encode_appearance=False, use_viewdirs=True):
        super().__init__()
        if (D, W, d_in, d_in_view, multires, multires_view, list(skips), use_viewdirs) != (8, 256, 3, 3, 0, 4, [4], True):
            raise NotImplementedError("the HIP sky layer implements the shipped NeRF(D=8, W=256, skips=[4], "
                                      "multires=0, multires_view=4) only (models.py:85-92)")
        self.D, self.W, self.skips = D, W, skips
        self.input_ch, self.input_ch_view = 3, 3 + 3 * 2 * multires_view
        self.pts_linears = nn.ModuleList(
            [nn.Linear(self.input_ch, W)] +
            [nn.Linear(W, W) if i not in skips else nn.Linear(W + self.input_ch, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(self.input_ch_view + W, W // 2)])
        self.feature_linear = nn.Linear(W, W)
        self.alpha_linear = nn.Linear(W, 1)
        self.rgb_linear = nn.Linear(W // 2, 3)
        self._desc = self._key = self._packed = None

    def _descriptor(self):
        ws = [p for p in self.parameters()]
        for w in ws:
            _lib.require_device(w, "skynerf parameter")
        key = tuple((w.data_ptr(), w._version) for w in ws)
        if key == self._key:
            return self._desc
        lib = _lib.load()
        d = _lib.UcnSky()
        for i in range(8):
            d.w_pts[i] = self.pts_linears[i].weight.data_ptr()
            d.b_pts[i] = self.pts_linears[i].bias.data_ptr()
        d.w_alpha, d.b_alpha = self.alpha_linear.weight.data_ptr(), self.alpha_linear.bias.data_ptr()
        d.w_feat, d.b_feat = self.feature_linear.weight.data_ptr(), self.feature_linear.bias.data_ptr()
        d.w_view, d.b_view = self.views_linears[0].weight.data_ptr(), self.views_linears[0].bias.data_ptr()
        d.w_rgb, d.b_rgb = self.rgb_linear.weight.data_ptr(), self.rgb_linear.bias.data_ptr()
        n = lib.ucn_sky_packed_floats()
        dev = ws[0].device
        if self._packed is None or self._packed.device != dev:
            self._packed = torch.empty(n, dtype=torch.float32, device=dev)
        d.packed = self._packed.data_ptr()
        _lib.check(lib.ucn_sky_pack(ctypes.byref(d), _lib.stream()))
        self._desc, self._key = d, key
        return d

    @torch.no_grad()
    def render(self, origins, directions, cam_dirs, far, far0=None, mixed=False):
        """models.py:326-337: near = batch.far, far = 1.5 * near[0] -> rgb_map [N,3].  `far0`: the tensor whose first
        element stands for near[0] when `far` is a subset of the batch (Model.sky_min_background).  `mixed`: bf16 MFMA
        layers (what the reference's nn.Linear layers are under a bf16 autocast) instead of the fp32-class ones."""
        lib = _lib.load()
        N = origins.shape[0]
        dev = origins.device
        d = self._descriptor()
        far = far.reshape(N).contiguous()
        far0 = float((far if far0 is None else far0.reshape(-1))[0].detach().cpu().item()) * 1.5   # the reference's host sync (models.py:329)
        ws = torch.empty(lib.ucn_sky_workspace_floats(N), device=dev)
        out = torch.empty(N, 3, device=dev)
        _lib.check(lib.ucn_sky_render(ctypes.byref(d), origins.data_ptr(), directions.data_ptr(), cam_dirs.data_ptr(),
                                      far.data_ptr(), far0, _t_vals(dev).data_ptr(), N, ws.data_ptr(), out.data_ptr(),
                                      int(bool(mixed)), _lib.stream()))
        return out

    def forward(self, input_pts, input_views):
        raise NotImplementedError("per-sample NeRF.forward is fused into NeRF.render on this build")


def render_rays(ray_batch, network_fn, N_samples=120, retraw=False, lindisp=False, perturb=0., N_importance=0,
                network_fine=None, white_bkgd=False, raw_noise_std=0., verbose=False, pytest=False):
    """ref models.py:852-904 with its call-site defaults (:334).  ray_batch [N, 11] =
    origins(3) directions(3) near(1) far(1) cam_dirs(3); `far` must be the constant 1.5*near[0]."""
    if (N_samples, lindisp, perturb, white_bkgd, raw_noise_std) != (120, False, 0., False, 0.):
        raise NotImplementedError("render_rays: only the reference's call-site configuration is implemented")
    o = ray_batch[:, 0:3].contiguous()
    d = ray_batch[:, 3:6].contiguous()
    near = ray_batch[:, 6].contiguous()
    cam = ray_batch[:, -3:].contiguous()
    return {'rgb_map': network_fn.render(o, d, cam, near)}
