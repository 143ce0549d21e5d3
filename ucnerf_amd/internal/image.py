"""Image metrics on the device -- host side of `ucn_image_metrics`.

`MetricHarness` has the call shape of the reference's (internal/image.py:114-133: `harness(rgb_pred, rgb_gt, name_fn)` ->
{'psnr', 'ssim', 'lpips'}) but takes the rendered frame where `render_image` left it, as a device tensor (numpy arrays are
uploaded), and computes PSNR and SSIM there in one pass each; the two floats are the only thing that crosses PCIe.  LPIPS
is computed only when the third-party `lpips` package (and its downloaded VGG weights) is importable, exactly as upstream;
otherwise the key is absent."""
import ctypes

import numpy as np
import torch

from .. import _lib


def mse_to_psnr(mse):
    """image.py:11-13."""
    return -10. / np.log(10.) * np.log(mse)


def image_metrics(rgb_pred, rgb_gt):
    """(psnr, ssim) of two [H, W, 3] images in [0, 1] with the reference's conventions (uint8 quantisation, data range 255,
    SSIM on OpenCV's grey conversion with skimage's defaults)."""
    lib = _lib.load()
    dev = rgb_pred.device if torch.is_tensor(rgb_pred) and rgb_pred.is_cuda else torch.device("cuda")
    p = torch.as_tensor(rgb_pred).to(dev).float().contiguous()
    g = torch.as_tensor(rgb_gt).to(dev).float().contiguous()
    if p.shape != g.shape or p.dim() != 3 or p.shape[-1] != 3:
        raise RuntimeError(f"image_metrics: expected two [H, W, 3] images, got {tuple(p.shape)} and {tuple(g.shape)}")
    H, W = int(p.shape[0]), int(p.shape[1])
    ws = torch.empty(lib.ucn_image_metrics_ws_bytes(H, W), dtype=torch.uint8, device=dev)
    out = torch.empty(3, dtype=torch.float64, device=dev)
    _lib.check(lib.ucn_image_metrics(p.data_ptr(), g.data_ptr(), H, W, ws.data_ptr(), out.data_ptr(), _lib.stream()))
    psnr, ssim, _ = out.cpu().tolist()
    return psnr, ssim


class MetricHarness:
    def __init__(self):
        self._lpips = None
        try:
            import lpips
            self._lpips = lpips.LPIPS(net="vgg")
        except Exception as e:                              # noqa: BLE001 -- package or weights absent, a corrupt / partial weight cache
            # (RuntimeError / pickle errors from torch.load), version skew (AttributeError): 'lpips' is reported as nan, key always
            # present (a narrower guard crashed eval.py / train.py at harness construction); the reason is logged once
            import warnings
            warnings.warn(f"MetricHarness: LPIPS unavailable ({type(e).__name__}: {e}); 'lpips' will be nan", stacklevel=2)
            self._lpips = None

    def __call__(self, rgb_pred, rgb_gt, name_fn=lambda s: s):
        psnr, ssim = image_metrics(rgb_pred, rgb_gt)
        res = {name_fn('psnr'): float(psnr), name_fn('ssim'): float(ssim)}
        res[name_fn('lpips')] = float('nan')                  # the reference always returns the key (image.py:127-133)
        if self._lpips is not None:
            # image.py:119-120: the prediction is clipped before quantisation, the ground truth is quantised as it is
            # (numpy's astype(uint8) wraps values >= 256 where torch's .to(uint8) is implementation-defined: wrap explicitly)
            gt_ = (torch.remainder((torch.as_tensor(rgb_gt).float().cpu() * 255).trunc(), 256).to(torch.uint8).float() / 255).permute(2, 0, 1).unsqueeze(0) * 2 - 1.0
            pr_ = ((torch.as_tensor(rgb_pred).float().cpu().clamp(0, 1) * 255).to(torch.uint8).float() / 255).permute(2, 0, 1).unsqueeze(0) * 2 - 1.0
            res[name_fn('lpips')] = float(self._lpips(gt_, pr_).detach().item())
        return res
