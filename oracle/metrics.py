"""PSNR / SSIM as the reference's MetricHarness computes them (image.py:114-133), TEST INFRASTRUCTURE.

The reference delegates to third-party code that is not installed here -- cv2.cvtColor(RGB2GRAY) on uint8,
skimage.metrics.peak_signal_noise_ratio(data_range=255) and skimage.metrics.structural_similarity(data_range=255) with its
defaults -- so this is a restatement of their PUBLISHED algorithms (OpenCV's 8-bit fixed-point grey conversion: coefficients
4899 / 9617 / 1868 at 14 fractional bits with rounding; skimage >= 0.19: float64 arithmetic, 7 x 7 uniform window, sample
covariance NP / (NP - 1), K1 = 0.01, K2 = 0.03, mean over the image cropped by (win - 1) // 2), not a check against them:
parity with the reference's numbers is UNPINNED for this row; the device kernel is pinned to this file."""
import numpy as np


def quantise(pred, gt):
    p = (np.clip(pred, 0.0, 1.0) * 255).astype(np.uint8)             # image.py:119
    g = (np.asarray(gt) * 255).astype(np.uint8)                      # image.py:120
    return p, g


def to_gray(rgb_u8):
    r, g, b = (rgb_u8[..., c].astype(np.uint32) for c in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


def psnr(p, g):
    err = np.mean((p.astype(np.float64) - g.astype(np.float64)) ** 2)
    return float(10 * np.log10(255.0 ** 2 / err)) if err > 0 else float("inf")


def ssim(a, b, win=7):
    a, b = a.astype(np.float64), b.astype(np.float64)
    H, W = a.shape

    def box(x):                                                      # window sums over the fully-inside positions
        c = np.cumsum(np.cumsum(np.pad(x, ((1, 0), (1, 0))), axis=0), axis=1)
        return (c[win:, win:] - c[:-win, win:] - c[win:, :-win] + c[:-win, :-win])
    NP = win * win
    cov = NP / (NP - 1.0)
    ux, uy = box(a) / NP, box(b) / NP
    vx = cov * (box(a * a) / NP - ux * ux)
    vy = cov * (box(b * b) / NP - uy * uy)
    vxy = cov * (box(a * b) / NP - ux * uy)
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2))
    return float(S.mean())


def metric_harness(pred, gt):
    p, g = quantise(pred, gt)
    return dict(psnr=psnr(p, g), ssim=ssim(to_gray(p), to_gray(g)))
