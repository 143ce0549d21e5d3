"""SURVEY 8 row f4 (remainder): the evaluation metrics of image.py:114-133.  CPU: the restatement against a brute-force
evaluation of the published formulas.  GPU: ucn_image_metrics == the restatement."""
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import metrics as om  # noqa: E402


def images(h, w, seed, noise=0.05):
    g = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.linspace(0, 1, h), np.linspace(0, 1, w), indexing="ij")
    gt = np.stack([0.5 + 0.5 * np.sin(7 * xx + 3 * yy), xx * yy, 0.3 + 0.4 * np.cos(11 * yy)], -1).astype(np.float32)
    gt = np.clip(gt + 0.02 * g.normal(size=gt.shape).astype(np.float32), 0, 1)
    pred = (gt + noise * g.normal(size=gt.shape)).astype(np.float32)          # leaves [0, 1] in places: the clip matters
    return pred, gt


def test_restatement_against_brute_force_and_limits():
    pred, gt = images(19, 23, 1)
    p, g = om.quantise(pred, gt)
    assert p.dtype == np.uint8 and p.min() >= 0 and p.max() <= 255
    assert om.to_gray(np.array([[[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255]]], np.uint8)).tolist() == [[255, 76, 150, 29]]
    a, b = om.to_gray(p).astype(np.float64), om.to_gray(g).astype(np.float64)
    S = []
    for y in range(3, 19 - 3):
        for x in range(3, 23 - 3):
            wa, wb = a[y - 3:y + 4, x - 3:x + 4], b[y - 3:y + 4, x - 3:x + 4]
            ux, uy = wa.mean(), wb.mean()
            vx, vy = wa.var(ddof=1), wb.var(ddof=1)
            vxy = ((wa - ux) * (wb - uy)).sum() / 48
            C1, C2 = 6.5025, 58.5225
            S.append((2 * ux * uy + C1) * (2 * vxy + C2) / ((ux * ux + uy * uy + C1) * (vx + vy + C2)))
    assert abs(om.ssim(om.to_gray(p), om.to_gray(g)) - np.mean(S)) <= 1e-12
    assert om.ssim(om.to_gray(g), om.to_gray(g)) == 1.0 and om.psnr(g, g) == float("inf")
    mse = np.mean((p.astype(np.float64) - g.astype(np.float64)) ** 2)
    assert abs(om.psnr(p, g) - 10 * np.log10(65025 / mse)) <= 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,seed,noise", [(64, 80, 2, 0.05), (7, 7, 3, 0.2), (203, 331, 4, 0.01), (40, 40, 5, 0.0)])
def test_device_metrics_are_the_restatement(h, w, seed, noise):
    from ucnerf_amd.internal import image
    pred, gt = images(h, w, seed, noise)
    want = om.metric_harness(pred, gt)
    psnr, ssim = image.image_metrics(torch.from_numpy(pred).cuda(), torch.from_numpy(gt).cuda())
    if np.isinf(want["psnr"]):
        assert np.isinf(psnr)
    else:
        assert abs(psnr - want["psnr"]) <= 1e-9
    assert abs(ssim - want["ssim"]) <= 1e-12
    res = image.MetricHarness()(pred, gt, name_fn=lambda s: "m/" + s)          # numpy in, like the reference's callers
    assert abs(res["m/ssim"] - want["ssim"]) <= 1e-12 and set(res) >= {"m/psnr", "m/ssim"}
