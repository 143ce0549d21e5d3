import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import helpers as H
from oracle import raymarch as rm
from ucnerf_amd import _lib
from test_gpu_parity import _resample, dev
lib = _lib.load()
fx = H.load("stepfun.npz")
t, w, dil = dev(fx["t"]), dev(fx["w"]), float(fx["dilation"])
knots = fx["t_dilate"][..., 1:-1]
for frac in ("1.0", "0.25"):
    f = float(frac)
    got = _resample(lib, t, w, dil, 10 * f / (9 * f + 1), 128)
    want = fx[f"sample_eval_{frac}"]
    cdf = rm.cdf_of_weights(torch.softmax(fx[f"logits_{frac}"], dim=-1))
    qg, qw = rm.interp_sorted(got, knots, cdf), rm.interp_sorted(want, knots, cdf)
    bad = ~(((got - want).abs() <= 2e-6) | ((qg - qw).abs() <= 2e-6))
    print(frac, "bad count", int(bad.sum()), "rows", torch.nonzero(bad.any(1)).flatten().tolist())
    for r, c in torch.nonzero(bad)[:6].tolist():
        # local pdf of the bin containing want
        i = int(torch.searchsorted(knots[r].contiguous(), want[r, c:c+1].contiguous(), right=True)) - 1
        i = max(0, min(i, knots.shape[1] - 2))
        print(f"  r{r} c{c} got {got[r,c]:.8f} want {want[r,c]:.8f} dq {float(qg[r,c]-qw[r,c]):.2e} bin[{i}] width {float(knots[r,i+1]-knots[r,i]):.3e} mass {float(cdf[r,i+1]-cdf[r,i]):.3e}")
