"""Fuzz: k_resample vs the oracle chain (accepted when as close to the float64 evaluation as the float32 oracle) over 150 (n_prev, S, dilation) combinations (run on the GPU box)."""
import sys, itertools
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import helpers as H
from oracle import raymarch as rm
from ucnerf_amd import _lib
from test_gpu_parity import _resample, dev
lib = _lib.load()
bad = 0
for n_prev, S, dil, seed in itertools.product((1, 2, 3, 5, 63, 64, 65, 129, 255, 256), (2, 3, 64, 127, 511), (0.0, 0.0103, 0.2), (0,)):
    g = torch.Generator().manual_seed(seed * 1000 + n_prev * 7 + S)
    N = 37
    t = torch.sort(torch.rand(N, n_prev + 1, generator=g), dim=-1).values
    t[:, 0], t[:, -1] = 0.0, 1.0
    w = torch.rand(N, n_prev, generator=g) ** 3 + 1e-4
    w = w / w.sum(-1, keepdim=True)
    anneal = 1.0
    jit = torch.rand(N, 1, generator=g)
    def chain(t, w, jit):
        if dil > 0:
            td, wd = rm.dilate_weights(t, w, dil, 0.0, 1.0)
            td, wd = td[..., 1:-1], wd[..., 1:-1]
        else:                                          # dilation 0 = the reference's use_dilation == False branch (models.py:167-168)
            td, wd = t, w
        logits = torch.where(td[..., 1:] > td[..., :-1], anneal * torch.log(wd), torch.full_like(wd, -torch.inf))
        return rm.sample_fenceposts(td, logits, S, 0.0, 1.0, jit)
    want = chain(t, w, jit); truth = chain(t.double(), w.double(), jit.double())
    got = _resample(lib, dev(t), dev(w), dil, anneal, S, jitter=dev(jit))
    e_ref = (want.double() - truth).abs().amax(-1, keepdim=True); e_hip = (got.double() - truth).abs()
    # ... or within one float ulp of the CDF (6e-8) times the amplification width / weight of the bin the sample falls in:
    # inverse-CDF sampling is ill-conditioned there whatever the implementation (a bin of weight 2e-6 and width 8e-3
    # turns 3e-8 of CDF into 1e-4 of position)
    if dil > 0:
        tb, wb = rm.dilate_weights(t.double(), w.double(), dil, 0.0, 1.0)
        tb, wb = tb[..., 1:-1], wb[..., 1:-1]
    else:
        tb, wb = t.double(), w.double()
    wb = wb / wb.sum(-1, keepdim=True)
    bin_ = (torch.searchsorted(tb.contiguous(), truth.contiguous()).clamp(1, tb.shape[-1] - 1) - 1)
    amp = (torch.gather(tb, -1, bin_ + 1) - torch.gather(tb, -1, bin_)) / torch.gather(wb, -1, bin_).clamp_min(1e-30)
    amp = torch.maximum(amp, torch.maximum(torch.roll(amp, 1, -1), torch.roll(amp, -1, -1)))     # a fencepost = midpoint of two samples
    ok = ((got - want).abs() <= 2e-6) | (e_hip <= 4 * e_ref + 1e-5) | (e_hip <= 6e-8 * amp)
    mono = bool((got[:, 1:] >= got[:, :-1]).all())
    if not (ok.all() and mono and float(got.min()) >= 0 and float(got.max()) <= 1):
        bad += 1
        print("MISMATCH", n_prev, S, dil, float((got - want).abs().max()), float(e_hip.max()), float(e_ref.max()), mono)
print("resample fuzz done, mismatches:", bad)
