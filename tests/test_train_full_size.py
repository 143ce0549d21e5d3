"""BASELINE.json configs[2] (the training step) at FULL table size against the CPU oracle's autograd (-m gpu).

The r01-r04 training tests pin the hand-written backward to the reference only at the "tiny" spec (G10: T = 2^12 rows
per level) and check the size-dependent machinery -- fixed-point row blocks, the > 32-row-block path, the persistent
per-XCD queues, the lane-paired forward fetch of incoherent rays -- against the repo's own atomic kernel.  Here the SAME
full-size state (bench.build_model: NeRF grid L16 / C2 / T = 2^19, proposal L6; and once the waymo.gin grid L10 / C4 /
T = 2^21 with 256 row blocks per level) goes through

  * oracle/raymarch.py model_forward(training=True) + the reference's losses (train.py:173-216 with the waymo defaults)
    + torch-CPU autograd, the table gradient through oracle/grid_oracle.c's restatement of gridencoder.cu:248-340;
  * the HIP fp32 route (Model.forward(rand=True), every random draw pinned through the batch);
  * the HIP bf16 autocast route,

on 1024 rays.  Bars (fp32 route, as G10's): loss terms <= 2e-4 relative, dense-layer gradients <= 2e-2 of their scale,
table gradients per LEVEL: |g| sums <= 2e-2 and the difference's L2 norm a stated fraction of the level's norm (a row
block landing in the wrong level or at the wrong rows moves both; the aggregate alone would not see it).  The bf16 route
is held to the oracle (not to the repo's fp32 route) at cosine >= 0.99 per tensor.
"""
import types

import numpy as np
import pytest
import torch

import bench
import helpers as H
from oracle import raymarch as rm
from ucnerf_amd.internal import dense_f32 as D

pytestmark = pytest.mark.gpu

CFG = types.SimpleNamespace(data_loss_type='charb', charb_padding=0.001, data_loss_mult=1.0, data_coarse_loss_mult=0.,
                            anti_interlevel_loss_mult=0.01, pulse_width=[0.03, 0.003], distortion_loss_mult=0.005,
                            hash_decay_mults=0.1, disable_multiscale_loss=False)
N_RAYS = 1024


def _losses(tu, batch, rend, hist):
    return dict(data=tu.compute_data_loss(batch, rend, CFG)[0], anti_interlevel=tu.anti_interlevel_loss(hist, CFG),
                distortion=tu.distortion_loss(hist, CFG), hash_decay=tu.hash_decay_loss(hist, CFG))


def oracle_step(spec, sd, rays, target, noise, train_frac):
    """The reference's step on the host: forward (training branches), losses, backward.  Returns losses and gradients."""
    from ucnerf_amd.internal import train_utils as tu
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if v.is_floating_point() and not k.endswith("grid_sizes")}
    state = dict(sd)
    state.update(params)
    rend, hist = rm.model_forward(spec, state, rays, noise, train_frac=train_frac, compute_extras=False, training=True)
    batch = {k: v[:, None, None, :] for k, v in rays.items()}
    batch['rgb'] = target[:, None, None, :]
    rend = [{k: (v[:, None, None] if torch.is_tensor(v) else v) for k, v in r.items()} for r in rend]
    hist = [{k: (v[:, None, None] if torch.is_tensor(v) and v.dim() >= 1 and k != 'loss_hash_decay' else v)
             for k, v in h.items()} for h in hist]
    losses = _losses(tu, batch, rend, hist)
    sum(losses.values()).backward()
    return {k: float(v.detach()) for k, v in losses.items()}, {k: p.grad for k, p in params.items() if p.grad is not None}


def hip_step(model, rays, target, noise, train_frac, bf16):
    from ucnerf_amd.internal import train_utils as tu
    model.train()
    model.zero_grad(set_to_none=True)
    batch = {k: v[:, None, None, :].cuda() for k, v in rays.items()}
    batch['rgb'] = target[:, None, None, :].cuda()
    batch = H.pin_noise(batch, noise)
    batch['rand_vec'] = batch['rand_vec'][:, None, None, :]
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=bf16):
        rend, hist = model(True, batch, train_frac, False, zero_glo=False)
        losses = _losses(tu, batch, rend, hist)
        total = sum(losses.values())
    total.backward()
    torch.cuda.synchronize()
    return ({k: float(v.detach()) for k, v in losses.items()},
            {k: p.grad.detach().float().cpu() for k, p in model.named_parameters() if p.grad is not None})


def per_level(g, offsets):
    return [g[int(offsets[i]):int(offsets[i + 1])].double() for i in range(offsets.numel() - 1)]


def _case(grid):
    dev = torch.device("cuda", 0)
    model, _, sd = bench.build_model(dev, grid=grid)
    spec = rm.make_spec(grid)
    rays = rm.synthetic_rays(N_RAYS, seed=41)
    g = torch.Generator().manual_seed(42)
    target = torch.rand(N_RAYS, 3, generator=g)
    noise = [rm.draw_level_noise(spec, N_RAYS, lvl, True, g) for lvl in range(2)]
    return model, spec, sd, rays, target, noise


def _check_fp32(spec, want_l, want_g, got_l, got_g, table_bar):
    for k, v in want_l.items():
        assert abs(got_l[k] - v) <= 2e-4 * max(1.0, abs(v)), (k, got_l[k], v)
    assert set(want_g) <= set(got_g), set(want_g) - set(got_g)
    report = {}
    for name, w in want_g.items():
        g = got_g[name]
        if name.endswith("encoder.embeddings"):
            fs = spec.nerf if name.startswith("nerf_mlp") else spec.props[0]
            _, offsets, _, _ = fs.layout()
            for lvl, (a, b) in enumerate(zip(per_level(g, offsets), per_level(w, offsets))):
                sa, sb = float(a.abs().sum()), float(b.abs().sum())
                rel = float((a - b).norm() / (b.norm() + 1e-30))
                report[f"{name}[{lvl}]"] = (sa, sb, rel)
                assert abs(sa - sb) <= 2e-2 * sb + 1e-12, (name, lvl, sa, sb)
                assert abs(float(a.sum()) - float(b.sum())) <= 2e-2 * sb + 1e-12, (name, lvl)
                assert rel <= table_bar(name, lvl), (name, lvl, rel)
        else:
            scale = float(w.abs().max())
            assert H.maxdiff(g, w) <= 2e-2 * scale + 1e-12, (name, H.maxdiff(g, w), scale)
    return report


def _check_bf16(want_l, want_g, got_l, got_g):
    for k, v in want_l.items():
        assert abs(got_l[k] - v) <= 3e-2 * abs(v) + 1e-6, (k, got_l[k], v)
    worst = (1.0, None, 0.0, None)
    for name, w in want_g.items():
        g, w = got_g[name].double().reshape(-1), w.double().reshape(-1)
        cos = float((g * w).sum() / (g.norm() * w.norm() + 1e-30))
        ratio = float(g.norm()) / float(w.norm())
        if cos < worst[0]:
            worst = (cos, name, worst[2], worst[3])
        if abs(ratio - 1.0) > worst[2]:
            worst = (worst[0], worst[1], abs(ratio - 1.0), name)
        assert cos >= 0.99, (name, cos)
        assert abs(ratio - 1.0) <= 5e-2, (name, float(g.norm()), float(w.norm()))
    print(f"bf16 route vs oracle: lowest cosine {worst[0]:.5f} ({worst[1]}), largest norm deviation {worst[2]:.4f} ({worst[3]})")


def _table_bar(name, lvl):
    # Rows of the dense (coarse) levels are sums over thousands of samples: fp32 summation order only.  On the hashed
    # levels a row sees a handful of samples and the gradient reaching it went through the 16-level featurisation's
    # noise floor (DESIGN.md "Parity analysis": 1e-7 differences in a sample position, amplified by resolutions up to
    # 2^19), so single rows move by percents while the level's sums hold.
    return 3e-2 if lvl < 4 else 1e-1        # measured 0.4 - 1.4 % / 1.4 - 3.4 % (profiles/r05/full_train_test_vs_oracle.txt)


def test_config2_training_step_full_tables_vs_oracle_autograd():
    model, spec, sd, rays, target, noise = _case("B")
    want_l, want_g = oracle_step(spec, sd, rays, target, noise, 0.5)
    for engine in ("split", "exact"):            # r06: the fp32 route's dense layers on both engines, the same bars
        prev = D.set_engine(engine)
        try:
            got_l, got_g = hip_step(model, rays, target, noise, 0.5, bf16=False)
        finally:
            D.set_engine(prev)
        rep = _check_fp32(spec, want_l, want_g, got_l, got_g, _table_bar)
        print(f"fp32 route ({engine} engine), per-level table gradients (sum|g| hip, oracle, rel L2 of the difference):")
        for k, v in rep.items():
            print(f"  {k}: {v[0]:.6e} {v[1]:.6e} {v[2]:.3e}")
    b_l, b_g = hip_step(model, rays, target, noise, 0.5, bf16=True)
    _check_bf16(want_l, want_g, b_l, b_g)


def test_waymo_gin_grid_training_step_full_tables_vs_oracle_autograd():
    """The reference's own grid (waymo.gin: L10 / C4 / T = 2^21, 128 + 32 samples): 2^21 rows per hashed level = 256 row
    blocks per level in the table-gradient kernel."""
    model, spec, sd, rays, target, noise = _case("R")
    want_l, want_g = oracle_step(spec, sd, rays, target, noise, 0.5)
    for engine in ("split", "exact"):
        prev = D.set_engine(engine)
        try:
            got_l, got_g = hip_step(model, rays, target, noise, 0.5, bf16=False)
        finally:
            D.set_engine(prev)
        _check_fp32(spec, want_l, want_g, got_l, got_g, _table_bar)
    b_l, b_g = hip_step(model, rays, target, noise, 0.5, bf16=True)
    _check_bf16(want_l, want_g, b_l, b_g)
