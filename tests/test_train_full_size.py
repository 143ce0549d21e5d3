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


def _oracle_losses(batch, rend, hist):
    """the oracle side's OWN restatement of the reference's losses (oracle/losses.py, pinned to reference-generated values by
    tests/test_oracle_golden.py) -- not the product's train_utils, which the HIP side uses"""
    from oracle import losses as ol
    return dict(data=ol.data_loss(batch, rend, charb_padding=CFG.charb_padding, data_loss_mult=CFG.data_loss_mult,
                                  data_coarse_loss_mult=CFG.data_coarse_loss_mult),
                anti_interlevel=ol.anti_interlevel_loss(hist, pulse_width=CFG.pulse_width, anti_interlevel_loss_mult=CFG.anti_interlevel_loss_mult),
                distortion=ol.distortion_loss(hist, distortion_loss_mult=CFG.distortion_loss_mult),
                hash_decay=ol.hash_decay_loss(hist, hash_decay_mults=CFG.hash_decay_mults))


def oracle_step(spec, sd, rays, target, noise, train_frac):
    """The reference's step on the host: forward (training branches), losses, backward.  Returns losses and gradients."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if v.is_floating_point() and not k.endswith("grid_sizes")}
    state = dict(sd)
    state.update(params)
    rend, hist = rm.model_forward(spec, state, rays, noise, train_frac=train_frac, compute_extras=False, training=True)
    batch = {k: v[:, None, None, :] for k, v in rays.items()}
    batch['rgb'] = target[:, None, None, :]
    if 'lossmult' not in batch:
        batch['lossmult'] = torch.ones(N_RAYS, 1, 1, 1)
    rend = [{k: (v[:, None, None] if torch.is_tensor(v) else v) for k, v in r.items()} for r in rend]
    hist = [{k: (v[:, None, None] if torch.is_tensor(v) and v.dim() >= 1 and k != 'loss_hash_decay' else v)
             for k, v in h.items()} for h in hist]
    losses = _oracle_losses(batch, rend, hist)
    sum(losses.values()).backward()
    oracle_step.sdist = [h['sdist'].detach().reshape(N_RAYS, -1).clone() for h in hist]      # for the pinned-position run
    return {k: float(v.detach()) for k, v in losses.items()}, {k: p.grad for k, p in params.items() if p.grad is not None}


def hip_step(model, rays, target, noise, train_frac, bf16, sdist=None):
    from ucnerf_amd.internal import train_utils as tu
    model.train()
    model.zero_grad(set_to_none=True)
    batch = {k: v[:, None, None, :].cuda() for k, v in rays.items()}
    batch['rgb'] = target[:, None, None, :].cuda()
    batch = H.pin_noise(batch, noise)
    if sdist is not None:                        # the oracle's own sample fenceposts (see test_..._pinned_sample_positions)
        for lvl, s in enumerate(sdist):
            batch['march_noise'][lvl]['sdist'] = s.cuda()
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


def _nerf_level_features(model, spec, sd, rays, noise, sdist):
    """The NeRF level's damped, multisample-averaged grid features [N, S, L, C] at the SAME fenceposts, cone draws and tables: (a) the
    oracle's torch formulation (render.py:94-152 cone cast -> coord.py contraction -> grid lookup -> erf damping -> mean of six), (b) the
    fused HIP gather, called through the C ABI."""
    import ctypes
    from ucnerf_amd import _lib
    lib = _lib.load()
    fs = spec.nerf
    nz = noise[-1]
    tdist = sdist * rays['far'] + (1 - sdist) * rays['near']
    with torch.no_grad():
        means, stds, _ = rm.cone_multisamples(tdist, rays['origins'], rays['directions'], rays['cam_dirs'], rays['radii'], nz.rand_vec,
                                              spec.std_scale, nz.flip, nz.spin)
        _, _, _, feat = rm.field_density_features(fs, sd, means, stds)
    n, S = sdist.shape[0], sdist.shape[1] - 1
    L, C = fs.num_grid_levels, fs.grid_level_dim
    want = feat.reshape(n, S, L, C)
    dev = "cuda"
    g = {k: v.to(dev).contiguous() for k, v in rays.items()}
    basis = torch.empty(n, 6, device=dev)
    rv = nz.rand_vec.to(dev).contiguous()
    _lib.check(lib.ucn_cone_basis(g['cam_dirs'].data_ptr(), rv.data_ptr(), n, basis.data_ptr(), _lib.stream()))
    geom = (sdist.to(dev).contiguous(), g['near'].reshape(-1).contiguous(), g['far'].reshape(-1).contiguous(), g['origins'], g['directions'], basis,
            g['radii'].reshape(-1).contiguous(), nz.flip.to(dev).contiguous(), nz.spin.to(dev).contiguous())
    desc = _lib.UcnField.from_buffer_copy(model.nerf_mlp.field())
    out = torch.empty(L, n * S, C, device=dev)
    _lib.check(lib.ucn_march_features(ctypes.byref(desc), *[t.data_ptr() for t in geom], float(model.std_scale), n, S, 0, 0, out.data_ptr(),
                                      None, None, _lib.stream()))
    got = out.reshape(L, n, S, C).permute(1, 2, 0, 3).float().cpu()
    return got, want


def test_config2_table_gradient_with_pinned_sample_positions():
    """the hashed levels of the NeRF grid are held to the oracle at 1e-1 (measured 1.4 - 3.4 %) where the proposal
    grid agrees to 1e-7, and the stated reason -- positions that differ by an ulp, turned into other cells by levels up to 2^19 wide --
    was an argument.  Three measurements replace it:

    1. The same step with BOTH levels' sample fenceposts handed to the HIP graph from the oracle's own forward
       (batch['march_noise'][l]['sdist'], a test hook beside the pinned random draws): the per-level differences fall by about a third
       (0.23 - 2.4 % instead of 0.36 - 3.4 %), the losses agree to 2e-5.  So the resampling is part of it, NOT all of it.
    2. The dense layers between the loss and the grid: rgb layer 4e-5, colour layers 2-4e-4, density_layer.2 2e-4, density_layer.0 (the one
       whose weight gradient is d h0^T FEATURES) 4e-3 -- the distance grows towards the features.
    3. The features themselves, at the oracle's fenceposts, oracle formulation against the fused gather (both pinned elsewhere: the
       gather's table arithmetic bit for bit against grid_oracle.c, its geometry against the G3 / G4 fixtures): their relative
       difference DOUBLES from level to level -- 1.9e-7 at a grid side of 17, 1.6e-2 at 524 289 -- i.e. it is proportional to the
       resolution, at a constant 3e-8 of the unit cube: half an ulp of a coordinate near 0.5.  The six points of the cone cast are
       computed in-kernel in another operation order than torch's; an ulp of position is 3 % of a cell at the finest level, and the
       trilinear weights move by that much.  A sample sees a 1 % different top-level feature, hence a slightly different hidden
       state, hence a different gradient on EVERY level, the dense ones included: that is the 0.2 - 0.9 % of levels 0 - 3.
    The table-gradient kernel itself is the exact adjoint of the gather at the gather's own positions
    (tests/test_full_size.py::test_featurisation_is_linear_with_closed_form_on_ones_and_exact_adjoint, 1e-5 of the absolute sum) and equals
    the global-atomic kernel row for row; the proposal grid (resolution <= 512) agrees with the oracle to 1e-7 in this very run."""
    model, spec, sd, rays, target, noise = _case("B")
    want_l, want_g = oracle_step(spec, sd, rays, target, noise, 0.5)
    got_l, got_g = hip_step(model, rays, target, noise, 0.5, bf16=False, sdist=oracle_step.sdist)
    for k, v in want_l.items():
        assert abs(got_l[k] - v) <= 2e-5 * max(1.0, abs(v)), (k, got_l[k], v)
    rep = _check_fp32(spec, want_l, want_g, got_l, got_g, lambda name, lvl: 1.0)
    print("fp32 route with the oracle's sample positions, per-level table gradients (sum|g| hip, oracle, rel L2 of the difference):")
    for k, v in rep.items():
        print(f"  {k}: {v[0]:.6e} {v[1]:.6e} {v[2]:.3e}")
    dense = {}
    for name, w in want_g.items():
        if not name.endswith("encoder.embeddings"):
            dense[name] = float((got_g[name].double() - w.double()).norm() / (w.double().norm() + 1e-30))
            print(f"  dense {name}: rel L2 {dense[name]:.3e}")
    for k, v in rep.items():
        assert v[2] <= (2.5e-3 if k.startswith("prop") else 1.2e-2 if int(k.split("[")[1][:-1]) < 4 else 3e-2), (k, v)
    assert dense["nerf_mlp.rgb_layer.weight"] <= 2e-4 and dense["nerf_mlp.lin_second_stage_1.weight"] <= 1.5e-3
    assert dense["nerf_mlp.density_layer.0.weight"] >= 3 * dense["nerf_mlp.rgb_layer.weight"]          # grows towards the features
    assert all(v <= 2e-4 for k, v in dense.items() if k.startswith("prop_mlp_0"))
    # 3. the features at the oracle's fenceposts
    got, want = _nerf_level_features(model, spec, sd, rays, noise, oracle_step.sdist[-1])
    _, offsets, grid_sizes, _ = spec.nerf.layout()
    print("NeRF-level features at the oracle's fenceposts, oracle formulation vs fused gather, per level:")
    print("  level  grid side   rel L2     share of (sample, level) entries off by > 1e-4")
    off_any = torch.zeros(got.shape[:2], dtype=torch.bool)
    rel, share = [], []
    for l in range(got.shape[2]):
        d = (got[:, :, l] - want[:, :, l]).abs().amax(dim=-1)
        rel.append(float((got[:, :, l] - want[:, :, l]).norm() / (want[:, :, l].norm() + 1e-30)))
        share.append(float((d > 1e-4).float().mean()))
        off_any |= d > 1e-4
        print(f"  {l:5d}  {int(grid_sizes[l]):9d}  {rel[-1]:.3e}  {share[-1]:.5f}")
    print(f"  samples with at least one such level: {float(off_any.float().mean()):.4f}")
    # every level: no further from the oracle than a position difference of 1e-7 of the unit cube (1.7 ulp at 0.5) explains
    for l in range(got.shape[2]):
        assert rel[l] <= 1e-7 * float(grid_sizes[l]) + 2e-7, (l, rel[l], int(grid_sizes[l]))
    assert rel[-1] >= 30 * rel[5]                              # ... and proportional to the resolution, not a constant floor


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
