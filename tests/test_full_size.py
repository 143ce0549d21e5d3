"""Size-independent properties of the HIP path at BASELINE.json's full sizes (-m gpu).

The oracle cannot run config B (16 levels x 2^19 rows, 65 536-ray tiles) in seconds, so at these sizes the path is
held to what the domain guarantees for ANY size:
  * a ray's result does not depend on which rays share its launch (chunking, permutation): bit-exact;
  * step functions stay sorted in [0, 1], weights are a sub-probability vector, colours stay in the padded range;
  * the featurisation is LINEAR in the hash table, a table of ones turns it into the mean erf damping of the six
    multisamples (closed form from the oracle's geometry, the trilinear weights of a cell sum to 1), and the table
    gradient is its exact adjoint: <g, F(T)> == <F^T(g), T>.
"""
import ctypes

import numpy as np
import pytest
import torch

import bench
from oracle import raymarch as rm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene():
    dev = torch.device("cuda", 0)
    model, cfg, sd = bench.build_model(dev)
    rays = bench.frame_rays(dev)
    n_total = bench.H_IMG * bench.W_IMG
    flat = {k: v.reshape(n_total, -1) for k, v in rays.items()}
    return model, flat, n_total


def _pick(flat, n_total, n, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    idx = torch.randperm(n_total, device="cuda", generator=g)[:n]
    batch = {k: v[idx].contiguous() for k, v in flat.items()}
    batch["rand_vec"] = torch.randn(n, 6, device="cuda", generator=g)          # pinned cone-basis draws
    return batch


def test_rays_are_independent_of_their_launch_and_outputs_stay_in_range(scene):
    model, flat, n_total = scene
    n = 65536                                                                  # one render_image tile of config B
    batch = _pick(flat, n_total, n, seed=1)
    with torch.no_grad():
        rend, hist = model(False, batch, 1.0, False)
        full = {k: rend[-1][k].reshape(n, -1).clone() for k in ("rgb", "depth", "acc")}
        parts = []
        for s in range(0, n, 16384):                                           # 4 launches of a quarter tile
            r, _ = model(False, {k: v[s:s + 16384].contiguous() for k, v in batch.items()}, 1.0, False)
            parts.append({k: r[-1][k].reshape(16384, -1).clone() for k in full})
        perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
        rp, _ = model(False, {k: v[perm].contiguous() for k, v in batch.items()}, 1.0, False)
    for k in full:
        assert torch.equal(torch.cat([p[k] for p in parts]), full[k]), k
        assert torch.equal(rp[-1][k].reshape(n, -1), full[k][perm]), k
    pad = float(model.nerf_mlp.rgb_padding)
    assert float(full["acc"].min()) >= 0.0 and float(full["acc"].max()) <= 1.0 + 1e-5
    assert float(full["rgb"].min()) >= -pad - 1e-5 and float(full["rgb"].max()) <= 1.0 + pad + 1e-5
    assert torch.isfinite(full["depth"]).all()
    for h in hist:
        sd = h["sdist"].reshape(n, -1)
        assert bool((sd[:, 1:] >= sd[:, :-1]).all()) and float(sd.min()) >= 0.0 and float(sd.max()) <= 1.0
        w = h["weights"].reshape(n, -1)
        assert float(w.min()) >= 0.0 and float(w.sum(-1).max()) <= 1.0 + 1e-5


def _features(lib, _lib, desc, geom, n, S, std_scale, L, C):
    out = torch.empty(L, n * S, C, device="cuda")
    _lib.check(lib.ucn_march_features(ctypes.byref(desc), *[_lib.ptr(t) for t in geom], std_scale, n, S, 0, 0,
                                      out.data_ptr(), None, None, _lib.stream()))
    return out


def test_featurisation_is_linear_with_closed_form_on_ones_and_exact_adjoint(scene):
    from ucnerf_amd import _lib
    lib = _lib.load()
    model, flat, n_total = scene
    n = 8192                                                                   # BASELINE configs[2]: the training batch
    batch = _pick(flat, n_total, n, seed=3)
    with torch.no_grad():
        _, hist = model(False, batch, 1.0, True)
    mlp = model.nerf_mlp
    enc = mlp.encoder
    L, C = enc.num_levels, enc.level_dim
    sdist = hist[-1]["sdist"].reshape(n, -1).contiguous()
    S = sdist.shape[1] - 1
    basis = torch.empty(n, 6, device="cuda")
    rvec = batch["rand_vec"][:, 3:6].contiguous()
    _lib.check(lib.ucn_cone_basis(batch["cam_dirs"].data_ptr(), rvec.data_ptr(), n, basis.data_ptr(), _lib.stream()))
    near, far, rad = (batch[k].reshape(-1).contiguous() for k in ("near", "far", "radii"))
    geom = (sdist, near, far, batch["origins"], batch["directions"], basis, rad, None, None)
    std_scale = float(model.std_scale)

    def desc_for(table):
        d = _lib.UcnField.from_buffer_copy(mlp.field())
        d.embeddings = table.data_ptr()
        return d

    T = enc.embeddings.detach()
    g = torch.Generator(device="cuda").manual_seed(4)
    T2 = torch.rand(T.shape, device="cuda", generator=g) * 2 - 1
    ones = torch.ones_like(T)
    Tsum = T + T2
    f1 = _features(lib, _lib, desc_for(T), geom, n, S, std_scale, L, C)
    f2 = _features(lib, _lib, desc_for(T2), geom, n, S, std_scale, L, C)
    f12 = _features(lib, _lib, desc_for(Tsum), geom, n, S, std_scale, L, C)
    fo = _features(lib, _lib, desc_for(ones), geom, n, S, std_scale, L, C)
    # linear in the table (fp32 products re-associated: 48 addends of magnitude <= 2)
    assert float((f12 - (f1 + f2)).abs().max()) <= 2e-5

    # table of ones -> mean over the six multisamples of erf(1 / sqrt(8 std^2 res^2)) (models.py:495-496): the
    # oracle's cone geometry + contraction on the host, no table involved
    k = 1024                                                                   # rays checked against the host formula
    c = {kk: v[:k].cpu() for kk, v in batch.items()}
    tdist = sdist[:k].cpu() * c["far"] + (1 - sdist[:k].cpu()) * c["near"]
    means, stds, _ = rm.cone_multisamples(tdist, c["origins"], c["directions"], c["cam_dirs"], c["radii"], rvec[:k].cpu(),
                                          std_scale)
    _, cs = rm.contract_points(means.reshape(-1, 3), stds.reshape(-1))
    cs = cs.reshape(stds.shape) / 2
    want = rm.level_damping(cs, torch.from_numpy(enc._sizes_np.copy())).mean(dim=-2)          # [k, S, L]
    got = fo.reshape(L, n, S, C)[:, :k].permute(1, 2, 0, 3).cpu()
    assert float((got - want[..., None]).abs().max()) <= 2e-5

    # adjoint: <grad, F(T)> == <F^T(grad), T>, sums in float64 (the backward is the compacted row-block kernel)
    grad = torch.randn(L, n * S, C, device="cuda", generator=g)
    ws = torch.empty(lib.ucn_march_features_backward_ws_floats(ctypes.byref(mlp.field()), n, S), device="cuda")
    gt = torch.zeros_like(T)
    _lib.check(lib.ucn_march_features_backward(ctypes.byref(mlp.field()), *[_lib.ptr(t) for t in geom], std_scale, n, S,
                                               0, 0, grad.data_ptr(), gt.data_ptr(), ws.data_ptr(), _lib.stream()))
    lhs = float((grad.double() * f2.double()).sum())
    rhs = float((gt.double() * T2.double()).sum())
    scale = float((grad.double() * f2.double()).abs().sum())
    assert abs(lhs - rhs) <= 1e-5 * scale, (lhs, rhs, scale)
    # ... per level too (a level-local error cannot hide in the total)
    off = enc._offsets_np
    for l in range(L):
        a = float((grad[l].double() * f2[l].double()).sum())
        b = float((gt[int(off[l]):int(off[l + 1])].double() * T2[int(off[l]):int(off[l + 1])].double()).sum())
        s = float((grad[l].double() * f2[l].double()).abs().sum())
        assert abs(a - b) <= 1e-5 * s, (l, a, b, s)


def test_lane_paired_fetch_is_bit_identical_to_the_per_lane_fetch(scene):
    """k_march_features fetches the x0 / x0 + 1 corners of the levels finer than 2048 through lane PAIRS (level_accumulate_lanepairs:
    lane i and lane i + 32 of one load instruction serve one point, v_permlane32_swap sorts rows and values) -- only in waves whose 64
    lanes are all active; the one partial wave of a launch keeps the per-lane fetch.  Same values, same fmaf order: a sample must get
    bit-identical features on either route.  129 rays x 65 samples = 131 x 64 + 1: the LAST sample of the launch is alone in its wave
    (per-lane route); with the rays in reverse order the same sample is lane 0 of a full wave (lane-paired route).  fp32 and half tables."""
    from ucnerf_amd import _lib
    lib = _lib.load()
    model, flat, n_total = scene
    n, S = 129, 65
    batch = _pick(flat, n_total, n, seed=21)
    mlp = model.nerf_mlp
    enc = mlp.encoder
    L, C = enc.num_levels, enc.level_dim
    g = torch.Generator(device="cuda").manual_seed(22)
    sdist = torch.sort(torch.rand(n, S + 1, device="cuda", generator=g), dim=-1).values.contiguous()
    basis = torch.empty(n, 6, device="cuda")
    _lib.check(lib.ucn_cone_basis(batch["cam_dirs"].data_ptr(), batch["rand_vec"][:, 3:6].contiguous().data_ptr(), n, basis.data_ptr(), _lib.stream()))
    near, far, rad = (batch[k].reshape(-1).contiguous() for k in ("near", "far", "radii"))
    emb16 = enc.embeddings.detach().to(torch.half)

    def run(order, half):
        geom = [t[order].contiguous() for t in (sdist, near, far, batch["origins"], batch["directions"], basis, rad)]
        d = _lib.UcnField.from_buffer_copy(mlp.field())
        if half:
            d.embeddings = emb16.data_ptr()
        out = torch.empty(L, S * n, C, device="cuda")                          # layout 2: b = s * n + ray
        _lib.check(lib.ucn_march_features(ctypes.byref(d), *[t.data_ptr() for t in geom], None, None, float(model.std_scale), n, S, 0,
                                          2 | (_lib.TABLE_F16 if half else 0), out.data_ptr(), None, None, _lib.stream()))
        return out.reshape(L, S, n, C)
    fwd = torch.arange(n, device="cuda")
    rev = torch.flip(fwd, dims=[0])
    for half in (False, True):
        a, b = run(fwd, half), run(rev, half)
        assert torch.equal(a, torch.flip(b, dims=[2])), half                  # every sample, whichever wave and lane it landed in
        assert float(a[:, S - 1, n - 1].abs().max()) > 0                     # (the lone sample of the partial wave is a live one)


def test_fixed_point_row_blocks_against_the_float_row_blocks(scene):
    """UCN_BWD_FIXED_POINT (the autocast training step's table gradient; r04): int32 fixed-point row blocks, one 64-bit LDS add
    per channel pair.  On the training batch of config B (8192 rays x 128 samples, all 16 levels) against the exact-fp32-add
    route: (i) no row differs by more than the rounding model allows -- a task's resolution is q <= 2^-29 of ITS summed
    max_c |g| (power-of-two scale from a guaranteed bound), every addend is rounded to nearest (+- q / 2, unbiased), a row
    collects a few hundred of them; (ii) the adjoint identity holds to 1e-4 of the absolute sum (the float route: 1e-5);
    (iii) the error is unbiased: its mean over the touched rows is << its rms; (iv) non-finite gradients poison, not vanish."""
    from ucnerf_amd import _lib
    lib = _lib.load()
    model, flat, n_total = scene
    n = 8192
    batch = _pick(flat, n_total, n, seed=9)
    with torch.no_grad():
        _, hist = model(False, batch, 1.0, True)
    mlp = model.nerf_mlp
    enc = mlp.encoder
    L, C = enc.num_levels, enc.level_dim
    sdist = hist[-1]["sdist"].reshape(n, -1).contiguous()
    S = sdist.shape[1] - 1
    basis = torch.empty(n, 6, device="cuda")
    rvec = batch["rand_vec"][:, 3:6].contiguous()
    _lib.check(lib.ucn_cone_basis(batch["cam_dirs"].data_ptr(), rvec.data_ptr(), n, basis.data_ptr(), _lib.stream()))
    near, far, rad = (batch[k].reshape(-1).contiguous() for k in ("near", "far", "radii"))
    geom = (sdist, near, far, batch["origins"], batch["directions"], basis, rad, None, None)
    std_scale = float(model.std_scale)
    g = torch.Generator(device="cuda").manual_seed(10)
    grad = torch.randn(n * S, L * C, device="cuda", generator=g) * 1e-3          # autograd's layout; a realistic magnitude
    grad[::5] = 0
    ws = torch.empty(lib.ucn_march_features_backward_ws_floats(ctypes.byref(mlp.field()), n, S), device="cuda")

    def run(flag, gr=grad):
        out = torch.zeros_like(enc.embeddings)
        _lib.check(lib.ucn_march_features_backward(ctypes.byref(mlp.field()), *[_lib.ptr(t) for t in geom], std_scale, n, S,
                                                   0, 1 | flag, gr.data_ptr(), out.data_ptr(), ws.data_ptr(), _lib.stream()))
        return out
    want = run(0)
    got = run(_lib.BWD_FIXED_POINT)
    err = (got - want).double()
    # a task (row block x sample part) sees at most B / split samples; bound its resolution by the WHOLE call's sum (looser)
    l1 = float(grad.reshape(n * S, L, C).abs().amax(-1).double().sum(0).max())
    q = l1 * 2.0 ** -29
    touched = want != 0
    n_touched = int(touched.sum())
    assert n_touched > 1e6
    rms = float(err[touched].pow(2).mean().sqrt())
    print(f"fixed point: resolution bound {q:.3e}, max |err| {float(err.abs().max()):.3e}, rms {rms:.3e}, mean {float(err[touched].mean()):.3e}, "
          f"max |grad row| {float(want.abs().max()):.3e}")
    assert float(err.abs().max()) <= 40 * q                    # a few hundred addends x q / 2, random signs (level 0: more addends, run-merged)
    assert abs(float(err[touched].mean())) <= 0.05 * rms + 1e-12     # round to nearest: no drift
    assert rms <= 2e-3 * float(want[touched].double().pow(2).mean().sqrt())
    # adjoint identity through the fixed-point route
    T2 = torch.rand(enc.embeddings.shape, device="cuda", generator=g) * 2 - 1
    d2 = _lib.UcnField.from_buffer_copy(mlp.field())
    d2.embeddings = T2.data_ptr()
    f2 = _features(lib, _lib, d2, geom, n, S, std_scale, L, C)                  # [L][B][C]
    gl = grad.reshape(n * S, L, C).permute(1, 0, 2)
    lhs = float((gl.double() * f2.double()).sum())
    rhs = float((got.double() * T2.double()).sum())
    scale = float((gl.double() * f2.double()).abs().sum())
    assert abs(lhs - rhs) <= 1e-4 * scale, (lhs, rhs, scale)
    # the same call twice: the row blocks are order-independent; what remains is the float flush of `split` parts per block
    again = run(_lib.BWD_FIXED_POINT)
    assert float((again - got).abs().max()) <= 1e-6 * float(want.abs().max())
    # a NaN / inf in the feature gradient must not vanish in the integer conversion
    bad = grad.clone()
    bad[12345, 3] = float("nan")
    bad[54321, 20] = float("inf")
    out = run(_lib.BWD_FIXED_POINT, bad)
    assert not bool(torch.isfinite(out).all())


@pytest.mark.gpu
def test_coresident_launch_shapes_give_identical_pixels():
    """Model.overlap_streams: featurisation of pass i + 1 on a second HIP stream beside the MLP of pass i, both in their
    co-resident launch shapes (UCN_LAUNCH_CORESIDENT: 512-thread featurisation workgroups, 64 KiB weight ring).  A ray's
    result must not depend on the launch it rides in: bit-identical frames."""
    import bench
    dev = torch.device("cuda", 0)
    model, cfg, _ = bench.build_model(dev)
    batch = bench.frame_rays(dev)
    n = 3 * 4096 + 517                                        # several passes and a ragged last one
    flat = {k: v.reshape(-1, v.shape[-1])[:n].contiguous() for k, v in batch.items()}
    flat["rand_vec"] = torch.randn(n, 6, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    model.max_chunk_rays = 4096
    out = {}
    with torch.no_grad():
        for ov in (False, True, 2):                           # 2: the same two streams with the plain launch shapes
            model.overlap_streams = ov
            r, _ = model(False, flat, 1.0, True)
            torch.cuda.synchronize()
            out[ov] = {k: r[-1][k].clone() for k in ("rgb", "depth", "acc", "weights")}
    model.overlap_streams = False
    for k in out[False]:
        assert torch.equal(out[False][k], out[True][k]), k
        assert torch.equal(out[False][k], out[2][k]), k


@pytest.mark.gpu
def test_bench_two_ranks_flow_on_one_gpu():
    """`python bench.py --gpus 2` with no launcher around it (the driver's N = 1 command form; bench.py re-executes itself under
    `python -m torch.distributed.run --nproc-per-node 2 ...`, which is also the driver's N > 1 form, and refuses a WORLD_SIZE that
    differs from --gpus) end to end on what a 1-GPU box allows: two ranks sharing the device with UCN_DIST_BACKEND=gloo (RCCL refuses two ranks on one
    GPU; its collective is covered by test_the_frame_exchange_runs_on_rccl).  Rank 0 prints ONE JSON line whose value is
    the whole job's rays/s, n_gpus = 2, scaling = strong; the frame each rank returns is the single-process frame."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UCN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    # the driver's N = 1 command form with N = 2: plain `python bench.py --gpus 2`, NO launcher -- bench.py must start the ranks itself
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=repo)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, p.stdout[-2000:] + p.stderr[-3000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 1 and r["scaling"] == "strong" and r["unit"] == "rays/s"
    assert r["launch"].startswith("self-launched") and r["dist_ranks"] == 2 and r["dist_backend"] == "gloo" and r["rccl_ranks"] == 0
    assert r["all_gather"]["frames"] == 1 and r["all_gather"]["ms_per_frame"] > 0
    assert r["all_gather"]["bytes_sent_per_rank"] == 1280 * 1920 // 2 * 36
    assert r["config"]["rays_per_step"] == 1280 * 1920 and "x2" in r["config"]["parallelism"]
    assert 1e5 < r["value"] < 2e7 and abs(r["value"] - 1280 * 1920 / (r["ms_per_step"] * 1e-3)) <= 1e-3 * r["value"]
    assert r["roofline"]["bound"] in ("hbm", "mfma") and 0 < r["roofline"]["frac"] <= 1.2
    assert "cpu_baseline" not in r and "train_step" not in r          # rank-0-at-N=1-only extras stay out of the N > 1 line
    d = r["train_step_ddp"]                                            # BASELINE's second metric at N > 1: the DDP training step
    assert "error" not in d, d
    assert d["n_gpus"] == 2 and d["rays"] == 8192 and d["rays_per_rank"] == 4096 and 1.0 < d["ms"] < 2000.0
    # both gradient exchanges are timed: DDP all-reduce, and reduce-scatter -> sharded Adam -> all-gather
    assert set(d["modes"]) == {"all_reduce", "reduce_scatter"} and all("error" not in m for m in d["modes"].values()), d["modes"]
    rs, ar = d["modes"]["reduce_scatter"]["grad_exchange"], d["modes"]["all_reduce"]["grad_exchange"]
    assert rs["mode"] == "reduce_scatter" and rs["table_gradient_bytes"] == (7131240 + 1888360) * 2 * 4 == ar["table_gradient_bytes"]
    assert rs["bytes_on_the_wire_per_rank_before_the_optimiser"] < ar["bytes_on_the_wire_per_rank_before_the_optimiser"]


def test_item_list_backward_is_the_same_adjoint():
    """The fine levels' second route through the table gradient (UCN_BWD_LISTS=1: counting sort of (point, (y, z) combination)
    items into per-block lists, march_features.hip k_bwd_bin / k_bwd_list; off by default on measurement) must satisfy the
    same exact-adjoint and parity tests as the compacted kernel.  The switch is read once per process: a child pytest."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UCN_BWD_LISTS="1")
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-k",
                        "(adjoint or features_backward or train_graph_matches_reference_step) and not item_list and not mask_plane_routes",
                        os.path.join(repo, "tests", "test_full_size.py"), os.path.join(repo, "tests", "test_gpu_parity.py"),
                        os.path.join(repo, "tests", "test_train_step.py")],
                       capture_output=True, text=True, timeout=1200, env=env, cwd=repo)
    assert p.returncode == 0 and " passed" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]


@pytest.mark.parametrize("knobs", [dict(UCN_BWD_BYTE_MASKS="0", UCN_BWD_COARSE_RES="512"), dict(UCN_BWD_BYTE_MASKS="3")],
                         ids=["r05_plan_nibble_planes", "byte_scan_with_corner_walk"])
def test_mask_plane_routes_of_the_table_gradient_are_the_same_adjoint(knobs):
    """r06: the point-item levels of the table gradient read BYTE planes (MaskPlan::fine_kind 3, cmp_block_bytes), and under fixed-point
    rows the hashed levels of resolution 84 ... 446 are point items as well -- the defaults every other test of this suite runs.  The
    routes behind the switches (the r05 plan: nibble planes + sample items up to 512; the byte scan with the corner walk, fine_kind 4)
    must pass the same adjoint / parity / fixed-point / full-size training tests.  The switches are read once per process: a child pytest."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **knobs)
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-k",
                        "(adjoint or features_backward or fixed_point or random_grid_shapes or train_graph_matches_reference_step) "
                        "and not item_list and not mask_plane_routes",
                        os.path.join(repo, "tests", "test_full_size.py"), os.path.join(repo, "tests", "test_gpu_parity.py"),
                        os.path.join(repo, "tests", "test_train_step.py")],
                       capture_output=True, text=True, timeout=1200, env=env, cwd=repo)
    assert p.returncode == 0 and " passed" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]


def test_backward_of_a_grid_with_more_than_32_row_blocks_per_level():
    """The reference's own waymo.gin grid (L = 10, C = 4, T = 2^21: 256 row blocks of 8192 rows per hashed level) goes through the
    row-block kernel as well (multi-word sample masks, march_features.hip `coarse == 3`): same table gradient as the global-atomic
    kernel (other summation order), and the exact adjoint of the forward, level by level."""
    from ucnerf_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    model, cfg, sd = bench.build_model(dev, grid="R")
    rays = bench.frame_rays(dev)
    n_total = bench.H_IMG * bench.W_IMG
    flat = {k: v.reshape(n_total, -1) for k, v in rays.items()}
    n = 4096
    batch = _pick(flat, n_total, n, seed=11)
    with torch.no_grad():
        _, hist = model(False, batch, 1.0, True)
    for mlp, lvl in ((model.nerf_mlp, -1), (model.prop_mlp_0, 0)):
        enc = mlp.encoder
        L, C = enc.num_levels, enc.level_dim
        assert C == 4 and int(enc._offsets_np[-1] - enc._offsets_np[-2]) == 2 ** 21
        sdist = hist[lvl]["sdist"].reshape(n, -1).contiguous()
        S = sdist.shape[1] - 1
        basis = torch.empty(n, 6, device="cuda")
        rvec = (batch["rand_vec"][:, 0:3] if lvl == 0 else batch["rand_vec"][:, 3:6]).contiguous()   # (any basis: forward and backward share it)
        _lib.check(lib.ucn_cone_basis(batch["cam_dirs"].data_ptr(), rvec.data_ptr(), n, basis.data_ptr(), _lib.stream()))
        near, far, rad = (batch[k].reshape(-1).contiguous() for k in ("near", "far", "radii"))
        geom = (sdist, near, far, batch["origins"], batch["directions"], basis, rad, None, None)
        std_scale = float(model.std_scale)
        g = torch.Generator(device="cuda").manual_seed(12)
        T2 = torch.rand(enc.embeddings.shape, device="cuda", generator=g) * 2 - 1
        d = _lib.UcnField.from_buffer_copy(mlp.field())
        d.embeddings = T2.data_ptr()
        f2 = _features(lib, _lib, d, geom, n, S, std_scale, L, C)
        grad = torch.randn(L, n * S, C, device="cuda", generator=g)
        grad[:, ::7] = 0.0                                                     # samples without gradient are dropped by the masks
        ws = torch.empty(lib.ucn_march_features_backward_ws_floats(ctypes.byref(mlp.field()), n, S), device="cuda")
        gt, ga = torch.zeros_like(T2), torch.zeros_like(T2)
        _lib.check(lib.ucn_march_features_backward(ctypes.byref(mlp.field()), *[_lib.ptr(t) for t in geom], std_scale, n, S,
                                                   0, 0, grad.data_ptr(), gt.data_ptr(), ws.data_ptr(), _lib.stream()))
        _lib.check(lib.ucn_march_features_backward(ctypes.byref(mlp.field()), *[_lib.ptr(t) for t in geom], std_scale, n, S,
                                                   1, 0, grad.data_ptr(), ga.data_ptr(), None, _lib.stream()))
        assert float(gt.abs().max()) > 0
        assert float((gt - ga).abs().max()) <= 2e-5 * max(1.0, float(ga.abs().max()))
        off = enc._offsets_np
        for l in range(L):
            a = float((grad[l].double() * f2[l].double()).sum())
            b = float((gt[int(off[l]):int(off[l + 1])].double() * T2[int(off[l]):int(off[l + 1])].double()).sum())
            s = float((grad[l].double() * f2[l].double()).abs().sum())
            assert abs(a - b) <= 1e-5 * s, (l, a, b, s)


def test_table_gradient_over_random_grid_shapes():
    """tools/fuzz_backward_grids.py: the row-block table gradient (persistent workgroups; point items, sample items, the path for
    levels of more than 32 row blocks) == the global-atomic kernel over random (L, C, T, resolutions, N, S): C = 1 / 2 / 4 / 8, 1 to 256
    blocks per level, dense / hashed / strided levels, N * S from 1 to 640 000, zero-gradient samples, rays outside the unit cube."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(repo, "tools", "fuzz_backward_grids.py"), "32"], capture_output=True, text=True,
                       timeout=900, cwd=repo)
    assert p.returncode == 0 and "mismatches: 0" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]


def _check_other_keys(got, want, n, mixed=False, tol_scale=1.0):
    """acc, depth, distance_median of one level's rendering against the oracle's: acc <= 1e-4; depth <= 1e-3 (t-units; the clip of
    render.py:205-207 and the division by acc amplify the weights' 1e-5) on rays whose acc is not within 2e-3 of the 0.6 switch of
    render.py:208-213 -- both sides must agree on WHICH rays carry the sentinel there --; distance_median: the weighted-percentile
    interpolation (stepfun.py:329-339) is piecewise linear in the cumulative weights, 1e-3 away from its own breakpoints.
    The mixed route (bf16 dense layers) gets 50 x."""
    f = (50.0 if mixed else 1.0) * tol_scale
    g_acc, w_acc = got["acc"].reshape(n).float().cpu(), want["acc"].reshape(n)
    assert float((g_acc - w_acc).abs().max()) <= 1e-4 * f, float((g_acc - w_acc).abs().max())
    g_d, w_d = got["depth"].reshape(n).float().cpu(), want["depth"].reshape(n)
    safe = (w_acc - 0.6).abs() > 2e-3 * f
    assert bool(((g_d == 300.0) == (w_d == 300.0))[safe].all()), "depth sentinel on different rays"
    assert float((g_d - w_d)[safe].abs().max()) <= 1e-3 * f, float((g_d - w_d)[safe].abs().max())
    if "distance_median" in got and "distance_median" in want:
        g_m, w_m = got["distance_median"].reshape(n).float().cpu(), want["distance_median"].reshape(n)
        d = (g_m - w_m).abs()
        # a percentile that falls on a breakpoint of the CDF may land in either neighbouring interval: allow 0.5 % of the rays to differ more
        assert float(torch.quantile(d, 0.995)) <= 1e-3 * f, float(torch.quantile(d, 0.995))


def _oracle_vs_gpu_on_the_frame(heads, autocast, mlp_mode=None, n=2048, grid="B"):
    """`n` strided rays of bench.py's frame (the headline workload's own rays, weights and cone-basis draws) through the product
    path and through the CPU oracle at FULL table size (T = 2^19, 16 levels, 64 + 128 samples, 256-wide colour MLP): the
    headline parity as a test, not only as a number bench.py prints.  ~2 s of oracle time at ~2 k rays/s."""
    from ucnerf_amd.internal import models
    dev = torch.device("cuda", 0)
    model, cfg, sd = bench.build_model(dev, heads=heads, grid=grid)
    saved = models.MLP.mlp_mode
    try:
        if mlp_mode is not None:
            for m in (model.nerf_mlp, model.prop_mlp_0):
                m.mlp_mode = mlp_mode
        rays = bench.frame_rays(dev, virtual=heads)
        n_total = bench.H_IMG * bench.W_IMG
        idx = torch.linspace(0, n_total - 1, n).long()
        flat = {k: v.reshape(n_total, -1)[idx.to(dev)].contiguous() for k, v in rays.items()}
        rand_vec = torch.randn(n, 6, generator=torch.Generator().manual_seed(1))
        batch = dict(flat, rand_vec=rand_vec.to(dev))
        eval_camidx = torch.tensor([7]) if heads else None
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            rend, _ = model(False, batch, 1.0, True, eval_camidx=eval_camidx)
        got = rend[-1]["rgb"].reshape(n, 3).float().cpu()
        spec = rm.make_spec(grid, model_sky=True, brightness_correction=True, training_views=210) if heads else rm.make_spec(grid)
        torch.set_num_threads(min(32, torch.get_num_threads()))
        noise = [rm.LevelNoise(rand_vec=rand_vec[:, 3 * l:3 * l + 3]) for l in range(2)]
        with torch.no_grad():
            want_all, _ = rm.model_forward(spec, sd, {k: v.cpu() for k, v in flat.items()}, noise, eval_camidx=eval_camidx)
        want = want_all[-1]["rgb"].reshape(n, 3)
        # r06: the other keys the reference returns (render.py:177-244) at full size too -- acc, depth with its
        # acc < 0.6 -> 300 sentinel (compared away from the switch), the median distance of the extras
        _check_other_keys(rend[-1], want_all[-1], n, autocast)
    finally:
        models.MLP.mlp_mode = saved
    linf = float((got - want).abs().max())
    psnr = float(-10 * np.log10(max(float(((got - want) ** 2).mean()), 1e-20)))
    return linf, psnr


@pytest.mark.parametrize("mlp_mode", [1, 0])
def test_config_B_full_tables_vs_oracle(mlp_mode):
    """BASELINE configs[1] (the headline): RGB L-inf <= 1e-4 against the oracle (north_star's fp32 bar), in the default split-f16
    MFMA mode (fp32-class) and in the exact fp32-input MFMA mode."""
    linf, psnr = _oracle_vs_gpu_on_the_frame(heads=False, autocast=False, mlp_mode=mlp_mode)
    print(f"config B, mlp_mode {mlp_mode}: rgb L-inf {linf:.3e}, PSNR {psnr:.1f} dB")
    assert linf <= 1e-4 and psnr >= 95.0, (linf, psnr)


def test_config_B_with_heads_full_tables_vs_oracle():
    """BASELINE configs[4]'s model (sky NeRF layer + colour-correction head, rays of perturbed poses, eval_camidx = 7) in fp32-class
    arithmetic: the same 1e-4 bar."""
    linf, psnr = _oracle_vs_gpu_on_the_frame(heads=True, autocast=False)
    print(f"config B + heads: rgb L-inf {linf:.3e}, PSNR {psnr:.1f} dB")
    assert linf <= 1e-4 and psnr >= 95.0, (linf, psnr)


def test_config_B_mixed_precision_route_has_a_stated_bar():
    """The 'mixed bf16/fp32' route of BASELINE configs[4] (render under a bf16 autocast: half tables, bf16 MFMA dense layers, fp32
    resampling / compositing) is NOT the fp32 route and is not held to 1e-4.  Its bar: RGB L-inf <= 2e-3 (half a bf16 step of
    a value in [0.5, 1): one rounding of the output's size) and PSNR >= 70 dB against the fp32 oracle -- an 8-bit image differs
    from its source by 48 dB.  The measured figures are printed (driver's r03 line: L-inf 1.5e-4 on a 5-camera frame)."""
    linf, psnr = _oracle_vs_gpu_on_the_frame(heads=True, autocast=True)
    print(f"config B + heads, mixed: rgb L-inf {linf:.3e}, PSNR {psnr:.1f} dB")
    assert linf <= 2e-3 and psnr >= 70.0, (linf, psnr)


def test_waymo_gin_grid_full_tables_vs_oracle():
    """The reference's own waymo.gin grid (L = 10, C = 4, T = 2^21, 128 + 32 samples) at full table size: the same 1e-4 bar."""
    linf, psnr = _oracle_vs_gpu_on_the_frame(heads=False, autocast=False, n=1024, grid="R")
    print(f"waymo.gin grid: rgb L-inf {linf:.3e}, PSNR {psnr:.1f} dB")
    assert linf <= 1e-4 and psnr >= 95.0, (linf, psnr)


def test_config3_five_camera_frame_row_tiles_vs_oracle():
    """BASELINE configs[3] inside -m gpu: the 5-camera 1280x1920 frame of bench.py (frame_rays(n_cams=5): five poses of the
    trajectory, camera c = rows 1280 c ... 1280 c + 1279).  One 8-row tile of EACH camera (taken at a different height per
    camera) goes through models.render_image -- the frame's own code path: tile-major march order, chunking, the packed
    exchange buffer -- and 2048 strided rays of the 76,800 are checked against the CPU oracle at the 1e-4 bar."""
    from ucnerf_amd.internal import models
    dev = torch.device("cuda", 0)
    model, cfg, sd = bench.build_model(dev)
    cfg.render_ray_tile = 8
    cfg.render_gather_weights = False
    rays = bench.frame_rays(dev, 5)
    rows = torch.cat([torch.arange(8) + 1280 * c + 8 * (17 + 29 * c) for c in range(5)]).to(dev)
    batch = {k: v[rows].contiguous() for k, v in rays.items()}
    n = rows.numel() * bench.W_IMG
    rand_vec = torch.randn(n, 6, generator=torch.Generator().manual_seed(1))
    batch["rand_vec"] = rand_vec.reshape(rows.numel(), bench.W_IMG, 6).to(dev)
    out = models.render_image(model, bench.Ranks(1, 0), batch, False, 1.0, cfg, verbose=False, eval_camidx=0)
    got = out["rgb"].reshape(n, 3).float().cpu()
    assert torch.isfinite(got).all()
    # the five cameras really differ -- a rig: one position, five yaw angles (bench.frame_cameras) -- a frame_rays that repeated
    # camera 0 five times would pass a per-ray oracle check
    c = batch["cam_dirs"].reshape(5, -1, 3)[:, 0]
    assert float((c[:, None, :] - c[None, :, :]).norm(dim=-1)[~torch.eye(5, dtype=torch.bool, device=dev)].min()) > 0.1
    idx = torch.linspace(0, n - 1, 2048).long()
    flat = {k: v.reshape(n, -1)[idx.to(dev)].cpu() for k, v in batch.items() if k != "rand_vec"}
    noise = [rm.LevelNoise(rand_vec=rand_vec[idx, 3 * l:3 * l + 3]) for l in range(2)]
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        want, _ = rm.model_forward(rm.make_spec("B"), sd, flat, noise)
    want = want[-1]["rgb"].reshape(-1, 3)
    per_cam = [(float((got[idx][c * 2048 // 5:(c + 1) * 2048 // 5] - want[c * 2048 // 5:(c + 1) * 2048 // 5]).abs().max())) for c in range(5)]
    print("configs[3] row tiles, rgb L-inf per camera:", ["%.2e" % v for v in per_cam])
    assert max(per_cam) <= 1e-4, per_cam


def test_fitted_field_march_and_compaction_vs_oracle():
    """VERDICT r05 missing #2 / weak #4: every other oracle comparison runs on random-initialised weights with U(-1, 1) tables.  Here
    bench.build_model() starts from the reference's own initialisation (tables +-1e-4, grid.py:151-153) and is FITTED for 300 steps to
    tools/fit_scene.py's analytic scene (ground plane + spheres: opaque surfaces, empty space in front of them) with this repo's
    training graph; then 1 024 strided rays of the benchmark frame go through the product's inference march (the route render_image
    takes: no per-sample history) and through the CPU oracle on the fitted state_dict (reference semantics: every sample evaluated,
    models.py:221-311, render.py:155-244).

    1. The fitted field as it is, with compact_min_weight = 0 and with 4e-8 (north_star's early-termination sample compaction: ballot /
       prefix-sum alive list, the colour layers only for the samples whose compositing weight reaches the threshold): acc <= 1e-4,
       depth <= 1e-3 away from the 0.6 switch, rgb: 99 % of the rays <= 1e-4 (north_star's bar) and the worst ray <= 3e-4 (+ num_nerf_samples x threshold).
       The worst-ray bar is NOT north_star's 1e-4, and the numbers say why: over fits of 100-300 steps and two seeds the worst of 1 024
       rays measured 0.8e-4 ... 1.6e-4 (median ray 8e-6; the fit itself is not bit-reproducible: its table gradients meet in float
       atomics) -- on a fitted field a few rays graze a surface, where one ulp in a resampled position moves weight between
       differently coloured samples.  That is fp32 itself, not the split-f16 engine: the exact-fp32-product mode is asserted to
       be as far from the oracle (mode 1 <= 1.5 x mode 0 + 2e-5), and both engines agree with each other to 5e-5 (measured 1.9e-5).  (The benchmark's
       random-init field holds 1e-4 with a factor 3 to spare: test_config_B_full_tables_vs_oracle.)
       MEASURED alive fraction: 1.0 -- after such a fit the proposal resampling has moved all 128 samples onto the surfaces, every
       sample weighs >= 1e-5 -- so on this field the compacted route is exercised end to end but removes nothing (the reason it is
       off by default).
    2. The same field with its density logits SHARPENED x 16 (row 0 of density_layer.2: surfaces turn hard, empty space emptier): 68 %
       of the samples reach 4e-8, 38 % reach 1e-5 -- the regime compaction is for.  Densities of 1e3-1e4 amplify a 1-ulp difference in
       a sample position into 1e-3 of a pixel on BOTH dense-layer engines (printed: the exact-fp32 mode is as far from the oracle as
       the split-f16 mode), so this field's bar against the oracle is 5e-3; what is asserted of the compaction is that it adds at
       most num_nerf_samples x threshold to the uncompacted route's distance from the ORACLE, and that the alive fraction is a real
       reduction."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fit_scene
    dev = torch.device("cuda", 0)
    torch.manual_seed(20)
    model, cfg, _ = bench.build_model(dev)
    for mlp in (model.nerf_mlp, model.prop_mlp_0):
        mlp.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
    fit_scene.fit(model, dev, 300)
    rays = bench.frame_rays(dev)
    n_total, n = bench.H_IMG * bench.W_IMG, 1024
    idx = torch.linspace(0, n_total - 1, n).long()
    flat = {k: v.reshape(n_total, -1)[idx.to(dev)].contiguous() for k, v in rays.items()}
    rand_vec = torch.randn(n, 6, generator=torch.Generator().manual_seed(1))
    batch = dict(flat, rand_vec=rand_vec.to(dev))
    torch.set_num_threads(min(32, torch.get_num_threads()))
    noise = [rm.LevelNoise(rand_vec=rand_vec[:, 3 * l:3 * l + 3]) for l in range(2)]
    S = model.num_nerf_samples

    def oracle():
        sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        with torch.no_grad():
            want, hist = rm.model_forward(rm.make_spec("B"), sd, {k: v.cpu() for k, v in flat.items()}, noise)
        return want[-1], hist[-1]["weights"].reshape(n, -1)

    def march(thr, mode=1):
        model.compact_min_weight = thr
        model._alive_stats = [] if thr > 0 else None
        keep = model.nerf_mlp.mlp_mode
        model.nerf_mlp.mlp_mode = mode
        try:
            with torch.no_grad():
                rend, _ = model._march(False, batch, 1.0, True, None, want_history=False)
            alive = (sum(a for a, _ in model._alive_stats) / max(1, sum(b for _, b in model._alive_stats))) if thr > 0 else None
        finally:
            model.compact_min_weight = 0.0
            model._alive_stats = None
            model.nerf_mlp.mlp_mode = keep
        return rend[-1], alive

    linf = lambda got, want: float((got["rgb"].reshape(n, 3).float().cpu() - want["rgb"].reshape(n, 3)).abs().max())
    # ---- 1. the fitted field
    want, w = oracle()
    true_rgb, _ = fit_scene.scene_colour(flat["origins"], flat["directions"])
    mse = float(((want["rgb"].reshape(n, 3) - true_rgb.cpu()) ** 2).mean())
    print(f"fitted field (300 steps): oracle PSNR vs the analytic scene {-10 * np.log10(max(mse, 1e-12)):.2f} dB, acc mean {float(want['acc'].mean()):.3f}, "
          f"samples with weight >= 4e-8: {float((w >= 4e-8).float().mean()):.4f}")
    assert float(want["acc"].mean()) > 0.3, "the fit did not produce surfaces"
    per_ray = lambda got: (got["rgb"].reshape(n, 3).float().cpu() - want["rgb"].reshape(n, 3)).abs().max(dim=1).values
    exact, _ = march(0.0, mode=0)
    e_exact = linf(exact, want)
    for thr in (0.0, 4e-8):
        got, alive = march(thr)
        e, q99 = linf(got, want), float(torch.quantile(per_ray(got), 0.99))
        d_modes = float((got["rgb"].reshape(n, 3).float() - exact["rgb"].reshape(n, 3).float()).abs().max())
        print(f"  compact_min_weight {thr:g}: rgb vs oracle: worst ray {e:.3e}, 99 % of the rays <= {q99:.3e}, median {float(per_ray(got).median()):.2e}; "
              f"exact-fp32 mode: worst ray {e_exact:.3e}; split vs exact mode {d_modes:.2e}" + ("" if alive is None else f"; alive fraction {alive:.4f}"))
        assert e <= 3e-4 + S * thr and q99 <= 1e-4 + S * thr, (thr, e, q99)
        assert e <= 1.5 * e_exact + 2e-5 + S * thr and d_modes <= 5e-5 + S * thr, (e, e_exact, d_modes)
        _check_other_keys(got, want, n)
    # ---- 2. the sharpened field
    with torch.no_grad():
        model.nerf_mlp.density_layer[2].weight[0].mul_(16.0)
        model.nerf_mlp.density_layer[2].bias[0].mul_(16.0)
    want, w = oracle()
    plain, _ = march(0.0)
    exact, _ = march(0.0, mode=0)
    e0, e_exact = linf(plain, want), linf(exact, want)
    print(f"sharpened x 16: samples with weight >= 4e-8 (oracle): {float((w >= 4e-8).float().mean()):.4f}; uncompacted rgb L-inf vs oracle: "
          f"split-f16 {e0:.3e}, exact fp32 {e_exact:.3e}")
    assert e0 <= 5e-3 and e_exact <= 5e-3, (e0, e_exact)
    assert e0 <= 3 * e_exact + 1e-4, (e0, e_exact)                 # the split engine is not what the distance comes from
    for thr in (4e-8, 1e-5):
        got, alive = march(thr)
        e, d = linf(got, want), float((got["rgb"].reshape(n, 3).float() - plain["rgb"].reshape(n, 3).float()).abs().max())
        print(f"  compact_min_weight {thr:g}: alive fraction {alive:.4f}, rgb L-inf vs oracle {e:.3e}, vs the uncompacted march {d:.3e}")
        assert e <= e0 + S * thr and d <= S * thr, (thr, e, e0, d)
        assert abs(alive - float((w >= thr).float().mean())) <= 0.02, (alive, float((w >= thr).float().mean()))   # the alive list = the oracle's count
        assert alive < 0.8, alive
        assert torch.equal(got["acc"], plain["acc"]) and torch.equal(got["depth"], plain["depth"])     # density side untouched
