"""bench.py -- rays/s of the forward render hot path on N MI355X GPUs (one node).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the hot path over one synthetic 1280x1920 frame (2,457,600 rays):
BASELINE.json configs[1] -- proposal 64 + NeRF 128 samples, NeRF grid L=16 C=2 T=2^19, proposal grid
L=6 C=2 T=2^19, 256-wide colour MLP, fp32, rand=False, compute_extras=True (what render_image does).
Rays and weights are resident in HBM before the timed region.  With N > 1 the frame's rays are
sharded row-contiguously over the ranks (strong scaling) and the finished buffers are exchanged with
one packed RCCL all-gather per frame, inside the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

H_IMG, W_IMG, FOCAL = 1280, 1920, 2000.0
S_PROP, S_NERF = 64, 128
# algorithmic costs per ray (DESIGN.md "Rooflines")
GATHER_BYTES_NERF = S_NERF * 6 * 16 * 8 * 2 * 4          # 786,432 B: samples x multisamples x levels x corners x C x 4
GATHER_BYTES_PROP = S_PROP * 6 * 6 * 8 * 2 * 4           # 147,456 B
MAC_NERF = 32 * 64 + 64 * 256 + 283 * 256 + 539 * 256 + 256 * 3      # 229,632 MAC / sample (reference formulation)
FLOP_NERF_RAY = S_NERF * 2 * MAC_NERF
PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
PROF_EVERY = 8                # live per-kernel times: HIP events around every 8th pass of the timed frames (see main())
PEAK_F32_MFMA_TF = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
PEAK_F16_MFMA_TF = 2500.0    # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak
# MACs the split-f16 kernel issues per sample: first layer padded to 64 inputs; the bottleneck is composed away
# (field_mlp_h.hip), so both colour layers see [h0 (64) | direction+bias tile (32)] = 96 inputs, layer 1 also h1 (256)
MAC_NERF_SPLIT = 64 * 64 + 96 * 256 + (256 + 96) * 256


def frame_cameras(n_cams=1, virtual=False):
    """Pinhole cameras with Waymo-like intrinsics: (pixtocams, camtoworlds, distortion, ndc) as the reference's
    Dataset.cameras tuple (datasets.py:346-349; pixtocam = inv(K) in float64, datasets.py:855).  Camera 0 is the
    single-camera benchmark pose; the others fan out in yaw like a five-camera rig (front, +-50, +-100 degrees).
    virtual=True perturbs every pose the way the reference builds its virtual cameras (datasets.py:983-1063: a small
    rotation about each axis and a shift of a few percent of the scene radius)."""
    K = np.array([[FOCAL, 0.0, W_IMG / 2], [0.0, FOCAL, H_IMG / 2], [0.0, 0.0, 1.0]])
    rig = [0.3, 0.3 + 0.87, 0.3 - 0.87, 0.3 + 1.75, 0.3 - 1.75, 0.3 + 2.6, 0.3 - 2.6, 0.3 + 3.1]
    rng = np.random.default_rng(7)
    c2ws = []
    for i in range(n_cams):
        yaw = rig[i % len(rig)] + 0.05 * (i // len(rig))
        R = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
        t = np.array([0.1, -0.05, 0.2])
        if virtual:
            ax, ay, az = rng.uniform(-0.03, 0.03, 3)
            Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
            Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
            Rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
            R = Rz @ Ry @ Rx @ R
            t = t + rng.uniform(-0.03, 0.03, 3)
        c2ws.append(np.concatenate([R, t[:, None]], axis=1))
    return (np.repeat(np.linalg.inv(K)[None], n_cams, 0), np.stack(c2ws), None, None)


def frame_rays(device, n_cams=1, virtual=False):
    """Every pixel of every camera as the model's ray batch ([n_cams * H, W, .]: the cameras stacked along the rows,
    so that row-tile sharding hands whole-camera or partial-camera tiles to the ranks), generated ON the device by the
    path's own ray generator (ucnerf_amd/internal/camera_utils.py -> ucn_generate_rays, bit-identical
    to the reference's camera_utils.pixels_to_rays + datasets._make_ray_batch); outside the timed region, timed
    separately."""
    from ucnerf_amd.internal import camera_utils
    cams = frame_cameras(n_cams, virtual)
    keys = ("origins", "directions", "viewdirs", "cam_dirs", "radii", "near", "far", "cam_idx", "lossmult")
    per = [camera_utils.generate_ray_batch(cams, i, W_IMG, H_IMG, 0.0, 8.0, device=device) for i in range(n_cams)]
    return {k: torch.cat([b[k] for b in per], dim=0) for k in keys}


def ray_generation_ms(device, steps=20):
    """one full frame of rays per launch; 68 B written per ray, nothing read but two 3x4 matrices."""
    from ucnerf_amd.internal import camera_utils
    cams = frame_cameras()
    for _ in range(3):
        camera_utils.generate_ray_batch(cams, 0, W_IMG, H_IMG, 0.0, 8.0, device=device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        camera_utils.generate_ray_batch(cams, 0, W_IMG, H_IMG, 0.0, 8.0, device=device)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    n = H_IMG * W_IMG
    return dict(ms_per_frame=ms, rays_per_s=n / (ms * 1e-3), bytes_per_ray=76, achieved_GBps=n * 76 / (ms * 1e-3) / 1e9,
                peak_GBps=PEAK_HBM_GBS, kernel="k_generate_rays (float64 arithmetic, float32 stores; incl. output allocation)",
                reference="numpy on DataLoader workers + 64 B/ray over PCIe (157 MB per frame)")


def virtual_warp_ms(device, steps=20):
    """train_utils.img_warping on one full 1280x1920 depth map (what datasets.py:527 does on the CPU
    every training step): 4 B read, 9 B written per pixel."""
    from ucnerf_amd.internal import train_utils as tu
    g = torch.Generator(device=device).manual_seed(3)
    depth = torch.rand(H_IMG, W_IMG, device=device, generator=g) * 10 + 2
    depth[torch.rand(H_IMG, W_IMG, device=device, generator=g) < 0.4] = 0
    K = np.array([[FOCAL, 0.0, W_IMG / 2], [0.0, FOCAL, H_IMG / 2], [0.0, 0.0, 1.0]])
    ref = np.eye(4)
    src = np.eye(4)
    src[:3, 3] = [0.3, 0.35, 0.1]
    for _ in range(3):
        tu.img_warping(ref, src, depth, K)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        pts, mask = tu.img_warping(ref, src, depth, K)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    n = H_IMG * W_IMG
    return dict(ms_per_frame=ms, pixels_per_s=n / (ms * 1e-3), achieved_GBps=n * 13 / (ms * 1e-3) / 1e9, peak_GBps=PEAK_HBM_GBS,
                valid_fraction=float(mask.float().mean()), kernel="k_img_warp (incl. the host-side 4x4 inverse and the bool cast)")


def density_query_ms(model, device, side=256, chunk=1 << 21):
    """the field as extract.py:28-64 `evaluate_density` consumes it (marching cubes / TSDF): a dense
    side^3 lattice of points through nerf_mlp.predict_density(no_warp=True) in chunks."""
    lin = torch.linspace(-1, 1, side, device=device)
    pts = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(-1, 3)
    mlp = model.nerf_mlp

    def sweep():
        out = []
        for p in torch.split(pts, chunk, dim=0):
            raw = mlp.predict_density(p[:, None], torch.zeros_like(p[:, :1]), no_warp=True)[0]
            out.append(torch.nn.functional.softplus(raw + mlp.density_bias))
        return torch.cat(out)
    with torch.no_grad():
        sweep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        z = sweep()
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    # ... and the mesh the reference takes from that lattice (extract.py:452 skimage.measure.marching_cubes on the host after
    # a device-to-host copy): csrc/mesh.hip on the device, at the median density (a surface through the whole random field)
    from ucnerf_amd.internal import mesh
    vol = z.reshape(side, side, side)
    level = float(vol.median())
    mesh.marching_cubes(vol, level)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    v, f, n, _ = mesh.marching_cubes(vol, level, spacing=(2 / (side - 1),) * 3)
    torch.cuda.synchronize()
    mc_ms = (time.perf_counter() - t0) * 1e3
    return dict(points=pts.shape[0], ms=ms, points_per_s=pts.shape[0] / (ms * 1e-3), finite=bool(torch.isfinite(z).all()),
                call="nerf_mlp.predict_density(means[:, None], stds[:, None], no_warp=True)  (extract.py:54)",
                marching_cubes=dict(ms=mc_ms, lattice=f"{side}^3", level=level, vertices=int(v.shape[0]), triangles=int(f.shape[0]),
                                    cells_per_s=(side - 1) ** 3 / (mc_ms * 1e-3),
                                    kernel="k_mc_count / k_mc_scan / k_mc_verts / k_mc_faces (indexed mesh, incl. the host read of the two counts)"))


def tsdf_fusion_ms(device, resolution=512, views=4):
    """TSDF.integrate_tsdf (reference tsdf.py:115-219) of `views` 1280x1920 depth + colour images into a
    512^3 volume (the reference's default): one launch, one read-modify-write of the volume (36 B per voxel)."""
    import types
    from ucnerf_amd.internal.tsdf import TSDF
    cfg = types.SimpleNamespace(tsdf_radius=2.0, tsdf_resolution=resolution, truncation_margin=5.0, tsdf_max_radius=10.0)
    acc = types.SimpleNamespace(device=device, num_processes=1, process_index=0, is_main_process=True)
    vol = TSDF(cfg, acc)
    g = torch.Generator(device=device).manual_seed(4)
    depth = torch.rand(views, 1, H_IMG, W_IMG, device=device, generator=g) * 4 + 1
    color = torch.rand(views, 3, H_IMG, W_IMG, device=device, generator=g)
    c2w = torch.eye(4, device=device)[None].repeat(views, 1, 1)
    c2w[:, 0, 3] = torch.linspace(-0.3, 0.3, views, device=device)
    K = torch.tensor([[FOCAL, 0.0, W_IMG / 2], [0.0, FOCAL, H_IMG / 2], [0.0, 0.0, 1.0]], device=device)
    vol.integrate_tsdf(c2w, K, depth, color)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        vol.integrate_tsdf(c2w, K, depth, color)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    n = resolution ** 3
    return dict(ms=ms, voxels=n, views=views, voxel_views_per_s=n * views / (ms * 1e-3), volume_GBps=n * 56 / (ms * 1e-3) / 1e9,
                peak_GBps=PEAK_HBM_GBS, updated_fraction=float((vol.weights > 0).float().mean()),
                kernel="k_tsdf_integrate (thread = voxel, the call's views in registers; 16 B world + 20 B state read, 20 B written)")


def hbm_probe(device, n_floats=1 << 28, steps=10):
    """What a plain device-to-device copy kernel (`ucn_probe_copy`, float4 per lane) reaches on this part: the
    practical ceiling behind the 8 TB/s spec figure that `roofline.peak` uses."""
    import ctypes  # noqa: F401
    from ucnerf_amd import _lib
    lib = _lib.load()
    src = torch.empty(n_floats, device=device).normal_()
    dst = torch.empty_like(src)
    for _ in range(2):
        _lib.check(lib.ucn_probe_copy(src.data_ptr(), dst.data_ptr(), n_floats, _lib.stream()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        _lib.check(lib.ucn_probe_copy(src.data_ptr(), dst.data_ptr(), n_floats, _lib.stream()))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    os.environ["UCN_PROBE_COPY_UNROLL"] = "1"           # the same copy with four loads in flight per thread (4096 workgroups)
    try:
        for _ in range(2):
            _lib.check(lib.ucn_probe_copy(src.data_ptr(), dst.data_ptr(), n_floats, _lib.stream()))
        e0.record()
        for _ in range(steps):
            _lib.check(lib.ucn_probe_copy(src.data_ptr(), dst.data_ptr(), n_floats, _lib.stream()))
        e1.record()
        torch.cuda.synchronize()
        ms4 = e0.elapsed_time(e1) / steps
    finally:
        del os.environ["UCN_PROBE_COPY_UNROLL"]
    return dict(bytes_read_plus_written=8 * n_floats, ms=ms, GBps=8 * n_floats / (ms * 1e-3) / 1e9, spec_peak_GBps=PEAK_HBM_GBS,
                GBps_four_loads_in_flight=8 * n_floats / (ms4 * 1e-3) / 1e9, guide_float4_copy_GBps=6290.0,
                shape=("k_copy4: a grid-stride float4 copy, 2048 workgroups x 256 threads, 1 GiB read + 1 GiB written, one load in flight per "
                       "thread (no unrolling, no non-temporal hints); /opt/skills/guides/MI355X_MICROARCH.md quotes 6.29 TB/s for its float4 "
                       "copy -- this probe is the repo's OWN plain-kernel yardstick and is the pessimistic one of the two; neither enters "
                       "`roofline.frac`, which is taken against the 8 TB/s spec peak.  GBps_four_loads_in_flight: the same copy, four "
                       "independent 16-byte loads per thread before the first store, 4096 workgroups (r06: 4.65 against 5.04 TB/s -- more "
                       "requests in flight do not lift a 1 GiB -> 1 GiB copy on this part)"))


class Ranks:
    """The three attributes render_image reads from an `accelerate.Accelerator`."""

    def __init__(self, world, rank):
        self.num_processes, self.process_index, self.is_main_process = world, rank, rank == 0


def build_model(device, heads=False, grid="B"):
    """Random-init weights of the reference's architecture (module init laws of models.py:438-483),
    same seed on every rank; hash tables re-drawn ~U(-1,1) so density varies (the
    reference's +-1e-4 table init makes the grid a no-op).  heads=True: BASELINE configs[4] -- the sky NeRF layer
    and the per-camera colour-correction head (210 training views) on top."""
    from ucnerf_amd.internal import configs, models
    torch.manual_seed(0)
    cfg = configs.Config(model_sky=True, brightness_correction=True, training_views=210) if heads else configs.Config()
    if grid == "R":                                  # the reference's own waymo.gin: class defaults (L = 10, C = 4, T = 2^21), 128 + 32 samples
        model = models.Model(config=cfg, num_levels=2, num_prop_samples=128, num_nerf_samples=32)
    else:
        kw = dict(grid_level_dim=2, grid_log2_hashmap_size=19)
        with models.bindings(NerfMLP=dict(grid_disired_resolution=524288, **kw), PropMLP=dict(**kw)):
            model = models.Model(config=cfg, num_levels=2, num_prop_samples=S_PROP, num_nerf_samples=S_NERF)
    for mlp in (model.nerf_mlp, model.prop_mlp_0):
        mlp.encoder.embeddings.data.uniform_(-1, 1)
    if heads:                                        # zero latent codes would make every camera's affine map the same
        model.brightness_corr.latent_code.data.normal_(0, 0.1)
        model.brightness_corr.sky_latent_code.data.normal_(0, 0.1)
        # a default-initialised sky NeRF is DEAD on these rays (alpha_linear's output is negative for every sample: relu(sigma) = 0,
        # the layer renders exactly 0 and receives exactly zero gradients): lift its density head so that the layer contributes
        # to the pixels the CPU oracle checks and to the gradients the training step computes
        model.skynerf.alpha_linear.bias.data.fill_(0.05)
    sd = {k: v.clone() for k, v in model.state_dict().items()}       # CPU copy for the CPU baseline
    return model.to(device).eval(), cfg, sd


def cpu_baseline(sd, rays_flat, rand_vec, gpu_rgb, n_sample=40960, per_call=8192, heads=False, eval_camidx=None, grid="B", thread_sweep=True):
    """The reference's path on the host cores: oracle/raymarch.py (== reference Python, bit-exact in
    the authoring container) + oracle/grid_oracle.c for the CUDA-only grid op, same rays / weights.
    This is the ONLY place bench.py touches oracle/ (as the timed baseline and the parity check).

    SURVEY.md 8(d): chunks of 8192 rays, one warm-up, MEDIAN of >= 5 chunks; the thread count is stated, and the
    all-cores and the 1-thread figures are reported beside it (`threads`).  `value` is the fastest of the three
    configurations (the fairest baseline): torch-CPU eager ops on these tensors stop scaling beyond a few dozen
    threads (all 256 host CPUs measured 82 rays/s in r03), so the all-cores run is bounded to one short chunk."""
    from oracle import raymarch as rm
    spec = rm.make_spec(grid, model_sky=True, brightness_correction=True, training_views=210) if heads else rm.make_spec(grid)
    ncpu = os.cpu_count() or 1
    idx = torch.linspace(0, rays_flat["origins"].shape[0] - 1, n_sample).long()
    sub = {k: v[idx].cpu() for k, v in rays_flat.items()}
    noise = [rm.LevelNoise(rand_vec=rand_vec[idx, 3 * l:3 * l + 3].cpu()) for l in range(2)]

    def run(sl):
        with torch.no_grad():
            rend, _ = rm.model_forward(spec, sd, {k: v[sl] for k, v in sub.items()},
                                       [rm.LevelNoise(rand_vec=n.rand_vec[sl]) for n in noise], eval_camidx=eval_camidx)
        return rend[-1]["rgb"].reshape(-1, 3)

    main_threads = min(ncpu, 32)
    threads_before = torch.get_num_threads()             # restored below: the GPU legs that follow keep the process default
    torch.set_num_threads(main_threads)
    run(slice(0, 256))                                   # warm-up
    rgb, secs = [], []
    for r0 in range(0, n_sample, per_call):              # the reference renders in chunks too (render_chunk_size)
        t0 = time.perf_counter()
        rgb.append(run(slice(r0, r0 + per_call)))
        secs.append(time.perf_counter() - t0)
    rgb = torch.cat(rgb)
    rate = per_call / float(np.median(secs))
    threads = {str(main_threads): dict(rays_per_s=rate, chunks=len(secs), rays_per_chunk=per_call, seconds=float(sum(secs)))}
    if thread_sweep:
        for nt, n in ((ncpu, 2048), (1, 1024)):
            if str(nt) in threads:
                continue
            torch.set_num_threads(nt)
            run(slice(0, 64))
            t0 = time.perf_counter()
            run(slice(0, n))
            dt = time.perf_counter() - t0
            threads[str(nt)] = dict(rays_per_s=n / dt, chunks=1, rays_per_chunk=n, seconds=dt)
        torch.set_num_threads(main_threads)
    torch.set_num_threads(threads_before)
    best = max(threads, key=lambda k: threads[k]["rays_per_s"])
    linf = float((rgb - gpu_rgb[idx].cpu()).abs().max())
    mse = float(((rgb - gpu_rgb[idx].cpu()) ** 2).mean())
    return dict(value=threads[best]["rays_per_s"], unit="rays/s", cores=int(best), host_cpu_count=ncpu, kind="port",
                sample=f"{n_sample} rays strided over the same frame, same weights and rand_vec: {len(secs)} chunks of {per_call} on "
                       f"{main_threads} threads (median chunk; {sum(secs):.1f} s)" +
                       ("; all-cores and 1-thread figures on one shorter chunk each (`threads`)" if thread_sweep else ""),
                threads=threads, rgb_linf_gpu_vs_cpu=linf, psnr_gpu_vs_cpu=float(-10 * np.log10(max(mse, 1e-20))))


def render_config(device, cameras, heads, autocast, n_cpu=8192, grid="B"):
    """One more BASELINE config through exactly the headline's code path (render_image on a full frame resident in HBM,
    one warm-up frame + one timed frame), with its own CPU-oracle check on n_cpu rays of that frame -- so that the
    driver's bench line carries rays/s AND the RGB L-inf for configs[3] / configs[4], not only builder-run files."""
    from ucnerf_amd.internal import models
    model, cfg, sd = build_model(device, heads=heads, grid=grid)
    cfg.render_ray_tile = 8
    cfg.render_gather_weights = False
    batch = frame_rays(device, cameras, virtual=heads)
    n_rays = cameras * H_IMG * W_IMG
    rand_vec = torch.randn(n_rays, 6, generator=torch.Generator().manual_seed(1))
    batch["rand_vec"] = rand_vec.reshape(cameras * H_IMG, W_IMG, 6).to(device)
    eval_camidx = torch.tensor([7]) if heads else 0

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            return models.render_image(model, Ranks(1, 0), batch, False, 1.0, cfg, verbose=False, eval_camidx=eval_camidx)
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    flat = {k: v.reshape(n_rays, -1) for k, v in batch.items() if k != "rand_vec"}
    cpu = cpu_baseline(sd, flat, rand_vec, out["rgb"].reshape(n_rays, 3), heads=heads,
                       eval_camidx=eval_camidx if heads else None, n_sample=n_cpu, grid=grid, thread_sweep=False)
    res = dict(rays=n_rays, cameras=cameras, ms_per_frame=dt * 1e3, rays_per_s=n_rays / dt, steps=1, warmup=1,
               dtype=DTYPE_MIXED if autocast else DTYPE_F32_CLASS,
               rgb_linf_gpu_vs_cpu=cpu["rgb_linf_gpu_vs_cpu"], psnr_gpu_vs_cpu=cpu["psnr_gpu_vs_cpu"],
               cpu_rays_per_s=cpu["value"], cpu_cores=cpu["cores"], cpu_sample=cpu["sample"])
    del model, batch, out, flat
    torch.cuda.empty_cache()
    return res


DTYPE_F32_CLASS = ("f32-class (dense layers: f16 MFMA on hi/lo-split f32 operands, 2^-22 per product, f32 accumulate; "
                   "grid, resampling, compositing f32)")
DTYPE_MIXED = "bf16 dense layers + f16 tables, f32 interpolation / resampling / compositing (autocast)"


def live_traffic(autocast):
    """roofline.traffic measured in THIS run when rocprofv3 is on PATH: HBM-side bytes per NeRF-level featurisation
    launch from two PMC passes (FETCH_SIZE, WRITE_SIZE -- separate passes as MI355X_MICROARCH.md prescribes; --kernel-trace
    only) over a child process of this same script that renders a 160-row slice of the frame (30 launches of 10,240
    rays: per-launch counters do not need the whole frame).  gfx950 corrections of the guide: both counters are in KiB
    units; FETCH_SIZE = TCC_EA0_RDREQ x 64 B (a 128-byte request counts 64: true figure between 1x and 2x); Infinity-Cache
    hits are counted.  Returns None when rocprofv3 is missing or a pass fails -- never raises."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if os.environ.get("UCN_BENCH_CHILD") or shutil.which("rocprofv3") is None:
        return None
    got = {}
    tmp = tempfile.mkdtemp(prefix="ucn_pmc_", dir="/tmp")
    env = dict(os.environ, UCN_BENCH_CHILD="1", TMPDIR="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child"] + (["--autocast"] if autocast else [])
            subprocess.run(cmd, cwd="/tmp", env=env, timeout=240, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            tot, n = 0.0, 0
            for f in glob.glob(os.path.join(out, "**", "p_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    name = r["Kernel_Name"]
                    half = "DF16_" in name or "_Float16" in name
                    nerf_level = "Lb0E" in name or ", false>" in name     # the FEW_LEVELS name tag (mangled / demangled): false = NeRF field
                    if (r["Counter_Name"] == ctr and "k_march_features" in name and "bwd" not in name and half == bool(autocast)
                            and nerf_level):
                        tot += float(r["Counter_Value"]) * 1024.0
                        n += 1
            if n == 0:
                return None
            got[ctr] = (tot / n, n)
    except Exception:                       # noqa: BLE001  (a profiler problem must not cost the bench line)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return dict(fetch_bytes_per_launch=got["FETCH_SIZE"][0], write_bytes_per_launch=got["WRITE_SIZE"][0],
                launches_sampled=got["FETCH_SIZE"][1])


def sky_layer_ms(batch_flat, device, n_rays=65536, steps=3):
    """SURVEY 8 row a12 / cfg5's extra term: the sky NeRF (120 samples x 562,688 MAC) on n_rays of the frame,
    random-init weights of the reference architecture (models.py:85-92).  Outside the timed region."""
    from ucnerf_amd.internal.sky import NeRF
    torch.manual_seed(1)
    net = NeRF(D=8, d_in_view=3, W=256, multires_view=4, output_ch=4, skips=[4]).to(device)
    tot = batch_flat["origins"].shape[0]
    idx = torch.arange(0, tot, tot // n_rays, device=device)[:n_rays]
    o, d, cam = (batch_flat[k][idx].contiguous() for k in ("origins", "directions", "cam_dirs"))
    far = batch_flat["far"][idx].reshape(-1).contiguous()
    for _ in range(2):
        net.render(o, d, cam, far)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        net.render(o, d, cam, far)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    for _ in range(2):
        net.render(o, d, cam, far, mixed=True)
    e0.record()
    for _ in range(steps):
        net.render(o, d, cam, far, mixed=True)
    e1.record()
    torch.cuda.synchronize()
    ms_bf = e0.elapsed_time(e1) / steps
    flops = 2.0 * 562688 * 120 * n_rays
    # executed bf16 MFMAs of the mixed kernel: 984 of 32 x 32 x 16 per 32 samples (the composed views layer saves 64 K MAC)
    flops_bf = 984 * 32768 / 32 * 120 * n_rays
    return dict(ms=ms, rays=n_rays, rays_per_s=n_rays / ms * 1e3, algorithmic_tflops=flops / ms / 1e9,
                frac_of_f16_mfma_peak=flops / ms / 1e9 / PEAK_F16_MFMA_TF, ms_per_frame=ms * tot / n_rays,
                kernel="k_sky_mlp (split-f16 MFMA, composed views layer) + k_sky_composite",
                mixed=dict(ms=ms_bf, rays_per_s=n_rays / ms_bf * 1e3, executed_mfma_tflops=flops_bf / ms_bf / 1e9,
                           frac_of_bf16_mfma_peak=flops_bf / ms_bf / 1e9 / PEAK_F16_MFMA_TF,
                           kernel="k_sky_mlp_bf (bf16 MFMA, two workgroups per CU; under autocast) + k_sky_composite"))


def train_step_ms(model, batch_flat, device, n_rays=8192, steps=12, heads=False, autocast=True):
    """BASELINE configs[2]: one training step on an 8192-ray batch -- Model.forward(rand=True) under bf16
    autocast, the losses of train.py:173-216 with waymo defaults, backward, nan_to_num on grads
    (train_utils.py:335-344), Adam(lr 0.01, betas (0.9, 0.99), eps 1e-8; waymo.gin:6).  Median of `steps`.
    heads=True: the step the reference's shipped launch trains (scripts/train_waymo.sh:11-12: model_sky +
    brightness_correction): sky NeRF forward / backward, per-ray colour-correction affines, sky-segment and
    identity losses (train.py:181-185, sky_weight = idt_weight = 0.002) on top."""
    import types
    from ucnerf_amd.internal import train_utils as tu
    cfg = types.SimpleNamespace(data_loss_type='charb', charb_padding=0.001, data_loss_mult=1.0, data_coarse_loss_mult=0.,
                                anti_interlevel_loss_mult=0.01, pulse_width=[0.03, 0.003], distortion_loss_mult=0.005,
                                hash_decay_mults=0.1, disable_multiscale_loss=False, sky_weight=0.002, idt_weight=0.002)
    seed = 2 + (torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0)
    g = torch.Generator(device=device).manual_seed(seed)        # every rank its own rays (datasets.py:278)
    # = create_optimizer (train_utils.py:347); the sharded form when dist.wrap_ddp(grad_exchange="reduce_scatter") marked the tables
    opt_cls = tu.ShardedFusedAdam if any(getattr(p, "_ucn_sharded", False) for p in model.parameters()) else tu.FusedAdam
    opt = opt_cls(model.parameters(), lr=0.01, betas=(0.9, 0.99), eps=1e-8)
    model.train()
    times = []
    n_total = batch_flat['origins'].shape[0]
    for it in range(steps + 2):
        idx = torch.randint(0, n_total, (n_rays,), device=device, generator=g)
        if os.environ.get("UCN_BENCH_SORT_RAYS") == "1" and n_total == H_IMG * W_IMG:
            # experiment (profiles/r06/sort_rays_ab.txt): the same random rays, visited in 8 x 8-pixel-tile order (the per-ray math and the
            # mean losses do not depend on the order inside the batch; the synthetic target is drawn after, so nothing is un-permuted)
            yy, xx = idx // W_IMG, idx % W_IMG
            idx = idx[torch.argsort(((yy // 8) * (W_IMG // 8) + xx // 8) * 64 + (yy % 8) * 8 + xx % 8)]
        batch = {k: v[idx][:, None, None, :] for k, v in batch_flat.items()}
        batch['rgb'] = torch.rand(n_rays, 1, 1, 3, device=device, generator=g)
        if heads:
            batch['cam_idx'] = torch.randint(0, 210, (n_rays, 1, 1, 1), device=device, generator=g)
            batch['sky_segs'] = (torch.rand(n_rays, 1, 1, device=device, generator=g) > 0.7).float()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=autocast):     # train.py:165-171: only the model call is autocast
            rend, hist = model(True, batch, 0.5, False, zero_glo=False)
        loss = (tu.compute_data_loss(batch, rend, cfg)[0] + tu.anti_interlevel_loss(hist, cfg)
                + tu.distortion_loss(hist, cfg) + tu.hash_decay_loss(hist, cfg))
        if heads:
            loss = loss + cfg.sky_weight * tu.sky_loss(batch, rend) + cfg.idt_weight * tu.transformIdentityLoss(rend)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        tu.clip_gradients(model, None, cfg)                         # train_utils.py:335-344: nan_to_num on every gradient
        opt.step()
        torch.cuda.synchronize()
        if it >= 2:
            times.append((time.perf_counter() - t0) * 1e3)
    model.eval()
    if not autocast:
        from ucnerf_amd.internal import dense_f32
        eng = dense_f32.engine()
        return dict(ms=float(np.median(times)), rays=n_rays, rays_per_s=n_rays / (np.median(times) * 1e-3), steps=steps,
                    dtype=("f32-class (dense layers: split-f16 products on v_mfma_f32_32x32x16_f16, every operand to 2^-22 at a per-tensor "
                           "power-of-two scale, fp32 accumulate; fp32 tables, exact-fp32-add table gradients, fp32 everything else)"
                           if eng == "split" and os.environ.get("UCN_F32_LIBRARY") != "1" else "f32"),
                    precision="no autocast (the reference's shipped launch: scripts/train_waymo.sh:3 has no --mixed_precision): fp32 tables, "
                              "exact-fp32-add table gradients, every dense layer forward / dgrad / wgrad on " +
                              ("csrc/gemm_h3.hip (r06: the split-f16 engine; rows < 4096 on csrc/gemm_f32.hip)" if eng == "split" else
                               "csrc/gemm_f32.hip (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate)") +
                              ("; UCN_F32_LIBRARY=1: the r03 graph (uncomposed colour MLP) on torch's library GEMMs (A/B)" if os.environ.get("UCN_F32_LIBRARY") == "1" else
                               "; r04: the activation-free bottleneck is composed into the colour layers (K = 64 instead of 256 for two of the three 256-wide GEMMs)"),
                    dense_engine=eng, heads=bool(heads))
    return dict(ms=float(np.median(times)), rays=n_rays, rays_per_s=n_rays / (np.median(times) * 1e-3), steps=steps,
                heads=("sky NeRF (120 samples x 8 x 256 MLP) + per-ray colour-correction affines + sky-segment and identity "
                       "losses (scripts/train_waymo.sh:11-12)" if heads else "none (BASELINE configs[2])"),
                autocast="bf16 dense layers; forward gather from a half copy of the tables (the reference's autocast policy, "
                         "gridencoder/grid.py:41-44) with fp32 interpolation; fp32 table gradients, compositing and losses",
                graph="HIP resample, fused featurisation fwd / bwd (LDS row blocks, no global atomics), NeRF-field dense forward and "
                      "dgrad as bf16 MFMA kernels (ucn_train_fwd / ucn_train_bwd, two workgroups per CU), proposal field as VALU "
                      "kernels, compositing fwd / bwd, distortion + interlevel + hash-decay losses, Adam (tables and small "
                      "parameters); weight + bias gradients by wgrad.hip (ds_read_b64_tr_b16 operand transposes + bf16 MFMA, split-K)")


def fitted_field_frames(device, cfg_template, steps=300, thr=4e-8):
    """north_star's early-termination sample compaction on a field that is NOT random: the config-B model from the reference's own
    initialisation (tables +-1e-4, grid.py:151-153), fitted `steps` steps to tools/fit_scene.py's analytic opaque-surface scene with
    this repo's training graph, then the headline frame timed with Model.compact_min_weight = 0 and = thr; the alive fraction is the
    share of NeRF-level samples whose compositing weight reaches thr.  (tests/test_full_size.py::test_fitted_field_march_and_
    compaction_vs_oracle holds the same field to the CPU oracle.)"""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import fit_scene
    from ucnerf_amd.internal import models
    torch.manual_seed(20)
    model, cfg, _ = build_model(device)
    for mlp in (model.nerf_mlp, model.prop_mlp_0):
        mlp.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
    t0 = time.perf_counter()
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()) as log:
        fit_scene.fit(model, device, steps)
    fit_s = time.perf_counter() - t0
    last = [l for l in log.getvalue().splitlines() if "psnr" in l][-1].strip()
    cfg.render_ray_tile = cfg_template.render_ray_tile
    cfg.render_gather_weights = False
    batch = frame_rays(device)
    n_rays = H_IMG * W_IMG
    batch["rand_vec"] = torch.randn(n_rays, 6, generator=torch.Generator().manual_seed(1)).reshape(H_IMG, W_IMG, 6).to(device)
    acc = Ranks(1, 0)

    def frame():
        models.render_image(model, acc, batch, False, 1.0, cfg, verbose=False, eval_camidx=0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        out = models.render_image(model, acc, batch, False, 1.0, cfg, verbose=False, eval_camidx=0)
        torch.cuda.synchronize()
        return time.perf_counter() - t1, out
    d0, out0 = frame()
    model.compact_min_weight = thr
    model._alive_stats = []
    models.render_image(model, acc, batch, False, 1.0, cfg, verbose=False, eval_camidx=0)          # the statistics frame (host reads: not timed)
    alive = sum(a for a, _ in model._alive_stats) / max(1, sum(b for _, b in model._alive_stats))
    model._alive_stats = None
    d1, out1 = frame()
    model.compact_min_weight = 0.0
    return dict(value_fitted=n_rays / d0, value_fitted_compacted=n_rays / d1, ms_fitted=d0 * 1e3, ms_fitted_compacted=d1 * 1e3,
                compact_min_weight=thr, alive_fraction=alive, rgb_linf_compacted_vs_plain=float((out0["rgb"] - out1["rgb"]).abs().max()),
                fit=dict(steps=steps, seconds=fit_s, last_logged=last, scene="tools/fit_scene.py: ground plane + six spheres, 24 poses, bf16 autocast steps of 8192 rays"),
                note=("after a fit of this length the proposal resampling has moved the 128 NeRF-level samples onto the surfaces: (nearly) every sample "
                      "still carries weight >= the threshold, so the compacted route -- density head, compositing weights, ballot / prefix-sum alive list, "
                      "colour layers of the alive samples -- is exercised end to end but removes (next to) nothing and pays its fixed cost; it stays off "
                      "by default (break-even alive fraction 0.73, profiles/r04/compaction_fit8000*.txt)"))


def fp32_step_both_engines(model, flat, device, **kw):
    """the non-autocast step on the default ("split": fp32-class) engine, with the exact-fp32-product engine's time beside it"""
    from ucnerf_amd.internal import dense_f32
    r = train_step_ms(model, flat, device, autocast=False, **kw)
    prev = dense_f32.set_engine("exact")
    try:
        kw2 = dict(kw, steps=min(kw.get("steps", 4), 3))
        r["exact_fp32_products_ms"] = train_step_ms(model, flat, device, autocast=False, **kw2)["ms"]
    finally:
        dense_f32.set_engine(prev)
    return r


def self_launch(n):
    """Re-run this command line as N ranks: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port <free> bench.py <the same arguments>`.  Returns the launcher's exit code.  RCCL wants one GPU per rank:
    fewer visible GPUs than ranks is an error here, not a silent N = 1 run (UCN_DIST_BACKEND=gloo is the functional
    check that lets ranks share a device)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("UCN_DIST_BACKEND", "nccl") == "nccl":
        print(f"bench.py: --gpus {n} needs {n} visible GPUs for RCCL (one rank per GPU), found {have}", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), UCN_BENCH_SELF_LAUNCHED="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); default: WORLD_SIZE when a launcher set it, else 1")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the post-timed configs[3] / configs[4] frames and the PMC traffic passes")
    ap.add_argument("--pmc-child", action="store_true", help="(internal) the short run live_traffic() profiles")
    ap.add_argument("--ray-major", action="store_true", help="lanes = consecutive samples of a ray (default: neighbouring rays)")
    ap.add_argument("--mlp-mode", type=int, default=None, help="0 fp32-input MFMA, 1 split-f16 MFMA (default: the package default)")
    ap.add_argument("--overlap", type=int, nargs="?", const=1, default=0,
                    help="featurisation of pass i+1 beside the MLP of pass i on 2 streams: 1 = co-resident launch shapes, 2 = plain shapes")
    ap.add_argument("--levels-per-block", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--compact", type=float, default=0.0,
                    help="Model.compact_min_weight: early-termination sample compaction (colour layers only for samples whose "
                         "compositing weight reaches this value; 4e-8 bounds the pixel error by 5e-6).  Default off: on the random-init "
                         "field AND on a fitted one every sample carries weight (tools/fit_scene.py, DESIGN.md)")
    ap.add_argument("--ray-tile", type=int, default=8,
                    help="config.render_ray_tile: render_image marches the frame in tile-major order (T x T pixel blocks per "
                         "wave; 1 = the frame's row-major order)")
    ap.add_argument("--autocast", action="store_true",
                    help="render under torch.autocast(bf16) like the reference's render_image under `accelerate --mixed_precision "
                         "bf16` (models.py:957): half tables in the gather, dense layers as bf16 MFMAs, fp32 compositing -- the "
                         "'mixed bf16/fp32' of BASELINE configs[4]; NOT the fp32 headline")
    ap.add_argument("--float-features", action="store_true", help="with --autocast: float features between gather and MLP (Model.autocast_bf16_features = False)")
    ap.add_argument("--sky-skip", type=float, default=0.0,
                    help="Model.sky_min_background (with --cfg5): sky layer only for rays whose background weight reaches this "
                         "value.  Default off: the reference returns sky_rgbs for every ray, and on the random-init field "
                         "every ray has background weight ~0.11")
    ap.add_argument("--fit-steps", type=int, default=0,
                    help="fit the model to tools/fit_scene.py's analytic scene for this many steps before the timed region "
                         "(a trained-like field instead of BASELINE's random-init one; reported in config.field)")
    ap.add_argument("--cameras", type=int, default=1,
                    help="cameras per frame: 5 = BASELINE configs[3] (full 5-camera Waymo frame, 12.29 M rays, row tiles over the ranks)")
    ap.add_argument("--cfg5", action="store_true",
                    help="BASELINE configs[4]: sky layer + colour-correction head on, rays of VIRTUAL (perturbed) poses; "
                         "implies --cameras 5 unless given")
    args = ap.parse_args()
    if args.cfg5 and args.cameras == 1:
        args.cameras = 5
    if args.gpus is None:                  # `torchrun --nproc-per-node N bench.py` with no --gpus: the launcher's world is the answer
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.pmc_child:
        args.steps, args.warmup, args.no_cpu_baseline, args.no_train, args.no_extras = 1, 0, True, True, True
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.pmc_child:
        # `python bench.py --gpus N` with no launcher around it: become the launcher.  One rank per GPU under
        # torch.distributed.run (the same form the driver uses for N > 1); rank 0's JSON line passes through on stdout.
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not args.pmc_child and world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE); "
                         "the line's n_gpus must be the number of ranks that really ran")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU path exists)"
    # UCN_DIST_BACKEND=gloo lets two ranks share one GPU: a functional check of the multi-rank flow on a 1-GPU box
    backend = os.environ.get("UCN_DIST_BACKEND", "nccl")             # nccl == RCCL on ROCm
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        # communicator set-up (RCCL builds its rings lazily, seconds on a cold node) stays out of the timed region
        # even with --warmup 0: one collective of each kind the frame uses
        w = torch.zeros(8, device=device)
        dist.all_reduce(w)
        dist.all_gather_into_tensor(torch.empty(8 * world, device=device), w)
        torch.cuda.synchronize()
    from ucnerf_amd.internal import models, dist as udist
    if args.mlp_mode is not None:
        models.MLP.mlp_mode = args.mlp_mode
    model, cfg, sd = build_model(device, heads=args.cfg5)
    if args.levels_per_block:
        model.levels_per_block = args.levels_per_block
    if args.chunk:
        model.max_chunk_rays = args.chunk
    if args.ray_major:
        model.rays_fastest = False
    if args.overlap:
        model.overlap_streams = args.overlap
    if args.fit_steps:
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import fit_scene
        for mlp in (model.nerf_mlp, model.prop_mlp_0):
            mlp.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
        fit_scene.fit(model, device, args.fit_steps)
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    model.compact_min_weight = args.compact
    model.sky_min_background = args.sky_skip
    model.autocast_bf16_features = not args.float_features
    cfg.render_ray_tile = args.ray_tile
    cfg.render_gather_weights = False      # the exchanged tile = pixels + depth / acc / distance statistics (36 B per ray); the
    #                                        [H, W, 128] per-sample weights (27x that) are computed but not part of the returned
    #                                        frame at ANY world size (no caller of the reference reads them)
    batch = frame_rays(device, args.cameras, virtual=args.cfg5)
    rows = args.cameras * H_IMG
    if args.pmc_child:                     # 160 rows = 30 passes of 10,240 rays: enough launches for per-launch counters
        rows = 160
        batch = {k: v[:rows].contiguous() for k, v in batch.items()}
    n_rays = rows * W_IMG
    g = torch.Generator().manual_seed(1)
    rand_vec = torch.randn(n_rays, 6, generator=g)                  # pinned cone-basis draws (render.py:140)
    batch["rand_vec"] = rand_vec.reshape(rows, W_IMG, 6).to(device)
    acc = Ranks(world, rank)
    # configs[4]: one colour-correction latent per frame, as render.py:146-147 / eval.py:140-141 pass it
    eval_camidx = torch.tensor([7]) if args.cfg5 else 0

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.autocast):
            return models.render_image(model, acc, batch, False, 1.0, cfg, verbose=False, eval_camidx=eval_camidx)

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    fence()
    model._prof = []
    model._prof_every = 1 if args.pmc_child else PROF_EVERY       # (the 30-pass counter child: every pass, so that the launch-count check below holds)
    udist.EXCHANGE_EVENTS = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    prof, model._prof = model._prof, None
    exch, udist.EXCHANGE_EVENTS = udist.EXCHANGE_EVENTS, None
    t = torch.tensor([dt], device=device)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())

    # per-kernel time from the HIP events recorded on the launch stream during the timed steps
    feat_ms = {0: 0.0, 1: 0.0}; mlp_ms = {0: 0.0, 1: 0.0}; rays_seen = {0: 0, 1: 0}; launches = {0: 0, 1: 0}
    for lvl, n, e0, e1, m0, e2 in prof:
        feat_ms[lvl] += e0.elapsed_time(e1); mlp_ms[lvl] += m0.elapsed_time(e2); rays_seen[lvl] += n; launches[lvl] += 1
    cpu_leg = None
    if rank == 0:
        lo, hi = udist.shard_bounds(n_rays, world, rank)
        rays_rank = (hi - lo) * args.steps
        assert 0 < rays_seen[1] <= rays_rank and launches[1] >= 8
        sampling = (f"HIP events around every {PROF_EVERY}th pass of the timed frames ({launches[1]} NeRF-level launches timed): an event record "
                    "costs the queue ~10 us of idle time at a kernel boundary -- around every pass (r01-r05) that was 9 ms of a 470 ms "
                    "frame, 1.9 % off `value` (tools/frame_gaps.py, profiles/r06/frame_gaps.txt)")
        gather = dict(bound="hbm", kernel="k_march_features<2, 256, float, false> (FEW_LEVELS = false: the NeRF-level launches)" + (", half tables" if args.autocast else ""),
                      achieved=rays_seen[1] * (GATHER_BYTES_NERF // 2 if args.autocast else GATHER_BYTES_NERF) / (feat_ms[1] * 1e-3) / 1e9,
                      peak=PEAK_HBM_GBS, unit="GB/s",
                      avg_launch_ms=feat_ms[1] / launches[1], rays_per_launch=rays_seen[1] / launches[1], launches_timed=launches[1],
                      timing=sampling, traffic=None)
        if args.autocast:
            gather["bytes_note"] = "half tables: 2-byte entries, half the algorithmic gather bytes of the fp32 path"
        gather["frac"] = gather["achieved"] / gather["peak"]
        # what actually binds the fine hashed levels: L1 / texture-address requests, not bytes
        gather["l2_request_bound"] = dict(
            lines_per_sample_level=36, fine_levels=10, l1_lines_per_clk_per_cu=1,
            ceiling_ms_per_fine_level_per_8_4M_samples=0.49,
            note="every corner pair of a hashed level is its own 64 B line: ~36 line requests per (sample, level); at one line per "
                 "clock and CU that is 0.49 ms per fine level per 8.4 M samples -- the kernel's real bound; the `achieved` GB/s is "
                 "served mostly from L2 (TCC hit ~91 %), so it may exceed what an HBM copy reaches (hbm_probe)")
        # HBM-side bytes per launch: measured live by two rocprofv3 PMC passes over a child run of this script when
        # rocprofv3 is on PATH (live_traffic); else from the committed PMC passes of this same command
        live = None if (args.no_extras or world > 1) else live_traffic(args.autocast)
        if live is not None:
            gather["traffic"] = live["fetch_bytes_per_launch"] + live["write_bytes_per_launch"]
            gather["traffic_note"] = (f"LIVE: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over a 160-row slice rendered by a child of "
                                      f"this run, {live['launches_sampled']} NeRF-level launches; KiB units, FETCH_SIZE = TCC_EA0_RDREQ x 64 B on "
                                      "gfx950 (true figure between 1x and 2x), Infinity-Cache hits included; below the algorithmic gather bytes "
                                      "because the 4 MiB level slices are re-read from L2")
            gather["traffic_detail"] = live
        if live is None:
            try:
                tj = json.load(open(os.path.join(REPO, "profiles", "r02c" if args.autocast else "r05", "traffic_autocast.json" if args.autocast else "traffic.json")))
                if abs(gather["rays_per_launch"] - tj["rays_per_launch"]) < 0.5 and (not args.autocast or model.autocast_bf16_features):
                    gather["traffic"] = tj["fetch_bytes_per_launch"] + tj["write_bytes_per_launch"]
                    gather["traffic_note"] = ("NOT live (rocprofv3 unavailable or a pass failed): bytes per launch from the committed rocprofv3 PMC "
                                              "passes of this command (profiles/r05/traffic.json, profiles/r02c/traffic_autocast.json): FETCH_SIZE + WRITE_SIZE")
            except (OSError, KeyError, ValueError):
                pass
        split = model.nerf_mlp.mlp_mode == 1
        # achieved = ALGORITHMIC flops of the reference formulation / kernel time; peak = the dense peak of
        # the MFMA instruction the kernel issues.  mode 0: exact fp32 products on v_mfma_f32_32x32x2_f32 (157.3 TF).
        # mode 1: v_mfma_f32_32x32x16_f16 (2500 TF); every fp32 product costs three f16 MFMA products
        # (hi*hi + hi*lo + lo*hi) while the composed layers need 0.52x the reference's MACs -- executed_mfma_tflops
        # is what the matrix cores actually ran.
        mlp = dict(bound="mfma", kernel=("k_field_mlp_h8<8,4,...> (two 4-wave workgroups per CU)" if split else "k_field_mlp<8,8>") + " (NeRF level)",
                   achieved=rays_seen[1] * FLOP_NERF_RAY / (mlp_ms[1] * 1e-3) / 1e12,
                   peak=PEAK_F16_MFMA_TF if (split or args.autocast) else PEAK_F32_MFMA_TF, unit="TFLOP/s",
                   avg_launch_ms=mlp_ms[1] / launches[1], rays_per_launch=rays_seen[1] / launches[1], traffic=None)
        mlp["frac"] = mlp["achieved"] / mlp["peak"]
        mlp["issue_model_note"] = ("per wave (32 samples): 696 MFMAs, 1463 VALU, 622 LDS reads, 118 LDS-DMA pieces, 29 barriers "
                                   "(profiles/r02c/pmc_table.txt); beside MFMAs a VALU instruction costs ~2-2.6 cycles of the SIMD "
                                   "(tools/mfma_valu_bench.hip), which puts the issue ceiling at ~0.87 of the MFMA peak before any "
                                   "wait; the executed fraction equals what the CDNA4 guide's hand-scheduled attention example "
                                   "reaches (0.50-0.56 of the same peak)")
        mlp["peak_note"] = ("dense f16 MFMA peak; fp32-class products = 3 f16 MFMAs each, composed layers = 0.52x MACs" if split
                            else "fp32-input MFMA peak (= fp32 vector rate on CDNA4)")
        if args.autocast:
            # mixed-precision render: ucn_train_fwd without stores, the direction tile in the stream: 276 bf16 MFMAs of
            # 32 x 32 x 16 per 32 samples, one product per MAC
            mlp["kernel"] = "k_train_fwd<1, AUX> (bf16 MFMA inference: composed colour layers, direction tile in the weight stream) (NeRF level)"
            mlp["peak_note"] = "dense bf16 MFMA peak"
            mlp["issue_model_note"] = "276 MFMAs per wave (32 samples), two workgroups per CU"
            mlp["executed_mfma_tflops"] = rays_seen[1] * S_NERF * (276 * 32768 / 32) / (mlp_ms[1] * 1e-3) / 1e12
            mlp["executed_frac"] = mlp["executed_mfma_tflops"] / PEAK_F16_MFMA_TF
        elif split:
            mlp["executed_mfma_tflops"] = 3 * rays_seen[1] * S_NERF * 2 * MAC_NERF_SPLIT / (mlp_ms[1] * 1e-3) / 1e12
            mlp["executed_frac"] = mlp["executed_mfma_tflops"] / PEAK_F16_MFMA_TF
        dominant, other = (gather, mlp) if feat_ms[1] >= mlp_ms[1] else (mlp, gather)
        total_ms = dt * 1e3 / args.steps
        if args.cfg5:
            workload = (f"BASELINE configs[4]: {args.cameras}-camera 1280x1920 frame ({n_rays:,} rays) of VIRTUAL (perturbed) poses, sky NeRF "
                        "layer + per-camera colour-correction head (210 views) on, proposal 64 + NeRF 128 samples, NeRF grid L=16 C=2 "
                        "T=2^19; " + ("MIXED bf16 / fp32 as the config names it: half tables, bf16 MFMA dense layers (fields and sky), "
                                      "fp32 resampling / compositing / colour head" if args.autocast else
                                      "dense layers on f16 MFMA with hi/lo operands and fp32 accumulation (fp32-class: >= the bf16 "
                                      "the config names), grid + compositing fp32"))
        elif args.cameras > 1:
            workload = (f"BASELINE configs[3]: full {args.cameras}-camera 1280x1920 frame ({n_rays:,} rays), row tiles over the ranks, "
                        "proposal 64 + NeRF 128 samples, NeRF grid L=16 C=2 T=2^19, proposal grid L=6 C=2 T=2^19, fp32 forward render")
        else:
            workload = ("BASELINE configs[1]: one 1280x1920 frame (2,457,600 rays), proposal 64 + NeRF 128 samples, "
                        "NeRF grid L=16 C=2 T=2^19, proposal grid L=6 C=2 T=2^19, fp32 forward render, compute_extras=True")
        res = {
            "metric": "rays/sec (fwd render), 1280x1920 @ 64+128 samples", "value": n_rays * args.steps / dt,
            "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": total_ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": DTYPE_MIXED if args.autocast else (DTYPE_F32_CLASS if split else "f32"), "data": "synthetic",
            "config": {"workload": workload, "rays_per_step": n_rays, "cameras": args.cameras,
                       "field": (f"fitted for {args.fit_steps} steps to the analytic scene of tools/fit_scene.py" if args.fit_steps
                                 else "random-init weights, tables U(-1,1) (BASELINE configs)"),
                       "compact_min_weight": args.compact, "ray_tile": args.ray_tile,
                       "render_gather_weights": bool(cfg.render_gather_weights),
                       "compact_evidence": ("sample compaction stays OFF by default: on a field fitted for 8000 steps to an opaque-surface analytic scene "
                                            "(tools/fit_scene.py: acc 0.995-0.999 on surface-hit rays) 88-92 % of the NeRF-level samples carry weight >= 4e-8 "
                                            "and 85-90 % >= 1e-6 (the proposal resampling has already moved the samples to the surface), while the compacted "
                                            "route costs 407 ms + 137 ms x alive fraction against 508 ms plain: it wins below an alive fraction of 0.73 "
                                            "(r03: 0.51, before k_compact_alive's one-atomic-per-tile form) -- profiles/r04/compaction_fit8000*.txt"),
                       **({"sky_min_background": args.sky_skip,
                           "sky_rays_kept": getattr(model, "_sky_kept", None)} if args.cfg5 else {}),
                       "parallelism": f"ray-tile shard x{world}, 1 packed all-gather per frame (36 B per ray: rgb, depth, acc, 4 distance statistics)",
                       "levels_per_block": model.levels_per_block, "chunk_rays": model.max_chunk_rays,
                       "mlp_mode": ("bf16 MFMA, composed colour layers (the training forward kernel without its stores)" if args.autocast else
                                    {0: "fp32-input MFMA", 1: "split-f16 MFMA (hi/lo operands, fp32 accumulate)"}[model.nerf_mlp.mlp_mode]),
                       **({"autocast": "bf16"} if args.autocast else {})},
            "roofline": dominant, "roofline_secondary": other,
            # whole-path fraction: value / (N x the rate at which ONE GPU could stream the path's algorithmic gather bytes at the HBM
            # spec peak); the MFMA-bound rate of the arithmetic actually issued is higher (the binding roof is HBM), both stated
            "end_to_end": dict(
                frac=n_rays * args.steps / dt / (world * PEAK_HBM_GBS * 1e9 / ((GATHER_BYTES_NERF + GATHER_BYTES_PROP) // (2 if args.autocast else 1) + 84)),
                hbm_bound_rays_per_s_per_gpu=PEAK_HBM_GBS * 1e9 / ((GATHER_BYTES_NERF + GATHER_BYTES_PROP) // (2 if args.autocast else 1) + 84),
                mfma_bound_rays_per_s_per_gpu=(PEAK_F16_MFMA_TF * 1e12 / (S_NERF * 276 * 32768 / 32) if args.autocast else
                                               PEAK_F16_MFMA_TF * 1e12 / (3 * S_NERF * 2 * MAC_NERF_SPLIT) if split else
                                               PEAK_F32_MFMA_TF * 1e12 / FLOP_NERF_RAY),
                note="frac = value / (n_gpus x min(HBM-bound, MFMA-bound) rate), SURVEY.md 8(d); gather and MLP kernels run back to back, "
                     "so 1 / (1 / hbm + 1 / mfma) is the serial ceiling"),
            "launch": ("self-launched: bench.py re-executed itself under torch.distributed.run" if os.environ.get("UCN_BENCH_SELF_LAUNCHED")
                       else ("torch.distributed.run (external launcher)" if world > 1 else "single process")),
            # (from the timed sample of passes: ms per ray of the sample x the rank's rays per step)
            "kernel_ms_per_step_rank0": {"features_prop": feat_ms[0] / max(rays_seen[0], 1) * (hi - lo), "mlp_prop": mlp_ms[0] / max(rays_seen[0], 1) * (hi - lo),
                                         "features_nerf": feat_ms[1] / rays_seen[1] * (hi - lo), "mlp_nerf": mlp_ms[1] / rays_seen[1] * (hi - lo)},
        }
        if world > 1:
            # proof that the collective library saw N ranks, and what the frame's one exchange cost on rank 0
            res["dist_backend"] = torch.distributed.get_backend()
            res["rccl_ranks"] = torch.distributed.get_world_size() if res["dist_backend"] == "nccl" else 0
            res["dist_ranks"] = torch.distributed.get_world_size()
            if exch:
                ms = [a.elapsed_time(b) for a, b, _ in exch]
                res["all_gather"] = dict(ms_per_frame=sum(ms) / len(ms), bytes_sent_per_rank=exch[0][2], frames=len(ms),
                                         bytes_received_per_rank=exch[0][2] * world,
                                         algbw_GBps=exch[0][2] * world / (sum(ms) / len(ms) * 1e-3) / 1e9,
                                         note="HIP events around the frame's one packed all_gather_into_tensor on rank 0 (includes waiting "
                                              "for the slowest rank's shard)")
        # the CPU leg runs AFTER the training legs when both are on: its 32- and 256-thread passes leave the host busy enough
        # (thread pools, a changed torch thread count) to show in the launch-heavy training steps that followed it -- r06: 9.37 ms
        # in the line against 8.7 ms for the same step run alone on the same box (profiles/r06/bench_order_note.txt)
        if world == 1 and not args.no_cpu_baseline:
            flat_cpu = {k: v.reshape(n_rays, -1) for k, v in batch.items() if k != "rand_vec"}
            rgb_cpu = out["rgb"].reshape(n_rays, 3).clone()
            cpu_leg = lambda: cpu_baseline(sd, flat_cpu, rand_vec, rgb_cpu, heads=args.cfg5, eval_camidx=eval_camidx if args.cfg5 else None,
                                           n_sample=8192 if args.cfg5 else 40960)
            if args.no_train or args.no_extras or args.cameras != 1 or args.cfg5:
                res["cpu_baseline"] = cpu_leg()
                cpu_leg = None
        if world == 1 and not args.no_train and args.cameras == 1 and not args.cfg5:
            flat = {k: v.reshape(n_rays, -1) for k, v in batch.items() if k != "rand_vec"}
            # the price of strict fp32 in the headline: the same frame with the exact-fp32 MFMA dense layers (--mlp-mode 0)
            if model.nerf_mlp.mlp_mode != 0:
                keep = model.nerf_mlp.mlp_mode
                model.nerf_mlp.mlp_mode = 0
                try:
                    step(); torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    out0 = step(); torch.cuda.synchronize()
                    d1 = time.perf_counter() - t1
                    res["value_exact_fp32"] = dict(rays_per_s=n_rays / d1, ms_per_frame=d1 * 1e3, steps=1, warmup=1, dtype="f32",
                                                   mlp_mode="fp32-input MFMA (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate)",
                                                   rgb_linf_vs_headline_frame=float((out0["rgb"] - out["rgb"]).abs().max()))
                    del out0
                finally:
                    model.nerf_mlp.mlp_mode = keep
            # the reference's returned key set (models.py:965-971 also hands back the [H, W, 128] per-sample `weights` of the last level,
            # which extract.py:355 reads): the same frame with Config.render_gather_weights = True -- at one GPU a 1.26 GB permute more,
            # at N GPUs 27 x the exchange payload; the headline `value` returns every other key
            cfg.render_gather_weights = True
            try:
                step(); torch.cuda.synchronize()
                t1 = time.perf_counter()
                outw = step(); torch.cuda.synchronize()
                d1 = time.perf_counter() - t1
                res["value_with_weights"] = dict(rays_per_s=n_rays / d1, ms_per_frame=d1 * 1e3, steps=1, warmup=1,
                                                 returned_keys=sorted(k for k in outw if torch.is_tensor(outw[k]) or isinstance(outw[k], list)),
                                                 weights_shape=list(outw["weights"].shape) if "weights" in outw else None)
                del outw
            finally:
                cfg.render_gather_weights = False
            res["train_step"] = train_step_ms(model, flat, device)       # outside the timed region
            # the reference's shipped precision: fp32 (no autocast), hand-written fp32 MFMA dense layers; and the same graph on the
            # library GEMMs for comparison
            res["train_step_fp32"] = fp32_step_both_engines(model, flat, device, steps=6)
            prev = os.environ.get("UCN_F32_LIBRARY")
            os.environ["UCN_F32_LIBRARY"] = "1"
            try:
                # NOT a like-for-like GEMM A/B: UCN_F32_LIBRARY=1 selects the r03 graph (uncomposed colour MLP) on torch's library GEMMs
                res["train_step_fp32"]["r03_graph_on_library_gemms_ms"] = train_step_ms(model, flat, device, steps=6, autocast=False)["ms"]
            finally:
                if prev is None:
                    os.environ.pop("UCN_F32_LIBRARY", None)
                else:
                    os.environ["UCN_F32_LIBRARY"] = prev
            res["sky_layer"] = sky_layer_ms(flat, device)
            res["ray_generation"] = ray_generation_ms(device)
            res["virtual_warp"] = virtual_warp_ms(device)
            res["density_query"] = density_query_ms(model, device)
            res["tsdf_fusion"] = tsdf_fusion_ms(device)
            res["hbm_probe"] = hbm_probe(device)
            if not args.no_extras:
                del out, batch
                torch.cuda.empty_cache()
                # what the reference's shipped launch trains: sky NeRF + colour-correction head on (train_waymo.sh:11-12)
                hmodel, _, _ = build_model(device, heads=True)
                res["train_step_sky"] = train_step_ms(hmodel, flat, device, heads=True)
                res["train_step_sky_fp32"] = fp32_step_both_engines(hmodel, flat, device, steps=4, heads=True)
                del hmodel
                torch.cuda.empty_cache()
                # the same step on the reference's OWN grid (waymo.gin / class defaults: L = 10, C = 4, T = 2^21 -- 256 row blocks
                # per hashed level -- and 128 + 32 samples): not a BASELINE config, but what a user of the reference trains
                rmodel, _, _ = build_model(device, grid="R")
                res["train_step_waymo_gin_grid"] = dict(train_step_ms(rmodel, flat, device, steps=8),
                                                        grid="L 10, C 4, T 2^21 (240 + 106 MB tables), 128 + 32 samples")
                del rmodel
                torch.cuda.empty_cache()
                # ... and that grid WITH the sky NeRF + colour head: scripts/train_waymo.sh as shipped (waymo.gin + model_sky + brightness_correction)
                rmodel, _, _ = build_model(device, heads=True, grid="R")
                res["train_step_waymo_gin_launch"] = dict(train_step_ms(rmodel, flat, device, steps=8, heads=True),
                                                          grid="L 10, C 4, T 2^21, 128 + 32 samples; sky NeRF + colour-correction head on")
                # ... and the reference's LITERAL shipped launch: scripts/train_waymo.sh:3 has no --mixed_precision (fp32 throughout),
                # waymo.gin:7 batch_size = 15000, waymo.gin:10-13 grid, :11-12 model_sky + brightness_correction
                res["train_step_waymo_gin_launch_fp32"] = dict(
                    fp32_step_both_engines(rmodel, flat, device, n_rays=15000, steps=4, heads=True),
                    grid="L 10, C 4, T 2^21, 128 + 32 samples; sky NeRF + colour-correction head on",
                    launch="scripts/train_waymo.sh as shipped: fp32 (no --mixed_precision), batch_size = 15000 (waymo.gin:7)")
                del rmodel, flat
                torch.cuda.empty_cache()
                # north_star's compaction beside the headline, on a fitted (non-random) field
                try:
                    res["fitted_field"] = fitted_field_frames(device, cfg)
                except Exception as e:                                    # noqa: BLE001 -- reported, never fatal to the line
                    res["fitted_field"] = dict(error=f"{type(e).__name__}: {e}"[:300])
                if cpu_leg is not None:
                    res["cpu_baseline"] = cpu_leg()
                    cpu_leg = None
                # the other BASELINE configs, one timed frame each, with their own CPU-oracle L-inf
                res["configs"] = {
                    "configs[3] 5-camera frame, fp32-class": render_config(device, 5, heads=False, autocast=False),
                    "configs[4] 5 cameras virtual poses + sky + colour head, fp32-class": render_config(device, 5, heads=True, autocast=False),
                    "configs[4] same, mixed bf16/fp32 (autocast)": render_config(device, 5, heads=True, autocast=True),
                    # not a BASELINE config: one frame on the reference's own waymo.gin grid (L 10, C 4, T 2^21; 128 + 32 samples)
                    "waymo.gin grid, 1 camera, fp32-class": render_config(device, 1, heads=False, autocast=False, grid="R"),
                }
    if rank == 0 and world == 1 and cpu_leg is not None:                 # (no path above should leave it pending)
        res["cpu_baseline"] = cpu_leg()
    if world > 1 and not args.no_train and args.cameras == 1 and not args.cfg5:
        # every rank takes part (collectives in the backward); after the timed render, outside it
        flat_t = {k: v.reshape(n_rays, -1) for k, v in batch.items() if k != "rand_vec"}
        ddp_res = ddp_train_step(device, flat_t, world, rank)
        if rank == 0:
            res["train_step_ddp"] = ddp_res
    if rank == 0:
        # the line is ~15 KB: a reader (or a driver) that keeps only its TAIL still gets every second-metric number from this last key
        pick = lambda k, f="ms": (res.get(k) or {}).get(f)
        res["summary"] = {
            "rays_per_s": res["value"], "ms_per_frame": res["ms_per_step"], "n_gpus": world,
            "roofline_frac_gather": res["roofline"].get("frac") if res["roofline"].get("bound") == "hbm" else res["roofline_secondary"].get("frac"),
            "rays_per_s_exact_fp32": pick("value_exact_fp32", "rays_per_s"),
            "rays_per_s_with_weights": pick("value_with_weights", "rays_per_s"),
            "rays_per_s_fitted": pick("fitted_field", "value_fitted"), "rays_per_s_fitted_compacted": pick("fitted_field", "value_fitted_compacted"),
            "fitted_alive_fraction": pick("fitted_field", "alive_fraction"),
            "cpu_rays_per_s": pick("cpu_baseline", "value"), "cpu_cores": pick("cpu_baseline", "cores"),
            "rgb_linf_gpu_vs_cpu": pick("cpu_baseline", "rgb_linf_gpu_vs_cpu"),
            "train_step_ms": {k: pick(k) for k in ("train_step", "train_step_fp32", "train_step_sky", "train_step_sky_fp32", "train_step_waymo_gin_grid",
                                                     "train_step_waymo_gin_launch", "train_step_waymo_gin_launch_fp32", "train_step_ddp") if pick(k) is not None},
        }
        print(json.dumps(res))
    if world > 1:
        torch.distributed.destroy_process_group()


def ddp_train_step(device, flat, world, rank, steps=8):
    """BASELINE's second metric at N > 1: the training step under DistributedDataParallel (what accelerator.prepare hands to
    train.py:95), 8192 rays per step over ALL ranks (each rank draws its 8192 / N, datasets.py:278).  Both gradient exchanges
    of internal/dist.py wrap_ddp are timed: "all_reduce" (dense gradients all-reduced over RCCL in one 128 MB bucket, full Adam
    pass on every rank) and "reduce_scatter" (tables out of DDP: reduce-scatter -> Adam on 1 / N of the rows -> all-gather of
    the parameters).  Every rank calls this; the MAX over ranks of the median step time is reported per
    mode.  Never raises: a failure comes back as {"error": ...} so that the headline line survives."""
    out = {}
    for mode in ("all_reduce", "reduce_scatter"):
        try:
            from ucnerf_amd.internal import dist as udist
            model, _, _ = build_model(device)
            ddp = udist.wrap_ddp(model, device_ids=[device.index], grad_exchange=mode)
            table_bytes = sum(p.numel() * 4 for p in model.parameters() if p.numel() >= udist.SHARD_MIN_NUMEL)
            small_bytes = sum(p.numel() * 4 for p in model.parameters() if p.numel() < udist.SHARD_MIN_NUMEL)
            r = train_step_ms(ddp, flat, device, n_rays=8192 // world, steps=steps)
            t = torch.tensor([r["ms"]], device=device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            del ddp, model
            torch.cuda.empty_cache()
            out[mode] = dict(ms=float(t.item()), rays=8192 // world * world, rays_per_rank=8192 // world, n_gpus=world, steps=steps,
                             rays_per_s=(8192 // world * world) / (float(t.item()) * 1e-3), autocast=r["autocast"],
                             grad_exchange=dict(
                                 mode=mode, table_gradient_bytes=table_bytes, dense_layer_gradient_bytes=small_bytes,
                                 # ring accounting per rank: an all-reduce moves 2 (N-1)/N x the buffer, each half of it (N-1)/N
                                 bytes_on_the_wire_per_rank_before_the_optimiser=int(
                                     (2 if mode == "all_reduce" else 1) * (world - 1) / world * table_bytes + 2 * (world - 1) / world * small_bytes),
                                 bytes_on_the_wire_per_rank_after_the_optimiser=int(0 if mode == "all_reduce" else (world - 1) / world * table_bytes),
                                 adam_rows_per_rank="all" if mode == "all_reduce" else f"1/{world} of the table rows",
                                 note=("DDP: one 128 MB bucket, gradient_as_bucket_view" if mode == "all_reduce" else
                                       "tables: reduce_scatter_tensor(AVG) -> ucn_adam_step on this rank's rows -> all_gather_into_tensor; "
                                       "dense layers stay in DDP's bucket")))
        except Exception as e:                                        # noqa: BLE001 -- reported, not raised (see docstring)
            out[mode] = dict(error=f"{type(e).__name__}: {e}"[:300])
    best = min((m for m in out if "ms" in out[m]), key=lambda m: out[m]["ms"], default=None)
    res = dict(out[best]) if best else dict(error="both gradient exchanges failed")
    res["modes"] = out
    return res


if __name__ == "__main__":
    main()
