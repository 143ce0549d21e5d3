"""CPU-side checks (-m "not gpu"): the C-ABI library builds, loads and exports every symbol the
header declares; host-side metadata logic; the multi-process exchange over gloo (world_size 2).
No kernel is launched here."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch



def H_free_port():
    """a TCP port nobody is listening on right now (fixed pid-derived ports collided with lingering sockets of earlier runs)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from ucnerf_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(REPO, "include", "ucnerf_march.h")).read()
    declared = set(re.findall(r"\b(ucn_[a-z0-9_]+)\s*\(", header))
    declared -= {"ucn_field_t", "ucn_sky_t", "ucn_stream_t"}
    assert len(declared) >= 20
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/ucnerf_march.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert lib.ucn_abi_version() == _lib.ABI_VERSION


def test_argument_errors_surface_as_messages_without_a_gpu():
    """Precondition failures are reported before any launch (the TORCH_CHECK analogue)."""
    from ucnerf_amd import _lib
    lib = _lib.load()
    off = np.array([0, 8, 16], dtype=np.int32)
    rc = lib.ucn_grid_encode_forward(1, 1, off.ctypes.data, 1, 4, 3, 3, 2, 1.0, 16, None, 0, 0, 0, 0, None)
    assert rc != 0 and b"C must be 1, 2, 4, or 8" in lib.ucn_last_error()
    rc = lib.ucn_grid_encode_forward(1, 1, off.ctypes.data, 1, 4, 7, 2, 2, 1.0, 16, None, 0, 0, 0, 0, None)
    assert rc != 0 and b"D must be" in lib.ucn_last_error()
    rc = lib.ucn_resample(None, None, 0, 0.0, 1.0, 0.0, 1, None, 0, 0.0, 4, 1, 1, None)
    assert rc != 0 and b"num_samples must be > 1" in lib.ucn_last_error()       # stepfun.py:271-272
    # the bf16 MLP's entry point: level-major / bf16 features belong to the inference form (no per-ray terms, no stores)
    fwd = lambda pr, lvl, h0=None: lib.ucn_train_fwd(1, 32, 1, 1, 1, 1, pr, pr, 4, 8, h0, None, None, None, 0, 1, None, None, None,
                                                    1, 1, None, None, None, lvl, None)
    assert fwd(1, 2) != 0 and b"inference layout" in lib.ucn_last_error()
    assert fwd(1, 2 | _lib.FEAT_BF16) != 0 and b"bf16 features are the inference form" in lib.ucn_last_error()
    assert fwd(None, 4 | _lib.FEAT_BF16) != 0 and b"level_dim 2" in lib.ucn_last_error()
    assert fwd(None, 2, h0=1) != 0 and b"no stores" in lib.ucn_last_error()
    # r06: the split engine's ReLU bit masks -- one bit per element of whole 32-row tiles; only behind the staged 128- / 256-wide epilogue
    assert lib.ucn_relu_bits_words(33, 256) == 2 * 4 * 32 and lib.ucn_relu_bits_words(32, 128) == 2 * 32 and lib.ucn_relu_bits_words(0, 256) == 0
    h3 = lambda N, bits_out, bits_in, flags=0, mask=None: lib.ucn_gemm_h3_x2(16, 64, 16, 16, 16, None, 64, N, 64, flags, 16, N, mask, N, None, 0, 0,
                                                                            None, 0, None, 0, bits_out, bits_in, None, None)
    assert h3(64, 16, None) != 0 and b"bit masks need the staged epilogue" in lib.ucn_last_error()
    assert h3(256, None, 16, flags=4, mask=16) != 0 and b"a float mask and a bit mask together" in lib.ucn_last_error()


def test_gridencoder_extension_module_exports_the_reference_operator():
    """`_gridencoder` as a real torch extension (pybind11 over the C ABI): the three names of
    bindings.cpp:5-9, importable the way grid.py:10 imports it, linked to THIS libucnerf_march.so, and the reference's
    TORCH_CHECK preconditions (gridencoder.cu:15-18, 449-465) surface as RuntimeError before any launch."""
    from ucnerf_amd import _lib
    from ucnerf_amd.gridencoder import native
    if not os.path.exists(native.path()):
        subprocess.check_call([os.path.join(REPO, "ucnerf_amd", "csrc", "ext", "build_ext.sh")])
    code = ("import sys, torch; sys.path.insert(0, %r); import _gridencoder as _backend; "
            "print(_backend.__file__); print(sorted(n for n in dir(_backend) if not n.startswith('_')))" % native.NATIVE_DIR)
    out = subprocess.check_output([sys.executable, "-c", code], text=True).splitlines()
    assert out[0] == native.path()
    assert out[1] == "['abi_version', 'grad_total_variation', 'grid_encode_backward', 'grid_encode_forward']"
    g = native.load()
    assert g.abi_version() == _lib.ABI_VERSION
    x, off = torch.zeros(4, 3), torch.zeros(3, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="inputs must be a CUDA tensor"):
        g.grid_encode_forward(x, x, off, x, 4, 3, 2, 2, 1.0, 16, None, 0, False, 0)
    with pytest.raises(RuntimeError, match="grad must be a CUDA tensor"):
        g.grid_encode_backward(x, x, x, off, x, 4, 3, 2, 2, 1.0, 16, None, None, 0, False, 0)
    with pytest.raises(RuntimeError, match="inputs must be a CUDA tensor"):
        g.grad_total_variation(x, x, x, off, 1.0, 4, 3, 2, 2, 1.0, 16, 0, False)
    with pytest.raises(TypeError):                    # pybind11 signature: positional arguments as in gridencoder.h:12-15
        g.grid_encode_forward(x, x, off, x)


def test_product_has_no_cpu_path():
    from ucnerf_amd.internal import configs, models
    with models.bindings(NerfMLP=dict(grid_log2_hashmap_size=10), PropMLP=dict(grid_log2_hashmap_size=10)):
        model = models.Model(config=configs.Config(), num_levels=2)
    batch = {k: torch.zeros(4, 3) for k in ("origins", "directions", "viewdirs", "cam_dirs")}
    batch.update({k: torch.zeros(4, 1) for k in ("radii", "near", "far")})
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        model(False, batch, 1.0, False)
    # nothing in the shipped package imports the oracle
    for root, _, files in os.walk(os.path.join(REPO, "ucnerf_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("CPU oracle", ""), os.path.join(root, f)


def test_state_dict_keys_match_the_reference_layout():
    """Checkpoint compatibility."""
    from ucnerf_amd.internal import configs, models
    cfg = configs.Config(model_sky=True, brightness_correction=True)
    with models.bindings(NerfMLP=dict(grid_log2_hashmap_size=10), PropMLP=dict(grid_log2_hashmap_size=10)):
        m = models.Model(config=cfg, num_levels=2, num_prop_samples=128, num_nerf_samples=32)
    sd = m.state_dict()
    for k, shape in {"nerf_mlp.density_layer.0.weight": (64, 40), "nerf_mlp.density_layer.2.weight": (256, 64),
                     "nerf_mlp.lin_second_stage_0.weight": (256, 283), "nerf_mlp.lin_second_stage_1.weight": (256, 539),
                     "nerf_mlp.rgb_layer.weight": (3, 256), "prop_mlp_0.density_layer.0.weight": (64, 24),
                     "prop_mlp_0.density_layer.2.weight": (1, 64), "skynerf.pts_linears.5.weight": (256, 259),
                     "skynerf.views_linears.0.weight": (128, 283), "skynerf.alpha_linear.weight": (1, 256),
                     "skynerf.rgb_linear.weight": (3, 128), "brightness_corr.latent_code": (210, 4),
                     "brightness_corr.sky_latent_code": (210, 4),
                     "brightness_corr.brightness_MLP.output_linear.weight": (12, 256)}.items():
        assert tuple(sd[k].shape) == shape, k
    assert sd["nerf_mlp.encoder.offsets"].dtype == torch.int32 and sd["nerf_mlp.encoder.idx"].dtype == torch.int64
    assert sd["nerf_mlp.encoder.grid_sizes"].tolist()[:3] == [17, 33, 65]
    # the reference's full-size table layout (no allocation of weights needed to check it)
    from ucnerf_amd.gridencoder import GridEncoder
    e = GridEncoder.__new__(GridEncoder)
    torch.nn.Module.__init__(e)
    from oracle import grid_cpu
    _, off, _, _ = grid_cpu.table_layout(10, 4, 16, 8192, 21)
    assert int(off[-1]) == 14995560


def test_shard_bounds_cover_the_frame_once():
    from ucnerf_amd.internal import dist as ud
    for n, world in [(2457600, 8), (384, 3), (10, 4), (7, 8)]:
        seen = np.zeros(n, int)
        for r in range(world):
            lo, hi = ud.shard_bounds(n, world, r)
            seen[lo:hi] += 1
            assert hi - lo <= ud.rows_per_rank(n, world)
        assert (seen == 1).all()


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from ucnerf_amd.internal import dist as ud
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
n = 101                                     # odd: the last shard is one row short (padding path)
full = {"rgb": torch.arange(n * 3, dtype=torch.float32).reshape(n, 3), "depth": torch.arange(n, dtype=torch.float32).reshape(n, 1) * 2}
lo, hi = ud.shard_bounds(n, 2, rank)
got = ud.all_gather_rows({k: v[lo:hi].clone() for k, v in full.items()}, n, 2, rank)
assert all(torch.equal(got[k], full[k]) for k in full), "row gather mismatch"
lv = [{"ray_sdist": torch.full((4, 5), float(rank)), "ray_rgbs": torch.full((4, 2, 3), float(rank) + 10)} for _ in range(2)]
b = ud.all_gather_bundles(lv, 2, rank)
assert b[0]["ray_sdist"].shape == (8, 5) and b[1]["ray_rgbs"].shape == (8, 2, 3)
assert b[0]["ray_sdist"][:4].eq(0).all() and b[0]["ray_sdist"][4:].eq(1).all() and b[1]["ray_rgbs"][4:].eq(11).all()
dist.barrier(); dist.destroy_process_group()
print("OK", rank)
'''


def test_all_gather_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = str(H_free_port())
    procs = [subprocess.Popen([sys.executable, str(script), REPO, port, str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert all("OK" in o for o in outs)


EXCHANGE_WORKER = r'''
import copy, os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from ucnerf_amd.internal import dist as ud, train_utils as tu
rank = int(sys.argv[3])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=rank, world_size=2)


class Field(torch.nn.Module):
    """a table (gathered rows, like the grid) + a small dense layer: the two kinds of parameters of the path"""
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.table = torch.nn.Parameter(torch.rand(4096, 2, generator=g) * 2 - 1)
        self.lin = torch.nn.Linear(2, 3)
        with torch.no_grad():
            self.lin.weight.copy_(torch.rand(3, 2, generator=g)); self.lin.bias.zero_()
    def forward(self, idx):
        return self.lin(self.table[idx]).square().mean() + 0.1 * self.table.square().mean()     # data term + a dense decay term


def run(mode):
    torch.manual_seed(0)
    model = Field()
    if rank == 1:                                                    # a rank that starts from other weights: wrapping must bring rank 0's
        with torch.no_grad():
            model.table.add_(1.0); model.lin.weight.add_(1.0)
    ddp = ud.wrap_ddp(model, grad_exchange=mode, shard_min_numel=1024)
    ref = Field()
    assert torch.equal(model.table, ref.table) and torch.equal(model.lin.weight, ref.lin.weight), "the state of rank 0 was not broadcast"
    assert ddp.grad_exchange == mode
    params = list(model.parameters())
    sharded = [p for p in params if getattr(p, "_ucn_sharded", False)]
    assert len(sharded) == (1 if mode == "reduce_scatter" else 0)
    cfg = type("C", (), dict(lr_init=0.01, lr_final=0.001, max_steps=10, lr_delay_steps=0, lr_delay_mult=1.0, adam_beta1=0.9,
                             adam_beta2=0.99, adam_eps=1e-8))
    opt, _ = tu.create_optimizer(cfg, model)
    assert type(opt).__name__ == ("ShardedFusedAdam" if mode == "reduce_scatter" else "FusedAdam")
    g = torch.Generator().manual_seed(100 + rank)                    # every rank its own rays
    for it in range(3):
        idx = torch.randint(0, 4096, (512,), generator=g)
        opt.zero_grad(set_to_none=True)
        loss = ddp(idx)
        if it == 1 and rank == 1:                                    # a non-finite LOCAL gradient on one rank (tensor hooks run before
            def poison(gr):                                          # DDP's reducer): the reference sanitises the REDUCED gradient
                gr = gr.clone(); gr[7, 0] = float("nan"); return gr
            h = model.table.register_hook(poison)
        loss.backward()
        if it == 1 and rank == 1:
            h.remove()
        tu.clip_gradients(model, None, type("C", (), dict(grad_max_norm=0.0, grad_max_val=0.0)))
        opt.step()
    return model, opt


a_model, a_opt = run("all_reduce")
b_model, b_opt = run("reduce_scatter")
for (n, a), (_, b) in zip(a_model.named_parameters(), b_model.named_parameters()):
    assert torch.equal(a, b), (n, float((a.detach() - b.detach()).abs().max()))         # two ranks: a + b == b + a, bit-identical
# both ranks hold the same table (every row was stepped on exactly one of them)
chk = torch.tensor([float(b_model.table.double().abs().sum())])
both = [torch.zeros_like(chk) for _ in range(2)]
dist.all_gather(both, chk)
assert float(both[0]) == float(both[1])
# moments: held for half the table only; the gathered state loads into the unsharded optimiser and equals it
shard_state = [s for s in b_opt.state.values() if s["exp_avg"].numel() == 4096]
assert len(shard_state) == 1
full = b_opt.gathered_state_dict()
ref = a_opt.state_dict()
for i in ref["state"]:
    for k in ("exp_avg", "exp_avg_sq"):
        assert torch.equal(full["state"][i][k].reshape(-1), ref["state"][i][k].reshape(-1)), (i, k)
assert b_opt.exchange_bytes["reduce_scatter_in"] == 4096 * 2 * 4
# ---- checkpoint round trip (ADVICE r05): state_dict() is the gathered, torch.optim.Adam-compatible state (collective); a fresh sharded
# optimiser loads it (every rank slices ITS rows), one more step on both routes -> still bit-identical.  The rank-local state is refused.
saved = b_opt.state_dict()
for i in ref["state"]:
    for k in ("exp_avg", "exp_avg_sq"):
        assert torch.equal(saved["state"][i][k].reshape(-1), ref["state"][i][k].reshape(-1)), (i, k)
import io
buf = io.BytesIO(); torch.save(saved, buf); buf.seek(0)
saved = torch.load(buf, weights_only=False)                          # through a file image, like accelerator.save_state / load_state
c_model = Field()
c_model.load_state_dict(b_model.state_dict())
c_ddp = ud.wrap_ddp(c_model, grad_exchange="reduce_scatter", shard_min_numel=1024, process_group=dist.new_group([0, 1]))
cfg = type("C", (), dict(lr_init=0.01, lr_final=0.001, max_steps=10, lr_delay_steps=0, lr_delay_mult=1.0, adam_beta1=0.9,
                         adam_beta2=0.99, adam_eps=1e-8))
c_opt, _ = tu.create_optimizer(cfg, c_model)
c_opt.load_state_dict(saved)
try:
    c_opt.load_state_dict(b_opt.local_state_dict())
    raise SystemExit("a rank-local optimiser state was accepted")
except ValueError:
    pass
c_opt.load_state_dict(saved)
a_ddp = ud.wrap_ddp(a_model, grad_exchange="all_reduce")
g = torch.Generator().manual_seed(300 + rank)
idx = torch.randint(0, 4096, (512,), generator=g)
for ddp_, opt_, mdl in ((a_ddp, a_opt, a_model), (c_ddp, c_opt, c_model)):
    opt_.zero_grad(set_to_none=True)
    ddp_(idx).backward()
    tu.clip_gradients(mdl, None, type("C", (), dict(grad_max_norm=0.0, grad_max_val=0.0)))
    opt_.step()
for (n, a), (_, c) in zip(a_model.named_parameters(), c_model.named_parameters()):
    assert torch.equal(a, c), ("after the checkpoint round trip", n, float((a.detach() - c.detach()).abs().max()))
# a rank without a table gradient still enters the collectives (zero contribution)
c_opt.zero_grad(set_to_none=True)
c_ddp(idx).backward()
if rank == 1:
    c_model.table.grad = None                                        # (what a batch that never touched the field leaves behind)
c_opt.step()
# gradient clipping needs the reduced gradient: refused with the tables sharded
try:
    tu.clip_gradients(b_model, type("A", (), dict(sync_gradients=True))(), type("C", (), dict(grad_max_norm=1.0, grad_max_val=0.0)))
    raise SystemExit("clip_gradients accepted norm clipping on un-reduced table gradients")
except NotImplementedError:
    pass
# re-wrapping for the all-reduce exchange drops the marks of the sharded wrap
d_ddp = ud.wrap_ddp(b_model, grad_exchange="all_reduce")
assert not any(getattr(p, "_ucn_sharded", False) for p in b_model.parameters())
assert type(tu.create_optimizer(cfg, b_model)[0]).__name__ == "FusedAdam"
dist.barrier(); dist.destroy_process_group()
print("OK", rank)
'''


def test_reduce_scatter_gradient_exchange_equals_all_reduce_world_size_2_gloo(tmp_path):
    """reduce-scatter -> sharded Adam (each rank steps 1 / N of the table rows) -> all-gather of the
    parameters, against DDP's all-reduce + the full Adam pass: identical parameters after 3 steps on 2 ranks (bit for bit),
    a rank-local NaN gradient handled like the reference handles it (nan_to_num on the REDUCED gradient), optimiser state
    interchangeable through state_dict() / load_state_dict() (r06: save -> a new process group -> load -> one more step, bit-identical
    to the all-reduce route; a rank without a table gradient does not hang the exchange; re-wrapping drops the shard marks)."""
    script = tmp_path / "worker.py"
    script.write_text(EXCHANGE_WORKER)
    port = str(H_free_port())
    procs = [subprocess.Popen([sys.executable, str(script), REPO, port, str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert all("OK" in o for o in outs)


def test_unwrap_model_sees_through_ddp_style_wrappers():
    """render_image is handed accelerate's prepared model (reference train.py:95,330): `.module` chains are unwrapped,
    anything else is rejected with a clear message."""
    from ucnerf_amd.internal import models

    class Core(torch.nn.Module):
        def _march(self):
            return "marched"

    class Wrap(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.module = m
    core = Core()
    assert models.unwrap_model(core) is core
    assert models.unwrap_model(Wrap(Wrap(core))) is core
    with pytest.raises(TypeError, match="expected a ucnerf_amd Model"):
        models.unwrap_model(torch.nn.Linear(2, 2))


def test_host_offsets_cache_lives_on_the_tensor():
    """the host copy of `offsets` must not be keyed on the device address (a freed encoder's block is
    handed to the next one by the caching allocator)."""
    from ucnerf_amd.gridencoder import _backend
    a = torch.tensor([0, 8, 24], dtype=torch.int32)
    ha = _backend.host_offsets(a)
    assert list(ha) == [0, 8, 24] and _backend.host_offsets(a) is ha          # cached on the object
    b = torch.tensor([0, 16, 48], dtype=torch.int32)                         # another tensor never sees a's copy
    assert list(_backend.host_offsets(b)) == [0, 16, 48]
    a.copy_(torch.tensor([0, 32, 64], dtype=torch.int32))                    # in-place update invalidates
    assert list(_backend.host_offsets(a)) == [0, 32, 64]


def test_fields_survive_deepcopy_and_pickle():
    """cached C descriptors (ctypes structs with raw pointers) must not travel with copy / pickle."""
    import copy
    import io
    from ucnerf_amd.internal import configs, models
    with models.bindings(NerfMLP=dict(grid_log2_hashmap_size=8, grid_disired_resolution=64),
                         PropMLP=dict(grid_log2_hashmap_size=8)):
        m = models.Model(config=configs.Config(), num_levels=2, num_prop_samples=8, num_nerf_samples=8)
    m.nerf_mlp._fields[1] = ("key", object(), None)            # stand-in for a cached descriptor
    m.nerf_mlp._grid_desc = object()
    c = copy.deepcopy(m)
    assert c.nerf_mlp._fields == {} and not hasattr(c.nerf_mlp, "_grid_desc")
    buf = io.BytesIO()
    m.nerf_mlp._fields[1] = ("key", ctypes.c_void_p(1), None)
    torch.save(m, buf)
    buf.seek(0)
    r = torch.load(buf, weights_only=False)
    assert r.nerf_mlp._fields == {} and torch.equal(r.nerf_mlp.encoder.embeddings, m.nerf_mlp.encoder.embeddings)


def test_instance_keeps_its_bound_configuration_after_the_binding_block():
    """A field built under bindings(NerfMLP=dict(bottleneck_width=64, ...)) must describe itself as 64-wide for its
    whole life (the C descriptor is filled from instance attributes per call), not fall back to the restored class
    default -- found by the configs[0] test: a 64-wide field was packed as 256-wide."""
    from ucnerf_amd.internal import configs, models
    with models.bindings(NerfMLP=dict(bottleneck_width=64, net_width_viewdirs=64, grid_log2_hashmap_size=8,
                                      grid_disired_resolution=64), PropMLP=dict(grid_log2_hashmap_size=8)):
        m = models.Model(config=configs.Config(), num_levels=2, num_prop_samples=8, num_nerf_samples=8)
    assert models.NerfMLP.bottleneck_width == 256                       # class default restored
    assert (m.nerf_mlp.bottleneck_width, m.nerf_mlp.net_width_viewdirs) == (64, 64)
    assert m.nerf_mlp.lin_second_stage_0.weight.shape == (64, 64 + 27)
    assert m.prop_mlp_0.disable_rgb and not m.nerf_mlp.disable_rgb and m.num_nerf_samples == 8


def test_fused_heads_weight_preparation_is_one_gather():
    """train_graph._head_gather_index: ONE gather over the flat bf16 copy of the NeRF field's dense parameters (the colour
    layers composed with the activation-free bottleneck for the forward) yields the forward and the dgrad fragment
    streams, the direction blocks / biases in accumulator order -- element for element what the per-matrix packers build
    from the sliced, transposed, concatenated and woven matrices."""
    from ucnerf_amd.internal import train_graph as tg
    torch.manual_seed(0)
    dt = torch.bfloat16
    for F_in in (32, 40):
        NB = NW = 256
        E, T = 27, 448
        Wd0, Wd1, W0 = torch.randn(64, F_in), torch.randn(NB, 64), torch.randn(NW, NB + E)
        W1, Wr = torch.randn(NW, NW + NB + E), torch.randn(3, NW)
        bd0, bd1, b0, b1, br = torch.randn(64), torch.randn(NB), torch.randn(NW), torch.randn(NW), torch.randn(3)
        Wc0, Wc1 = W0[:, :NB] @ Wd1, W1[:, NW:NW + NB] @ Wd1
        b0c, b1c = b0 + W0[:, :NB] @ bd1, b1 + W1[:, NW:NW + NB] @ bd1
        idx, n = tg._head_gather_index(F_in, NB, NW, E, T, "cpu")
        src = torch.cat([t.reshape(-1) for t in (Wd0, Wd1, W0, W1, Wr, bd0, bd1, b0c, b1c, br, Wc0, Wc1)] + [torch.zeros(1)]).to(dt)
        assert src.numel() == n
        got = src[idx]
        W0x, W0e = W0[:, :NB].to(dt), W0[:, NB:].to(dt)
        W1h, W1x, W1e = W1[:, :NW].to(dt), W1[:, NW:NW + NB].to(dt), W1[:, NW + NB:].to(dt)
        Wd0b, Wd1b, Wrb = Wd0.to(dt), Wd1.to(dt), Wr.to(dt)

        def packed(mats, weave):
            # per-matrix fragment streams of the per-matrix packer, the consumer layer's fragments woven behind each
            # output-tile pair of its producer (field_train.hip), zero-padded to T fragments
            parts = tg._weave([tg._pack_fragments([m], "cpu") for m in mats], *weave)
            flat = torch.cat(parts)
            return torch.cat([flat, flat.new_zeros(T * 512 - flat.numel())])
        fwd = packed([(Wd0b, True), (Wd1b, False), (Wc0.to(dt), False), (torch.cat([W1h, Wc1.to(dt)], 1), False), (Wrb, False)], (3, 4))
        bwd = packed([(Wrb.t(), True), (W1h.t(), False), (torch.cat([W1x.t(), W0x.t()], 1), False), (Wd1b.t(), False),
                      (Wd0b.t(), False)], (2, 3))
        assert torch.equal(got[:T * 512], fwd) and torch.equal(got[T * 512:2 * T * 512], bwd)
        assert int((fwd.reshape(T, 512) != 0).any(dim=1).sum()) <= 248          # what the forward kernel reads: 244 / 248 fragments
        o = 2 * T * 512
        We, be = got[o:o + 2 * NW * E].view(2 * NW, E), got[o + 2 * NW * E:o + 2 * NW * E + 2 * NW]
        bv = got[o + 2 * NW * E + 2 * NW:].float()
        eb = torch.randn(5, E).to(dt)
        for half, (b, We_ref) in enumerate(((b0c, W0e), (b1c, W1e))):
            pr = torch.addmm(be[half * NW:(half + 1) * NW], eb, We[half * NW:(half + 1) * NW].t()).float()
            assert torch.equal(pr, tg._acc_vec(torch.addmm(b.to(dt), eb, We_ref.t()).float(), NW, "cpu"))
        for got_v, (b, w) in zip((bv[:64], bv[64:64 + NB], bv[64 + NB:]), ((bd0, 64), (bd1, NB), (br, 32))):
            assert torch.equal(got_v, tg._acc_vec(b.to(dt).float(), w, "cpu"))


def test_data_loss_levels_and_lazy_stats():
    """compute_data_loss (train_utils.py:171-230): every level through one set of launches; value and stats equal the
    per-level loop of the reference, and the mse statistics are fetched on first read (no host sync inside the loss)."""
    import types
    from ucnerf_amd.internal import train_utils as tu
    g = torch.Generator().manual_seed(3)
    batch = dict(rgb=torch.rand(50, 1, 1, 3, generator=g), lossmult=torch.rand(50, 1, 1, 1, generator=g) + 0.5)
    rend = [dict(rgb=torch.rand(50, 1, 1, 3, generator=g)) for _ in range(3)]
    for kind in ("charb", "mse"):
        for multiscale_off in (False, True):
            cfg = types.SimpleNamespace(data_loss_type=kind, charb_padding=0.001, data_loss_mult=0.7, data_coarse_loss_mult=0.3,
                                        disable_multiscale_loss=multiscale_off)
            loss, stats = tu.compute_data_loss(batch, rend, cfg)
            lm = torch.ones_like(batch['rgb']) if multiscale_off else torch.broadcast_to(batch['lossmult'], batch['rgb'].shape)
            per, mses = [], []
            for r in rend:
                sq = (r['rgb'] - batch['rgb']) ** 2
                mses.append(float((lm * sq).sum() / lm.sum()))
                per.append((lm * (sq if kind == "mse" else torch.sqrt(sq + 0.001 ** 2))).sum() / lm.sum())
            want = 0.3 * sum(per[:-1]) + 0.7 * per[-1]
            assert abs(float(loss) - float(want)) <= 1e-6
            got = stats['mses']                                                      # numpy on any device, like the reference's
            assert isinstance(got, np.ndarray) and not stats.pending('mses')
            assert np.allclose(np.asarray(got), mses, atol=1e-6) and list(stats.keys()) == ['mses']
            stats['psnr'] = 1.0                                                      # train.py:226-227 adds keys
            assert set(dict(stats)) == {'mses', 'psnr'} and len(stats) == 2


def test_hash_decay_on_host_tensors_is_the_reference_reduction():
    """models.py:297-306: mean over (level, channel) of the per-level mean of embeddings^2 -- the host-tensor form of
    train_graph.hash_decay (the device form is ucn_hash_decay, pinned by the G10 loss values)."""
    from ucnerf_amd.internal import configs, models, train_graph
    with models.bindings(NerfMLP=dict(grid_log2_hashmap_size=10), PropMLP=dict(grid_log2_hashmap_size=10)):
        model = models.Model(config=configs.Config(), num_levels=2)
    enc = model.nerf_mlp.encoder
    enc.embeddings.data.normal_()
    off = enc._offsets_np
    per_level = torch.stack([(enc.embeddings[off[i]:off[i + 1]] ** 2).mean(dim=0) for i in range(len(off) - 1)])
    got = train_graph.hash_decay(model.nerf_mlp)
    assert abs(float(got) - float(per_level.mean())) <= 1e-6 * float(per_level.mean())
    got.backward()
    assert enc.embeddings.grad is not None and float(enc.embeddings.grad.abs().sum()) > 0


def test_tile_order_is_a_permutation_of_the_frame_in_blocks():
    """models._tile_order: every pixel once, T x T blocks contiguous (ragged at the edges), inverse undoes it."""
    from ucnerf_amd.internal import models
    for Hh, Ww, T in ((16, 24, 8), (7, 13, 4), (8, 8, 8), (3, 50, 8)):
        perm, inv = models._tile_order(Hh, Ww, T, "cpu")
        assert sorted(perm.tolist()) == list(range(Hh * Ww))
        assert torch.equal(perm[inv], torch.arange(Hh * Ww)) and torch.equal(inv[perm], torch.arange(Hh * Ww))
        r, c = perm // Ww, perm % Ww
        block = (r // T) * ((Ww + T - 1) // T) + c // T
        assert bool((block[1:] >= block[:-1]).all())                      # blocks are contiguous runs
        first = perm[:min(T, Ww)]
        assert first.tolist() == list(range(min(T, Ww)))                 # row-major inside the first block


def test_bench_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` (no launcher): bench.py starts N ranks itself with the driver's own launcher form; a launcher
    that started a different number of ranks than --gpus says is refused (the line's n_gpus is never a flag echoed back);
    fewer GPUs than RCCL ranks is an error, not a silent one-rank run.  CPU-only: the launcher is intercepted."""
    import subprocess
    import sys
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    assert bench.self_launch(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and seen["env"]["UCN_BENCH_SELF_LAUNCHED"] == "1"
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    monkeypatch.delenv("UCN_DIST_BACKEND", raising=False)
    assert bench.self_launch(4) == 2                                  # RCCL: one GPU per rank
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0 and "launcher started 1 rank" in p.stderr
