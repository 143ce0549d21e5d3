"""Which Python lines of the training step launch the small torch kernels (copies, fills, adds, casts)?
a TorchDispatchMode over ONE step of bench.py's configs[2] step (argv[1]: '' | heads | fp32 | heads_fp32), grouped by
(op, innermost ucnerf_amd / bench frame).  Prints ops by launch count."""
import collections, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from ucnerf_amd.internal import train_utils as tu
dev = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else ""
heads = mode in ("heads", "heads_fp32")
fp32 = mode in ("fp32", "heads_fp32")
model, cfg0, sd = bench.build_model(dev, heads=heads, grid="B")
fr = bench.frame_rays(dev)
n = bench.H_IMG * bench.W_IMG
flat = {k: v.reshape(n, -1) for k, v in fr.items()}
cfg = types.SimpleNamespace(data_loss_type='charb', charb_padding=0.001, data_loss_mult=1.0, data_coarse_loss_mult=0.,
                            anti_interlevel_loss_mult=0.01, pulse_width=[0.03, 0.003], distortion_loss_mult=0.005,
                            hash_decay_mults=0.1, disable_multiscale_loss=False, sky_weight=0.002, idt_weight=0.002)
g = torch.Generator(device=dev).manual_seed(2)
opt = tu.FusedAdam(model.parameters(), lr=0.01, betas=(0.9, 0.99), eps=1e-8)
model.train()
R = 8192

def make():
    idx = torch.randint(0, n, (R,), device=dev, generator=g)
    b = {k: v[idx][:, None, None, :] for k, v in flat.items()}
    b['rgb'] = torch.rand(R, 1, 1, 3, device=dev, generator=g)
    if heads:
        b['cam_idx'] = torch.randint(0, 210, (R, 1, 1, 1), device=dev, generator=g)
        b['sky_segs'] = (torch.rand(R, 1, 1, device=dev, generator=g) > 0.7).float()
    return b

def step(b):
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=not fp32):
        rend, hist = model(True, b, 0.5, False, zero_glo=False)
    loss = (tu.compute_data_loss(b, rend, cfg)[0] + tu.anti_interlevel_loss(hist, cfg) + tu.distortion_loss(hist, cfg) + tu.hash_decay_loss(hist, cfg))
    if heads:
        loss = loss + cfg.sky_weight * tu.sky_loss(b, rend) + cfg.idt_weight * tu.transformIdentityLoss(rend)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    tu.clip_gradients(model, None, cfg)
    opt.step()

for _ in range(3):
    step(make())
b = make()
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
sites = collections.Counter()
SKIP = ("aten::view", "aten::_unsafe_view", "aten::reshape", "aten::t", "aten::transpose", "aten::permute", "aten::expand", "aten::slice", "aten::select",
        "aten::unsqueeze", "aten::squeeze", "aten::detach", "aten::alias", "aten::as_strided", "aten::empty", "aten::unflatten", "aten::split", "aten::unbind",
        "aten::empty_like", "aten::empty_strided", "aten::new_empty", "aten::_local_scalar_dense", "aten::is_", "aten::stride", "aten::size", "aten::sym_")

class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types_, args=(), kwargs=None):
        name = func.name()
        if not name.startswith(SKIP):
            frame = "(autograd engine / no python frame)"
            for fs in reversed(traceback.extract_stack()[:-1]):
                if ("ucnerf_amd" in fs.filename or fs.filename.endswith("bench.py")) and not fs.filename.endswith("_lib.py"):
                    frame = f"{fs.filename.split('ucnerf_amd/')[-1]}:{fs.lineno} {fs.line[:90] if fs.line else ''}"
                    break
            sites[(name, frame)] += 1
        return func(*args, **(kwargs or {}))

with Spy():
    step(b)
torch.cuda.synchronize()
print(f"mode {mode!r}: {sum(sites.values())} dispatched ops (views and allocations skipped) in one step")
for (name, frame), cnt in sorted(sites.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{cnt:4d}  {name:36s} {frame}")
