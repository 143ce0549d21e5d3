"""A/B of the sky forward training kernel's one-tile and two-tile builds inside one process: every buffer the backward consumes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import raymarch as rm
from ucnerf_amd import _lib
if os.environ.get("UCN_TOOL_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["UCN_TOOL_LIB"])
from ucnerf_amd.internal import train_graph as tg
from ucnerf_amd.internal.sky import NeRF

torch.manual_seed(5)
net = NeRF(D=8, d_in_view=3, W=256, multires_view=4, output_ch=4, skips=[4]).cuda()
with torch.no_grad():
    for p in net.parameters():
        if p.dim() == 1:
            p.normal_(0, 0.1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rays = rm.synthetic_rays(n, seed=12)
o, d, cam = (rays[k].cuda() for k in ("origins", "directions", "cam_dirs"))
far = rays["far"].cuda()
got = {}
for tag in ("1a", "2a", "1b", "2b", "1c", "2c"):
    tiles = tag[0]
    os.environ["UCN_SKY_FWD_TILES"] = tiles
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = tg.sky_forward_fused(net, o, d, cam, far)
    torch.cuda.synchronize()
    fn = out.grad_fn
    while fn is not None and "SkyFused" not in type(fn).__name__:
        fn = fn.next_functions[0][0] if fn.next_functions else None
    packed, raw, d_, far_, act, mask, mask_v = fn.saved_tensors
    got[tag] = dict(out=out.detach().float().clone(), raw=raw.clone(), act=act.float().clone(), mask=mask.clone(), mask_v=mask_v.clone())
for x, y in (("1a", "1b"), ("1a", "1c"), ("2a", "2b"), ("2a", "2c"), ("1a", "2a")):
    print(x, y, {k: int((got[x][k][..., :2208] != got[y][k][..., :2208]).sum()) if k == "act" else int((got[x][k] != got[y][k]).sum()) for k in got[x]})
a, b = got["1a"], got["2a"]
ld = a["act"].shape[1]
print("act ld", ld, "M", a["act"].shape[0])
for k in ("out", "raw"):
    print(k, "max diff", float((a[k] - b[k]).abs().max()))
da = (a["act"] - b["act"]).abs()
print("act: differing elements", int((da > 0).sum()), "max", float(da.max()))
cols = (da > 0).any(0).nonzero().flatten()
rows = (da > 0).any(1).nonzero().flatten()
print("  columns with differences:", cols[:8].tolist(), "...", cols[-8:].tolist(), "count", cols.numel())
print("  rows with differences:", rows[:8].tolist(), "...", rows[-8:].tolist(), "count", rows.numel())
for blk in range(0, ld, 256):
    print("  block at col", blk, "differing", int((da[:, blk:blk + 256] > 0).sum()))
dm = (a["mask"] != b["mask"])
print("mask: differing words per layer", dm.reshape(8, -1).sum(1).tolist())
print("mask_v differing", int((a["mask_v"] != b["mask_v"]).sum()))
