import sys, os, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import helpers as H
from oracle import raymarch as rm, grid_cpu
from ucnerf_amd import _lib
lib = _lib.load()
spec = rm.make_spec("tiny"); sd = rm.init_state(spec, seed=1)
model, _ = H.hip_model(spec, sd)
mlp = model.nerf_mlp
d = mlp.field()
g = torch.Generator().manual_seed(0)
B = 4096
pts = torch.randn(B, 1, 3, generator=g) * torch.logspace(-1, 1, B)[:, None, None]
std = torch.rand(B, 1, generator=g) * 1e-3
L, C = mlp.encoder.num_levels, mlp.encoder.level_dim
feat = torch.empty(L * B * C, device="cuda"); coord = torch.empty(B, 3, device="cuda")
pts_d, std_d = pts.cuda().contiguous(), std.cuda().contiguous()
_lib.check(lib.ucn_points_features(ctypes.byref(d), pts_d.data_ptr(), std_d.data_ptr(), B, 1, 1, 1, feat.data_ptr(), coord.data_ptr(), _lib.stream()))
torch.cuda.synchronize()
cm, cs = rm.contract_points(pts.reshape(-1, 3), std.reshape(-1))
want = cm / 2
got = coord.cpu()
inside = (pts.reshape(-1, 3) ** 2).sum(-1) <= 1
print("coord bit-equal inside:", bool(torch.equal(got[inside], want[inside])), " outside:", bool(torch.equal(got[~inside], want[~inside])),
      "maxdiff outside", float((got[~inside] - want[~inside]).abs().max()), "n mismatch", int((got != want).any(-1).sum()), "of", B)
# features: oracle with oracle coords vs HIP
f = feat.reshape(L, B, C).permute(1, 0, 2).reshape(B, L * C).cpu()
_, _, _, ofeat = rm.field_density_features(spec.nerf, sd, pts, std)
dd = (f - ofeat.reshape(B, -1)).abs().reshape(B, L, C)
print("feature maxdiff per level (all):", [f"{float(x):.1e}" for x in dd.amax(dim=(0, 2))])
same = ~(got != want).any(-1)
print("feature maxdiff per level (bit-equal coords only):", [f"{float(x):.1e}" for x in dd[same].amax(dim=(0, 2))])
# erf weights alone
pls, offsets, sizes, _ = spec.nerf.layout()
damp = rm.level_damping(cs / 2, sizes)
bad = (got != want).any(-1)
i = torch.nonzero(bad)[:5, 0]
for k in i.tolist():
    print("mismatch", k, pts.reshape(-1,3)[k].tolist(), got[k].tolist(), want[k].tolist(), float((pts.reshape(-1,3)[k]**2).sum()))
print("damp range", float(damp.min()), float(damp.max()))
