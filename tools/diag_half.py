import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import grid_cpu
from ucnerf_amd.gridencoder import _backend
D, C, gridtype, align, interp = 3, 1, 0, False, 0
rng = np.random.default_rng(600 + D * 10 + C)
L, T = 7, 11
pls, offsets, sizes, _ = grid_cpu.table_layout(L, C, 16, 1024, T, input_dim=D, align_corners=align)
table = torch.from_numpy((rng.random((int(offsets[-1]), C), dtype=np.float32) * 2 - 1).astype(np.float16))
B = 2000
x = rng.random((B, D), dtype=np.float32)
x = torch.from_numpy(x)
S = np.log2(pls)
want = torch.empty(L, B, C, dtype=torch.float16)
wjac = torch.empty(B, L * D * C, dtype=torch.float16)
grid_cpu.grid_encode_forward_half(x, table, offsets, want, B, D, C, L, S, 16, wjac, gridtype, align, interp)
got = torch.empty(L, B, C, device="cuda", dtype=torch.float16)
gjac = torch.empty(B, L * D * C, device="cuda", dtype=torch.float16)
_backend.grid_encode_forward(x.cuda(), table.cuda(), offsets.cuda(), got, B, D, C, L, S, 16, gjac, gridtype, align, interp)
g = got.cpu()
bad = (g.view(torch.int16) != want.view(torch.int16)).nonzero()
print("mismatch", len(bad), "of", g.numel(), offsets)
for i in bad[:10]:
    l, b, c = i.tolist()
    print(l, b, float(g[l, b, c]), float(want[l, b, c]), x[b])
print("per level mismatches", [(g[l].view(torch.int16) != want[l].view(torch.int16)).sum().item() for l in range(L)])
