import torch, time
dev = torch.device("cuda", 0)
M, LD = 1 << 20, 864
dt = torch.bfloat16
act = torch.randn(M, LD, device=dev, dtype=dt)
d1 = torch.randn(M, 256, device=dev, dtype=dt)
dense = torch.randn(M, 544, device=dev, dtype=dt)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for c in (2048, 4096, 8192, 16384):
    B = M // c
    a3 = act.view(B, c, LD); d3 = d1.view(B, c, 256); dn = dense.view(B, c, 544)
    print(c, "a  d1^T act[256:800]      ", round(t(lambda: torch.bmm(d3.transpose(1, 2), a3[:, :, 256:800]))))
    print(c, "a' d1^T dense544          ", round(t(lambda: torch.bmm(d3.transpose(1, 2), dn))))
    print(c, "b  two: [256:512],[512:800]", round(t(lambda: (torch.bmm(d3.transpose(1, 2), a3[:, :, 256:512]), torch.bmm(d3.transpose(1, 2), a3[:, :, 512:800])))))
    print(c, "c  act^T d1               ", round(t(lambda: torch.bmm(a3[:, :, 256:800].transpose(1, 2), d3))))
    print(c, "e  d1^T act[256:768] (512)", round(t(lambda: torch.bmm(d3.transpose(1, 2), a3[:, :, 256:768]))))
    print(c, "f  d1^T act[512:800] (288)", round(t(lambda: torch.bmm(d3.transpose(1, 2), a3[:, :, 512:800]))))
    print(c, "g  d1^T act[768:864] (96) ", round(t(lambda: torch.bmm(d3.transpose(1, 2), a3[:, :, 768:864]))))
    print(c, "h  d1^T act[768:800] (32) ", round(t(lambda: torch.bmm(d3.transpose(1, 2), a3[:, :, 768:800]))))
    print(c, "i  d1^T act[0:256]        ", round(t(lambda: torch.bmm(d3.transpose(1, 2), a3[:, :, 0:256]))))
