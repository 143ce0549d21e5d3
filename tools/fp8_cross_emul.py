"""CPU emulation of the NeRF field MLP with the split product's two cross terms (hi*lo + lo*hi) taken in fp8 (e4m3 with MX block
scales of 32, e5m2 = the high byte of the f16 operands) instead of f16: rgb / density error against float64 for three weight scales.
Companion of tools/fp8_cross_bench.hip (layout + rate of v_mfma_scale_f32_32x32x64_f8f6f4); results in profiles/r02c/fp8_cross.txt."""
import torch, math
torch.manual_seed(0)
torch.set_num_threads(8)
def q_e5m2(x, rn=True):
    # x float32 tensor holding f16-representable values -> e5m2 (keep 2 mantissa bits)
    h = x.to(torch.float16).view(torch.int16).to(torch.int32) & 0xffff
    if rn:
        h = h + 0x7f + ((h >> 8) & 1)       # round to nearest even on the dropped byte
    h = h & 0xff00
    h = torch.where(h >= 0x8000, h - 0x10000, h).to(torch.int16)
    return h.view(torch.float16).float()
def q_e4m3_block(x, axis_blocks):
    # x [..., K]; per block of 32 along last axis: shared power-of-two scale so that block max <= 448; e4m3 rounding (3 mantissa bits, min normal 2^-6, subnormal 2^-9)
    K0 = x.shape[-1]
    pad = (-K0) % 32
    if pad:
        x = torch.nn.functional.pad(x, (0, pad))
    sh = x.shape
    xb = x.reshape(*sh[:-1], sh[-1] // 32, 32)
    mx = xb.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    e = torch.floor(torch.log2(mx)) - 8          # scale so that max in [256, 512) -> clamp 448
    s = torch.pow(2.0, e)
    y = xb / s
    # quantize to e4m3
    a = y.abs().clamp_max(448.0)
    ex = torch.floor(torch.log2(a.clamp_min(2.0 ** -9))).clamp_min(-6)
    step = torch.pow(2.0, ex - 3)
    q = torch.round(a / step) * step
    q = torch.sign(y) * q.clamp_max(448.0)
    return (q * s).reshape(sh)[..., :K0]
def split(x):
    hi = x.to(torch.float16).float()
    lo = (x - hi).to(torch.float16).float()
    return hi, lo
def mm(a, w, mode):
    # a [M,K], w [N,K] -> a w^T with the product model
    if mode == 'f64':
        return (a.double() @ w.double().t())
    if mode == 'f32':
        return (a @ w.t()).double()
    ah, al = split(a); wh, wl = split(w)
    main = ah.double() @ wh.double().t()
    if mode == 'split3':
        return main + ah.double() @ wl.double().t() + al.double() @ wh.double().t()
    if mode == 'hihi':
        return main
    if mode in ('e5m2rn', 'e5m2tr'):
        rn = mode == 'e5m2rn'
        c = q_e5m2(ah, rn).double() @ q_e5m2(wl, rn).double().t() + q_e5m2(al, rn).double() @ q_e5m2(wh, rn).double().t()
        return main + c
    if mode == 'e4m3':
        c = q_e4m3_block(ah, 32).double() @ q_e4m3_block(wl, 32).double().t() + q_e4m3_block(al, 32).double() @ q_e4m3_block(wh, 32).double().t()
        return main + c
def lin(out_f, in_f, scale=1.0):
    l = torch.nn.Linear(in_f, out_f)
    return l.weight.detach() * scale, l.bias.detach()
def run(mode, P, feat, enc):
    (Wd0, bd0), (Wd1, bd1), (W0, b0), (W1, b1), (Wr, br) = P
    f = lambda t: t.float()
    h0 = torch.relu(mm(feat, Wd0, mode) + bd0.double())
    x = mm(f(h0), Wd1, mode) + bd1.double()
    in0 = torch.cat([f(x), enc], -1)
    h1 = torch.relu(mm(in0, W0, mode) + b0.double())
    in1 = torch.cat([f(h1), in0], -1)
    h2 = torch.relu(mm(in1, W1, mode) + b1.double())
    y = mm(f(h2), Wr, mode) + br.double()
    rgb = torch.sigmoid(y) * 1.002 - 0.001
    dens = torch.nn.functional.softplus(x[:, 0] - 1)
    return rgb, dens
M = 60000
for wscale in (1.0, 3.0, 6.0):
    P = [lin(64, 32, wscale), lin(256, 64, wscale), lin(256, 283, wscale), lin(256, 539, wscale), lin(3, 256, wscale)]
    feat = (torch.rand(M, 32) * 2 - 1) * 0.5
    d = torch.nn.functional.normalize(torch.randn(M, 3), dim=-1)
    enc = torch.cat([d] + [fn(d * s) for s in (1, 2, 4, 8) for fn in (torch.sin, torch.cos)], -1)
    ref, dref = run('f64', P, feat, enc)
    print('weight scale', wscale, 'logit range', float(torch.logit(((ref+0.001)/1.002).clamp(1e-9,1-1e-9)).abs().max()))
    for mode in ('f32', 'split3', 'e4m3', 'e5m2rn', 'e5m2tr', 'hihi'):
        r, dn = run(mode, P, feat, enc)
        e = (r - ref).abs(); ed = ((dn - dref).abs() / dref.abs().clamp_min(1e-3))
        print('  %-7s rgb err max %.2e rms %.2e | density rel err max %.2e' % (mode, float(e.max()), float(e.pow(2).mean().sqrt()), float(ed.max())))
