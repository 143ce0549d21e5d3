"""csrc/gemm_f32.hip + csrc/gemm_h3.hip (-m gpu): the fp32 training route's dense layers on hand-written MFMA kernels, against float64
matmuls and against torch's own autograd of F.linear, on both engines of internal/dense_f32.py.  Bars: "exact" = exact fp32 products +
fp32 accumulation => a dot product of length K differs from the float64 result by ~sqrt(K) ulp of its absolute-value sum; "split" =
three f16-half products per term (each operand to 2^-22 relative, the lo x lo term dropped: <= 3 x 2^-22 = 7.2e-7 of |x| |w| per term,
random in sign) + the same fp32 accumulation: 1.2e-6 of the absolute-value sum."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["exact", "split"])
def engine(request):
    from ucnerf_amd.internal import dense_f32 as D
    prev, prev_rows = D.set_engine(request.param), D.H3_MIN_ROWS
    D.H3_MIN_ROWS = 4096                # (the product keeps operands below 32768 rows on the exact kernels: launch-bound; here the split kernels take them too)
    yield request.param
    D.set_engine(prev)
    D.H3_MIN_ROWS = prev_rows


def _eps(engine, base):
    return base if engine == "exact" else max(base, 1.2e-6)


def _ref(x, w, b=None, acc=None, relu=False):
    y = x.double() @ w.double().t()
    if b is not None:
        y = y + b.double()
    if acc is not None:
        y = y + acc.double()
    return torch.relu(y) if relu else y


@pytest.mark.parametrize("M,N,K", [(1000, 3, 256), (777, 256, 544), (4096, 64, 32), (130, 128, 28), (8192, 256, 256), (33, 12, 256),
                                   (5, 1, 4), (65536, 256, 64),
                                   # r05: the weight-resident persistent kernel (N K small, >= 64 row tiles) incl. ragged last tiles,
                                   # fewer tiles than waves, K not a multiple of the 32-wide chunk, N below the column tile
                                   (100003, 64, 256), (70001, 4, 256), (3000, 128, 128), (2049, 256, 64), (40000, 60, 36), (8191, 20, 252)])
def test_gemm_f32_against_float64(M, N, K, engine):
    from ucnerf_amd.internal import dense_f32 as D
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    scale = (x.double().abs() @ w.double().abs().t()).max().item() + 1.0
    e = _eps(engine, 4e-7)
    y = D.gemm(x, w, b)
    assert float((y.double() - _ref(x, w, b)).abs().max()) <= e * scale * max(1.0, K ** 0.5 / 4)
    if engine == "split" and M >= D.H3_MIN_ROWS:                  # the epilogue recorded the product's absolute maximum on the device
        assert float(y._ucn_amax[0]) == float(y.abs().max())
    y2 = D.gemm(x, w, None, D.RELU)
    assert float((y2.double() - _ref(x, w, relu=True)).abs().max()) <= e * scale * max(1.0, K ** 0.5 / 4)
    base = torch.randn(M, N, device="cuda", generator=g)
    y3 = base.clone()
    D.gemm(x, w, b, D.ACCUMULATE, out=y3)
    assert float((y3.double() - _ref(x, w, b, base)).abs().max()) <= e * scale * max(1.0, K ** 0.5 / 4)
    # column-slice views of wider buffers in and out (leading dimensions), the reference's concatenated layer inputs
    wide_x = torch.randn(M, K + 8, device="cuda", generator=g)
    wide_y = torch.zeros(M, N + 5, device="cuda")
    D.gemm(wide_x[:, 4:4 + K], w, b, out=wide_y[:, 2:2 + N])
    assert float((wide_y[:, 2:2 + N].double() - _ref(wide_x[:, 4:4 + K], w, b)).abs().max()) <= e * scale * max(1.0, K ** 0.5 / 4)
    assert float(wide_y[:, :2].abs().max()) == 0 and float(wide_y[:, 2 + N:].abs().max()) == 0


@pytest.mark.parametrize("M,N,K,ldm_extra", [(8192, 256, 256, 0), (100003, 64, 256, 8), (5000, 256, 64, 0), (333, 7, 256, 1), (70000, 128, 32, 4)])
def test_gemm_f32_mask_epilogue(M, N, K, ldm_extra, engine):
    """UCN_GEMM_MASK: y = mask > 0 ? (x w^T + bias) : 0 -- the ReLU derivative of the layer below as the epilogue of its d X GEMM;
    exactly the unmasked product where the mask is positive, exactly 0 elsewhere (also with ACCUMULATE: masked after the sum)."""
    from ucnerf_amd.internal import dense_f32 as D
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    mwide = torch.relu(torch.randn(M, N + ldm_extra, device="cuda", generator=g))         # ~half zeros, like a ReLU output
    mask = mwide[:, :N]
    plain = D.gemm(x, w, b)
    got = D.gemm(x, w, b, mask=mask)
    assert torch.equal(got, torch.where(mask > 0, plain, torch.zeros_like(plain)))
    base = torch.randn(M, N, device="cuda", generator=g)
    acc_plain = base.clone(); D.gemm(x, w, b, D.ACCUMULATE, out=acc_plain)
    acc_got = base.clone(); D.gemm(x, w, b, D.ACCUMULATE, out=acc_got, mask=mask)
    assert torch.equal(acc_got, torch.where(mask > 0, acc_plain, torch.zeros_like(acc_plain)))


@pytest.mark.parametrize("M,N,K,group", [(120 * 700, 128, 256, 120), (4096, 256, 64, 128), (1000, 6, 256, 7)])
def test_gemm_f32_row_group_bias(M, N, K, group, engine):
    """rowbias: + rowbias[row // group] before the ReLU -- the per-RAY term of the sky NeRF's view layer under a per-sample GEMM."""
    from ucnerf_amd.internal import dense_f32 as D
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    rb = torch.randn((M + group - 1) // group, N, device="cuda", generator=g)
    got = D.gemm(x, w, b, D.RELU, rowbias=rb, rgroup=group)
    want = torch.relu(x.double() @ w.double().t() + b.double() + rb.double().repeat_interleave(group, dim=0)[:M])
    scale = (x.double().abs() @ w.double().abs().t()).max().item() + 1.0
    assert float((got.double() - want).abs().max()) <= _eps(engine, 4e-7) * scale * max(1.0, K ** 0.5 / 4)


@pytest.mark.parametrize("M,N,K", [(1000, 4, 256), (8191, 256, 544), (4096, 64, 32), (100000, 256, 256), (31, 12, 28), (262144, 128, 284)])
def test_wgrad_f32_against_float64_and_deterministic(M, N, K, engine):
    from ucnerf_amd.internal import dense_f32 as D
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    gy = torch.randn(M, N, device="cuda", generator=g)
    x = torch.randn(M, K, device="cuda", generator=g)
    gw, gb = D.wgrad(gy, x, True)
    want = gy.double().t() @ x.double()
    scale = float((gy.double().abs().t() @ x.double().abs()).max())
    assert float((gw.double() - want).abs().max()) <= _eps(engine, 2e-7) * scale * max(1.0, M ** 0.5 / 16 if engine == "exact" else 1.0)
    assert float((gb.double() - gy.double().sum(0)).abs().max()) <= 2e-7 * float(gy.double().abs().sum(0).max()) * max(1.0, M ** 0.5 / 16)
    gw2, gb2 = D.wgrad(gy, x, True)
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)                  # fixed-order partial sums


@pytest.mark.parametrize("lead,N,K,relu,bias", [((1024, 120), 256, 256, True, True), ((3000,), 3, 256, False, True), ((210,), 256, 4, True, True),
                                                ((64, 128), 1, 256, False, True), ((8192,), 256, 27, False, False), ((500, 7), 128, 283, True, True)])
def test_hip_linear_is_f_linear_with_the_same_gradients(lead, N, K, relu, bias, engine):
    from ucnerf_amd.internal import dense_f32 as D
    g = torch.Generator(device="cuda").manual_seed(N * 7 + K)
    x = torch.randn(*lead, K, device="cuda", generator=g, requires_grad=True)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).requires_grad_(True)
    b = torch.randn(N, device="cuda", generator=g, requires_grad=True) if bias else None
    up = torch.randn(*lead, N, device="cuda", generator=g)

    def run(fn):
        for t in (x, w, b):
            if t is not None:
                t.grad = None
        y = fn(x, w, b)
        (y * up).sum().backward()
        return y.detach(), x.grad.clone(), w.grad.clone(), None if b is None else b.grad.clone()
    got = run(lambda x_, w_, b_: D.hip_linear(x_, w_, b_, relu=relu))
    xd, wd, bd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True), None if b is None else b.detach().double().requires_grad_(True)
    yd = torch.nn.functional.linear(xd, wd, bd)
    if relu:                                    # the float32 forward's own mask: a pre-activation within an ulp of 0 may fall either way
        yd = yd * (got[0] > 0)
    (yd * up.double()).sum().backward()
    want = (yd.detach(), xd.grad, wd.grad, None if bd is None else bd.grad)
    M = int(np.prod(lead))
    for name, a, c in zip(("y", "gx", "gw", "gb"), got, want):
        if a is None:
            continue
        tol = 3e-6 * max(1.0, float(c.abs().max())) * (max(1.0, M ** 0.5 / 16) if name in ("gw", "gb") else 1.0)
        assert a.shape == c.shape and float((a.double() - c).abs().max()) <= tol, (name, float((a.double() - c).abs().max()), tol)


def test_dense_f32_refuses_host_tensors():
    from ucnerf_amd.internal import dense_f32 as D
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        D._HipLinear.apply(torch.randn(4, 4), torch.randn(4, 4), None, False)


@pytest.mark.parametrize("lo,hi", [(-40, -40), (30, 30), (-20, 10), (0, 30)])
def test_split_engine_operand_range(lo, hi):
    """The split engine's operands are f16 halves: without the per-tensor power-of-two scales 1e-12-sized gradients would flush to zero
    and 1e9-sized ones overflow.  Rows of d Y span 2^lo .. 2^hi (up to 2^30 within one tensor); reference = float64.  fp32-class is stated
    as in test_split_f16_range: as close to the float64 product as the exact-fp32 kernel is, up to a factor and a floor -- the floor is
    relative to the TENSOR's largest row (the scale is per tensor: an element 2^-15 below the maximum still carries a normal lo half,
    smaller ones an absolute error of 2^-37 of the maximum), which is what the sums over samples that consume these products see."""
    from ucnerf_amd.internal import dense_f32 as D
    g = torch.Generator(device="cuda").manual_seed(100 + lo * 7 + hi)
    M, N, K = 16384, 256, 256
    gy = torch.randn(M, N, device="cuda", generator=g)
    expo = torch.linspace(lo, hi, M, device="cuda").round()
    gy = gy * torch.exp2(expo)[:, None]
    w = torch.randn(N, K, device="cuda", generator=g) / N ** 0.5
    x = torch.randn(M, K, device="cuda", generator=g)
    want = gy.double() @ w.double()                                  # d X = d Y W
    want_w = gy.double().t() @ x.double()
    out = {}
    prev_rows, D.H3_MIN_ROWS = D.H3_MIN_ROWS, 4096
    for eng in ("exact", "split"):
        prev = D.set_engine(eng)
        try:
            out[eng] = (D.gemm(gy, w.t().contiguous()).double(), D.wgrad(gy, x)[0].double())
        finally:
            D.set_engine(prev)
    D.H3_MIN_ROWS = prev_rows
    assert torch.isfinite(out["split"][0]).all() and torch.isfinite(out["split"][1]).all(), "f16 operand overflow"
    assert float(out["split"][0].abs().max()) > 0
    row_scale = (gy.double().abs() @ w.double().abs()).amax(dim=1, keepdim=True)            # per-row absolute-value sums
    top = float(row_scale.max())
    for eng in ("exact", "split"):
        err = (out[eng][0] - want).abs()
        # rows within 2^-15 of the tensor's largest: relative to the row itself; all rows: relative to the largest row
        big = (row_scale[:, 0] >= top * 2.0 ** -14)
        rel_rows = float((err[big] / row_scale[big]).max())
        rel_top = float(err.max()) / top
        bar = 4e-7 * 4 if eng == "exact" else 1.2e-6 * 4
        assert rel_rows <= bar and rel_top <= bar, (eng, rel_rows, rel_top)
    scale_w = float((gy.double().abs().t() @ x.double().abs()).max())
    e_exact = float((out["exact"][1] - want_w).abs().max()) / scale_w
    e_split = float((out["split"][1] - want_w).abs().max()) / scale_w
    assert e_split <= 3 * e_exact + 1.2e-6, (e_split, e_exact)


def test_split_engine_amax_travels_and_goes_stale_safely():
    """the recorded maximum follows reshapes of the same storage, is recomputed when the tensor has been written through torch
    (version counter) and dropped by forget() after a raw-pointer write"""
    from ucnerf_amd.internal import dense_f32 as D
    prev = D.set_engine("split")
    prev_rows, D.H3_MIN_ROWS = D.H3_MIN_ROWS, 4096
    try:
        g = torch.Generator(device="cuda").manual_seed(5)
        x = torch.randn(8192, 64, device="cuda", generator=g)
        w = torch.randn(128, 64, device="cuda", generator=g)
        y = D.gemm(x, w)
        slot = y._ucn_amax[0]
        assert D.amax_of(y) is slot
        y3 = y.view(64, 128, 128)
        y3._ucn_amax = y._ucn_amax                                   # (what _HipLinear does for its reshaped output)
        assert D.amax_of(D._rows(y3)) is slot
        y.mul_(1000.0)                                               # torch write: version bump -> recomputed
        s2 = D.amax_of(y)
        assert s2 is not slot and float(s2) == float(y.abs().max())
        w2 = torch.randn(32, 128, device="cuda", generator=g)
        z = D.gemm(y, w2)
        assert float((z.double() - y.double() @ w2.double().t()).abs().max()) <= 1.2e-6 * float((y.double().abs() @ w2.double().abs().t()).max())
        D.forget(y)
        assert not hasattr(y, "_ucn_amax")
    finally:
        D.set_engine(prev)
        D.H3_MIN_ROWS = prev_rows


@pytest.mark.parametrize("M,N,K,relu,masked", [(120 * 300, 256, 256, True, False), (8192, 256, 128, False, True), (5000, 128, 64, True, True), (1000, 256, 256, True, False)])
def test_gemm_second_narrow_operand_in_the_epilogue(M, N, K, relu, masked, engine):
    """x2 / w2 (r06): out = x w^T + bias + x2[M, 4] w2[N, 4]^T, then ReLU / mask -- the sky NeRF's [hidden | 3-d point] skip layer and its
    [view branch | density row] gradient as ONE pass over the output (split engine: inside ucn_gemm_h3_x2's epilogue, exact fp32 FMAs;
    exact engine and short operands: two accumulating calls).  Against float64."""
    from ucnerf_amd.internal import dense_f32 as D
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    x2 = torch.zeros(M, 4, device="cuda"); x2[:, :3] = torch.randn(M, 3, device="cuda", generator=g)
    w2 = torch.zeros(N, 4, device="cuda"); w2[:, :3] = torch.randn(N, 3, device="cuda", generator=g)
    mask = torch.relu(torch.randn(M, N, device="cuda", generator=g)) if masked else None
    got = D.gemm(x, w, b, D.RELU if relu else 0, mask=mask, x2=x2, w2=w2)
    want = x.double() @ w.double().t() + b.double() + x2.double() @ w2.double().t()
    if relu:
        want = torch.relu(want)
    if masked:
        want = torch.where(mask > 0, want, torch.zeros_like(want))
    scale = (x.double().abs() @ w.double().abs().t() + x2.double().abs() @ w2.double().abs().t()).max().item() + 1.0
    assert float((got.double() - want).abs().max()) <= _eps(engine, 4e-7) * scale * max(1.0, K ** 0.5 / 4)
    if engine == "split" and M >= D.H3_MIN_ROWS and hasattr(got, "_ucn_amax"):
        assert float(got._ucn_amax[0]) == float(got.abs().max())


@pytest.mark.parametrize("M,N,K,x2", [(120 * 300, 256, 256, False), (8192 + 17, 256, 128, True), (5000, 128, 64, False), (70001, 128, 4, False),
                                      (32 * 1000 + 31, 256, 64, False)])
def test_relu_derivative_as_bit_mask_is_the_float_mask(M, N, K, x2):
    """r06: a ReLU product on the split engine leaves "out > 0" as one bit per element; a masked product whose mask is that output reads
    the bits (ucn_gemm_h3_x2 relu_bits_out / mask_bits) -- EXACTLY the result of reading the fp32 mask, on ragged row counts, with the
    second operand pair in the epilogue, and after the mask went through save_for_backward; a mask rewritten since falls back."""
    from ucnerf_amd.internal import dense_f32 as D
    prev, prev_rows = D.set_engine("split"), D.H3_MIN_ROWS
    D.H3_MIN_ROWS = 4096
    try:
        g = torch.Generator(device="cuda").manual_seed(M + N + K)
        x = torch.randn(M, K, device="cuda", generator=g)
        w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
        b = torch.randn(N, device="cuda", generator=g)
        h = D.gemm(x, w, b, D.RELU)
        bits = h._ucn_relu_bits[0]
        assert bits.numel() == (M + 31) // 32 * (N // 64) * 32
        # every bit: 32-bit word ((row tile) * (N/64) + column block) * 64 + lane, bit 4 u + j <-> row 32 tile + 4 u + lane // 16, column
        # 64 block + 4 (lane % 16) + j
        words = bits.view(torch.int32).view(-1, N // 64, 64).cpu().numpy().astype(np.uint32)
        bit = np.arange(32, dtype=np.uint32)
        got = ((words[..., None] >> bit) & np.uint32(1)).astype(bool).reshape(-1, N // 64, 4, 16, 8, 4)   # [tile, cb, lane // 16, lane % 16, u, j]
        hp = torch.zeros((M + 31) // 32 * 32, N, device="cuda")
        hp[:M] = h
        want = (hp > 0).cpu().numpy().reshape(-1, 8, 4, N // 64, 16, 4)                     # [tile, u, lane // 16, cb, lane % 16, j]
        want = want.transpose(0, 3, 2, 4, 1, 5)
        assert np.array_equal(got, want)
        gy = torch.randn(M, 64, device="cuda", generator=g)
        wt = torch.randn(N, 64, device="cuda", generator=g)
        kw = {}
        if x2:
            kw = dict(x2=torch.randn(M, 4, device="cuda", generator=g), w2=torch.randn(N, 4, device="cuda", generator=g))
        with_bits = D.gemm(gy, wt, mask=h, **kw)
        saved = h.detach().clone()                      # no recorded bits: the fp32 mask
        assert not hasattr(saved, "_ucn_relu_bits")
        with_floats = D.gemm(gy, wt, mask=saved, **kw)
        assert torch.equal(with_bits, with_floats)
        assert float(with_bits._ucn_amax[0]) == float(with_floats._ucn_amax[0])
        # through autograd's saved tensors

        class Node(torch.autograd.Function):
            @staticmethod
            def forward(ctx, a):
                ctx.save_for_backward(h)
                D.stash_amax(ctx, (h,))
                return a * 1.0

            @staticmethod
            def backward(ctx, ga):
                (hh,) = ctx.saved_tensors
                D.restore_amax(ctx, (hh,))
                assert D._bits_of(hh, M, N) is not None
                Node.out = D.gemm(gy, wt, mask=hh, **kw)
                return ga

        a = torch.ones(1, device="cuda", requires_grad=True)
        Node.apply(a).sum().backward()
        assert torch.equal(Node.out, with_floats)
        # a mask rewritten in place (version bump) no longer matches its bits: the float path, with the NEW values
        h.mul_(-1.0).add_(0.01)
        assert D._bits_of(h, M, N) is None
        again = D.gemm(gy, wt, mask=h, **kw)
        assert torch.equal(again, D.gemm(gy, wt, mask=h.clone(), **kw)) and not torch.equal(again, with_floats)
    finally:
        D.set_engine(prev)
        D.H3_MIN_ROWS = prev_rows
