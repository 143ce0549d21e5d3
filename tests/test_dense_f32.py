"""csrc/gemm_f32.hip (-m gpu): the fp32 training route's dense layers on hand-written fp32 MFMA kernels, against float64 matmuls
and against torch's own autograd of F.linear.  Bars: exact fp32 products + fp32 accumulation => a dot product of length K differs
from the float64 result by ~sqrt(K) ulp of its absolute-value sum."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


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
def test_gemm_f32_against_float64(M, N, K):
    from ucnerf_amd.internal import dense_f32 as D
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    scale = (x.double().abs() @ w.double().abs().t()).max().item() + 1.0
    y = D.gemm(x, w, b)
    assert float((y.double() - _ref(x, w, b)).abs().max()) <= 4e-7 * scale * max(1.0, K ** 0.5 / 4)
    y2 = D.gemm(x, w, None, D.RELU)
    assert float((y2.double() - _ref(x, w, relu=True)).abs().max()) <= 4e-7 * scale * max(1.0, K ** 0.5 / 4)
    base = torch.randn(M, N, device="cuda", generator=g)
    y3 = base.clone()
    D.gemm(x, w, b, D.ACCUMULATE, out=y3)
    assert float((y3.double() - _ref(x, w, b, base)).abs().max()) <= 4e-7 * scale * max(1.0, K ** 0.5 / 4)
    # column-slice views of wider buffers in and out (leading dimensions), the reference's concatenated layer inputs
    wide_x = torch.randn(M, K + 8, device="cuda", generator=g)
    wide_y = torch.zeros(M, N + 5, device="cuda")
    D.gemm(wide_x[:, 4:4 + K], w, b, out=wide_y[:, 2:2 + N])
    assert float((wide_y[:, 2:2 + N].double() - _ref(wide_x[:, 4:4 + K], w, b)).abs().max()) <= 4e-7 * scale * max(1.0, K ** 0.5 / 4)
    assert float(wide_y[:, :2].abs().max()) == 0 and float(wide_y[:, 2 + N:].abs().max()) == 0


@pytest.mark.parametrize("M,N,K,ldm_extra", [(8192, 256, 256, 0), (100003, 64, 256, 8), (5000, 256, 64, 0), (333, 7, 256, 1), (70000, 128, 32, 4)])
def test_gemm_f32_mask_epilogue(M, N, K, ldm_extra):
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
def test_gemm_f32_row_group_bias(M, N, K, group):
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
    assert float((got.double() - want).abs().max()) <= 4e-7 * scale * max(1.0, K ** 0.5 / 4)


@pytest.mark.parametrize("M,N,K", [(1000, 4, 256), (8191, 256, 544), (4096, 64, 32), (100000, 256, 256), (31, 12, 28), (262144, 128, 284)])
def test_wgrad_f32_against_float64_and_deterministic(M, N, K):
    from ucnerf_amd.internal import dense_f32 as D
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    gy = torch.randn(M, N, device="cuda", generator=g)
    x = torch.randn(M, K, device="cuda", generator=g)
    gw, gb = D.wgrad(gy, x, True)
    want = gy.double().t() @ x.double()
    scale = float((gy.double().abs().t() @ x.double().abs()).max())
    assert float((gw.double() - want).abs().max()) <= 2e-7 * scale * max(1.0, M ** 0.5 / 16)
    assert float((gb.double() - gy.double().sum(0)).abs().max()) <= 2e-7 * float(gy.double().abs().sum(0).max()) * max(1.0, M ** 0.5 / 16)
    gw2, gb2 = D.wgrad(gy, x, True)
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)                  # fixed-order partial sums


@pytest.mark.parametrize("lead,N,K,relu,bias", [((1024, 120), 256, 256, True, True), ((3000,), 3, 256, False, True), ((210,), 256, 4, True, True),
                                                ((64, 128), 1, 256, False, True), ((8192,), 256, 27, False, False), ((500, 7), 128, 283, True, True)])
def test_hip_linear_is_f_linear_with_the_same_gradients(lead, N, K, relu, bias):
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
