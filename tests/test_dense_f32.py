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
                                   (5, 1, 4), (65536, 256, 64)])
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
