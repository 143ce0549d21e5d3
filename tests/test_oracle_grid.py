"""The hash-grid oracle (C) against an independent numpy restatement of gridencoder.cu.

The reference's kernel is CUDA-only and untested upstream ("parity unpinned"); this pins the
restatement against itself twice over, on random, cell-boundary, domain-edge and out-of-range
inputs, in dense and hashed levels, for every (D, C) the reference dispatches on.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import grid_cpu, grid_numpy

HERE = os.path.dirname(os.path.abspath(__file__))


def layout(L, C, T, desired, D=3):
    pls, offsets, sizes, idx = grid_cpu.table_layout(L, C, 16, desired, T, input_dim=D)
    return pls, offsets, sizes


def hard_points(B, D, rng, scale_hint=4095.0):
    x = rng.random((B, D), dtype=np.float32)
    x[0] = 0.0
    x[1] = 1.0
    x[2] = np.nextafter(np.float32(1.0), np.float32(2.0))        # just outside -> zeros
    x[3] = -1e-7                                                   # just outside -> zeros
    x[4, 0] = np.nan if False else 0.5
    # exact lattice positions of a fine level: (k - 0.5) / scale
    k = rng.integers(1, 4000, size=(64, D)).astype(np.float32)
    x[5:69] = (k - np.float32(0.5)) / np.float32(scale_hint)
    x[69:80] = rng.random((11, D), dtype=np.float32) * 3 - 1      # mostly out of range
    return x


@pytest.mark.parametrize("D,C", [(3, 2), (3, 4), (2, 2), (3, 1), (3, 8), (4, 2)])
@pytest.mark.parametrize("interp", [0, 1])
def test_forward_bitexact_vs_numpy(D, C, interp):
    rng = np.random.default_rng(100 + D * 10 + C)
    L, T = 9, 12
    pls, offsets, sizes = layout(L, C, T, 4096, D)
    table = (rng.random((int(offsets[-1]), C), dtype=np.float32) * 2 - 1)
    x = hard_points(600, D, rng)
    S = np.log2(pls)
    out = torch.empty(L, len(x), C)
    grid_cpu.grid_encode_forward(torch.from_numpy(x), torch.from_numpy(table), offsets, out, len(x), D, C, L,
                                 S, 16, None, 0, False, interp)
    ref = grid_numpy.forward(x, table, offsets.numpy(), S, 16, interp=interp)
    assert np.array_equal(out.numpy(), ref), float(np.abs(out.numpy() - ref).max())
    # out-of-range rows are exactly zero on every level (gridencoder.cu:110-135)
    assert (out[:, 2] == 0).all() and (out[:, 3] == 0).all()
    assert out.abs().sum() > 0


def test_levels_switch_from_dense_to_hash():
    pls, offsets, sizes = layout(16, 2, 19, 524288)
    scale, res = grid_cpu.level_constants(offsets, np.log2(pls), 16)
    assert pls == 2.0
    assert list(sizes[:4]) == [17, 33, 65, 129] and int(sizes[-1]) == 524289
    assert np.array_equal(scale, (16 * 2.0 ** np.arange(16) - 1).astype(np.float32))
    assert np.array_equal(res, (16 * 2 ** np.arange(16)).astype(np.uint32))
    rows = (offsets[1:] - offsets[:-1]).numpy()
    assert rows[0] == 4920 and rows[1] == 35944 and (rows[3:] == 2 ** 19).all()
    assert int(offsets[-1]) == 7131240                       # B nerf table rows
    # waymo.gin nerf table (Appendix B.5)
    _, off_r, _ = layout(10, 4, 21, 8192)
    assert int(off_r[-1]) == 14995560 and list(off_r[:5]) == [0, 4920, 40864, 315496, 2412648]


def test_backward_matches_float64_scatter():
    rng = np.random.default_rng(7)
    L, C, T, D = 8, 2, 11, 3
    pls, offsets, sizes = layout(L, C, T, 2048)
    B = 500
    x = hard_points(B, D, rng)
    table = rng.random((int(offsets[-1]), C), dtype=np.float32)
    grad = rng.standard_normal((L, B, C)).astype(np.float32)
    g = torch.zeros(int(offsets[-1]), C)
    grid_cpu.grid_encode_backward(torch.from_numpy(grad), torch.from_numpy(x), torch.from_numpy(table), offsets, g,
                                  B, D, C, L, np.log2(pls), 16, None, None, 0, False, 0)
    ref = grid_numpy.backward_table(grad, x, offsets.numpy(), np.log2(pls), 16, int(offsets[-1]))
    # float32 sequential accumulation vs float64: a few ulp of the largest partial sums
    assert np.abs(g.numpy() - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    assert np.abs(ref).sum() > 0


def test_input_jacobian_matches_finite_difference():
    rng = np.random.default_rng(9)
    L, C, T, D = 5, 2, 14, 3
    pls, offsets, sizes = layout(L, C, T, 256)
    table = torch.from_numpy(rng.random((int(offsets[-1]), C), dtype=np.float32))
    x = torch.from_numpy(rng.random((50, D), dtype=np.float32) * 0.9 + 0.05)
    out, jac = grid_cpu.encode(x, table, offsets, pls, 16, want_jacobian=True)
    jac = jac.reshape(50, L, D, C)
    h = 1e-4
    for d in range(D):
        xp = x.clone(); xp[:, d] += h
        xm = x.clone(); xm[:, d] -= h
        fd = (grid_cpu.encode(xp, table, offsets, pls, 16) - grid_cpu.encode(xm, table, offsets, pls, 16)) / (2 * h)
        fd = fd.reshape(50, L, C)
        # piecewise-linear field: FD is exact unless the +-h stencil straddles a cell face
        close = (fd - jac[:, :, d, :]).abs() <= 2e-2 * jac[:, :, d, :].abs().clamp_min(1.0)
        assert close.float().mean() > 0.9
    # grad_inputs = sum_l,c grad * dy_dx   (gridencoder.cu:343-369)
    grad = torch.from_numpy(rng.standard_normal((L, 50, C)).astype(np.float32))
    gi = torch.zeros(50, D)
    grid_cpu.grid_encode_backward(grad, x.contiguous(), table, offsets, torch.zeros_like(table), 50, D, C, L,
                                  np.log2(pls), 16, jac.reshape(50, -1).contiguous(), gi, 0, False, 0)
    want = torch.einsum('lbc,bldc->bd', grad, jac)
    assert (gi - want).abs().max() < 1e-3 * want.abs().max()


def test_total_variation_gradient_is_finite_and_local():
    rng = np.random.default_rng(11)
    L, C, T, D = 4, 2, 14, 3
    pls, offsets, sizes = layout(L, C, T, 128)
    table = torch.from_numpy(rng.random((int(offsets[-1]), C), dtype=np.float32))
    x = torch.from_numpy(rng.random((200, D), dtype=np.float32))
    g = torch.zeros_like(table)
    grid_cpu.grad_total_variation(x, table, g, offsets, 1e-2, 200, D, C, L, np.log2(pls), 16, 0, False)
    assert torch.isfinite(g).all() and (g != 0).sum() > 0
    assert (g != 0).any(dim=1).sum() <= 200 * L            # one centre row per (point, level)
    # a constant table has zero variation => zero gradient
    g2 = torch.zeros_like(table)
    grid_cpu.grad_total_variation(x, torch.ones_like(table), g2, offsets, 1e-2, 200, D, C, L, np.log2(pls), 16, 0, False)
    assert (g2 == 0).all()


# ------------------------------------------------------------------ the op's other dispatch axes
@pytest.mark.parametrize("D,C,gridtype,align", [(3, 2, 1, False), (3, 2, 0, True), (3, 4, 1, True), (5, 2, 0, False),
                                                (5, 1, 1, False), (2, 8, 1, True), (4, 4, 0, True)])
def test_forward_tiled_aligned_and_5d_bitexact_vs_numpy(D, C, gridtype, align):
    """gridtype 1 ('tiled': dense index modulo the table, gridencoder.cu:78-82), align_corners (no +0.5, side =
    resolution, :70,:148) and D = 5 (:381-385): the C oracle and the numpy restatement must agree bit for bit."""
    rng = np.random.default_rng(300 + D * 10 + C + gridtype)
    L, T = 6, 11
    pls, offsets, sizes, _ = grid_cpu.table_layout(L, C, 16, 512, T, input_dim=D, align_corners=align)
    table = (rng.random((int(offsets[-1]), C), dtype=np.float32) * 2 - 1)
    x = hard_points(400, D, rng, scale_hint=511.0)
    S = np.log2(pls)
    out = torch.empty(L, len(x), C)
    grid_cpu.grid_encode_forward(torch.from_numpy(x), torch.from_numpy(table), offsets, out, len(x), D, C, L, S, 16,
                                 None, gridtype, align, 0)
    ref = grid_numpy.forward(x, table, offsets.numpy(), S, 16, gridtype=gridtype, align_corners=align)
    assert np.array_equal(out.numpy(), ref), float(np.abs(out.numpy() - ref).max())
    assert out.abs().sum() > 0


# ------------------------------------------------------------------ fp16 tables (grid.py:43-44)
def test_software_half_conversion_is_ieee():
    bits = np.arange(65536, dtype=np.uint16)
    f = grid_cpu.half_bits_to_float(bits)
    ref = bits.view(np.float16).astype(np.float32)
    fin = ~np.isnan(ref)
    assert np.array_equal(f.view(np.uint32)[fin], ref.view(np.uint32)[fin]) and np.isnan(f[~fin]).all()
    rng = np.random.default_rng(0)
    mags = rng.choice(np.array([1e-8, 1e-5, 1e-3, 1, 100, 6e4, 1e6], np.float32), 50000)
    halves = bits[:0x7c00].view(np.float16).astype(np.float64)
    v = np.concatenate([rng.standard_normal(50000).astype(np.float32) * mags,
                        ((halves[:-1] + halves[1:]) / 2).astype(np.float32),                  # every rounding tie
                        np.array([65504, 65519.99, 65520, 65536, np.inf, 0, 2.0 ** -25, 2.0 ** -25 * 1.0000001,
                                  2.0 ** -24, 6.1e-5, 6.097e-5], np.float32)])
    v = np.concatenate([v, -v])
    with np.errstate(over='ignore'):
        want = v.astype(np.float16).view(np.uint16)
    assert np.array_equal(grid_cpu.float_to_half_bits(v), want)


def _c10_half_pin(tmp_path):
    inc = os.path.join(os.path.dirname(torch.__file__), "include")
    if not os.path.exists(os.path.join(inc, "torch", "headeronly", "util", "Half.h")):
        pytest.skip("torch/headeronly/util/Half.h not shipped with this torch")
    so = str(tmp_path / "libhalfpin.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-I" + inc, "-o", so,
                           os.path.join(HERE, "pins", "half_semantics.cpp")])
    lib = ctypes.CDLL(so)
    u16, f32 = ctypes.c_uint16, ctypes.c_float
    lib.pin_forward_step.restype, lib.pin_forward_step.argtypes = u16, [u16, f32, u16]
    lib.pin_jacobian_step.restype, lib.pin_jacobian_step.argtypes = u16, [u16, f32, u16, u16, f32]
    lib.pin_input_backward_step.restype, lib.pin_input_backward_step.argtypes = u16, [u16, u16, u16]
    lib.pin_table_backward_step.restype, lib.pin_table_backward_step.argtypes = u16, [u16, f32, u16]
    return lib


def test_half_arithmetic_of_the_reference_is_pinned_by_the_c10_header(tmp_path):
    """Expressions of the reference's shape, evaluated with the real c10::Half operator set, against the
    roundings oracle/grid_oracle.c spells out (header comment there)."""
    pin = _c10_half_pin(tmp_path)
    rng = np.random.default_rng(5)
    h = lambda a: np.float32(np.float16(np.float32(a)))
    n = 4000
    hb = lambda: rng.standard_normal(n).astype(np.float16).view(np.uint16)
    acc, g, r, l = hb(), hb(), hb(), hb()
    w = rng.random(n, dtype=np.float32)
    pd = (rng.random(n, dtype=np.float32) * 1.5).astype(np.float32)
    f = lambda bits: bits.view(np.float16).astype(np.float32)
    with np.errstate(over='ignore'):
        for i in range(n):
            a, gi, ri, li, wi, pi = f(acc[i:i + 1])[0], f(g[i:i + 1])[0], f(r[i:i + 1])[0], f(l[i:i + 1])[0], w[i], pd[i]
            want = np.float16(a + h(wi * gi)).view(np.uint16)
            assert pin.pin_forward_step(int(acc[i]), float(wi), int(g[i])) == want
            want = np.float16(a + h(np.float32(wi * h(ri - li)) * pi)).view(np.uint16)
            assert pin.pin_jacobian_step(int(acc[i]), float(wi), int(r[i]), int(l[i]), float(pi)) == want
            want = np.float16(a + h(gi * ri)).view(np.uint16)
            assert pin.pin_input_backward_step(int(acc[i]), int(g[i]), int(r[i])) == want
            want = np.float16(a + h(wi * gi)).view(np.uint16)
            assert pin.pin_table_backward_step(int(acc[i]), float(wi), int(g[i])) == want


@pytest.mark.parametrize("D,C,gridtype,align,interp", [(3, 2, 0, False, 0), (3, 4, 0, False, 1), (3, 2, 1, True, 0),
                                                       (2, 8, 0, False, 0), (5, 2, 0, False, 0)])
def test_forward_half_bitexact_vs_numpy(D, C, gridtype, align, interp):
    rng = np.random.default_rng(400 + D * 10 + C)
    L, T = 7, 11
    pls, offsets, sizes, _ = grid_cpu.table_layout(L, C, 16, 1024, T, input_dim=D, align_corners=align)
    table = (rng.random((int(offsets[-1]), C), dtype=np.float32) * 2 - 1).astype(np.float16)
    x = hard_points(300, D, rng, scale_hint=1023.0)
    S = np.log2(pls)
    out = torch.empty(L, len(x), C, dtype=torch.float16)
    jac = torch.empty(len(x), L * D * C, dtype=torch.float16)
    grid_cpu.grid_encode_forward_half(torch.from_numpy(x), torch.from_numpy(table), offsets, out, len(x), D, C, L, S, 16,
                                      jac, gridtype, align, interp)
    ref = grid_numpy.forward_half(x, table, offsets.numpy(), S, 16, gridtype=gridtype, align_corners=align, interp=interp)
    assert np.array_equal(out.numpy().view(np.uint16), ref.view(np.uint16))
    # and it is the fp32 result to half precision (8 corner sums of O(1) values, each rounded twice)
    want32 = torch.empty(L, len(x), C)
    grid_cpu.grid_encode_forward(torch.from_numpy(x), torch.from_numpy(table.astype(np.float32)), offsets, want32, len(x),
                                 D, C, L, S, 16, None, gridtype, align, interp)
    assert (out.float() - want32).abs().max() <= (1 << D) * 2 ** -10
    assert torch.isfinite(jac.float()).all()


def test_backward_half_close_to_float64_scatter():
    rng = np.random.default_rng(17)
    L, C, T, D = 6, 2, 11, 3
    pls, offsets, sizes = layout(L, C, T, 512)
    B = 300
    x = hard_points(B, D, rng, scale_hint=511.0)
    grad = (rng.standard_normal((L, B, C)) * 0.1).astype(np.float16)
    g = torch.zeros(int(offsets[-1]), C, dtype=torch.float16)
    grid_cpu.grid_encode_backward_half(torch.from_numpy(grad), torch.from_numpy(x), offsets, g, B, D, C, L, np.log2(pls),
                                       16, None, None, 0, False, 0)
    ref = grid_numpy.backward_table(grad.astype(np.float32), x, offsets.numpy(), np.log2(pls), 16, int(offsets[-1]))
    # every addend and every partial sum is rounded to half: a few half ulps of the largest row sums
    assert np.abs(g.float().numpy() - ref).max() <= 2e-2 * max(1.0, np.abs(ref).max())
    assert np.abs(ref).sum() > 0
