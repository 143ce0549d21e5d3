"""The hash-grid oracle (C) against an independent numpy restatement of gridencoder.cu.

The reference's kernel is CUDA-only and untested upstream ("parity unpinned"); this pins the
restatement against itself twice over, on random, cell-boundary, domain-edge and out-of-range
inputs, in dense and hashed levels, for every (D, C) the reference dispatches on.
"""
import numpy as np
import pytest
import torch

from oracle import grid_cpu, grid_numpy


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
