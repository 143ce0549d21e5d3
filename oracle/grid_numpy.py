"""Second, independent restatement of the reference's hash-grid kernel (vectorised numpy).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Written from
/root/reference/nerf/gridencoder/src/gridencoder.cu:50-245 without looking at
grid_oracle.c's control flow, so that an indexing mistake in one shows up as a
disagreement with the other (tests/test_oracle_grid.py demands bit equality).

The CUDA source's contracted multiply-adds (nvcc -fmad=true) are emulated with a float64
product-sum rounded once to float32: the product of two float32 is exact in float64, and
the float64 sum is within half a float64 ulp of exact, so the result equals fmaf() except
in ~2^-29 of cases (double rounding) -- none occur in the seeded tests.
"""
import numpy as np

PRIMES = np.array([1, 2654435761, 805459861, 3674653429, 2097192037, 1434869437, 2165219737],
                  dtype=np.uint32)


def _fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def level_geometry(offsets, S, H):
    L = len(offsets) - 1
    lv = np.arange(L, dtype=np.uint32).astype(np.float32) * np.float32(S)
    scale = (np.exp2(lv).astype(np.float32) * np.float32(H) - np.float32(1.0)).astype(np.float32)
    res = np.ceil(scale).astype(np.uint32) + np.uint32(1)
    rows = (np.asarray(offsets[1:]) - np.asarray(offsets[:-1])).astype(np.uint32)
    return scale, res, rows


def rows_of(cells, rows, res, gridtype=0, align_corners=False):
    """cells uint32 [...,D] -> table row (gridencoder.cu:66-84)."""
    D = cells.shape[-1]
    side = np.uint32(res if align_corners else res + 1)
    stride = np.uint32(1)
    dense = np.zeros(cells.shape[:-1], np.uint32)
    d = 0
    while d < D and stride <= rows:
        dense = dense + cells[..., d] * stride           # uint32 wrap-around arithmetic
        stride = np.uint32((int(stride) * int(side)) & 0xFFFFFFFF)
        d += 1
    if gridtype == 0 and stride > rows:
        h = np.zeros(cells.shape[:-1], np.uint32)
        for k in range(D):
            h ^= cells[..., k] * PRIMES[k]
        dense = h
    return dense % np.uint32(rows)


def forward(x, table, offsets, S, H, gridtype=0, align_corners=False, interp=0):
    """x float32 [B,D] in [0,1]; table float32 [rows_total,C] -> [L,B,C] (gridencoder.cu:87-199)."""
    x = np.asarray(x, np.float32)
    B, D = x.shape
    C = table.shape[1]
    scale, res, rows = level_geometry(offsets, S, H)
    L = len(scale)
    out = np.zeros((L, B, C), np.float32)
    oob = ((x < 0) | (x > 1)).any(axis=1)
    with np.errstate(over='ignore'):
        for l in range(L):
            half = np.float32(0.0 if align_corners else 0.5)
            p = _fma(x, np.broadcast_to(scale[l], x.shape), np.broadcast_to(half, x.shape))
            cell = np.floor(p).astype(np.uint32)
            f = (p - cell.astype(np.float32)).astype(np.float32)
            if interp == 1:
                f = (f * f * (np.float32(3.0) - np.float32(2.0) * f)).astype(np.float32)
            acc = np.zeros((B, C), np.float32)
            tab = table[offsets[l]:offsets[l + 1]]
            for k in range(1 << D):
                w = np.ones(B, np.float32)
                corner = cell.copy()
                for d in range(D):
                    if k & (1 << d):
                        w = (w * f[:, d]).astype(np.float32)
                        corner[:, d] += np.uint32(1)
                    else:
                        w = (w * (np.float32(1.0) - f[:, d])).astype(np.float32)
                r = rows_of(corner, rows[l], res[l], gridtype, align_corners)
                acc = _fma(w[:, None], tab[r], acc)
            acc[oob] = 0
            out[l] = acc
    return out


def backward_table(grad, x, offsets, S, H, n_rows, gridtype=0, align_corners=False, interp=0):
    """d(loss)/d(table) in float64 accumulation (order-free comparison target for
    gridencoder.cu:248-340).  grad [L,B,C]."""
    x = np.asarray(x, np.float32)
    B, D = x.shape
    L, _, C = grad.shape
    scale, res, rows = level_geometry(offsets, S, H)
    g = np.zeros((n_rows, C), np.float64)
    ok = ~((x < 0) | (x > 1)).any(axis=1)
    with np.errstate(over='ignore'):
        for l in range(L):
            half = np.float32(0.0 if align_corners else 0.5)
            p = _fma(x, np.broadcast_to(scale[l], x.shape), np.broadcast_to(half, x.shape))
            cell = np.floor(p).astype(np.uint32)
            f = (p - cell.astype(np.float32)).astype(np.float32)
            if interp == 1:
                f = (f * f * (np.float32(3.0) - np.float32(2.0) * f)).astype(np.float32)
            for k in range(1 << D):
                w = np.ones(B, np.float32)
                corner = cell.copy()
                for d in range(D):
                    if k & (1 << d):
                        w = (w * f[:, d]).astype(np.float32)
                        corner[:, d] += np.uint32(1)
                    else:
                        w = (w * (np.float32(1.0) - f[:, d])).astype(np.float32)
                r = rows_of(corner, rows[l], res[l], gridtype, align_corners).astype(np.int64) + int(offsets[l])
                contrib = (w[:, None] * grad[l]).astype(np.float32).astype(np.float64)
                np.add.at(g, r[ok], contrib[ok])
    return g


def _h(a):
    """round float32 -> binary16 -> float32 (numpy's conversion is IEEE round-to-nearest-even)"""
    with np.errstate(over='ignore'):
        return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def forward_half(x, table16, offsets, S, H, gridtype=0, align_corners=False, interp=0):
    """kernel_grid with scalar_t = at::Half (grid.py:43-44 under autocast): table16 float16 [rows_total,C] ->
    float16 [L,B,C].  c10::Half has no fused or mixed accumulate: `results += w * grid[i]` (gridencoder.cu:187)
    rounds the float product to half, adds in float, rounds again (torch/headeronly/util/Half.h)."""
    x = np.asarray(x, np.float32)
    B, D = x.shape
    C = table16.shape[1]
    scale, res, rows = level_geometry(offsets, S, H)
    L = len(scale)
    out = np.zeros((L, B, C), np.float16)
    oob = ((x < 0) | (x > 1)).any(axis=1)
    with np.errstate(over='ignore'):
        for l in range(L):
            half = np.float32(0.0 if align_corners else 0.5)
            p = _fma(x, np.broadcast_to(scale[l], x.shape), np.broadcast_to(half, x.shape))
            cell = np.floor(p).astype(np.uint32)
            f = (p - cell.astype(np.float32)).astype(np.float32)
            if interp == 1:
                f = (f * f * (np.float32(3.0) - np.float32(2.0) * f)).astype(np.float32)
            acc = np.zeros((B, C), np.float32)                       # always holds a half-representable value
            tab = table16[offsets[l]:offsets[l + 1]].astype(np.float32)
            for k in range(1 << D):
                w = np.ones(B, np.float32)
                corner = cell.copy()
                for d in range(D):
                    if k & (1 << d):
                        w = (w * f[:, d]).astype(np.float32)
                        corner[:, d] += np.uint32(1)
                    else:
                        w = (w * (np.float32(1.0) - f[:, d])).astype(np.float32)
                r = rows_of(corner, rows[l], res[l], gridtype, align_corners)
                prod = _h((w[:, None] * tab[r]).astype(np.float32))
                acc = _h((acc + prod).astype(np.float32))
            acc[oob] = 0
            out[l] = acc.astype(np.float16)
    return out
