"""ctypes binding of oracle/grid_oracle.c (TEST INFRASTRUCTURE, see oracle/__init__.py).

Exposes the three entry points of the reference's ``_gridencoder`` pybind module
(/root/reference/nerf/gridencoder/src/bindings.cpp:5-9, gridencoder.h:12-15) with the
same positional signatures, but for float32 CPU tensors, so that the reference's own
``grid.py`` can run against it unchanged when generating golden vectors.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgrid_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "grid_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _chk(t, name, dtype=torch.float32):
    if t.device.type != "cpu":
        raise RuntimeError(f"{name} must be a CPU tensor for the oracle")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}")


def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx,
                        gridtype, align_corners, interp):
    _chk(inputs, "inputs"); _chk(embeddings, "embeddings"); _chk(outputs, "outputs")
    _chk(offsets, "offsets", torch.int32)
    if dy_dx is not None:
        _chk(dy_dx, "dy_dx")
    lib().grid_oracle_forward(_p(inputs), _p(embeddings), _p(offsets), _p(outputs),
                              ctypes.c_uint32(B), ctypes.c_uint32(D), ctypes.c_uint32(C),
                              ctypes.c_uint32(L), ctypes.c_float(float(S)), ctypes.c_uint32(H),
                              _p(dy_dx), ctypes.c_uint32(gridtype), ctypes.c_int(bool(align_corners)),
                              ctypes.c_uint32(interp))


def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H,
                         dy_dx, grad_inputs, gridtype, align_corners, interp):
    _chk(grad, "grad"); _chk(inputs, "inputs"); _chk(embeddings, "embeddings")
    _chk(grad_embeddings, "grad_embeddings"); _chk(offsets, "offsets", torch.int32)
    lib().grid_oracle_backward(_p(grad), _p(inputs), _p(embeddings), _p(offsets),
                               _p(grad_embeddings), ctypes.c_uint32(B), ctypes.c_uint32(D),
                               ctypes.c_uint32(C), ctypes.c_uint32(L), ctypes.c_float(float(S)),
                               ctypes.c_uint32(H), _p(dy_dx), _p(grad_inputs),
                               ctypes.c_uint32(gridtype), ctypes.c_int(bool(align_corners)),
                               ctypes.c_uint32(interp))


def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, C, L, S, H, gridtype,
                         align_corners):
    _chk(inputs, "inputs"); _chk(embeddings, "embeddings"); _chk(grad, "grad")
    _chk(offsets, "offsets", torch.int32)
    lib().grid_oracle_total_variation(_p(inputs), _p(embeddings), _p(grad), _p(offsets),
                                      ctypes.c_float(float(weight)), ctypes.c_uint32(B),
                                      ctypes.c_uint32(D), ctypes.c_uint32(C), ctypes.c_uint32(L),
                                      ctypes.c_float(float(S)), ctypes.c_uint32(H),
                                      ctypes.c_uint32(gridtype), ctypes.c_int(bool(align_corners)))


# ---------------------------------------------------------------- fp16 tables (grid.py:43-44 autocast policy)
def grid_encode_forward_half(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners,
                             interp):
    """kernel_grid with scalar_t = at::Half: embeddings / outputs / dy_dx are torch.float16 CPU tensors."""
    _chk(inputs, "inputs"); _chk(embeddings, "embeddings", torch.float16); _chk(outputs, "outputs", torch.float16)
    _chk(offsets, "offsets", torch.int32)
    if dy_dx is not None:
        _chk(dy_dx, "dy_dx", torch.float16)
    lib().grid_oracle_forward_h(_p(inputs), _p(embeddings), _p(offsets), _p(outputs), ctypes.c_uint32(B),
                                ctypes.c_uint32(D), ctypes.c_uint32(C), ctypes.c_uint32(L), ctypes.c_float(float(S)),
                                ctypes.c_uint32(H), _p(dy_dx), ctypes.c_uint32(gridtype),
                                ctypes.c_int(bool(align_corners)), ctypes.c_uint32(interp))


def grid_encode_backward_half(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype,
                              align_corners, interp):
    _chk(grad, "grad", torch.float16); _chk(inputs, "inputs"); _chk(grad_embeddings, "grad_embeddings", torch.float16)
    _chk(offsets, "offsets", torch.int32)
    lib().grid_oracle_backward_h(_p(grad), _p(inputs), _p(offsets), _p(grad_embeddings), ctypes.c_uint32(B),
                                 ctypes.c_uint32(D), ctypes.c_uint32(C), ctypes.c_uint32(L), ctypes.c_float(float(S)),
                                 ctypes.c_uint32(H), _p(dy_dx), _p(grad_inputs), ctypes.c_uint32(gridtype),
                                 ctypes.c_int(bool(align_corners)), ctypes.c_uint32(interp))


def half_bits_to_float(bits):
    """numpy uint16 array -> float32 through the oracle's software conversion (checked against numpy in the tests)."""
    f = lib().grid_oracle_h2f
    f.restype, f.argtypes = ctypes.c_float, [ctypes.c_uint16]
    return np.array([f(int(b)) for b in np.asarray(bits).reshape(-1)], dtype=np.float32)


def float_to_half_bits(vals):
    f = lib().grid_oracle_f2h
    f.restype, f.argtypes = ctypes.c_uint16, [ctypes.c_float]
    return np.array([f(float(v)) for v in np.asarray(vals, dtype=np.float32).reshape(-1)], dtype=np.uint16)


def level_constants(offsets, S, H):
    L = offsets.numel() - 1
    scale = np.zeros(L, np.float32)
    res = np.zeros(L, np.uint32)
    lib().grid_oracle_level_constants(_p(offsets), ctypes.c_uint32(L), ctypes.c_float(float(S)),
                                      ctypes.c_uint32(H), scale.ctypes.data_as(ctypes.c_void_p),
                                      res.ctypes.data_as(ctypes.c_void_p))
    return scale, res


def table_layout(num_levels, level_dim, base_resolution, desired_resolution, log2_hashmap_size,
                 input_dim=3, align_corners=False, per_level_scale=2.0):
    """Row layout of the embedding table (grid.py:105-106, 122-147).

    Returns (per_level_scale, offsets int32[L+1], grid_sizes int32[L], idx int64[rows])."""
    if desired_resolution is not None:
        per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
    cap = 2 ** log2_hashmap_size
    offs, sizes, off = [], [], 0
    for i in range(num_levels):
        r = int(np.ceil(base_resolution * per_level_scale ** i))
        r = r if align_corners else r + 1
        n = min(cap, r ** input_dim)
        n = int(np.ceil(n / 8) * 8)
        sizes.append(r)
        offs.append(off)
        off += n
    offs.append(off)
    offsets = torch.from_numpy(np.array(offs, dtype=np.int32))
    idx = torch.empty(off, dtype=torch.long)
    for i in range(num_levels):
        idx[offs[i]:offs[i + 1]] = i
    return per_level_scale, offsets, torch.from_numpy(np.array(sizes, dtype=np.int32)), idx


def encode(points01, embeddings, offsets, per_level_scale, base_resolution, want_jacobian=False,
           gridtype=0, align_corners=False, interp=0):
    """[B,D] in [0,1] -> [B, L*C] (the permute of grid.py:57 included)."""
    pts = points01.contiguous().float()
    B, D = pts.shape
    L = offsets.numel() - 1
    C = embeddings.shape[1]
    S = np.log2(per_level_scale)
    out = torch.empty(L, B, C)
    jac = torch.empty(B, L * D * C) if want_jacobian else None
    grid_encode_forward(pts, embeddings.contiguous(), offsets, out, B, D, C, L, S, base_resolution,
                        jac, gridtype, align_corners, interp)
    out = out.permute(1, 0, 2).reshape(B, L * C)
    return (out, jac) if want_jacobian else out
