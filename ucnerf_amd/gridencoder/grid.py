"""GridEncoder: multi-resolution hash-grid module (host side of the HIP operator).

Interface mirror of /root/reference/nerf/gridencoder/grid.py:24-198 -- same constructor
arguments, buffers (`offsets`, `idx`, `grid_sizes`), parameter (`embeddings`), forward(inputs,
bound) and grad_total_variation -- so checkpoints and callers are interchangeable.  The table
layout math (grid.py:105-147) is metadata, computed on the host exactly as upstream; all tensor
arithmetic happens in ucnerf_amd/csrc/grid_op.hip.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _backend

_GRIDTYPE = {"hash": 0, "tiled": 1}
_INTERP = {"linear": 0, "smoothstep": 1}


class _GridEncode(Function):
    """autograd bridge; ref grid.py:24-89.  Output layout [B, L*C] like upstream (the level-major
    kernel output is exposed to the fused ray-march directly, without this permute)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False,
                gridtype=0, align_corners=False, interpolation=0):
        inputs = inputs.contiguous()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = np.log2(per_level_scale)
        H = base_resolution
        # upstream policy (grid.py:43-44): under autocast the table is read as fp16 when C is even
        if torch.is_autocast_enabled() and C % 2 == 0:
            embeddings = embeddings.to(torch.half)
        outputs = torch.empty(L, B, C, device=inputs.device, dtype=embeddings.dtype)
        dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=embeddings.dtype) if calc_grad_inputs else None
        _backend.grid_encode_forward(inputs, embeddings.contiguous(), offsets, outputs, B, D, C, L, S, H, dy_dx,
                                     gridtype, align_corners, interpolation)
        outputs = outputs.permute(1, 0, 2).reshape(B, L * C)
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = [B, D, C, L, S, H, gridtype, interpolation]
        ctx.align_corners = align_corners
        return outputs

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype, interpolation = ctx.dims
        grad = grad.view(B, L, C).permute(1, 0, 2).contiguous()
        grad_embeddings = torch.zeros_like(embeddings)
        grad_inputs = torch.zeros_like(inputs, dtype=embeddings.dtype) if dy_dx is not None else None
        _backend.grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx,
                                      grad_inputs, gridtype, ctx.align_corners, interpolation)
        if dy_dx is not None:
            grad_inputs = grad_inputs.to(inputs.dtype)
        return grad_inputs, grad_embeddings, None, None, None, None, None, None, None


grid_encode = _GridEncode.apply


class GridEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, gridtype="hash", align_corners=False,
                 interpolation="linear", init_std=1e-4):
        super().__init__()
        if desired_resolution is not None:
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype = gridtype
        self.gridtype_id = _GRIDTYPE[gridtype]
        self.interpolation = interpolation
        self.interp_id = _INTERP[interpolation]
        self.align_corners = align_corners
        self.init_std = init_std
        self.max_params = 2 ** log2_hashmap_size
        # rows per level: min(T, side^D) rounded up to a multiple of 8 (grid.py:127-135)
        sides, starts, row = [], [], 0
        for lvl in range(num_levels):
            side = int(np.ceil(base_resolution * per_level_scale ** lvl))
            side = side if align_corners else side + 1
            rows = int(np.ceil(min(self.max_params, side ** input_dim) / 8) * 8)
            sides.append(side)
            starts.append(row)
            row += rows
        starts.append(row)
        offsets = torch.from_numpy(np.array(starts, dtype=np.int32))
        self.register_buffer("offsets", offsets)
        self.register_buffer("idx", torch.repeat_interleave(torch.arange(num_levels), (offsets[1:] - offsets[:-1]).long()))
        self.register_buffer("grid_sizes", torch.from_numpy(np.array(sides, dtype=np.int32)))
        self.n_params = offsets[-1] * level_dim
        self.embeddings = nn.Parameter(torch.empty(row, level_dim))
        self.reset_parameters()
        # host copies of the metadata the C ABI wants on the host
        self._offsets_np = np.ascontiguousarray(np.array(starts, dtype=np.int32))
        self._sizes_np = np.ascontiguousarray(np.array(sides, dtype=np.int32))

    def reset_parameters(self):
        self.embeddings.data.uniform_(-self.init_std, self.init_std)

    def extra_repr(self):
        # (nn.Module's own hook: `print(model)` lists the table's geometry; nothing reads this string)
        finest = int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))
        return (f"{self.input_dim}-d points, {self.num_levels} levels x {self.level_dim} features, side {self.base_resolution} .. {finest}, "
                f"{self.embeddings.shape[0]:,} rows, {self.gridtype} / {self.interpolation}" + (", align_corners" if self.align_corners else ""))

    def forward(self, inputs, bound=1):
        inputs = (inputs + bound) / (2 * bound)
        prefix = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        out = grid_encode(inputs, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution,
                          inputs.requires_grad, self.gridtype_id, self.align_corners, self.interp_id)
        return out.view(prefix + [self.output_dim])

    @torch.amp.autocast("cuda", enabled=False)
    def grad_total_variation(self, weight=1e-7, inputs=None, bound=1, B=1000000):
        D, C, L = self.input_dim, self.embeddings.shape[1], self.offsets.shape[0] - 1
        S, H = np.log2(self.per_level_scale), self.base_resolution
        if inputs is None:
            inputs = torch.rand(B, self.input_dim, device=self.embeddings.device)
        else:
            inputs = ((inputs + bound) / (2 * bound)).view(-1, self.input_dim)
            B = inputs.shape[0]
        if self.embeddings.grad is None:
            raise ValueError("grad is None, should be called after loss.backward() and before optimizer.step()!")
        _backend.grad_total_variation(inputs.contiguous(), self.embeddings, self.embeddings.grad, self.offsets, weight,
                                      B, D, C, L, S, H, self.gridtype_id, self.align_corners)
