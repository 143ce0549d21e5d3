"""The reference's `_gridencoder` extension module, re-expressed over the C ABI.

ref: /root/reference/nerf/gridencoder/src/bindings.cpp:5-9, gridencoder.h:12-15.
Same three functions, same positional arguments, same preconditions -> RuntimeError
(gridencoder.cu:15-18, 449-465, 474-496).  `import _gridencoder` resolves here through
ucnerf_amd/compat/_gridencoder.py, so the reference's own grid.py runs on this unchanged.
"""
import numpy as np
import torch

from .. import _lib

_F = (torch.float32, torch.float16, torch.float64)


def _checks(**tensors):
    for name, t in tensors.items():
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor")
        if not t.is_contiguous():
            raise RuntimeError(f"{name} must be a contiguous tensor")
        if name == "offsets":
            if t.dtype != torch.int32:
                raise RuntimeError("offsets must be an int tensor")
        elif t.dtype not in _F:
            raise RuntimeError(f"{name} must be a floating tensor")


def host_offsets(offsets):
    """Host copy of a level-offsets tensor (the C ABI takes level offsets as HOST metadata).

    Cached ON the tensor object together with the tensor's version counter: the copy dies with the tensor and
    an in-place update (load_state_dict) invalidates it.  (A cache keyed on data_ptr would hand a freed
    encoder's offsets to the next tensor the caching allocator places at that address.)"""
    hit = getattr(offsets, "_ucn_host_offsets", None)
    if hit is None or hit[0] != offsets._version:
        hit = (offsets._version, np.ascontiguousarray(offsets.detach().cpu().numpy().astype(np.int32)))
        try:
            offsets._ucn_host_offsets = hit
        except AttributeError:          # a tensor subclass without __dict__: fetch every time
            pass
    return hit[1]


def _dtype_code(t):
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.float16:
        return 1
    raise RuntimeError("embeddings must be float32 or float16 on this build (float64 tables are not supported)")


def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners,
                        interp):
    _checks(inputs=inputs, embeddings=embeddings, offsets=offsets, outputs=outputs, dy_dx=dy_dx)
    if inputs.dtype != torch.float32:
        raise RuntimeError("inputs must be float32 (gridencoder.cu:469 reads them as float)")
    if outputs.dtype != embeddings.dtype:
        raise RuntimeError("outputs must have the embeddings' dtype")
    lib = _lib.load()
    off = host_offsets(offsets)
    _lib.check(lib.ucn_grid_encode_forward(inputs.data_ptr(), embeddings.data_ptr(), off.ctypes.data, outputs.data_ptr(),
                                           B, D, C, L, float(S), H, _lib.ptr(dy_dx), gridtype, int(bool(align_corners)),
                                           interp, _dtype_code(embeddings), _lib.stream()))


def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs,
                         gridtype, align_corners, interp):
    _checks(grad=grad, inputs=inputs, embeddings=embeddings, offsets=offsets, grad_embeddings=grad_embeddings,
            dy_dx=dy_dx, grad_inputs=grad_inputs)
    if grad.dtype != grad_embeddings.dtype:
        raise RuntimeError("grad and grad_embeddings must have the same dtype")
    lib = _lib.load()
    off = host_offsets(offsets)
    _lib.check(lib.ucn_grid_encode_backward(grad.data_ptr(), inputs.data_ptr(), embeddings.data_ptr(), off.ctypes.data,
                                            grad_embeddings.data_ptr(), B, D, C, L, float(S), H, _lib.ptr(dy_dx),
                                            _lib.ptr(grad_inputs), gridtype, int(bool(align_corners)), interp,
                                            _dtype_code(grad), _lib.stream()))


def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, C, L, S, H, gridtype, align_corners):
    _checks(inputs=inputs, embeddings=embeddings, grad=grad, offsets=offsets)
    if embeddings.dtype != torch.float32 or inputs.dtype != torch.float32 or grad.dtype != torch.float32:
        raise RuntimeError("grad_total_variation runs in float32 (grid.py:176-177 disables autocast)")
    lib = _lib.load()
    off = host_offsets(offsets)
    _lib.check(lib.ucn_grad_total_variation(inputs.data_ptr(), embeddings.data_ptr(), grad.data_ptr(), off.ctypes.data,
                                            float(weight), B, D, C, L, float(S), H, gridtype, int(bool(align_corners)),
                                            _lib.stream()))
