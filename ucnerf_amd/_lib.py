"""ctypes loader for libucnerf_march.so (the C ABI of include/ucnerf_march.h).

The HIP library is the product: if it is missing or stale this module raises -- there is no
CPU or eager-PyTorch fallback anywhere in ucnerf_amd.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# UCN_LIB_PATH: an experiment build of the same ABI (tools/build_variant.sh) for A/B measurements; default = the in-tree product
LIB_PATH = os.environ.get("UCN_LIB_PATH") or os.path.join(_HERE, "csrc", "libucnerf_march.so")
ABI_VERSION = 26
LAUNCH_CORESIDENT = 0x100
TABLE_F16 = 0x200
RAYS_INCOHERENT = 0x1000   # ucn_march_features layout flag: random (training) rays -> lane-paired fetch on every hashed level
GFEAT_LEVEL_MAJOR4 = 0x20000  # ... for level_dim 4: [F / 4][M][4], / 6
GFEAT_LEVEL_MAJOR = 0x10000  # ucn_train_bwd F flag: gfeat as [F / 2][M][2], / 6 = ucn_march_features_backward's layout 4 (include/ucnerf_march.h)
BWD_FIXED_POINT = 0x800    # ucn_march_features_backward layout flag: int32 fixed-point row blocks (include/ucnerf_march.h UCN_BWD_FIXED_POINT)
FEATURES_BF16 = 0x400      # ucn_march_features layout flag: features as [L][B] bf16 pairs (half tables, level_dim 2)
FEAT_BF16 = 0x100          # ucn_train_fwd feat_level_dim flag: the features are those pairs          # include/ucnerf_march.h UCN_LAUNCH_CORESIDENT

c_u32, c_u64, c_i32, c_f32, c_vp = ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p


class UcnField(ctypes.Structure):
    """struct ucn_field (include/ucnerf_march.h)."""
    _fields_ = [
        ("embeddings", c_vp), ("offsets_host", c_vp), ("grid_sizes_host", c_vp),
        ("num_levels", c_u32), ("level_dim", c_u32), ("base_resolution", c_u32),
        ("log2_per_level_scale", c_f32),
        ("w_d0", c_vp), ("b_d0", c_vp), ("w_d1", c_vp), ("b_d1", c_vp),
        ("n_bottleneck", c_u32),
        ("w_c0", c_vp), ("b_c0", c_vp), ("w_c1", c_vp), ("b_c1", c_vp), ("w_rgb", c_vp), ("b_rgb", c_vp),
        ("n_width", c_u32), ("n_dir", c_u32),
        ("density_bias", c_f32), ("rgb_premultiplier", c_f32), ("rgb_bias", c_f32), ("rgb_padding", c_f32),
        ("packed", c_vp),
        ("mlp_mode", c_u32),
    ]


class UcnSky(ctypes.Structure):
    """struct ucn_sky (include/ucnerf_march.h)."""
    _fields_ = [
        ("w_pts", c_vp * 8), ("b_pts", c_vp * 8),
        ("w_alpha", c_vp), ("b_alpha", c_vp), ("w_feat", c_vp), ("b_feat", c_vp),
        ("w_view", c_vp), ("b_view", c_vp), ("w_rgb", c_vp), ("b_rgb", c_vp),
        ("packed", c_vp),
    ]


class UcnSkyTrain(ctypes.Structure):
    """struct ucn_sky_train (include/ucnerf_march.h)."""
    _fields_ = [
        ("w_pts", c_vp * 8), ("b_pts", c_vp * 8), ("m5", c_vp), ("mv", c_vp),
        ("w_alpha", c_vp), ("b_alpha", c_vp), ("w_rgb", c_vp), ("b_rgb", c_vp), ("packed", c_vp),
    ]


# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/ucnerf_march.h
SIGNATURES = {
    "ucn_last_error": [],
    "ucn_abi_version": [],
    "ucn_probe_copy": [c_vp, c_vp, c_u64, c_vp],
    "ucn_grid_encode_forward": [c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, c_u32, c_f32, c_u32, c_vp, c_u32, c_i32,
                                c_u32, c_i32, c_vp],
    "ucn_grid_encode_backward": [c_vp, c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, c_u32, c_f32, c_u32, c_vp, c_vp,
                                 c_u32, c_i32, c_u32, c_i32, c_vp],
    "ucn_grad_total_variation": [c_vp, c_vp, c_vp, c_vp, c_f32, c_u32, c_u32, c_u32, c_u32, c_f32, c_u32, c_u32, c_i32,
                                 c_vp],
    "ucn_field_packed_floats": [ctypes.POINTER(UcnField)],
    "ucn_field_pack": [ctypes.POINTER(UcnField), c_vp],
    "ucn_resample": [c_vp, c_vp, c_u32, c_f32, c_f32, c_f32, c_vp, c_vp, c_u32, c_f32, c_u32, c_u32, c_vp, c_vp],
    "ucn_cone_basis": [c_vp, c_vp, c_u32, c_vp, c_vp],
    "ucn_march_features": [ctypes.POINTER(UcnField), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32,
                           c_u32, c_u32, c_u32, c_i32, c_vp, c_vp, c_vp, c_vp],
    "ucn_march_features_backward": [ctypes.POINTER(UcnField), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                    c_f32, c_u32, c_u32, c_u32, c_i32, c_vp, c_vp, c_vp, c_vp],
    "ucn_march_features_backward_ws_floats": [ctypes.POINTER(UcnField), c_u32, c_u32],
    "ucn_march_features_backward_row_blocks": [ctypes.POINTER(UcnField), c_u32, c_u32],
    "ucn_cast_probe": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_u32, c_u32, c_vp, c_vp],
    "ucn_contract_probe": [c_vp, c_vp, c_u32, c_vp, c_vp, c_vp],
    "ucn_points_features": [ctypes.POINTER(UcnField), c_vp, c_vp, c_u32, c_u32, c_i32, c_u32, c_vp, c_vp, c_vp],
    "ucn_field_dir_floats": [ctypes.POINTER(UcnField), c_u32],
    "ucn_field_dir_bias": [ctypes.POINTER(UcnField), c_vp, c_u32, c_vp, c_vp],
    "ucn_field_mlp": [ctypes.POINTER(UcnField), c_vp, c_u32, c_u32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp],
    "ucn_composite": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_u32, c_u32, c_vp, c_vp, c_vp, c_vp],
    "ucn_tsdf_integrate": [c_vp, c_u32, c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, c_f32, c_vp, c_vp, c_vp, c_vp],
    "ucn_compact_alive": [c_vp, c_u32, c_u32, c_i32, c_f32, c_vp, c_vp, c_vp],
    "ucn_field_rgb_compacted": [c_vp, c_vp, c_u32, c_u32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp],
    "ucn_composite_backward": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_u32, c_u32, c_vp, c_vp, c_vp, c_vp, c_vp],
    "ucn_generate_rays": [c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_u32, c_u32, c_u32, c_u32, c_f32, c_f32, c_vp, c_vp, c_vp,
                          c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "ucn_adam_step": [c_vp, c_vp, c_vp, c_vp, c_u64, c_f32, c_f32, c_f32, c_f32, c_u32, c_i32, c_vp],
    "ucn_distortion_loss": [c_vp, c_vp, c_u32, c_u32, c_vp, c_vp, c_vp],
    "ucn_img_warping": [c_vp, c_vp, c_vp, c_u32, c_u32, c_vp, c_vp, c_vp, c_vp],
    "ucn_warp_scatter_depth": [c_vp, c_vp, c_vp, c_u32, c_u32, c_vp, c_vp, c_vp],
    "ucn_interlevel_loss": [c_vp, c_vp, c_u32, c_vp, c_vp, c_u32, c_f32, c_u32, c_vp, c_vp, c_vp],
    "ucn_bias_relu": [c_vp, c_vp, c_u32, c_u32, c_u32, c_i32, c_vp],
    "ucn_relu_backward_reduce": [c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, c_i32, c_vp],
    "ucn_nan_to_num_many": [c_vp, c_vp, c_u32, c_vp],
    "ucn_adam_step_many": [c_vp, c_vp, c_vp, c_vp, c_vp, c_u32, c_f32, c_f32, c_f32, c_f32, c_u32, c_i32, c_vp],
    "ucn_hash_decay": [c_vp, c_vp, c_u32, c_u32, c_vp, c_vp, c_vp, c_vp],
    "ucn_prop_train_fwd": [c_vp, c_u32, c_u32, c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_u64, c_vp, c_u32, c_u32, c_vp],
    "ucn_prop_train_bwd_ws_floats": [c_u32, c_u64],
    "ucn_prop_train_bwd": [c_vp, c_u32, c_u32, c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_u64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                           c_vp, c_vp],
    "ucn_train_fwd_fragments": [],
    "ucn_train_fwd": [c_vp, c_u32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_vp, c_vp, c_vp, c_vp, c_u32, c_vp, c_vp, c_vp,
                      ctypes.POINTER(c_f32), c_vp, c_vp, c_vp, c_vp, c_vp, c_u32, c_vp],
    "ucn_train_bwd": [c_vp, c_vp, ctypes.POINTER(c_f32), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, c_vp, c_vp, c_vp,
                      c_vp, c_vp, c_u32, c_vp, c_vp],
    "ucn_sky_packed_floats": [],
    "ucn_sky_pack": [ctypes.POINTER(UcnSky), c_vp],
    "ucn_sky_workspace_floats": [c_u32],
    "ucn_sky_render": [ctypes.POINTER(UcnSky), c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_u32, c_vp, c_vp, c_i32, c_vp],
    "ucn_sky_train_packed_bytes": [],
    "ucn_sky_train_act_ld": [],
    "ucn_sky_train_grad_ld": [],
    "ucn_sky_train_pack": [ctypes.POINTER(UcnSkyTrain), c_vp],
    "ucn_sky_train_fwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_u32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "ucn_sky_train_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_u32, c_vp, c_vp, c_vp, c_vp, c_vp],
    "ucn_wgrad_ws_floats": [c_u32, c_u32, c_u64],
    "ucn_wgrad_bf16": [c_vp, c_u32, c_u32, c_vp, c_u32, c_u32, c_vp, c_u32, c_u32, c_u64, c_vp, c_vp, c_vp],
    "ucn_gemm_f32": [c_vp, c_u32, c_vp, c_u32, c_vp, c_u32, c_u32, c_u32, c_i32, c_vp, c_u32, c_vp],
    "ucn_gemm_f32_ex": [c_vp, c_u32, c_vp, c_u32, c_vp, c_u32, c_u32, c_u32, c_i32, c_vp, c_u32, c_vp, c_u32, c_vp, c_u32, c_u32, c_vp],
    "ucn_wgrad_f32_ws_floats": [c_u32, c_u32, c_u64],
    "ucn_wgrad_f32": [c_vp, c_u32, c_vp, c_u32, c_u32, c_u32, c_u32, c_vp, c_vp, c_vp, c_vp],
    "ucn_amax_f32": [c_vp, c_u32, c_u64, c_u32, c_vp, c_vp],
    "ucn_pack_h3_bytes": [c_u32, c_u32],
    "ucn_pack_h3": [c_vp, c_u32, c_u32, c_u32, c_i32, c_vp, c_vp, c_vp],
    "ucn_gemm_h3": [c_vp, c_u32, c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, c_i32, c_vp, c_u32, c_vp, c_u32, c_vp, c_u32, c_u32, c_vp, c_vp],
    "ucn_gemm_h3_x2": [c_vp, c_u32, c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, c_i32, c_vp, c_u32, c_vp, c_u32, c_vp, c_u32, c_u32, c_vp, c_u32,
                       c_vp, c_u32, c_vp, c_vp, c_vp, c_vp],
    "ucn_relu_bits_words": [c_u64, c_u32],
    "ucn_wgrad_h3_ws_floats": [c_u32, c_u32, c_u64],
    "ucn_wgrad_h3": [c_vp, c_u32, c_vp, c_u32, c_vp, c_vp, c_u32, c_u32, c_u32, c_vp, c_vp, c_vp, c_vp],
    "ucn_marching_cubes_ws_bytes": [c_u32, c_u32, c_u32],
    "ucn_marching_cubes_count": [c_vp, c_u32, c_u32, c_u32, c_f32, c_vp, c_vp, c_vp],
    "ucn_marching_cubes_emit": [c_vp, c_u32, c_u32, c_u32, c_f32, c_f32, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp],
    "ucn_image_metrics_ws_bytes": [c_u32, c_u32],
    "ucn_image_metrics": [c_vp, c_vp, c_u32, c_u32, c_vp, c_vp, c_vp],
    "ucn_dense": [c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, c_i32, c_vp, c_vp],
    "ucn_apply_affine": [c_vp, c_vp, c_vp, c_vp, c_u32, c_vp, c_vp, c_u32, c_vp, c_vp],
    "ucn_affine_blend": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_u32, ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "ucn_data_loss": [c_vp, c_u32, c_vp, c_vp, c_vp, c_vp, c_u32, ctypes.c_float, c_vp, c_vp, c_vp, c_vp],
    "ucn_sky_loss": [c_vp, c_u32, c_vp, c_u32, c_vp, c_vp, c_vp, c_vp],
    "ucn_identity_loss": [c_vp, c_vp, c_u32, c_vp, c_vp, c_vp, c_vp, c_vp],
}
_RESTYPES = {"ucn_last_error": ctypes.c_char_p, "ucn_abi_version": c_u32, "ucn_field_packed_floats": c_u64,
             "ucn_field_dir_floats": c_u64, "ucn_march_features_backward_ws_floats": c_u64,
             "ucn_sky_packed_floats": c_u64, "ucn_sky_workspace_floats": c_u64, "ucn_train_fwd_fragments": c_u64,
             "ucn_prop_train_bwd_ws_floats": c_u64, "ucn_sky_train_packed_bytes": c_u64, "ucn_wgrad_ws_floats": c_u64, "ucn_wgrad_f32_ws_floats": c_u64, "ucn_wgrad_h3_ws_floats": c_u64, "ucn_pack_h3_bytes": c_u64, "ucn_relu_bits_words": c_u64, "ucn_marching_cubes_ws_bytes": c_u64, "ucn_image_metrics_ws_bytes": c_u64, "ucn_sky_train_act_ld": c_u32,
             "ucn_sky_train_grad_ld": c_u32}

_lib = None


def load():
    """dlopen the in-tree library and bind every symbol of the header (ImportError if absent)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with ucnerf_amd/csrc/build.sh (hipcc --offload-arch=gfx950). "
            "ucnerf_amd has no CPU / eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError = header/library mismatch: fail loudly
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, ctypes.c_int)
    got = lib.ucn_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"libucnerf_march.so ABI {got} != expected {ABI_VERSION}: rebuild")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise RuntimeError(load().ucn_last_error().decode())


def ptr(t):
    """Device/host address of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    """The current HIP stream handle (the reference launches on the default stream,
    gridencoder.cu:374-382; here the caller's current torch stream is honoured)."""
    return torch.cuda.current_stream().cuda_stream


def require_device(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what} must be a CUDA tensor")      # CHECK_CUDA, gridencoder.cu:15
