"""ctypes binding of ``libflocoder_amd.so`` (C ABI in ``include/flocoder_amd.h``).

There is no CPU or eager-PyTorch fallback anywhere behind this module: if the shared library has not been
built, or the device is not a gfx950 part, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# FLOCODER_AMD_LIB: another build of the same ABI (A/B timing of two builds inside one box session: tools/ab.sh)
LIB_PATH = os.environ.get("FLOCODER_AMD_LIB") or os.path.join(_HERE, "_lib", "libflocoder_amd.so")

FC_OK, FC_E_ARG, FC_E_SHAPE, FC_E_ARCH, FC_E_HIP, FC_E_STATE = 0, -1, -2, -3, -4, -5
FC_METHOD_EULER, FC_METHOD_RK4 = 0, 1
TILE_AUTO = -1
TILES = {"M128N32": 0, "M128N64": 1, "M64N32K2": 2, "M32N32K4": 3, "M64N64K2": 4, "M256N64": 5}


class fc_unet_config(C.Structure):
    _fields_ = [("dim", C.c_int), ("channels", C.c_int), ("n_levels", C.c_int), ("dim_mults", C.c_int * 8),
                ("groups", C.c_int), ("n_classes", C.c_int), ("mask_cond", C.c_int)]


_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64
_pi, _pf = C.POINTER(C.c_int), C.POINTER(C.c_float)

# name -> (restype, argtypes); mirrors include/flocoder_amd.h one to one (tests/test_abi.py checks it)
SIGNATURES = {
    "fc_abi_version": (_i, []),
    "fc_last_error": (C.c_char_p, []),
    "fc_check_device": (_i, [_i]),
    "fc_unet_create": (_i, [C.POINTER(fc_unet_config), _i, C.POINTER(_vp)]),
    "fc_unet_destroy": (None, [_vp]),
    "fc_unet_param_count": (_i, [_vp]),
    "fc_unet_param_info": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(_i64 * 4), C.POINTER(_i64)]),
    "fc_unet_param_numel": (_i64, [_vp]),
    "fc_unet_load_params": (_i, [_vp, _vp, _i64, _i, _vp]),
    "fc_unet_reserve": (_i, [_vp, _i, _i, _i]),
    "fc_unet_reserved": (_i, [_vp, _pi, _pi, _pi]),
    "fc_unet_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp]),
    "fc_unet_integrate": (_i, [_vp, _i, _vp, _i, _i, _i, _pf, _i, _f, _f, _vp, _f, _vp, _i, _vp]),
    "fc_unet_chains": (_i, [_vp, _pi]),
    "fc_unet_plan_launches": (_i, [_vp]),
    "fc_unet_flops_per_sample": (C.c_double, [_vp]),
    "fc_unet_op_info": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_double)]),
    "fc_unet_profile_ops": (_i, [_vp, _i, _i, _pf, _i, _vp]),
    "fc_unet_set_time_freqs": (_i, [_vp, _pf, _i]),
    "fc_unet_debug_tensor": (_i, [_vp, C.c_char_p, C.POINTER(_vp), _pi, _pi, _pi]),
    "fc_debug_copy": (_i, [_vp, _vp, _i64, _vp]),
    "fc_debug_set_poison": (_i, [_i]),
    "fc_debug_poison_check": (_i, [_pi, _pi]),
    "fc_debug_set_conv_stamps": (_i, [_vp]),
    "fc_debug_set_stamp_op": (_i, [_i, _vp]),
    "fc_debug_conv": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _pi, _pf, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _pf, _vp]),
    "fc_debug_set_fused_tail": (_i, [_i]),
    "fc_unet_fused_tail_errors": (_i, [_vp, C.POINTER(_i)]),
    "fc_unet_set_shared": (_i, [_vp, _i]),
    "fc_unet_meeting_launches": (_i, [_vp]),
    "fc_unet_check": (_i, [_vp, _vp, _i]),
    "fc_debug_unet_break_meeting": (_i, [_vp]),
    "fc_debug_unet_break_meeting_kind": (_i, [_vp, _i]),
    "fc_unet_train_reserve": (_i, [_vp, _i, _i, _i]),
    "fc_unet_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "fc_unet_backward_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i64, _vp, _vp, _i, _i, _i, _vp]),
    "fc_unet_backward_parts": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i64, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "fc_unet_grad_buckets": (_i, [_vp, C.POINTER(_i64)]),
    "fc_unet_set_grad_buckets": (_i, [_vp, _i]),
    "fc_unet_arena_serial": (C.c_uint64, [_vp]),
    "fc_unet_class_param_range": (_i, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "fc_flow_interp": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _vp]),
    "fc_flow_prepare": (_i, [_vp, _vp, _vp, _vp, _f, _f, _f, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i64, _vp]),
    "fc_mse_loss_grad": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "fc_grad_clip_coef": (_i, [_vp, _i64, _vp, _i64, C.c_float, _vp, _vp, _vp]),
    "fc_adam_ema_step": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, C.c_float, C.c_float, C.c_float, C.c_float, _i, C.c_float, _i, _vp]),
    "fc_adam_ema_step_guarded": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, C.c_float, C.c_float, C.c_float, C.c_float, _i, C.c_float, _i, _vp, _vp]),
    "fc_debug_conv_wgrad": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "fc_vae_create": (_i, [_i, C.POINTER(_vp)]),
    "fc_vae_destroy": (None, [_vp]),
    "fc_vae_param_count": (_i, [_vp]),
    "fc_vae_param_numel": (_i64, [_vp]),
    "fc_vae_param_info": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(_i64 * 4), C.POINTER(_i64)]),
    "fc_vae_load_params": (_i, [_vp, _vp, _i64, _i, _vp]),
    "fc_vae_reserve_encode": (_i, [_vp, _i, _i, _i]),
    "fc_vae_reserve_decode": (_i, [_vp, _i, _i, _i]),
    "fc_vae_encode": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "fc_vae_decode": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "fc_vae_flops_per_sample": (C.c_double, [_vp, _i]),
    "fc_vae_plan_launches": (_i, [_vp, _i]),
    "fc_vqvae_create": (_i, [_i, _i, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "fc_vqvae_create_ex": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "fc_na2d": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "fc_vqvae_destroy": (None, [_vp]),
    "fc_vqvae_param_count": (_i, [_vp]),
    "fc_vqvae_param_numel": (_i64, [_vp]),
    "fc_vqvae_param_info": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(_i64 * 4), C.POINTER(_i64)]),
    "fc_vqvae_load_params": (_i, [_vp, _vp, _i64, _i, _vp]),
    "fc_vqvae_reserve_encode": (_i, [_vp, _i, _i, _i]),
    "fc_vqvae_reserve_decode": (_i, [_vp, _i, _i, _i]),
    "fc_vqvae_encode": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "fc_vqvae_decode": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "fc_vqvae_flops_per_sample": (C.c_double, [_vp, _i]),
    "fc_vae_op_info": (_i, [_vp, _i, _i, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_double)]),
    "fc_vae_set_precision": (_i, [_vp, _i]),
    "fc_vqvae_set_precision": (_i, [_vp, _i]),
    "fc_debug_set_conv_precision": (_i, [_i]),
    "fc_vae_op_bytes": (_i, [_vp, _i, _i, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "fc_unet_op_bytes": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "fc_vqvae_plan_launches": (_i, [_vp, _i]),
    "fc_vqvae_op_info": (_i, [_vp, _i, _i, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "fc_vqvae_profile_ops": (_i, [_vp, _i, _vp, _vp, _i, _i, C.POINTER(C.c_float), _i, _vp]),
    "fc_vae_profile_ops": (_i, [_vp, _i, _vp, _vp, _i, _i, C.POINTER(C.c_float), _i, _vp]),
    "fc_rvq_quantize": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "fc_mask_encoder_create": (_i, [_i, C.POINTER(_vp)]),
    "fc_mask_encoder_destroy": (None, [_vp]),
    "fc_mask_encoder_param_count": (_i, [_vp]),
    "fc_mask_encoder_param_numel": (_i64, [_vp]),
    "fc_mask_encoder_param_info": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(_i64 * 4), C.POINTER(_i64)]),
    "fc_mask_encoder_load_params": (_i, [_vp, _vp, _i64, _i, _vp]),
    "fc_mask_encoder_reserve": (_i, [_vp, _i, _i, _i]),
    "fc_mask_encoder_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "fc_mask_encoder_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "fc_mask_blend": (_i, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "fc_sinkhorn_divergence": (_i, [_vp, _vp, _i, _i, _i64, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double),
                               _pi, _vp]),
    "fc_ot_pairing": (_i, [_vp, _vp, _i, _i64, _vp, _vp, _vp]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """The loaded library.  Raises RuntimeError when it has not been built -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"flocoder_amd: {LIB_PATH} is missing. Build it with `python -m flocoder_amd.build` "
                "(needs hipcc); there is no CPU or PyTorch fallback for this path.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.fc_abi_version() != 1:
            raise RuntimeError("flocoder_amd: shared library ABI version mismatch; rebuild it")
        _lib = handle
    return _lib


def check(rc: int) -> None:
    """0 -> ok; argument/shape errors -> ValueError, everything else -> RuntimeError (the reference raises plain
    Python exceptions from its codec / model constructors, codecs.py:725-728)."""
    if rc == FC_OK:
        return
    msg = lib().fc_last_error().decode(errors="replace")
    if rc in (FC_E_ARG, FC_E_SHAPE):
        raise ValueError(f"flocoder_amd: {msg}")
    raise RuntimeError(f"flocoder_amd: {msg} (code {rc})")


def ptr(t) -> Optional[int]:
    """data_ptr of a contiguous tensor, or None."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("flocoder_amd: tensors crossing the C ABI must be contiguous")
    return t.data_ptr()


def current_stream(device) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream
