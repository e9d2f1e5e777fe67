"""ctypes binding of libishap_hip.so (the C ABI declared in include/ishap.h).

The product path has no CPU fallback: if the library is missing or a call fails,
this module raises.  Build with `python -m ishapediting_amd.build`.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libishap_hip.so")

c_float_p = C.POINTER(C.c_float)
c_void_p = C.c_void_p


class UNetConfigC(C.Structure):
    _fields_ = [("image_size", C.c_int), ("in_channels", C.c_int), ("model_channels", C.c_int),
                ("out_channels", C.c_int), ("num_res_blocks", C.c_int), ("n_mult", C.c_int),
                ("channel_mult", C.c_int * 8), ("n_att", C.c_int), ("attention_ds", C.c_int * 8),
                ("num_head_channels", C.c_int), ("max_batch", C.c_int)]


class StepCoefs(C.Structure):
    _fields_ = [("min_log", C.c_float), ("max_log", C.c_float), ("sqrt_recip", C.c_float),
                ("sqrt_recipm1", C.c_float), ("coef1", C.c_float), ("coef2", C.c_float),
                ("nonzero", C.c_float), ("clip_denoised", C.c_int), ("mode", C.c_int),
                ("ddim_a", C.c_float), ("ddim_b", C.c_float), ("ddim_sigma", C.c_float),
                ("rng", C.c_int), ("rng_seed", C.c_ulonglong), ("rng_offset", C.c_ulonglong), ("noise_out", c_void_p)]


class DragArgsC(C.Structure):
    _fields_ = [("W", C.c_int), ("ld", C.c_int), ("Cc", C.c_int), ("chmap", c_void_p), ("sources", c_void_p),
                ("targets", c_void_p), ("B", C.c_int), ("r", C.c_int), ("voxel", C.c_float), ("cof", C.c_float),
                ("l1", C.c_int), ("touched", c_void_p), ("nmask", c_void_p), ("acc", c_void_p), ("grad_fx", c_void_p), ("chan_weight", c_void_p)]


class DecoderWeightsC(C.Structure):
    _fields_ = [("B", c_void_p), ("W1", c_void_p), ("b1", c_void_p), ("W2", c_void_p), ("b2", c_void_p),
                ("w3", c_void_p), ("b3", c_void_p)]


# every symbol include/ishap.h declares: (restype, argtypes)
SYMBOLS = {
    "ishap_last_error": (C.c_char_p, []),
    "ishap_version": (C.c_int, []),
    "ishap_device_status": (C.c_int, []),
    "ishap_rendezvous_would_grant": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ishap_group_norm32_scratch_bytes": (C.c_longlong, [C.c_int, C.c_int, C.c_int]),
    "ishap_group_norm32": (C.c_int, [c_void_p, c_void_p, c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     c_void_p, c_void_p, c_void_p, c_void_p]),
    "ishap_group_norm32_backward": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_int, c_void_p, c_void_p, c_void_p]),
    "ishap_group_norm32_parts": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "ishap_unet_create": (C.c_int, [C.POINTER(UNetConfigC), C.c_int, C.POINTER(c_void_p)]),
    "ishap_unet_destroy": (None, [c_void_p]),
    "ishap_unet_num_params": (C.c_int, [c_void_p]),
    "ishap_unet_param_info": (C.c_int, [c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int),
                                        C.POINTER(C.c_longlong)]),
    "ishap_unet_load_param": (C.c_int, [c_void_p, C.c_char_p, c_void_p, C.c_longlong, c_void_p]),
    "ishap_unet_params_loaded": (C.c_int, [c_void_p]),
    "ishap_unet_forward": (C.c_int, [c_void_p, c_void_p, c_float_p, C.c_int, C.c_int, c_void_p, c_void_p, C.c_int,
                                     c_void_p]),
    "ishap_unet_tap_shape": (C.c_int, [c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ishap_unet_tap_ptr": (c_void_p, [c_void_p]),
    "ishap_unet_copy_tap": (C.c_int, [c_void_p, c_void_p, c_void_p]),
    "ishap_unet_block_output": (C.c_int, [c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), c_void_p,
                                          c_void_p]),
    "ishap_unet_workspace_bytes": (C.c_longlong, [c_void_p]),
    "ishap_unet_join_tail": (C.c_int, [c_void_p, c_void_p]),
    "ishap_unet_run_tail": (C.c_int, [c_void_p]),
    "ishap_unet_marks": (C.c_int, [c_void_p, c_void_p, c_void_p, C.c_int, c_void_p, c_void_p]),
    "ishap_unet_prepare_timesteps": (C.c_int, [c_void_p, c_void_p, C.c_int, c_void_p]),
    "ishap_unet_backward_input": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ishap_unet_backward_from_output": (C.c_int, [c_void_p, c_void_p, C.c_int, c_void_p, c_void_p, c_void_p]),
    "ishap_triplane_points_loss_grad": (C.c_int, [c_void_p, C.c_int, C.POINTER(DecoderWeightsC), c_void_p, c_void_p, c_void_p,
                                                  c_void_p, C.c_longlong, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ishap_x0_grad_to_cotangent": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, C.c_float, C.c_float, C.c_int, C.c_int,
                                             c_void_p, c_void_p, c_void_p]),
    "ishap_ddpm_step": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, C.POINTER(StepCoefs), C.c_int, C.c_int,
                                  C.c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ishap_ddpm_step_guided": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, C.POINTER(StepCoefs), C.c_int, C.c_int, C.c_int,
                                         c_void_p, C.c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ishap_guided_update": (C.c_int, [c_void_p, c_void_p, c_void_p, C.c_float, c_void_p, C.c_longlong, c_void_p,
                                      c_void_p]),
    "ishap_axpby": (C.c_int, [c_void_p, c_void_p, C.c_float, C.c_float, C.c_longlong, c_void_p, c_void_p]),
    "ishap_drag_setup": (C.c_int, [C.POINTER(DragArgsC), c_void_p]),
    "ishap_drag_loss_grad": (C.c_int, [C.POINTER(DragArgsC), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ishap_drag_loss_cotangent": (C.c_int, [C.POINTER(DragArgsC), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p]),
    "ishap_grad_to_scaled_f16": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, C.c_longlong, c_void_p]),
    "ishap_planes_prepare": (C.c_int, [c_void_p, c_void_p, c_void_p, C.c_int, c_void_p, c_void_p]),
    "ishap_triplane_decode_points": (C.c_int, [c_void_p, C.c_int, C.POINTER(DecoderWeightsC), c_void_p, C.c_longlong,
                                               c_void_p, c_void_p]),
    "ishap_surface_scratch_bytes": (C.c_longlong, [C.c_int]),
    "ishap_surface_count": (C.c_int, [c_void_p, C.c_int, C.c_float, C.c_int, c_void_p, c_void_p, c_void_p]),
    "ishap_surface_emit": (C.c_int, [c_void_p, C.c_int, C.c_float, C.c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ishap_mesh_smooth_scratch_bytes": (C.c_longlong, [C.c_longlong, C.c_longlong]),
    "ishap_mesh_smooth": (C.c_int, [c_void_p, C.c_longlong, c_void_p, C.c_longlong, C.c_int, C.c_float, c_void_p, C.c_longlong, c_void_p]),
    "ishap_chamfer": (C.c_int, [c_void_p, C.c_longlong, c_void_p, C.c_longlong, c_void_p, c_void_p, c_void_p]),
    "ishap_mesh_tri_areas": (C.c_int, [c_void_p, c_void_p, C.c_longlong, c_void_p, c_void_p]),
    "ishap_mesh_points_on_tris": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, C.c_longlong, c_void_p, c_void_p]),
    "ishap_mesh_occupancy": (C.c_int, [c_void_p, c_void_p, C.c_longlong, c_void_p, C.c_longlong, c_void_p, c_void_p]),
    "ishap_profile_begin": (C.c_int, []),
    "ishap_profile_end": (C.c_int, [C.POINTER(C.c_double), C.c_int]),
    "ishap_profile_shapes": (C.c_int, [C.c_char_p, C.c_int]),
    "ishap_triplane_decode_grid": (C.c_int, [c_void_p, C.c_int, C.POINTER(DecoderWeightsC), c_void_p, C.c_int,
                                             c_void_p, c_void_p]),
}

_lib = None


def lib():
    """Load (once) and return the shared library; raise if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is required (no CPU fallback exists). "
                "Build it with `python -m ishapediting_amd.build`.")
        # One HIP runtime per process: torch ships its own libamdhip64 and must be the one that is resident
        # (torch owns the device memory and streams we are handed), so load it before our library resolves
        # its libamdhip64.so.N dependency by SONAME.
        import torch  # noqa: F401
        tlib = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(tlib):
            C.CDLL(tlib, mode=C.RTLD_GLOBAL)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)     # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if l.ishap_version() < 3:     # 3: ishap_step_coefs ends with the rng fields this module's StepCoefs declares
            raise RuntimeError(f"{LIB_PATH} is an older build (ABI {l.ishap_version()} < 3): rebuild with `python -m ishapediting_amd.build`")
        _lib = l
    return _lib


def check(rc: int):
    if rc != 0:
        msg = lib().ishap_last_error()
        raise RuntimeError(f"libishap_hip: {msg.decode() if msg else 'error'} (code {rc})")


def ptr(t):
    """Device (or host) address of a torch tensor; None -> NULL."""
    return None if t is None else t.data_ptr()


def stream_ptr(device=None):
    """hipStream_t of torch's current stream, so library work is ordered with torch ops."""
    import torch
    return torch.cuda.current_stream(device).cuda_stream
