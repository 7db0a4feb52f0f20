"""ctypes binding of libanimnerf_hip.so (the C ABI declared in include/animnerf_hip.h).

There is deliberately no fallback: if the library is missing, or a tensor is not on a GPU,
the calls raise.  The oracle under /oracle is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# ANIMNERF_HIP_LIB points at an alternative build of the SAME library (kernel experiments); never a fallback
LIB_PATH = os.environ.get("ANIMNERF_HIP_LIB") or os.path.join(_HERE, "libanimnerf_hip.so")

ANR_MLP_F32 = 0
ANR_MLP_BF16 = 1
ANR_MLP_FLAG_NO_DMA = 0x100
ANR_MLP_FLAG_SIGMA_ONLY = 0x400
ANR_MLP_FLAG_TANGENT = 0x800
ANR_MLP_FLAG_ACCUMULATE = 0x1000
ANR_MLP_FLAG_VIEW = 0x2000
ANR_MLP_FLAG_BACKGROUND = 0x4000
ANR_MLP_FLAG_NO_FILL = 0x10000
ANR_MLP_FLAG_BITS_ONLY = ANR_MLP_FLAG_ENC_ONLY = 0x8000
ANR_MAX_SAMPLES = 256


class HipLibraryMissing(RuntimeError):
    pass


class HipCallFailed(RuntimeError):
    pass


# struct anr_mlp_params { w_trunk[8]; b_trunk[8]; w_sigma; b_sigma; w_final; b_final; w_dir; b_dir; w_rgb; b_rgb; }
class AnrMlpParams(C.Structure):
    _fields_ = [
        ("w_trunk", C.c_void_p * 8), ("b_trunk", C.c_void_p * 8),
        ("w_sigma", C.c_void_p), ("b_sigma", C.c_void_p),
        ("w_final", C.c_void_p), ("b_final", C.c_void_p),
        ("w_dir", C.c_void_p), ("b_dir", C.c_void_p),
        ("w_rgb", C.c_void_p), ("b_rgb", C.c_void_p),
    ]


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float


class AnrLossArgs(C.Structure):
    _fields_ = ([(k, _P) for k in ("rgb", "acc", "rgb_fine", "acc_fine", "target_rgb", "target_alpha", "s", "s_fine", "quads",
                                   "quads_fine")]
                + [(k, _L) for k in ("R", "prior_rows", "nv", "normal_sets", "quad_rows")]
                + [("n_fg", C.c_int32), ("n_bg", C.c_int32)]
                + [(k, _F) for k in ("k", "delta", "lambda_alphas", "lambda_foreground", "lambda_background", "lambda_normals")]
                + [("s_stride", C.c_int32), ("s_count", _P), ("s_count_fine", _P), ("s_grad_rows", C.c_int32), ("quad_grad_rows", C.c_int32)])


class AnrDrawPlan(C.Structure):
    _fields_ = ([(k, _P) for k in ("t_rand", "noise_c", "u_fine", "noise_f")] + [(k, _L) for k in ("n_t", "n_nc", "n_u", "n_nf")]
                + [("t_scale", _F), ("noise_scale", _F), ("verts_template", _P), ("n_v3", _L), ("point_scale", _F), ("neighbour_scale", _F)]
                + [(k, _P) for k in ("n0", "n1", "pair", "quads")])


class AnrLossGrads(C.Structure):
    _fields_ = [(k, _P) for k in ("rgb", "acc", "rgb_fine", "acc_fine", "s", "s_fine", "quads", "quads_fine")]


# name -> (restype, argtypes); every symbol include/animnerf_hip.h declares
SIGNATURES = {
    "anr_version": (_I, []),
    "anr_last_error": (C.c_char_p, []),
    "anr_ray_gen": (_I, [_P, _P, _P, _I, _I, _F, _F, _P, _P]),
    "anr_smpl_forward": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "anr_frame_backward": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _I, _I, _P, _P, _P, _P, _P]),
    "anr_frame_backward_ws_floats": (_L, [_I, _I]),
    "anr_frame_backward_adjoint": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _I, _I, _P, _P, _P, _P, _P]),
    "anr_frame_backward_adjoint_values": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P]),
    "anr_frame_backward_ws_zero_floats": (_L, [_I]),
    "anr_to_root_frame": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "anr_rays_to_body": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "anr_ober2cano": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _L, _P]),
    "anr_knn_index_bytes": (_L, [_I]),
    "anr_knn_index_build": (_I, [_P, _P, _I, _I, _P, _P]),
    "anr_knn_index_build_reach": (_I, [_P, _P, _I, _I, _F, _P, _P]),
    "anr_knn": (_I, [_P, _P, _I, _I, _L, _P, _P, _P]),
    "anr_knn_k": (_I, [_P, _P, _I, _I, _I, _L, _I, _P, _P, _P]),
    "anr_sample_coarse": (_I, [_P, _I, _P, _P, _L, _I, _P, _P]),
    "anr_warp_ws_ints": (_L, [_I, _L]),
    "anr_warp_points_lean": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _L, _F, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "anr_warp_points_reuse": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _L, _F, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I,
                                   _P, _P, _P]),
    "anr_warp_points_cells": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _L, _F, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I,
                                   _P, _P, _P, _L, _P]),
    "anr_warp_points": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _L, _F, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "anr_warp_backward": (_I, [_P, _P, _I, _P, _I, _P, _P, _P, _I, _I, _L, _P, _P, _P, _P]),
    "anr_points_from_rays": (_I, [_P, _I, _P, _I, _L, _P, _P]),
    "anr_mlp_pack_bytes": (_L, [_I]),
    "anr_mlp_pack": (_I, [C.POINTER(AnrMlpParams), _I, _P, _P]),
    "anr_mlp_pack_pair": (_I, [C.POINTER(AnrMlpParams), C.POINTER(AnrMlpParams), _I, _P, _P, _P]),
    "anr_mlp_bwd_pack_pair": (_I, [C.POINTER(AnrMlpParams), C.POINTER(AnrMlpParams), _I, _P, _P, _P]),
    "anr_mlp_forward": (_I, [_P, _I, _P, _L, _P, _P]),
    "anr_mlp_forward_rays": (_I, [_P, _I, _P, _I, _P, _I, _L, _P, _P]),
    "anr_mlp_forward_embedded": (_I, [_P, _I, _P, _L, _P, _P, _P]),
    "anr_mlp_forward_rays_steps": (_I, [_P, _I, _P, _I, _P, _I, _L, _P, _P]),
    "anr_compact_valid": (_I, [_P, _L, _P, _P, _P, _I, _P]),
    "anr_mlp_forward_indexed": (_I, [_P, _I, _P, _P, _P, _L, _P, _P]),
    "anr_mlp_act_cols": (_I, []),
    "anr_mlp_pack_view": (_I, [C.POINTER(AnrMlpParams), _I, _I, _P, _P]),
    "anr_mlp_forward_view": (_I, [_P, _I, _P, _P, _I, _P, _P, _L, _P, _P]),
    "anr_mlp_forward_save": (_I, [_P, _I, _P, _L, _P, _P, _P]),
    "anr_mlp_forward_save_indexed": (_I, [_P, _I, _P, _P, _P, _L, _P, _P, _P]),
    "anr_encode": (_I, [_P, _I, _L, _I, _P, _P]),
    "anr_encode64": (_I, [_P, _I, _L, _I, _P, _P]),
    "anr_mlp_wgrad_floats": (_L, []),
    "anr_mlp_wgrad_ws_floats": (_L, [_L]),
    "anr_mlp_denc": (_I, [_I, _P, _P, _P, _L, _P, _P]),
    "anr_mlp_dpoints": (_I, [_P, _I, _P, _P, _L, _P, _P, _P]),
    "anr_warp_backward_compact": (_I, [_P, _P, _P, _I, _P, _I, _P, _P, _P, _I, _I, _L, _P, _P, _P, _P]),
    "anr_mlp_wgrad": (_I, [_I, _P, _P, _P, _P, _L, _P, _P, _P]),
    "anr_encode_backward": (_I, [_P, _I, _P, _L, _P, _P]),
    "anr_mlp_bwd_pack_bytes": (_L, [_I]),
    "anr_mlp_bwd_pack": (_I, [C.POINTER(AnrMlpParams), _I, _P, _P]),
    "anr_mlp_backward": (_I, [_P, _I, _P, _P, _P, _L, _P]),
    "anr_mlp_backward_feature": (_I, [_P, _I, _P, _P, _P, _P, _L, _P]),
    "anr_grid_points": (_I, [_I, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _P, _L, _L, _P, _P]),
    "anr_composite_masked": (_I, [_P, _P, _P, _I, _P, _P, _L, _I, _I, _P, _P, _P, _P, _P]),
    "anr_composite": (_I, [_P, _P, _P, _I, _P, _L, _I, _I, _P, _P, _P, _P, _P]),
    "anr_composite_indexed": (_I, [_P, _P, _P, _P, _I, _P, _L, _I, _I, _P, _P, _P, _P, _P]),
    "anr_composite_backward_indexed": (_I, [_P, _P, _P, _P, _I, _P, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "anr_composite_backward_compact": (_I, [_P, _P, _P, _P, _P, _I, _P, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "anr_composite_backward": (_I, [_P, _P, _P, _I, _P, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "anr_sample_fine_merge": (_I, [_P, _P, _P, _I, _L, _I, _I, _P, _P, _P, _P]),
    "anr_sample_fine_merge_u8": (_I, [_P, _P, _P, _I, _L, _I, _I, _P, _P, _P, _P]),
    "anr_compact_ws_ints": (_L, [_L]),
    "anr_compact_ordered": (_I, [_P, _L, _P, _P, _P, _P, _P, _P]),
    "anr_expand_rows": (_I, [_P, _P, _L, _I, _F, _P, _P]),
    "anr_mlp_head_grad": (_I, [_P, _P, _P, _P, _L, _L, _I, _P, _P]),
    "anr_tangent_quads": (_I, [_P, _L, _L, _P, _P]),
    "anr_mc_classify": (_I, [_P, _I, _I, _I, _F, _P, _P, _P, _P, _P]),
    "anr_mc_emit": (_I, [_P, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P, _P]),
    "anr_mlp_head_grad_counted": (_I, [_P, _P, _P, _P, _P, _L, _I, _P, _P]),
    "anr_mlp_backward_counted": (_I, [_P, _I, _P, _P, _P, _L, _P, _P]),
    "anr_encode64_counted": (_I, [_P, _I, _L, _P, _I, _P, _P]),
    "anr_mlp_wgrad_counted": (_I, [_I, _P, _P, _P, _P, _L, _P, _P, _P, _P]),
    "anr_mlp_denc_counted": (_I, [_I, _P, _P, _P, _L, _P, _P, _P]),
    "anr_encode_backward_counted": (_I, [_P, _I, _P, _L, _P, _P, _P]),
    "anr_sample_coarse_backward": (_I, [_P, _P, _P, _L, _I, _P, _P]),
    "anr_merge_backward": (_I, [_P, _P, _L, _I, _I, _P, _P]),
    "anr_train_loss_ws_floats": (_L, []),
    "anr_train_loss": (_I, [C.POINTER(AnrLossArgs), _P, _P, _P]),
    "anr_adam_chunk_floats": (_I, []),
    "anr_adam_chunk_bytes": (_I, []),
    "anr_adam_step": (_I, [_P, _I, _P, C.POINTER(_F), _I, C.c_double, C.c_double, C.c_double, _P]),
    "anr_compact_ordered_riders": (_I, [_P, _L, _P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P]),
    "anr_compact_state_words": (_L, [_L]),
    "anr_compact_ordered_single": (_I, [_P, _L, _P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P]),
    "anr_train_draws": (_I, [_P, _P, _P]),
    "anr_gather_frame_params": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P]),
    "anr_scatter_frame_param_grads": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "anr_merge_backward2": (_I, [_P, _P, _P, _L, _I, _I, _P, _P]),
    "anr_sample_coarse_backward_acc": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _P, _P]),
    "anr_to_root_frame_strided": (_I, [_P, _L, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "anr_zero_fill": (_I, [_P, _L, _P]),
    "anr_add_inplace": (_I, [_P, _P, _L, _P]),
    "anr_copy_segments": (_I, [_P, _P, _P, _I, _P]),
    "anr_zero_segments": (_I, [_P, _P, _I, _P]),
    "anr_add_segments": (_I, [_P, _P, _P, _I, _P]),
    "anr_mlp_wgrad_sigma_floats": (_L, []),
    "anr_warp_ws_zero_range": (_I, [_I, _L, C.POINTER(_L), C.POINTER(_L)]),
    "anr_frame_setup": (_I, [_P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I]
                        + [_P] * 13 + [_P, _P]),
    "anr_frame_setup_rows": (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I]
                             + [_P] * 13 + [_P, _P]),
    "anr_ray_march": (_I, [_P, _P, _I, _P, _I, _L, _P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "anr_ray_march_warp": (_I, [_P, _P, _I, _P, _I, _I, _L, _P, _I, _P, _I, _I, _P, _P, _P, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P]),
    "anr_knn_within": (_I, [_P, _P, _I, _I, _L, _F, _P, _P]),
    "anr_grid_points_cells": (_I, [_I, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _P, _P, _L, _P, _P, _P]),
    "anr_scatter_relu": (_I, [_P, _P, _L, _L, _L, _P, _P]),
    "anr_adam_step_counting": (_I, [_P, _I, _P, _P, _I, _P, C.POINTER(_F), _I, C.c_double, C.c_double, C.c_double, _P]),
    "anr_train_loss_backward": (_I, [C.POINTER(AnrLossArgs), _P, C.POINTER(AnrLossGrads), _P]),
    "anr_composite_sample": (_I, [_P, _P, _P, _P, _I, _P, _P, _I, _L, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
}

_lock = threading.Lock()
_lib = None


def load():
    """dlopen the library and type every entry point.  Raises HipLibraryMissing if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise HipLibraryMissing(
                    f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(hipcc, gfx950). There is no CPU or PyTorch fallback for the rendering path.")
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)        # AttributeError = header/library mismatch
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().anr_last_error()
        raise HipCallFailed(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
