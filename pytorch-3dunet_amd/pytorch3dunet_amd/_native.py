"""ctypes binding of the C-ABI in include/u3d.h (libu3d_hip.so, gfx950 HIP kernels).

The product path has NO fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# (U3D_LIB_PATH: another build of the same library — same-box A/B runs of compile-time kernel experiments, tools/ab_libs.sh)
LIB_PATH = os.environ.get("U3D_LIB_PATH") or os.path.join(_HERE, "lib", "libu3d_hip.so")

U3D_OK = 0


class U3DSrc(ctypes.Structure):
    """mirror of u3d_src_t (include/u3d.h)"""

    _fields_ = [
        ("p0", c_void_p),
        ("p1", c_void_p),
        ("zmap", c_void_p),
        ("ymap", c_void_p),
        ("xmap", c_void_p),
        ("affine", c_void_p),
        ("C0", c_int32),
        ("C1", c_int32),
        ("D1", c_int32),
        ("H1", c_int32),
        ("W1", c_int32),
    ]


class U3DAdamDesc(ctypes.Structure):
    """mirror of u3d_adam_desc_t (include/u3d.h)"""

    _fields_ = [("p", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p), ("first", c_int64), ("numel", c_int64)]


class U3DGnBwdJob(ctypes.Structure):
    """mirror of u3d_gn_bwd_job_t (include/u3d.h)"""

    _fields_ = [("gstats_lo", c_void_p), ("gstats_hi", c_void_p), ("mean_rstd", c_void_p), ("gamma", c_void_p), ("dgamma", c_void_p),
                ("dbeta", c_void_p), ("coef", c_void_p), ("coef_hi", c_void_p), ("count", ctypes.c_double), ("C0", c_int32),
                ("C1", c_int32), ("N", c_int32), ("G", c_int32), ("hi_scale", ctypes.c_float), ("reps_lo", c_int32), ("reps_hi", c_int32), ("reserved", c_int32)]


class U3DPackDesc(ctypes.Structure):
    """mirror of u3d_pack_desc_t (include/u3d.h)"""

    _fields_ = [("w", c_void_p), ("packed", c_void_p), ("first", c_int64), ("Cout", c_int32), ("Cin", c_int32),
                ("mode", c_int32), ("cin_stride", c_int32)]


_PROTOS = {
    # name: (restype, argtypes)
    "u3d_version": (c_int, []),
    "u3d_debug_stream_pass": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_double]),
    "u3d_debug_mfma_f32_rate": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, POINTER(c_double)]),
    "u3d_last_error": (c_char_p, []),
    "u3d_check_device": (c_int, [c_int]),
    "u3d_set_tuning": (c_int, [c_int, c_int]),
    "u3d_set_profile_buffer": (c_int, [c_void_p, c_size_t]),
    "u3d_packed_weight_floats": (c_size_t, [c_int, c_int, c_int]),
    "u3d_pack_weights": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "u3d_pack_weights_batch": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int64]),
    "u3d_pack_weights_cells_blocks": (c_int64, [c_void_p, c_int, c_int, c_int, c_int]),
    "u3d_pack_weights_batch_cells": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int64]),
    "u3d_conv3d": (
        c_int,
        [c_int, c_void_p, POINTER(U3DSrc), c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
         POINTER(U3DSrc), c_void_p],
    ),
    "u3d_wgrad_workspace_floats": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "u3d_conv3d_variant": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "u3d_conv3d_wgrad_variant": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "u3d_conv3d_wgrad": (
        c_int,
        [c_int, c_void_p, POINTER(U3DSrc), c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t],
    ),
    "u3d_conv3d_small_cin_fwd": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    ),
    "u3d_conv3d_small_cin_fwd_reps": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int],
    ),
    "u3d_small_cin_bwd_workspace_floats": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "u3d_conv3d_small_cin_bwd": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
         c_int, c_void_p, c_size_t],
    ),
    "u3d_conv3d_naive": (
        c_int,
        [c_int, c_void_p, POINTER(U3DSrc), c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int],
    ),
    "u3d_chan_stats": (c_int, [c_int, c_void_p, POINTER(U3DSrc), c_int, c_int, c_int, c_int, c_void_p]),
    "u3d_chan_stats_reps": (c_int, [c_int, c_void_p, POINTER(U3DSrc), c_int, c_int, c_int, c_int, c_void_p, c_int]),
    "u3d_gn_finalize": (
        c_int,
        [c_int, c_void_p, c_void_p, c_int, c_double, c_void_p, c_int, c_double, c_int, c_int, c_double, c_void_p,
         c_void_p, c_float, c_void_p, c_void_p],
    ),
    "u3d_gn_bwd_finalize": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p],
    ),
    "u3d_gn_finalize_split": (
        c_int,
        [c_int, c_void_p, c_void_p, c_int, c_double, c_void_p, c_int, c_double, c_int, c_int, c_double, c_void_p,
         c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    ),
    "u3d_gn_bwd_finalize_split_supported": (c_int, [c_int, c_int, c_int]),
    "u3d_chan_stats_children": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "u3d_adam_step": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int64, c_double, c_double, c_double, c_double, c_double, c_int64]),
    "u3d_gn_bwd_finalize_split": (
        c_int,
        [c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_double, c_void_p, c_void_p,
         c_void_p, c_float, c_void_p],
    ),
    "u3d_gn_bwd_apply": (
        c_int,
        [c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_void_p],
    ),
    "u3d_gn_bwd_apply_up": (
        c_int,
        [c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
         c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    ),
    "u3d_maxpool2_fwd": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "u3d_maxpool2_bwd_merge": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
         c_int, c_void_p],
    ),
    "u3d_maxpool2_bwd_merge_gn": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int,
         c_int, c_int, c_int, c_int, c_void_p],
    ),
    "u3d_conv1x1_head_fwd": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p, c_void_p],
    ),
    "u3d_conv1x1_head_bwd": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p, c_void_p],
    ),
    "u3d_conv1x1_head_bwd_reps": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_int],
    ),
    "u3d_conv3d_wgrad_strided": (
        c_int,
        [c_int, c_void_p, POINTER(U3DSrc), c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t],
    ),
    "u3d_conv3d_ex_reps": (
        c_int,
        [c_int, c_void_p, POINTER(U3DSrc), c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, POINTER(U3DSrc),
         c_void_p, c_void_p, c_void_p, c_int64, c_int],
    ),
    "u3d_gn_finalize_reps": (
        c_int,
        [c_int, c_void_p, c_void_p, c_int, c_double, c_int, c_void_p, c_int, c_double, c_int, c_int, c_int, c_double, c_void_p, c_void_p,
         c_float, c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    ),
    "u3d_conv3d_wgrad_job_supported": (c_int, [c_int, c_int, c_int]),
    "u3d_conv3d_wgrad_job": (
        c_int,
        [c_int, c_void_p, POINTER(U3DSrc), c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t,
         POINTER(U3DGnBwdJob)],
    ),
    "u3d_subpixel_wgrad_workspace_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "u3d_subpixel_conv_wgrad": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
         c_void_p, c_int64],
    ),
    "u3d_convtr3d_subpixel_packed_floats": (c_int64, [c_int, c_int]),
    "u3d_pack_convtr3d_subpixel": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "u3d_convtr3d_fwd_subpixel": (
        c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int],
    ),
    "u3d_subpixel_packed_floats": (c_int64, [c_int, c_int]),
    "u3d_subpixel_dgrad_packed_floats": (c_int64, [c_int, c_int]),
    "u3d_pack_subpixel_dgrad_weights": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "u3d_subpixel_conv_dgrad": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int],
    ),
    "u3d_subpixel_conv_dgrad_reps": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int],
    ),
    "u3d_pack_subpixel_weights": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    # round 5: decoder levels that upsample n -> 2n + 1 (sub-pixel kernels on a window + the general kernels on the boundary slab)
    "u3d_subpixel_conv_fwd_win": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                          c_int, c_int, POINTER(c_int)]),
    "u3d_subpixel_conv_dgrad_win": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                            c_int, c_int, POINTER(c_int)]),
    "u3d_subpixel_conv_wgrad_win": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                            c_int, c_int, c_int, c_void_p, c_int64, POINTER(c_int)]),
    "u3d_conv3d_box": (c_int, [c_int, c_void_p, POINTER(U3DSrc), c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, POINTER(c_int),
                               POINTER(c_int)]),
    "u3d_conv3d_wgrad_box": (c_int, [c_int, c_void_p, POINTER(U3DSrc), c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                     c_size_t, POINTER(c_int)]),
    "u3d_gn_bwd_apply_children": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                          c_int, c_int, c_int, c_int, c_void_p]),
    "u3d_nearest_childsum_add": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                         c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int]),
    "u3d_subpixel_fwd_workspace_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "u3d_subpixel_conv_fwd": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
         c_void_p, c_int64],
    ),
    "u3d_conv3d_workspace_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "u3d_conv3d_ex": (
        c_int,
        [c_int, c_void_p, POINTER(U3DSrc), c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
         POINTER(U3DSrc), c_void_p, c_void_p, c_void_p, c_int64],
    ),
    "u3d_conv3d_residual": (
        c_int,
        [c_int, c_void_p, POINTER(U3DSrc), c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    ),
    "u3d_gn_bwd_apply_add": (
        c_int,
        [c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p],
    ),
    "u3d_conv1x1_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p]),
    "u3d_conv1x1_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "u3d_pack_convtr_weights": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "u3d_convtr3d_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "u3d_convtr3d_bwd": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
         c_void_p],
    ),
    "u3d_nearest_add_fwd": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
         c_int, c_void_p, c_void_p],
    ),
    "u3d_nearest_sum_bwd": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
         c_void_p],
    ),
    "u3d_se_gate_fwd": (
        c_int,
        [c_int, c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
         c_void_p],
    ),
    "u3d_se_apply_fwd": (
        c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "u3d_se_bwd_reduce": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p,
         c_void_p],
    ),
    "u3d_se_gate_bwd": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_void_p,
         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "u3d_se_bwd_apply": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int,
         c_int, c_void_p],
    ),
    "u3d_se_apply_fwd_b16": (
        c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "u3d_se_bwd_reduce_b16": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p,
         c_void_p],
    ),
    "u3d_se_bwd_apply_b16": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int,
         c_int, c_void_p],
    ),
    "u3d_bce_dice_scratch_doubles": (c_int64, [c_int, c_int, c_int64]),
    "u3d_bce_dice_fwd": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_float, c_float, c_float, c_void_p, c_void_p,
         c_void_p],
    ),
    "u3d_bce_dice_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "u3d_conv3d_bf16_supported": (c_int, [c_int, c_int]),
    "u3d_packed_weight_bf16_elems": (c_int64, [c_int, c_int, c_int]),
    "u3d_packed_weight_f32s_elems": (c_int64, [c_int, c_int, c_int]),
    "u3d_pack_weights_f32s": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "u3d_conv3d_f32s": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                                                 c_int64],
    ),
    "u3d_pack_weights_bf16": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    # bf16 activation storage (include/u3d.h, "_b16" entry points)
    "u3d_conv3d_bf16_ex_b16": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                       c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64]),
    "u3d_conv3d_wgrad_bf16_b16": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                          c_int, c_void_p, c_int64]),
    "u3d_convtr3d_fwd_t8_b16": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int]),
    "u3d_convtr3d_dgrad_t8_b16": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                          c_int]),
    "u3d_convtr3d_wgrad_t8_b16": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                          c_void_p, c_int64]),
    "u3d_conv1x1_fwd_b16": (c_int, [c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int,
                                    c_void_p]),
    "u3d_conv1x1_mfma_b16_supported": (c_int, [c_int, c_int]),
    "u3d_conv1x1_fwd_mfma_b16": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int,
                                         c_void_p]),
    "u3d_conv1x1_bwd_mfma_b16_workspace_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "u3d_conv1x1_bwd_mfma_b16": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_int64]),
    "u3d_conv1x1_bwd_b16": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_void_p,
                                    c_void_p]),
    "u3d_maxpool2_fwd_b16": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "u3d_maxpool2_bwd_merge_b16": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                           c_int, c_int, c_int, c_int, c_void_p]),
    "u3d_nearest_add_fwd_t8_b16": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                           c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "u3d_nearest_sum_bwd_t8_b16": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                           c_int, c_int, c_int, c_void_p]),
    "u3d_gn_bwd_apply_b16": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int,
                                     c_void_p, c_void_p]),
    "u3d_conv1x1_head_fwd_b16": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p,
                                         c_void_p]),
    "u3d_conv1x1_head_bwd_b16": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p,
                                         c_void_p]),
    "u3d_pack_weights_bf16_blocks": (c_int64, [c_int, c_int, c_int]),
    "u3d_pack_weights_bf16_batch": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int64]),
    "u3d_conv3d_bf16": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
         c_void_p, c_void_p, c_void_p],
    ),
    "u3d_conv3d_bf16_workspace_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "u3d_conv3d_bf16_tile_variant": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "u3d_conv3d_wgrad_bf16_b16_variant": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "u3d_conv3d_wgrad_bf16_job_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int]),
    "u3d_conv3d_wgrad_bf16_job": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                          c_int, c_void_p, c_int64, POINTER(U3DGnBwdJob)]),
    "u3d_conv3d_wgrad_bf16_b16_job": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                              c_int, c_void_p, c_int64, POINTER(U3DGnBwdJob)]),
    "u3d_conv3d_bf16_ex": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
         c_void_p, c_void_p, c_void_p, c_void_p, c_int64],
    ),
    "u3d_conv3d_wgrad_bf16_supported": (c_int, [c_int, c_int]),
    "u3d_wgrad_bf16_workspace_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "u3d_conv3d_wgrad_bf16": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64],
    ),
    "u3d_convtr3d_t8_supported": (c_int, [c_int, c_int]),
    "u3d_convtr3d_t8_packed_elems": (c_int64, [c_int, c_int, c_int]),
    "u3d_pack_convtr3d_t8": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "u3d_convtr3d_fwd_t8": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int]),
    "u3d_convtr3d_dgrad_t8": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int]),
    "u3d_convtr3d_dgrad_t8_workspace_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "u3d_convtr3d_fwd_t8_workspace_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "u3d_convtr3d_fwd_t8_b16_ex": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                           c_void_p, c_int64]),
    "u3d_convtr3d_dgrad_t8_ex": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                         c_void_p, c_int64]),
    "u3d_convtr3d_dgrad_t8_b16_ex": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                             c_void_p, c_int64]),
    "u3d_convtr3d_wgrad_t8_workspace_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "u3d_convtr3d_wgrad_t8": (
        c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64]),
    "u3d_nearest_add_fwd_t8": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
         c_int, c_void_p, c_void_p],
    ),
    "u3d_nearest_sum_bwd_t8": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
         c_void_p],
    ),
    "u3d_act_fwd": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "u3d_act_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "u3d_affine_act_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_float, c_void_p]),
    "u3d_affine_add_act_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_float, c_void_p]),
    "u3d_resample2_fwd": (c_int, [c_int, c_void_p] + [c_void_p] * 7 + [c_int] * 8 + [c_void_p]),
    "u3d_resample2_bwd": (c_int, [c_int, c_void_p] + [c_void_p] * 10 + [c_int] * 8 + [c_void_p]),
    "u3d_nearest_cat_fwd": (c_int, [c_int, c_void_p] + [c_void_p] * 5 + [c_int] * 9 + [c_void_p]),
    "u3d_split_channels": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "u3d_bn_finalize": (c_int, [c_int, c_void_p, c_void_p, c_int, c_double, c_void_p, c_int, c_double, c_int, c_double, c_void_p,
                                c_void_p, c_float, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "u3d_bn_bwd_finalize": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_double, c_int, c_void_p, c_void_p,
                                    c_void_p]),
    "u3d_bias_table": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "u3d_bias_grad": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "u3d_mul": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "u3d_pair_stats": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p]),
    "u3d_cvt_f64_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64]),
    "u3d_cvt_f64_f32_sum": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int]),
    "u3d_ncdhw_to_ndhwc": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64]),
    "u3d_ndhwc_to_ncdhw": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS.keys())

_lib = None
_lock = threading.Lock()
launch_count = 0  # number of native calls issued (tests assert the HIP path really ran)


class U3DError(RuntimeError):
    pass


def lib_available() -> bool:
    return os.path.exists(LIB_PATH)


def get_lib():
    """Load libu3d_hip.so (once).  Raises U3DError if it has not been built — there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise U3DError(
                f"native library {LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). The MI355X path has no CPU/PyTorch fallback."
            )
        # torch bundles its own HIP runtime (torch/lib/libamdhip64.so): it must be in the process BEFORE this
        # library is loaded so that both share ONE runtime (streams, allocations); loading ours first would pull
        # in /opt/rocm's copy as a second, unusable runtime.
        import torch  # noqa: F401

        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        if lib.u3d_version() < 128:
            raise U3DError("libu3d_hip.so is older than the Python host code")
        for kv in os.environ.get("U3D_TUNE", "").split(","):  # A/B knobs of u3d_set_tuning, e.g. U3D_TUNE=8:256,9:1 (results never change)
            if ":" in kv:
                k, v = kv.split(":")
                if lib.u3d_set_tuning(int(k), int(v)) != 0:
                    raise U3DError(f"U3D_TUNE: bad knob {kv!r}")
        _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != U3D_OK:
        msg = get_lib().u3d_last_error()
        raise U3DError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")


class EventProfiler:
    """Per-entry-point device time from HIP events recorded on the launching stream (bench.py's roofline leg).
    Usage: nat.profiler = EventProfiler(); ...run...; torch.cuda.synchronize(); prof.summary()."""

    def __init__(self, flops_only: bool = False, only=None, prealloc: int = 0):
        import torch

        # creating a timing event costs ~0.1-0.2 ms of HOST time the first time (hipEventCreate; torch only pools events it has seen
        # freed): a timed region that creates ~50 events per step becomes host-bound (measured: 3.3 -> 12.4 ms of host time per
        # step, 17.5 -> 20 ms per step).  `prealloc` events are therefore created up front, outside any timed region.
        self._pool = [torch.cuda.Event(enable_timing=True) for _ in range(int(prealloc))]
        for ev in self._pool:
            ev.record()  # torch creates the HIP event lazily, at the first record(): force it NOW (measured: with few warm-up steps the
                         # timed region otherwise paid ~75 ms of event creation, 17.4 -> 24 ms per step at --steps 10 --warmup 1..3)
        if self._pool:
            torch.cuda.synchronize()
        self.records = []  # (name, flops, start_event, end_event)
        self.only = set(only) if only else None  # bracket just these entry points (bench.py: the dominant family in the timed region)
        # flops_only: time only the MFMA families (calls that declare FLOPs) — two event packets per call cost ~1 us of
        # device time each, 0.7 ms per step when every one of the ~320 calls is bracketed
        self.flops_only = flops_only

    def wrap(self, name, fn, args, flops):
        import torch

        if (self.flops_only and flops <= 0.0) or (self.only is not None and name not in self.only):
            return fn(*args)
        st = self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True)
        en = self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True)
        st.record()
        rc = fn(*args)
        en.record()
        self.records.append((name, flops, st, en))
        return rc

    def summary(self):
        out = {}
        for name, flops, st, en in self.records:
            d = out.setdefault(name, {"calls": 0, "ms": 0.0, "flops": 0.0})
            d["calls"] += 1
            d["ms"] += st.elapsed_time(en)
            d["flops"] += flops
        return out


profiler = None


def call(name: str, *args, flops: float = 0.0) -> None:
    """Call an int-returning entry point and raise on error."""
    global launch_count
    fn = getattr(get_lib(), name)
    launch_count += 1
    rc = fn(*args) if profiler is None else profiler.wrap(name, fn, args, flops)
    if rc != U3D_OK:
        check(rc, name)
