"""ctypes binding of libdpig_hip.so (the C ABI declared in include/dpig_hip.h).

The library is the product: there is NO fallback.  If the shared object is missing or a call
returns a non-zero status a RuntimeError is raised (mirroring the reference's `raise Exception`
convention, e.g. tflib/ops/deconv2d.py:38-39).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DPIG_LIB_PATH") or os.path.join(_HERE, "libdpig_hip.so")   # (override: A/B kernel experiments)

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2

c_float_p = ctypes.c_void_p  # device pointers travel as integers


class DpigConvDesc(ctypes.Structure):
    _fields_ = [
        ("N", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("C", ctypes.c_int32),
        ("K", ctypes.c_int32), ("R", ctypes.c_int32), ("S", ctypes.c_int32), ("stride", ctypes.c_int32),
        ("pad_t", ctypes.c_int32), ("pad_l", ctypes.c_int32),
        ("ldx", ctypes.c_int32), ("ldy", ctypes.c_int32), ("ldres", ctypes.c_int32), ("ldmask", ctypes.c_int32),
        ("act", ctypes.c_int32), ("alpha", ctypes.c_float),
        ("upsample2x", ctypes.c_int32), ("res_after_act", ctypes.c_int32), ("ldy2", ctypes.c_int32),
        ("res_class", ctypes.c_int32), ("split_k", ctypes.c_int32), ("compute", ctypes.c_int32),
    ]


class DpigCriticDesc(ctypes.Structure):
    _fields_ = [
        ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("Cin", ctypes.c_int32),
        ("dim", ctypes.c_int32), ("lrelu_alpha", ctypes.c_float), ("ln_eps", ctypes.c_float), ("lam", ctypes.c_float),
        ("compute", ctypes.c_int32),
    ]


class DpigCriticParams(ctypes.Structure):      # DpigCriticGrads has the same layout (non-const pointers)
    _fields_ = [
        ("w", ctypes.c_void_p * 4), ("b", ctypes.c_void_p * 4), ("ln_scale", ctypes.c_void_p * 3),
        ("ln_offset", ctypes.c_void_p * 3), ("w_out", ctypes.c_void_p),
    ]


# every symbol include/dpig_hip.h declares: name -> (restype, argtypes)
_vp, _i, _f, _i64, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64, ctypes.c_size_t
_dp = ctypes.POINTER(DpigConvDesc)
SYMBOLS = {
    "dpig_version": (_i, []),
    "dpig_last_error": (ctypes.c_char_p, []),
    "dpig_same_pad": (_i, [_i, _i, _i, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "dpig_conv2d_workspace_bytes": (_sz, [_dp, _i]),
    "dpig_conv2d_fwd": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpig_conv2d_dgrad": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpig_conv2d_wgrad": (_i, [_dp, _vp, _vp, _vp, _f, _vp, _f, _vp, _sz, _vp]),
    "dpig_conv2d_bf16_supported": (_i, [_dp, _i]),
    "dpig_conv2d_bf16_workspace_bytes": (_sz, [_dp, _i]),
    "dpig_conv2d_fwd_bf16": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpig_conv2d_dgrad_bf16": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpig_conv2d_wgrad_bf16": (_i, [_dp, _vp, _vp, _vp, _f, _vp, _f, _vp, _sz, _vp]),
    "dpig_conv_bf16_set_large_tile": (_i, [_i, _i]),
    "dpig_conv_bf16_set_large_tile_wgrad": (_i, [_i, _i]),
    "dpig_conv_bf16_set_wave8": (_i, [_i]),
    "dpig_wino_filter_elems": (_sz, [_i, _i]),
    "dpig_wino_filter_transform": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "dpig_wino_filter_jobs_plan": (_i, [_vp, _i]),
    "dpig_wino_filter_transform_jobs": (_i, [_vp, _i, _i, _vp]),
    "dpig_conv2d_wino_eligible": (_i, [_dp, _i]),
    "dpig_conv_wino_set_mode": (_i, [_i]),
    "dpig_conv_wino_get_mode": (_i, []),
    "dpig_conv2d_wino_workspace_bytes": (_sz, [_dp, _i]),
    "dpig_conv2d_fwd_wino": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpig_conv2d_dgrad_wino": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpig_wino4_filter_elems": (_sz, [_i, _i]),
    "dpig_wino4_filter_transform": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "dpig_wino4_filter_transform_jobs": (_i, [_vp, _i, _i, _vp]),
    "dpig_conv2d_wino4_eligible": (_i, [_dp, _i]),
    "dpig_conv_wino4_set_mode": (_i, [_i]),
    "dpig_conv_wino4_get_mode": (_i, []),
    "dpig_conv2d_wino4_workspace_bytes": (_sz, [_dp, _i]),
    "dpig_conv2d_fwd_wino4": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpig_conv2d_dgrad_wino4": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpig_conv2d_wgrad_wino_eligible": (_i, [_dp]),
    "dpig_conv2d_wgrad_wino_workspace_bytes": (_sz, [_dp]),
    "dpig_conv2d_wgrad_wino": (_i, [_dp, _vp, _vp, _vp, _f, _vp, _f, _vp, _sz, _vp]),
    "dpig_conv2d_fwd_thin_bf16": (_i, [_dp, _vp, _vp, _vp, _vp, _vp]),
    "dpig_conv2d_dgrad_thin_bf16": (_i, [_dp, _vp, _vp, _vp, _vp]),
    "dpig_conv2d_wgrad_thin_bf16": (_i, [_dp, _vp, _vp, _vp, _f, _vp, _f, _vp, _sz, _vp]),
    "dpig_cvt_f32_to_bf16": (_i, [_vp, _i, _vp, _i, _i64, _i, _vp]),
    "dpig_cvt_bf16_to_f32": (_i, [_vp, _i, _vp, _i, _i64, _i, _vp]),
    "dpig_cvt_f32_to_bf16_pad": (_i, [_vp, _i, _i, _vp, _i, _i, _i64, _vp]),
    "dpig_act_fwd_bf16": (_i, [_vp, _i, _vp, _i, _i64, _i, _i, _f, _vp]),
    "dpig_act_bwd_bf16": (_i, [_vp, _i, _vp, _i, _vp, _i, _i64, _i, _i, _f, _vp]),
    "dpig_filter_shadow_bf16_multi": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "dpig_filter_shadow_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "dpig_filter_shadow_split": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "dpig_filter_shadow_split_multi": (_i, [_vp, _vp, _vp, _i64, _vp, _i, _i, _vp]),
    "dpig_conv2d_fwd_x3": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpig_conv2d_dgrad_x3": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpig_conv2d_wgrad_x3": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _f, _vp, _sz, _vp]),
    "dpig_act_bwd_s32": (_i, [_vp, _i, _vp, _i, _vp, _i, _i64, _i, _i, _f, _vp, _vp]),
    "dpig_split32_bytes": (_sz, [_i64, _i]),
    "dpig_split32": (_i, [_vp, _i, _i64, _i, _vp, _vp]),
    "dpig_act_fwd": (_i, [_vp, _i, _vp, _i, _i64, _i, _i, _f, _vp]),
    "dpig_act_bwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _i64, _i, _i, _f, _vp]),
    "dpig_colsum_workspace_bytes": (_sz, [_i64, _i]),
    "dpig_colsum": (_i, [_vp, _i, _i64, _i, _vp, _f, _vp, _sz, _vp]),
    "dpig_border_class_sum_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "dpig_border_class_sum": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "dpig_border_class_sum_bf16": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "dpig_conv2d_bn_stats_tiles": (_i, [_dp]),
    "dpig_conv2d_fwd_stats": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dpig_conv2d_bn_stats_tiles_ws": (_i, [_dp]),
    "dpig_conv2d_fwd_stats_ws": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpig_bn_stats_finalize": (_i, [_vp, _i, _i64, _i, _i, _f, _vp, _vp, _vp]),
    "dpig_conv2d_bf16_bn_stats_tiles": (_i, [_dp]),
    "dpig_conv2d_fwd_bf16_stats": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dpig_bn_workspace_bytes": (_sz, [_i64, _i]),
    "dpig_bn_fwd": (_i, [_vp, _i, _i64, _i, _vp, _vp, _f, _i, _f, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "dpig_bn_bwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _i64, _i, _vp, _vp, _vp, _i, _f, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "dpig_bn_bf16_workspace_bytes": (_sz, [_i64, _i]),
    "dpig_bn_fwd_bf16": (_i, [_vp, _i, _i64, _i, _vp, _vp, _f, _i, _f, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "dpig_bn_apply_bf16": (_i, [_vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _i, _f, _vp, _i, _vp]),
    "dpig_bn_bwd_bf16": (_i, [_vp, _i, _vp, _i, _vp, _i, _i64, _i, _vp, _vp, _vp, _i, _f, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "dpig_bn_sqdev": (_i, [_vp, _i, _i64, _i, _vp, _vp, _vp, _sz, _vp]),
    "dpig_bn_apply": (_i, [_vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _i, _f, _vp, _i, _vp]),
    "dpig_bn_bwd_sums": (_i, [_vp, _i, _vp, _i, _vp, _i, _i64, _i, _vp, _vp, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    "dpig_bn_bwd_apply": (_i, [_vp, _i, _vp, _i, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _i, _f, _f, _vp, _i, _vp]),
    "dpig_ln_fwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "dpig_ln_fwd": (_i, [_vp, _i, _i, _i, _vp, _vp, _f, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpig_ln_fwd_bf16": (_i, [_vp, _i, _i, _i, _vp, _vp, _f, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpig_ln_workspace_bytes": (_sz, [_i, _i, _i]),
    "dpig_ln_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpig_ln_bwd_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpig_ln_bwd2_workspace_bytes": (_sz, [_i, _i, _i]),
    "dpig_ln_bwd2": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpig_ln_bwd2_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpig_linear_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "dpig_linear_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _sz, _vp]),
    "dpig_linear_dgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "dpig_linear_wgrad": (_i, [_vp, _vp, _vp, _f, _i, _i, _i, _vp, _sz, _vp]),
    "dpig_crop_resize_fwd": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "dpig_crop_resize_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "dpig_crop_resize_bwd": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "dpig_crop_resize_fwd_bf16": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "dpig_crop_resize_bwd_bf16": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "dpig_pose_points": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "dpig_pose_inflate": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "dpig_pose_rasterize": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "dpig_ssim_workspace_bytes": (_sz, [_i, _i, _i]),
    "dpig_ssim_gray_u8": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "dpig_gp_interpolate": (_i, [_vp, _vp, _vp, _i, _i64, _vp, _vp]),
    "dpig_gp_penalty_workspace_bytes": (_sz, [_i, _i64]),
    "dpig_gp_penalty": (_i, [_vp, _i, _i64, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpig_gp_double_backward_workspace_bytes": (_sz, [ctypes.POINTER(DpigCriticDesc)]),
    "dpig_gp_double_backward_slot": (_i, [ctypes.POINTER(DpigCriticDesc), _i, _i, ctypes.POINTER(_sz), ctypes.POINTER(_sz)]),
    "dpig_gp_double_backward": (_i, [ctypes.POINTER(DpigCriticDesc), ctypes.POINTER(DpigCriticParams), _vp, _vp, _vp, _f,
                                     ctypes.POINTER(DpigCriticParams), _vp, _vp, _vp, _sz, _vp]),
    "dpig_upsample2x_fwd": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "dpig_upsample2x_bwd": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "dpig_adam_step": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _i, _f, _vp]),
    "dpig_adam_step_dev": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _f, _f, _f, _f, _vp]),
    "dpig_adam_multi": (_i, [_vp, _vp, _i, _i64, _vp, _f, _f, _f, _i, _f, _vp]),
    "dpig_rmsprop_step": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f, _vp]),
    "dpig_clip": (_i, [_vp, _i64, _f, _f, _vp]),
    "dpig_mask_split_fwd": (_i, [_vp, _i, _vp, _i64, _i, _vp, _i, _vp, _i, _i, _vp]),
    "dpig_mask_split_bwd": (_i, [_vp, _i, _vp, _i, _vp, _i64, _i, _vp, _i, _i, _vp]),
    "dpig_roi_boxes": (_i, [_vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp]),
    "dpig_vis_concat_fwd": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "dpig_vis_concat_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "dpig_emb_class_weights_fwd": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "dpig_emb_class_weights_bwd": (_i, [_vp, _i, _i, _i, _vp, _f, _vp]),
    "dpig_axpby3d": (_i, [_vp, _i64, _i64, _vp, _i64, _i64, _i, _i, _i, _f, _vp]),
    "dpig_transpose12": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "dpig_act_bwd_pool2x": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "dpig_pose_stem_fwd": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _i, _vp]),
    "dpig_pose_stem_wgrad": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    "dpig_sce_mean": (_i, [_vp, _i, _f, _vp, _vp, _f, _vp]),
    "dpig_logit_mean": (_i, [_vp, _i, _i, _f, _vp, _vp, _f, _vp]),
    "dpig_l1_workspace_bytes": (_sz, [_i64]),
    "dpig_l1_mean": (_i, [_vp, _vp, _i64, _vp, _vp, _f, _vp, _sz, _vp]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle.  Fails loudly when the extension is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libdpig_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(h, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().dpig_last_error()
        raise RuntimeError("libdpig_hip %s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))


def ptr(t):
    """Device pointer of a tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


class _Workspace:
    """Grow-only scratch buffer per device and stream; the C ABI never allocates (SURVEY 8b ownership row)."""

    def __init__(self):
        self.buf = {}
        self.pinned = False      # a captured hipGraph has the buffer's address baked in (split-K partials, ...)
        self.retired = []

    def pin(self):
        """Called once a graph has been captured: from now on a buffer that has to grow is RETIRED, not freed -- graph
        replays keep writing to the old address, which must not be handed to anybody else by the caching allocator."""
        self.pinned = True

    def get(self, nbytes, device):
        if nbytes == 0:
            return None, 0
        # one buffer per (device, stream): launches on different streams (autograd.side_branch) may run concurrently
        key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
        b = self.buf.get(key)
        if b is None or b.numel() < nbytes:
            if b is not None and self.pinned:
                self.retired.append(b)
            # round up generously: reallocation is a sync point for the caching allocator
            size = max(int(nbytes * 1.25), 64 << 20)
            b = torch.empty(size, dtype=torch.uint8, device=device)
            self.buf[key] = b
        return b, b.numel()


workspace = _Workspace()


def same_pad(inp, k, stride):
    """TF 'SAME': (out, pad_before) -- tflib/ops/conv2d.py:110 semantics (host-side mirror)."""
    out = -(-inp // stride)
    total = max((out - 1) * stride + k - inp, 0)
    return out, total // 2
