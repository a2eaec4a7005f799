"""ctypes binding of ``libdir_hip.so`` (C-ABI declared in ``include/dir_hip.h``).

There is exactly one compute path: the hand-written gfx950 kernels in this library. If the
library is missing or an entry point fails, the caller gets an exception — there is no eager /
CPU fallback anywhere in the package.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdir_hip.so")

c_int, c_float, c_double, c_void_p, c_size_t, c_int64 = (
    ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int64)
c_longlong = ctypes.c_longlong
DIR_EUNSUPPORTED = -2

# name -> (restype, argtypes); must list every symbol of include/dir_hip.h
SIGNATURES = {
    "dir_abi_version": (c_int, []),
    "dir_error_string": (ctypes.c_char_p, [c_int]),
    "dir_fds_label_flags": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dir_fds_assign_bins": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dir_fds_bin_index": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dir_fds_scatter_stats_workspace": (c_size_t, [c_int, c_int, c_int]),
    "dir_fds_scatter_stats": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_size_t, c_void_p]),
    "dir_fds_finalize_update": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_double,
                                        c_void_p, c_void_p, c_void_p, c_void_p]),
    "dir_fds_finalize_update_groups": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_double,
                                       c_void_p, c_void_p, c_void_p, c_void_p]),
    "dir_fds_smooth_bins": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dir_fds_prepare_scale": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_void_p, c_void_p]),
    "dir_fds_calibrate_fwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dir_fds_calibrate_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "dir_fds_calibrate_fwd_lds": (c_int, [c_void_p, c_int, c_void_p, c_longlong, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dir_fds_calibrate_fwd_nchw": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_longlong, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dir_fds_calibrate_bwd_nchw": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_longlong, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dir_fds_smooth_fwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p]),
    "dir_weighted_loss_workspace": (c_size_t, [c_int]),
    "dir_weighted_loss": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_int,
                                  c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dir_scale_by_device_scalar": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "dir_bn_workspace": (c_size_t, [c_int, c_int64, c_int]),
    "dir_bn_fwd_train": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_double, c_double, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dir_bn_fwd_train_partials": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_double, c_double, c_int, c_void_p, c_void_p,
                                          c_void_p, c_size_t, c_void_p]),
    "dir_bn_fwd_eval": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_double, c_int, c_void_p, c_size_t, c_void_p]),
    "dir_bn_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p,
                           c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "dir_bn_bwd_partials": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "dir_bn_bwd_join": (c_int, [c_void_p] * 5 + [c_int, c_int64, c_int] + [c_void_p] * 11 + [c_size_t, c_void_p]),
    "dir_conv_dgrad_ex": (c_int, [c_void_p] * 7 + [c_int] * 8 + [c_void_p] * 6 + [c_int, c_int, c_void_p]),
    "dir_bn_fwd_train_bits": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dir_bn_apply_bits": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "dir_conv_dgrad_bnstats": (c_int, [c_void_p] * 6 + [c_int] * 8 + [c_void_p] * 6 + [c_int, c_void_p]),
    "dir_conv_dgrad_s2_bnstats": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p] * 6 + [c_int, c_void_p]),
    "dir_conv_dgrad_s2_ex": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p] * 6 + [c_int, c_int, c_void_p]),
    "dir_conv_plan_rows": (c_size_t, [c_int] * 11),
    "dir_bn_prepare_train": (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dir_bn_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_void_p]),
    "dir_conv_stats_rows": (c_size_t, [c_int, c_int, c_int]),
    "dir_adam_step": (c_int, [c_void_p, c_int, c_double, c_double, c_double, c_double, c_double, c_longlong, c_void_p]),
    "dir_sgd_step": (c_int, [c_void_p, c_int, c_double, c_double, c_double, c_double, c_int, c_int, c_void_p]),
    "dir_conv_prep_weights": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dir_conv_prep_weights_batched": (c_int, [c_void_p, c_int, c_void_p]),
    "dir_conv_prep_weights_ex": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "dir_conv_dgrad_s2": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dir_conv_fwd_fused": (c_int, [c_void_p] * 6 + [c_int] * 10 + [c_void_p]),
    "dir_stem_conv_stats_rows": (c_size_t, [c_int, c_int]),
    "dir_stem_conv_prep_weights": (c_int, [c_void_p, c_void_p, c_void_p]),
    "dir_stem_conv_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dir_stem_conv_wgrad_workspace": (c_size_t, [c_int, c_int]),
    "dir_stem_conv_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "dir_conv_dgrad_join": (c_int, [c_void_p] * 6 + [c_int] * 8 + [c_void_p]),
    "dir_conv_fwd": (c_int, [c_void_p] * 4 + [c_int] * 10 + [c_void_p]),
    "dir_conv_fwd_variant": (c_int, [c_void_p] * 4 + [c_int] * 11 + [c_void_p]),
    "dir_conv_wgrad_workspace": (c_size_t, [c_int] * 10),
    "dir_conv_wgrad": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 10 + [c_void_p, c_size_t, c_void_p]),
    "dir_conv_wgrad3x3_workspace": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "dir_conv_wgrad3x3": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "dir_conv_wgrad_reduce_splits": (c_int, [c_void_p, c_int, c_size_t, c_void_p, c_void_p]),
    "dir_conv_wgrad_partials": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 10 + [c_void_p, c_size_t, c_void_p]),
    "dir_conv_wgrad3x3_partials": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "dir_conv_wgrad_reduce_batched": (c_int, [c_void_p, c_int, c_void_p]),
    "dir_augment_u8": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dir_resize_ksize": (c_int, [c_int, c_int]),
    "dir_resize_u8_workspace": (c_size_t, [c_int, c_int, c_int, c_size_t]),
    "dir_resize_u8": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "dir_maxpool3x3s2_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dir_maxpool3x3s2_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dir_bn_relu_maxpool_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dir_bn_relu_maxpool_fwd_xmax": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dir_bn_relu_maxpool_bwd_xmax": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dir_bn_relu_maxpool_bwd_workspace": (c_size_t, [c_int]),
    "dir_bn_relu_maxpool_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dir_bn_bwd_finalize": (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p]),
    "dir_avgpool_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dir_avgpool_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dir_fds_bin_edges": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dir_fds_fill_empty_buckets": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dir_fds_prepare_scale_ex": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_int, c_void_p, c_void_p]),
    "dir_fds_bin_scaled": (c_int, [c_void_p, ctypes.c_longlong, c_float, c_int, c_int, c_void_p, c_void_p]),
    "dir_conv_f32_fwd": (c_int, [c_void_p] * 3 + [c_int] * 9 + [c_void_p]),
    "dir_conv_f32_dgrad": (c_int, [c_void_p] * 3 + [c_int] * 9 + [c_void_p]),
    "dir_conv_f32_dgrad_fused": (c_int, [c_void_p] * 6 + [c_int] * 9 + [c_void_p]),
    "dir_conv_f32_wgrad_workspace": (c_size_t, [c_int] * 9),
    "dir_conv_f32_wgrad": (c_int, [c_void_p] * 3 + [c_int] * 9 + [c_void_p, c_size_t, c_void_p]),
    "dir_conv_f32_fwd_variant": (c_int, [c_void_p] * 3 + [c_int] * 10 + [c_void_p]),
    "dir_conv_f32_stats_rows": (c_size_t, [c_int] * 3),
    "dir_conv_f32_fwd_stats": (c_int, [c_void_p] * 4 + [c_int] * 10 + [c_void_p]),
    "dir_conv_f32_fwd_stats_variant": (c_int, [c_void_p] * 4 + [c_int] * 11 + [c_void_p]),
    "dir_conv_f32_dgrad_variant": (c_int, [c_void_p] * 6 + [c_int] * 10 + [c_void_p]),
    "dir_conv_f32_wgrad_variant": (c_int, [c_void_p] * 3 + [c_int] * 9 + [c_void_p, c_size_t, c_int, c_void_p]),
    "dir_maxpool3x3s2_f32_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dir_maxpool3x3s2_f32_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dir_avgpool_f32_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dir_avgpool_f32_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dir_tail_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int] + [c_void_p] * 9),
    "dir_tail_bwd_workspace": (c_size_t, [c_int, c_int]),
    "dir_tail_bwd": (c_int, [c_void_p] * 6 + [c_int] * 4 + [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dir_lds_weights": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
}

ABI_VERSION = 4
DIR_F32, DIR_BF16 = 0, 1
WGRAD_AUTO, WGRAD_TRANSPOSE, WGRAD_DMA1, WGRAD_DMA2 = 0, 1, 2, 3                             # DIR_WGRAD_*
CONV_AUTO, CONV_TILE_REG, CONV_TILE_DMA, CONV_PATCH3, CONV_BIG = 0, 1, 2, 3, 5      # DIR_CONV_* of include/dir_hip.h
FLAG_HAS_LO, FLAG_HAS_HI, FLAG_NONINTEGER, FLAG_NAN = 1, 2, 4, 8
FACTOR_ZERO, FACTOR_MOMENTUM, FACTOR_COUNT = 0, 1, 2
LOSS_KINDS = {"mse": 0, "l1": 1, "focal_mse": 2, "focal_l1": 3, "huber": 4}
REWEIGHT = {"sqrt_inv": 1, "inverse": 2}

_lib = None


class DirHipError(RuntimeError):
    pass


def lib():
    """Load the shared library (once). Raises if it has not been built — never falls back."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise DirHipError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py` "
                f"(or `make -C imbalanced-regression_amd/csrc`). There is no fallback path.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError = ABI mismatch, fail loudly
            fn.restype = res
            fn.argtypes = args
        if handle.dir_abi_version() != ABI_VERSION:
            raise DirHipError(f"ABI version mismatch: {handle.dir_abi_version()}")
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().dir_error_string(rc).decode()
        if rc > 0:
            msg += f" [hipError_t {rc}]"
        raise DirHipError(f"{what} failed: {msg}")


def stream_ptr(device=None):
    """Raw hipStream_t of torch's current stream (kernels are enqueued where torch's work is). The C-ABI launches on the
    CURRENT device: tensors that live on another GPU are refused here instead of being launched on the wrong one
    (one process per GPU is the design; wrap foreign-device work in ``torch.cuda.device(t.device)``)."""
    if device is not None:
        idx = device.index if isinstance(device, torch.device) else int(device)
        if idx is not None and idx != torch.cuda.current_device():
            raise DirHipError(f"tensors on cuda:{idx} but the current device is cuda:{torch.cuda.current_device()}: "
                              f"call torch.cuda.set_device / use torch.cuda.device(...) around the call")
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def require_device_tensor(t, dtype, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise DirHipError(
            f"{name} is on {t.device}: the FDS/LDS hot path runs only as HIP kernels on an AMD GPU "
            f"(no CPU fallback). Move the tensor to the GPU.")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t
