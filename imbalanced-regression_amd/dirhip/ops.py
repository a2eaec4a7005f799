"""Thin tensor-level wrappers over the C-ABI (one function per entry point of dir_hip.h).

torch is used here for device memory and streams only; all arithmetic on the hot path happens
inside ``libdir_hip.so``.
"""
import ctypes

import numpy as np
import torch

from . import _lib as L

f32, f64, i32 = torch.float32, torch.float64, torch.int32


def _ws(nbytes, device):
    # torch's caching allocator returns >=256-B aligned blocks
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ---- K1 -------------------------------------------------------------------------------------------
def label_flags(labels, bucket_start, bucket_num, flags=None):
    labels = L.require_device_tensor(labels, f32, "labels")
    if flags is None:
        flags = torch.zeros(1, dtype=i32, device=labels.device)
    L.check(L.lib().dir_fds_label_flags(L.ptr(labels), labels.numel(), bucket_start, bucket_num,
                                        L.ptr(flags), L.stream_ptr(labels.device)), "dir_fds_label_flags")
    return flags


def assign_bins(labels, bucket_start, bucket_num, flags):
    labels = L.require_device_tensor(labels, f32, "labels")
    bins = torch.empty(labels.numel(), dtype=i32, device=labels.device)
    L.check(L.lib().dir_fds_assign_bins(L.ptr(labels), labels.numel(), bucket_start, bucket_num,
                                        L.ptr(flags), L.ptr(bins), L.stream_ptr(labels.device)), "dir_fds_assign_bins")
    return bins


def bin_index(labels, bucket_start, bucket_num):
    """labels [n] f32 -> (bins [n] int32, flags [1] int32)."""
    labels = L.require_device_tensor(labels, f32, "labels")
    bins = torch.empty(labels.numel(), dtype=i32, device=labels.device)
    flags = torch.empty(1, dtype=i32, device=labels.device)
    L.check(L.lib().dir_fds_bin_index(L.ptr(labels), labels.numel(), bucket_start, bucket_num,
                                      L.ptr(bins), L.ptr(flags), L.stream_ptr(labels.device)), "dir_fds_bin_index")
    return bins, flags


def bin_edges(labels, edges, bucket_start, bucket_num):
    """STS-B variant: labels [n] f32, edges [nedges] f32 (device) -> table rows [n] int32 (-1 = skipped)."""
    labels = L.require_device_tensor(labels, f32, "labels")
    edges = L.require_device_tensor(edges, f32, "edges")
    bins = torch.empty(labels.numel(), dtype=i32, device=labels.device)
    L.check(L.lib().dir_fds_bin_edges(L.ptr(labels), labels.numel(), L.ptr(edges), edges.numel(), bucket_start, bucket_num,
                                      L.ptr(bins), L.stream_ptr(labels.device)), "dir_fds_bin_edges")
    return bins


def bin_scaled(labels, mult, bucket_start, bucket_num):
    """NYUD2 variant: labels [n] f32 -> table rows clamp(int(label * mult), start, num - 1) - start (int32)."""
    labels = L.require_device_tensor(labels, f32, "labels")
    bins = torch.empty(labels.numel(), dtype=i32, device=labels.device)
    L.check(L.lib().dir_fds_bin_scaled(L.ptr(labels), labels.numel(), float(mult), bucket_start, bucket_num, L.ptr(bins),
                                       L.stream_ptr(labels.device)), "dir_fds_bin_scaled")
    return bins


def fill_empty_buckets(count, running_mean, running_var):
    nb, c = running_mean.shape
    L.check(L.lib().dir_fds_fill_empty_buckets(L.ptr(count), nb, c, L.ptr(running_mean), L.ptr(running_var),
                                               L.stream_ptr(running_mean.device)), "dir_fds_fill_empty_buckets")


# ---- K2 -------------------------------------------------------------------------------------------
def scatter_stats(feats, bins, nb):
    """feats [n, C] f32, bins [n] int32 -> count [nb], mean [nb, C], m2 [nb, C] (all f64)."""
    feats = L.require_device_tensor(feats, f32, "features")
    bins = L.require_device_tensor(bins, i32, "bins")
    n, c = feats.shape
    dev = feats.device
    count = torch.empty(nb, dtype=f64, device=dev)
    mean = torch.empty(nb, c, dtype=f64, device=dev)
    m2 = torch.empty(nb, c, dtype=f64, device=dev)
    nbytes = L.lib().dir_fds_scatter_stats_workspace(n, c, nb)
    ws = _ws(nbytes, dev)
    L.check(L.lib().dir_fds_scatter_stats(L.ptr(feats), L.DIR_F32, L.ptr(bins), n, c, nb, L.ptr(count),
                                          L.ptr(mean), L.ptr(m2), L.ptr(ws), ws.numel(), L.stream_ptr(dev)),
            "dir_fds_scatter_stats")
    return count, mean, m2


# ---- K3 -------------------------------------------------------------------------------------------
def finalize_update(count, mean, m2, factor_mode, momentum, running_mean, running_var, tracked):
    nb, c = running_mean.shape
    for t, dt, nm in ((count, f64, "count"), (mean, f64, "mean"), (m2, f64, "m2"), (running_mean, f32, "running_mean"),
                      (running_var, f32, "running_var"), (tracked, f32, "num_samples_tracked")):
        L.require_device_tensor(t, dt, nm)
    L.check(L.lib().dir_fds_finalize_update(L.ptr(count), L.ptr(mean), L.ptr(m2), nb, c, factor_mode,
                                            float(momentum), L.ptr(running_mean), L.ptr(running_var),
                                            L.ptr(tracked), L.stream_ptr(running_mean.device)),
            "dir_fds_finalize_update")


def finalize_update_groups(count, mean, m2, bin_ptr, factor_mode, momentum, running_mean, running_var, tracked):
    """K3 for non-integer labels: statistics per distinct-value group ([U], [U, C] f64), ``bin_ptr`` [nb + 1] int32 = first group of every bin."""
    nb, c = running_mean.shape
    for t, dt, nm in ((count, f64, "count"), (mean, f64, "mean"), (m2, f64, "m2"), (bin_ptr, i32, "bin_ptr"), (running_mean, f32, "running_mean"),
                      (running_var, f32, "running_var"), (tracked, f32, "num_samples_tracked")):
        L.require_device_tensor(t, dt, nm)
    assert bin_ptr.numel() == nb + 1 and mean.shape == (count.numel(), c)
    L.check(L.lib().dir_fds_finalize_update_groups(L.ptr(count), L.ptr(mean), L.ptr(m2), count.numel(), c, L.ptr(bin_ptr), nb, factor_mode,
                                                   float(momentum), L.ptr(running_mean), L.ptr(running_var), L.ptr(tracked),
                                                   L.stream_ptr(running_mean.device)), "dir_fds_finalize_update_groups")


# ---- K4 -------------------------------------------------------------------------------------------
def smooth_bins(mean, var, window):
    mean = L.require_device_tensor(mean, f32, "mean table")
    var = L.require_device_tensor(var, f32, "var table")
    window = L.require_device_tensor(window, f32, "kernel_window")
    nb, c = mean.shape
    smean, svar = torch.empty_like(mean), torch.empty_like(var)
    L.check(L.lib().dir_fds_smooth_bins(L.ptr(mean), L.ptr(var), L.ptr(window), window.numel(), nb, c,
                                        L.ptr(smean), L.ptr(svar), L.stream_ptr(mean.device)), "dir_fds_smooth_bins")
    return smean, svar


# ---- K5a / K5 / K6 --------------------------------------------------------------------------------
def prepare_scale(v1, v2, clip_min=0.1, clip_max=10.0, out=None, guard_mode=0):
    """guard_mode 0: age variant (untouched where v1 == 0); 1: STS-B / NYUD2 (untouched where v1 <= 0 or v2 < 0)."""
    v1 = L.require_device_tensor(v1, f32, "v1")
    v2 = L.require_device_tensor(v2, f32, "v2")
    nb, c = v1.shape
    scale = torch.empty_like(v1) if out is None else out
    if guard_mode == 0:
        L.check(L.lib().dir_fds_prepare_scale(L.ptr(v1), L.ptr(v2), nb, c, clip_min, clip_max, L.ptr(scale),
                                              L.stream_ptr(v1.device)), "dir_fds_prepare_scale")
    else:
        L.check(L.lib().dir_fds_prepare_scale_ex(L.ptr(v1), L.ptr(v2), nb, c, clip_min, clip_max, guard_mode, L.ptr(scale),
                                                 L.stream_ptr(v1.device)), "dir_fds_prepare_scale_ex")
    return scale


def calibrate_fwd_(x, bins, m1, scale, m2):
    b, c = x.shape
    if b >= LDS_STAGED_MIN_ROWS:
        return calibrate_fwd_lds_(x, bins, m1, scale, m2)
    L.check(L.lib().dir_fds_calibrate_fwd(L.ptr(x), L.DIR_F32, L.ptr(bins), b, c, L.ptr(m1), L.ptr(scale),
                                          L.ptr(m2), L.stream_ptr(x.device)), "dir_fds_calibrate_fwd")
    return x


LDS_STAGED_MIN_ROWS = 4096     # from here on the tables are staged in LDS (dir_fds_calibrate_fwd_lds); below, the launch dominates


def calibrate_fwd_lds_(x, bins, m1, scale, m2):
    """In place on x [B, C] with the three [nb, C] tables staged in LDS (any B; meant for B >= LDS_STAGED_MIN_ROWS)."""
    b, c = x.shape
    L.check(L.lib().dir_fds_calibrate_fwd_lds(L.ptr(x), L.DIR_F32, L.ptr(bins), b, c, m1.shape[0], L.ptr(m1), L.ptr(scale),
                                              L.ptr(m2), L.stream_ptr(x.device)), "dir_fds_calibrate_fwd_lds")
    return x


def calibrate_nchw(x, bins, m1, scale, m2, out=None):
    """NYUD2 dense variant on an NCHW float32 map x [N, C, H, W] (contiguous), bins [N*H*W] int32: returns the calibrated map (new
    tensor unless ``out``), or None when the kernel does not take the geometry (caller falls back to the row form)."""
    n, c, h, w = x.shape
    y = torch.empty_like(x) if out is None else out
    rc = L.lib().dir_fds_calibrate_fwd_nchw(L.ptr(x), L.ptr(y), L.DIR_F32, L.ptr(bins), n, c, h * w, m1.shape[0], L.ptr(m1), L.ptr(scale),
                                            L.ptr(m2), L.stream_ptr(x.device))
    if rc == L.DIR_EUNSUPPORTED:
        return None
    L.check(rc, "dir_fds_calibrate_fwd_nchw")
    return y


def calibrate_bwd_nchw(dy, bins, scale):
    n, c, h, w = dy.shape
    dx = torch.empty_like(dy)
    rc = L.lib().dir_fds_calibrate_bwd_nchw(L.ptr(dy), L.ptr(dx), L.DIR_F32, L.ptr(bins), n, c, h * w, scale.shape[0], L.ptr(scale),
                                            L.stream_ptr(dy.device))
    if rc == L.DIR_EUNSUPPORTED:
        return None
    L.check(rc, "dir_fds_calibrate_bwd_nchw")
    return dx


def calibrate_bwd(dy, bins, scale):
    dy = dy.contiguous()
    b, c = dy.shape
    dx = torch.empty_like(dy)
    L.check(L.lib().dir_fds_calibrate_bwd(L.ptr(dy), L.ptr(dx), L.DIR_F32, L.ptr(bins), b, c, L.ptr(scale),
                                          L.stream_ptr(dy.device)), "dir_fds_calibrate_bwd")
    return dx


def smooth_fwd_(x, labels, bucket_start, bucket_num, m1, scale, m2):
    """In place on x [B, C]; returns the bins [B + 1] int32 (last slot scratch) for the backward."""
    b, c = x.shape
    bins = torch.empty(b + 1, dtype=i32, device=x.device)
    L.check(L.lib().dir_fds_smooth_fwd(L.ptr(x), L.DIR_F32, L.ptr(labels), b, c, bucket_start, bucket_num,
                                       L.ptr(m1), L.ptr(scale), L.ptr(m2), L.ptr(bins), L.stream_ptr(x.device)),
            "dir_fds_smooth_fwd")
    return bins


# ---- K7 -------------------------------------------------------------------------------------------
def weighted_loss(kind, x, y, w, beta, gamma, activate, need_grad):
    """x, y, (w) flat f32 device tensors of equal numel -> (loss 0-dim f32, dx_unit or None)."""
    n = x.numel()
    dev = x.device
    loss = torch.empty((), dtype=f32, device=dev)
    dx = torch.empty_like(x) if need_grad else None
    nbytes = L.lib().dir_weighted_loss_workspace(n)
    ws = _ws(nbytes, dev) if nbytes else None
    L.check(L.lib().dir_weighted_loss(L.LOSS_KINDS[kind], L.ptr(x), L.ptr(y), L.ptr(w), n, beta, gamma,
                                      1 if activate == "tanh" else 0, L.ptr(loss), L.ptr(dx), L.ptr(ws),
                                      ws.numel() if ws is not None else 0, L.stream_ptr(dev)), "dir_weighted_loss")
    return loss, dx


def scale_by_device_scalar(t, scalar):
    out = torch.empty_like(t)
    L.check(L.lib().dir_scale_by_device_scalar(L.ptr(t), L.ptr(scalar), L.ptr(out), t.numel(),
                                               L.stream_ptr(t.device)), "dir_scale_by_device_scalar")
    return out


# ---- K8 (host) ------------------------------------------------------------------------------------
def lds_weights(labels, max_target, reweight, lds, window):
    """labels: any 1-D array-like (host). Returns np.float32 [n]."""
    lab = np.ascontiguousarray(np.asarray(labels, dtype=np.float64).reshape(-1))
    n = lab.shape[0]
    out = np.empty(n, dtype=np.float32)
    win = None if window is None else np.ascontiguousarray(np.asarray(window, dtype=np.float64))
    rc = L.lib().dir_lds_weights(lab.ctypes.data_as(ctypes.c_void_p), n, max_target, L.REWEIGHT[reweight],
                                 1 if lds else 0, None if win is None else win.ctypes.data_as(ctypes.c_void_p),
                                 0 if win is None else win.shape[0], out.ctypes.data_as(ctypes.c_void_p))
    L.check(rc, "dir_lds_weights")
    return out
