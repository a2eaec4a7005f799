"""Fused network tail — ``AvgPool2d(7)`` + view -> ``FDS.smooth`` -> ``Linear(2048, 1)`` of ``resnet.py:136-148`` as ONE
hand-written kernel forward (``dir_tail_fwd``) and a single backward node (``dir_tail_bwd``: data gradient in one launch,
weight / bias gradients by a fixed-order two-stage reduction). SURVEY.md §8f-3. Replaces the pool kernel, the calibration
kernel and the library gemv of the linear layer (and, in the backward, a scale kernel, two library gemv, the calibration
backward, the pool backward and a sum): the [B, 2048] encoding is written once, its gradient never.

``tail_forward(fmap, linear, fds, targets)`` returns ``(pred [B, 1], encoding [B, C])`` with ``encoding`` the CALIBRATED
features when ``fds`` is given (what the reference returns as ``encoding``).
"""
import torch

from . import _lib as L
from . import gradsink
from . import ops

_DT = {torch.bfloat16: L.DIR_BF16, torch.float32: L.DIR_F32}
FUSED_LABEL_SCAN_MAX_B = 2048          # beyond this the in-kernel label rescan (O(B^2) L2 reads) loses to a separate bin pass


class _TailFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fmap, weight, bias, labels, m1, scale, m2, bucket_start, bucket_num):
        ctx.set_materialize_grads(False)
        x = fmap if fmap.is_contiguous(memory_format=torch.channels_last) else fmap.contiguous(memory_format=torch.channels_last)
        b, c, h, w = x.shape
        dev = x.device
        w32 = weight.detach().reshape(-1)
        if w32.dtype != torch.float32 or not w32.is_contiguous():
            w32 = w32.float().contiguous()
        b32 = bias.detach().reshape(-1).float().contiguous()
        enc = torch.empty((b, c), dtype=torch.float32, device=dev)
        pred = torch.empty((b, 1), dtype=torch.float32, device=dev)
        bins = bins_in = None
        if m1 is not None:
            bins = torch.empty(b, dtype=torch.int32, device=dev)
            if b > FUSED_LABEL_SCAN_MAX_B:
                bins_in, _ = ops.bin_index(labels, bucket_start, bucket_num)
                labels = None
        L.check(L.lib().dir_tail_fwd(L.ptr(x), _DT[x.dtype], L.ptr(labels), L.ptr(bins_in), b, h * w, c, bucket_start, bucket_num,
                                     L.ptr(m1), L.ptr(scale), L.ptr(m2), L.ptr(w32), L.ptr(b32), L.ptr(enc), L.ptr(pred), L.ptr(bins),
                                     L.stream_ptr(dev)), "dir_tail_fwd")
        ctx.save_for_backward(enc, bins, scale, w32)
        ctx.in_shape, ctx.in_dtype = (b, c, h, w), x.dtype
        ctx.w_shape, ctx.b_shape = weight.shape, bias.shape
        ctx.sinks = (gradsink.lookup(weight), gradsink.lookup(bias))
        return pred, enc

    @staticmethod
    def backward(ctx, dpred, denc):
        enc, bins, scale, w32 = ctx.saved_tensors
        b, c, h, w = ctx.in_shape
        dev = enc.device
        if dpred is None:
            dpred = torch.zeros((b, 1), dtype=torch.float32, device=dev)
        dpred = dpred.reshape(-1).float().contiguous()
        if denc is not None:
            denc = denc.float().contiguous()
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dx = torch.empty((b, c, h, w), dtype=ctx.in_dtype, device=dev, memory_format=torch.channels_last) if need_x else None
        dw = gradsink.out_for(ctx.sinks[0], (c,), dev) if need_w else None      # (bucket slots of the data-parallel engine, if any)
        db = gradsink.out_for(ctx.sinks[1], (1,), dev) if need_w else None
        ws = torch.empty(max(int(L.lib().dir_tail_bwd_workspace(b, c)), 256), dtype=torch.uint8, device=dev) if need_w else None
        L.check(L.lib().dir_tail_bwd(L.ptr(dpred), L.ptr(denc), L.ptr(bins), L.ptr(scale), L.ptr(w32), L.ptr(enc), b, h * w, c,
                                     _DT[ctx.in_dtype], L.ptr(dx), L.ptr(dw), L.ptr(db), L.ptr(ws), 0 if ws is None else ws.numel(),
                                     L.stream_ptr(dev)), "dir_tail_bwd")
        return (dx, None if dw is None else dw.reshape(ctx.w_shape), None if db is None else db.reshape(ctx.b_shape),
                None, None, None, None, None, None)


def fusable(fmap, pool, linear):
    """The fused tail takes the reference's geometry: the pool window is the whole (7x7) map, one output unit."""
    k = pool.kernel_size if isinstance(pool.kernel_size, tuple) else (pool.kernel_size, pool.kernel_size)
    return (fmap.is_cuda and fmap.dim() == 4 and fmap.dtype in _DT and tuple(fmap.shape[2:]) == tuple(k) and fmap.shape[1] % 8 == 0
            and pool.padding in (0, (0, 0)) and linear.out_features == 1 and linear.in_features == fmap.shape[1]
            and linear.bias is not None and linear.weight.dtype == torch.float32)


def tail_forward(fmap, linear, fds=None, targets=None):
    """``fds``: the FDS module when its calibration is live for this call (training, epoch >= start_smooth), else None."""
    if fds is None:
        return _TailFn.apply(fmap, linear.weight, linear.bias, None, None, None, None, 0, 1)
    labels = targets.squeeze(1)                                   # fds.py:119: labels are [B, 1]
    if labels.dtype != torch.float32:
        labels = labels.float()
    labels = L.require_device_tensor(labels.contiguous(), torch.float32, "labels")
    assert labels.numel() == fmap.shape[0] and fds.feature_dim == fmap.shape[1]
    m1 = fds.running_mean_last_epoch if fds.running_mean_last_epoch.is_contiguous() else fds.running_mean_last_epoch.contiguous()
    m2 = fds.smoothed_mean_last_epoch if fds.smoothed_mean_last_epoch.is_contiguous() else fds.smoothed_mean_last_epoch.contiguous()
    return _TailFn.apply(fmap, linear.weight, linear.bias, labels, m1, fds._scale_table(), m2, fds.bucket_start, fds.bucket_num)
