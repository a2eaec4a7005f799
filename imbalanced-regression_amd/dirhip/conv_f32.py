"""Exact-float32 convolution / pooling nodes (``dir_conv_f32_*``, ``dir_*pool*_f32_*``) — the PARITY MODE of the
ResNet-50 stack and the general fallback of the bf16 path.

``resnet50`` run without autocast (``DataParallelEngine(amp_dtype=None)``) computes every ``nn.Conv2d`` of
``imdb-wiki-dir/resnet.py:44-49,79,112-116`` here: implicit GEMMs on ``v_mfma_f32_32x32x2_f32`` (float32 products, float32
accumulation — the arithmetic class of the reference's fp32 convolutions), with the same fused BatchNorm / join / FDS /
loss nodes around them as the bf16 product path. That is what lets the tests hold the whole hand-written stack to the
north_star's 1e-5 loss bar. The bf16 path also lands here (through float32 casts) for the shapes its MFMA kernels do not
take — channel counts that are not multiples of 64, strided data gradients of odd-sized maps, tensors beyond 32-bit
offsets — so that no library convolution is ever called.
"""
import contextlib
import threading

import torch

from . import _lib as L


def _nhwc(t):
    return t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)


def _f32(t):
    return t if t.dtype == torch.float32 else t.float()


GATHER, TILE = 1, 2                     # DIR_CONV_F32_GATHER / DIR_CONV_F32_TILE: force a kernel (0 = the product's choice: tile where applicable)
TILE_X3, TILE_X2 = 3, 4                 # DIR_CONV_F32_TILE_X3 / _X2: the tile kernels with split-bf16 arithmetic (float32-GRADE, not bit-equal; ABI 4)
ARITHMETICS = {"exact": 0, "x3": TILE_X3, "x2": TILE_X2}

# The arithmetic of the float32 graph nodes is a property of the FORWARD PASS that builds them (set by the engine for the span of one forward call,
# `DataParallelEngine(f32_arith=...)` / `train.py --amp fp32x3`), recorded in each node and used again by its backward: thread-local, no process-wide switch.
_SCOPE = threading.local()


def current_arith():
    """``variant`` code of the float32 convolutions built now: 0 (exact float32 MFMA), ``TILE_X3`` or ``TILE_X2``."""
    return getattr(_SCOPE, "variant", 0)


@contextlib.contextmanager
def arithmetic(name):
    """``with arithmetic("x3"):`` — float32 convolution nodes created inside run (forward AND backward) on the split-bf16 kernels wherever the tile
    kernels take the geometry (whole 16-channel K-steps; the 7x7 stem and odd shapes stay exact). ``"exact"`` / None: the parity arithmetic."""
    prev = current_arith()
    _SCOPE.variant = ARITHMETICS[name or "exact"]
    try:
        yield
    finally:
        _SCOPE.variant = prev


def _checked(fn_name, rc_of, variant):
    """Run ``rc_of(variant)``; a split arithmetic on a geometry the tile kernels do not take (DIR_EUNSUPPORTED) falls back to the product's exact choice."""
    rc = rc_of(variant)
    if rc == L.DIR_EUNSUPPORTED and variant in (TILE_X3, TILE_X2):
        rc = rc_of(0)
    L.check(rc, fn_name)


def stats_fusable(x, w):
    """Can the forward of this geometry carry the BatchNorm statistics of its result (tile kernel + its LDS store loop: whole 16-channel K-steps,
    Cout % 4 == 0, 32-bit byte offsets)? The rule of ``dir_conv_f32_fwd_stats``."""
    return x.shape[1] % 16 == 0 and w.shape[0] % 4 == 0 and x.numel() * 4 < 2 ** 31 and w.numel() * 4 < 2 ** 31


def conv2d_f32_fwd(x, w, stride, padding, variant=0, want_stats=False):
    """x [N, Cin, H, W], w [Cout, Cin, R, S]: float32 channels_last device tensors -> y float32 channels_last. ``want_stats``: returns
    ``(y, partial)`` with the per-channel (sum, sum of squares) partials ``[rows][2][Cout]`` of y for the BatchNorm that follows, formed in
    the kernel's store loop (None where the geometry runs on the gather kernel: the BatchNorm then counts itself)."""
    n, cin, h, wd = x.shape
    cout, cin2, r, s = w.shape
    assert cin == cin2, (x.shape, w.shape)
    ho = (h + 2 * padding - r) // stride + 1
    wo = (wd + 2 * padding - s) // stride + 1
    y = torch.empty((n, cout, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    if want_stats:
        if variant in (GATHER, TILE) or not stats_fusable(x, w):
            return conv2d_f32_fwd(x, w, stride, padding, variant), None
        rows = L.lib().dir_conv_f32_stats_rows(n, ho, wo)
        stats = torch.empty((rows, 2, cout), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _checked("dir_conv_f32_fwd_stats", lambda v: L.lib().dir_conv_f32_fwd_stats_variant(
                L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(stats), rows, n, h, wd, cin, cout, r, s, stride, padding, v, L.stream_ptr(x.device)), variant)
        return y, stats
    with torch.cuda.device(x.device):
        _checked("dir_conv_f32_fwd", lambda v: L.lib().dir_conv_f32_fwd_variant(L.ptr(x), L.ptr(w), L.ptr(y), n, h, wd, cin, cout, r, s, stride, padding, v,
                                                                             L.stream_ptr(x.device)), variant)
    return y


def conv2d_f32_dgrad(dy, w, in_hw, stride, padding, addend=None, addend_s2=None, relu_mask=None, variant=0):
    """Data gradient; optionally with the fused store epilogue of the bf16 path (``dir_conv_f32_dgrad_fused``):
    ``+ addend`` (same shape as dx), ``+ addend_s2`` (compact ``[N, Cin, H/2, W/2]``, added at the even pixels) and the
    ReLU backward ``* (relu_mask > 0)``."""
    n, cout = dy.shape[0], dy.shape[1]
    cin, r, s = w.shape[1], w.shape[2], w.shape[3]
    h, wd = in_hw
    dx = torch.empty((n, cin, h, wd), dtype=torch.float32, device=dy.device, memory_format=torch.channels_last)
    for t, shp in ((addend, (n, cin, h, wd)), (relu_mask, (n, cin, h, wd)), (addend_s2, (n, cin, h // 2, wd // 2))):
        assert t is None or (tuple(t.shape) == shp and t.dtype == torch.float32 and t.is_contiguous(memory_format=torch.channels_last)), shp
    with torch.cuda.device(dy.device):
        _checked("dir_conv_f32_dgrad", lambda v: L.lib().dir_conv_f32_dgrad_variant(
            L.ptr(dy), L.ptr(w), L.ptr(addend), L.ptr(addend_s2), L.ptr(relu_mask), L.ptr(dx), n, h, wd, cin, cout, r, s, stride, padding, v,
            L.stream_ptr(dy.device)), variant)
    return dx


def conv2d_f32_wgrad(dy, x, kernel_hw, stride, padding, variant=0):
    n, cin, h, wd = x.shape
    cout = dy.shape[1]
    r, s = kernel_hw
    nbytes = L.lib().dir_conv_f32_wgrad_workspace(n, h, wd, cin, cout, r, s, stride, padding)
    if nbytes == 0:
        raise L.DirHipError(f"dir_conv_f32_wgrad: unsupported shape {tuple(x.shape)} -> Cout={cout}")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    dw = torch.empty((cout, cin, r, s), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        _checked("dir_conv_f32_wgrad", lambda v: L.lib().dir_conv_f32_wgrad_variant(
            L.ptr(dy), L.ptr(x), L.ptr(dw), n, h, wd, cin, cout, r, s, stride, padding, L.ptr(ws), ws.numel(), v, L.stream_ptr(x.device)), variant)
    return dw


class _ConvF32Fn(torch.autograd.Function):
    """``F.conv2d(x, weight, None, stride, padding)`` (square stride / padding, no dilation / groups / bias) in exact
    float32 on the hand-written kernels. Inputs of another dtype are computed in float32 and returned in their dtype."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding):
        if not x.is_cuda:
            raise L.DirHipError(f"conv_f32: input on {x.device}; the convolutions run only as HIP kernels (no CPU fallback)")
        ctx.in_dtype = x.dtype
        ctx.stride, ctx.padding = stride, padding
        x32, w32 = _nhwc(_f32(x)), _nhwc(_f32(weight.detach()))
        ctx.variant = current_arith()
        y = conv2d_f32_fwd(x32, w32, stride, padding, variant=ctx.variant)
        ctx.save_for_backward(x32, w32)
        return y if x.dtype == torch.float32 else y.to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        x32, w32 = ctx.saved_tensors
        dy32 = _nhwc(_f32(dy))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = conv2d_f32_dgrad(dy32, w32, x32.shape[2:], ctx.stride, ctx.padding, variant=ctx.variant)
            if ctx.in_dtype != torch.float32:
                dx = dx.to(ctx.in_dtype)
        dw = conv2d_f32_wgrad(dy32, x32, w32.shape[2:], ctx.stride, ctx.padding, variant=ctx.variant) if ctx.needs_input_grad[1] else None
        return dx, dw, None, None


def conv_ok(conv):
    return (conv.bias is None and conv.groups == 1 and conv.dilation == (1, 1) and conv.stride[0] == conv.stride[1]
            and conv.padding[0] == conv.padding[1] and isinstance(conv.padding[0], int) and conv.padding_mode == "zeros")


def conv_f32(x, conv):
    """``conv(x)`` for a bias-free ``nn.Conv2d`` on the exact-float32 kernels."""
    if not conv_ok(conv):
        raise L.DirHipError(f"conv_f32: unsupported convolution {conv}")
    return _ConvF32Fn.apply(x, conv.weight, conv.stride[0], conv.padding[0])


class _MaxPoolF32Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _nhwc(x)
        n, c, h, w = x.shape
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = torch.empty((n, c, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        idx = torch.empty((n, c, ho, wo), dtype=torch.uint8, device=x.device, memory_format=torch.channels_last)
        with torch.cuda.device(x.device):
            L.check(L.lib().dir_maxpool3x3s2_f32_fwd(L.ptr(x), L.ptr(y), L.ptr(idx), n, h, w, c, L.stream_ptr(x.device)),
                    "dir_maxpool3x3s2_f32_fwd")
        ctx.save_for_backward(idx)
        ctx.in_shape = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        n, c, h, w = ctx.in_shape
        dy = _nhwc(_f32(dy))
        dx = torch.empty((n, c, h, w), dtype=torch.float32, device=dy.device, memory_format=torch.channels_last)
        with torch.cuda.device(dy.device):
            L.check(L.lib().dir_maxpool3x3s2_f32_bwd(L.ptr(dy), L.ptr(idx), L.ptr(dx), n, h, w, c, L.stream_ptr(dy.device)),
                    "dir_maxpool3x3s2_f32_bwd")
        return dx


def maxpool3x3s2_f32(x):
    return _MaxPoolF32Fn.apply(x)


class _GlobalAvgPoolF32Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _nhwc(x)
        n, c, h, w = x.shape
        y = torch.empty((n, c), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            L.check(L.lib().dir_avgpool_f32_fwd(L.ptr(x), L.ptr(y), n, h * w, c, L.stream_ptr(x.device)), "dir_avgpool_f32_fwd")
        ctx.in_shape = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w = ctx.in_shape
        dy = _f32(dy).contiguous()
        dx = torch.empty((n, c, h, w), dtype=torch.float32, device=dy.device, memory_format=torch.channels_last)
        with torch.cuda.device(dy.device):
            L.check(L.lib().dir_avgpool_f32_bwd(L.ptr(dy), L.ptr(dx), n, h * w, c, L.stream_ptr(dy.device)), "dir_avgpool_f32_bwd")
        return dx


def global_avgpool_flat_f32(x):
    return _GlobalAvgPoolF32Fn.apply(x)


# ---- the graph nodes of dirhip.conv, in float32 ------------------------------------------------------------------
# Same contracts as conv._ConvFn / conv._ProjectionPairFn (second output = the input itself so that the shortcut's
# gradient is accumulated inside the data-gradient kernel; ReLU backward of the producing relu(bn + shortcut) node applied
# on store; the stride-2 downsample gradient in compact form): resnet.py builds ONE graph, these are its float32 kernels.
class _ConvGraphF32Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, stride, padding, alias_input, relu_input, want_stats=False):
        ctx.set_materialize_grads(False)
        ctx.stride, ctx.padding, ctx.relu_input = stride, padding, relu_input
        x32, w32 = _nhwc(x), _nhwc(weight.detach())
        v = ctx.variant = current_arith()
        y, stats = conv2d_f32_fwd(x32, w32, stride, padding, v, want_stats=True) if want_stats else (conv2d_f32_fwd(x32, w32, stride, padding, v), None)
        ctx.save_for_backward(x32, w32)
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        return y, (x if alias_input else None), stats

    @staticmethod
    def backward(ctx, dy, dalias=None, dstats=None):
        x, w32 = ctx.saved_tensors
        if dy is None:                                                           # only the alias output was used
            assert not ctx.relu_input
            return dalias, None, None, None, None, None, None
        dy = _nhwc(_f32(dy))
        dalias = None if dalias is None else _nhwc(_f32(dalias))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = conv2d_f32_dgrad(dy, w32, x.shape[2:], ctx.stride, ctx.padding, addend=dalias,
                                  relu_mask=x if ctx.relu_input else None, variant=ctx.variant)
        elif dalias is not None:
            dx = dalias
        dw = conv2d_f32_wgrad(dy, x, w32.shape[2:], ctx.stride, ctx.padding, variant=ctx.variant)
        return dx, dw, None, None, None, None, None


def conv_graph_f32(x, conv, alias_input, relu_input, want_stats=False):
    """Returns ``(y, alias or None, BatchNorm statistic partials of y or None)``."""
    return _ConvGraphF32Fn.apply(x, conv.weight, conv.stride[0], conv.padding[0], alias_input, relu_input, want_stats)


class _ProjectionPairF32Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, wd, stride_d, relu_input, want_stats=False):
        ctx.set_materialize_grads(False)
        ctx.stride_d, ctx.relu_input = stride_d, relu_input
        x32, w1_32, wd_32 = _nhwc(x), _nhwc(w1.detach()), _nhwc(wd.detach())
        v = ctx.variant = current_arith()
        y1, s1 = conv2d_f32_fwd(x32, w1_32, 1, 0, v, want_stats=True) if want_stats else (conv2d_f32_fwd(x32, w1_32, 1, 0, v), None)
        yd, sd = conv2d_f32_fwd(x32, wd_32, stride_d, 0, v, want_stats=True) if want_stats else (conv2d_f32_fwd(x32, wd_32, stride_d, 0, v), None)
        ctx.save_for_backward(x32, w1_32, wd_32)
        for st in (s1, sd):
            if st is not None:
                ctx.mark_non_differentiable(st)
        return y1, yd, s1, sd

    @staticmethod
    def backward(ctx, dy1, dyd, ds1=None, dsd=None):
        x, w1, wd = ctx.saved_tensors
        if dy1 is None:
            raise L.DirHipError("projection pair: conv1's output received no gradient")
        dy1 = _nhwc(_f32(dy1))
        dyd = None if dyd is None else _nhwc(_f32(dyd))
        mask = x if ctx.relu_input else None
        # the downsample conv's data gradient in COMPACT form: a plain 1x1 stride-1 data gradient on its own dY grid
        v = ctx.variant
        compact = conv2d_f32_dgrad(dyd, wd, dyd.shape[2:], 1, 0, variant=v) if dyd is not None else None
        if compact is None:
            dx = conv2d_f32_dgrad(dy1, w1, x.shape[2:], 1, 0, relu_mask=mask, variant=v)
        elif ctx.stride_d == 1:
            dx = conv2d_f32_dgrad(dy1, w1, x.shape[2:], 1, 0, addend=compact, relu_mask=mask, variant=v)
        else:
            dx = conv2d_f32_dgrad(dy1, w1, x.shape[2:], 1, 0, addend_s2=compact, relu_mask=mask, variant=v)
        dw1 = conv2d_f32_wgrad(dy1, x, (1, 1), 1, 0, variant=v)
        dwd = conv2d_f32_wgrad(dyd, x, (1, 1), ctx.stride_d, 0, variant=v) if dyd is not None else None
        return (dx if ctx.needs_input_grad[0] else None), dw1, dwd, None, None, None


def projection_pair_f32(x, conv1, conv_d, relu_input, want_stats=False):
    """Returns ``(y1, yd, statistic partials of y1 or None, of yd or None)``."""
    return _ProjectionPairF32Fn.apply(x, conv1.weight, conv_d.weight, conv_d.stride[0], relu_input, want_stats)
