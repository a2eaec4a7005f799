"""MFMA implicit-GEMM convolution (``dir_conv_fwd``) — tensor-level wrapper.

``conv2d_igemm(x, w, stride, padding, want_stats)``: x ``[N, Cin, H, W]`` and w ``[Cout, Cin, R, S]``, both bf16 and
channels_last (so their memory is NHWC / ``[Cout][R][S][Cin]``); returns y (channels_last bf16) and, optionally,
the per-tile BatchNorm partial statistics ``[rows][2][Cout]`` float32.
"""
import torch

from . import _lib as L


def supported(cin, cout):
    return cin % 64 == 0 and cout % 64 == 0


def conv2d_igemm(x, w, stride=1, padding=0, want_stats=False, addend=None):
    if not x.is_cuda:
        raise L.DirHipError(f"conv2d_igemm: input on {x.device}; MFMA convolution runs only on the GPU (no CPU fallback)")
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    if not w.is_contiguous(memory_format=torch.channels_last):
        w = w.contiguous(memory_format=torch.channels_last)
    n, cin, h, wd = x.shape
    cout, cin2, r, s = w.shape
    assert cin == cin2
    ho = (h + 2 * padding - r) // stride + 1
    wo = (wd + 2 * padding - s) // stride + 1
    y = torch.empty((n, cout, ho, wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    stats = None
    if want_stats:
        rows = L.lib().dir_conv_stats_rows(n, ho, wo)
        stats = torch.empty((rows, 2, cout), dtype=torch.float32, device=x.device)
    if addend is not None:
        assert addend.shape == y.shape and addend.dtype == torch.bfloat16 and not want_stats
        if not addend.is_contiguous(memory_format=torch.channels_last):
            addend = addend.contiguous(memory_format=torch.channels_last)
    L.check(L.lib().dir_conv_fwd_add(L.ptr(x), L.ptr(w), L.ptr(addend), L.ptr(y), L.ptr(stats), n, h, wd, cin, cout, r, s,
                                     stride, padding, L.stream_ptr(x.device)), "dir_conv_fwd")
    return (y, stats) if want_stats else y


def conv2d_wgrad(dy, x, kernel_size, stride=1, padding=0):
    """Weight gradient (float32, channels_last ``[Cout, Cin, R, S]``) from bf16 channels_last ``dy`` and ``x``."""
    assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16
    if not dy.is_contiguous(memory_format=torch.channels_last):
        dy = dy.contiguous(memory_format=torch.channels_last)
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    n, cin, h, w = x.shape
    cout = dy.shape[1]
    r = s = kernel_size
    nbytes = L.lib().dir_conv_wgrad_workspace(n, h, w, cin, cout, r, s, stride, padding)
    if nbytes == 0:
        raise L.DirHipError(f"dir_conv_wgrad: unsupported shape Cin={cin} Cout={cout}")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    dw = torch.empty((cout, cin, r, s), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    L.check(L.lib().dir_conv_wgrad(L.ptr(dy), L.ptr(x), L.ptr(dw), n, h, w, cin, cout, r, s, stride, padding, L.ptr(ws),
                                   ws.numel(), L.stream_ptr(x.device)), "dir_conv_wgrad")
    return dw


class _ConvFn(torch.autograd.Function):
    """Forward = hand-written MFMA implicit GEMM (+ BatchNorm partial statistics in the epilogue); data gradient of
    stride-1 layers = the same kernel on dY with rotated/transposed weights; weight gradient = the MFMA split-K
    kernel ``dir_conv_wgrad``. Only the data gradient of the six stride-2 layers still uses the library kernel."""

    @staticmethod
    def forward(ctx, x, weight, w16, w16_rot, stride, padding, want_stats, alias_input):
        ctx.stride, ctx.padding = stride, padding
        ctx.alias_input = alias_input
        if want_stats:
            y, stats = conv2d_igemm(x, w16, stride, padding, want_stats=True)
            ctx.mark_non_differentiable(stats)
        else:
            y, stats = conv2d_igemm(x, w16, stride, padding), None
        ctx.save_for_backward(x, w16, w16_rot)
        if alias_input:
            # second output = the input itself (for the identity shortcut): its gradient arrives in backward() together
            # with dy, so the accumulation dX = dgrad(dy) + d_shortcut happens inside the dgrad kernel's store loop
            return y, stats, x
        return y, stats, None

    @staticmethod
    def backward(ctx, dy, _dstats, dalias=None):
        x, w16, w16_rot = ctx.saved_tensors
        if dalias is not None and dalias.dtype != torch.bfloat16:
            dalias = dalias.to(torch.bfloat16)
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dy = dy.contiguous(memory_format=torch.channels_last)
        need_dx = ctx.needs_input_grad[0]
        dx = None
        if need_dx and w16_rot is not None:
            # data gradient of a stride-1 convolution = the SAME implicit GEMM on dY with the 180-degree rotated,
            # in/out-transposed weights and padding R-1-pad
            dx = conv2d_igemm(dy, w16_rot, 1, w16.shape[2] - 1 - ctx.padding, addend=dalias)
            dalias = None
            need_dx = False
        dw = conv2d_wgrad(dy, x, w16.shape[2], ctx.stride, ctx.padding)          # float32, deterministic split-K
        if need_dx:                                                              # strided data gradient: library kernel for now
            dx = torch.ops.aten.convolution_backward(
                dy, x, w16, None, [ctx.stride, ctx.stride], [ctx.padding, ctx.padding], [1, 1], False, [0, 0], 1,
                [True, False, False])[0]
        if dalias is not None:                                                   # strided layer: eager accumulation
            dx = dalias if dx is None else dx + dalias
        return dx, dw, None, None, None, None, None, None


# Generation counter of "the weights may have changed": fused / multi-tensor optimizer kernels update parameters
# WITHOUT bumping Tensor._version (measured: torch 2.10 fused Adam leaves _version untouched), so the bf16 weight
# cache is keyed on (generation, _version, data_ptr) and every optimizer.step() anywhere bumps the generation.
_GENERATION = [0]


def invalidate_weight_cache(*_a, **_k):
    _GENERATION[0] += 1


from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post_hook  # noqa: E402
_reg_post_hook(invalidate_weight_cache)


def conv_bn_input(x, conv, want_stats, alias_input=False):
    """Apply ``conv`` (an ``nn.Conv2d`` with bias=False, Cin/Cout multiples of 64) to a bf16 channels_last tensor with
    the MFMA kernel. Returns ``(y, partial_stats or None)``. The bf16 copy of the fp32 master weight is cached until
    the next optimizer step / in-place edit (one cast per step instead of one per use: train forward, epoch-tail
    forward and the backward share it)."""
    w = conv.weight
    key = (_GENERATION[0], w._version, w.data_ptr())
    cache = getattr(conv, "_dir_w16", None)
    if cache is None or cache[0] != key:
        wd = w.detach()
        if not wd.is_contiguous(memory_format=torch.channels_last):
            wd = wd.contiguous(memory_format=torch.channels_last)
        cout, cin, r, s = wd.shape
        w16 = torch.empty((cout, cin, r, s), dtype=torch.bfloat16, device=wd.device, memory_format=torch.channels_last)
        w16_rot = None
        if conv.stride[0] == 1 and supported(cout, cin):
            # [Cin][R][S][Cout], taps rotated by 180 degrees: the weight of the data-gradient convolution
            w16_rot = torch.empty((cin, cout, r, s), dtype=torch.bfloat16, device=wd.device, memory_format=torch.channels_last)
        L.check(L.lib().dir_conv_prep_weights(L.ptr(wd), cout, r, s, cin, L.ptr(w16), L.ptr(w16_rot),
                                              L.stream_ptr(wd.device)), "dir_conv_prep_weights")
        conv._dir_w16 = (key, w16, w16_rot)
    else:
        w16, w16_rot = cache[1], cache[2]
    if x.dtype != torch.bfloat16:
        x = x.to(torch.bfloat16)
    y, stats, alias = _ConvFn.apply(x, w, w16, w16_rot if torch.is_grad_enabled() else None, conv.stride[0],
                                    conv.padding[0], want_stats, alias_input and torch.is_grad_enabled() and x.requires_grad)
    if alias_input:
        return y, stats, (alias if alias is not None else x)
    return y, stats
