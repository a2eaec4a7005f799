"""Pools of the ResNet stack (``resnet.py:82,85,131,136-137``) on hand-written kernels: 3x3 / stride-2 / pad-1 max pooling
(one argmax byte per element forward, atomic-free gather backward), the global average pool, and the fused stem tail
``maxpool(relu(bn(x)))``. bf16 channels_last activations use ``dir_maxpool3x3s2_*`` / ``dir_avgpool_*`` /
``dir_bn_relu_maxpool_*``; float32 activations (parity mode) use the ``*_f32_*`` kernels of ``conv_f32``. Geometries the
reference's ResNet never uses raise instead of silently falling back to a library kernel."""
import torch

from . import _lib as L
from . import gradsink


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
        n, c, h, w = x.shape
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = torch.empty((n, c, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        idx = torch.empty((n, c, ho, wo), dtype=torch.uint8, device=x.device, memory_format=torch.channels_last)
        L.check(L.lib().dir_maxpool3x3s2_fwd(L.ptr(x), L.ptr(y), L.ptr(idx), n, h, w, c, L.stream_ptr(x.device)),
                "dir_maxpool3x3s2_fwd")
        ctx.save_for_backward(idx)
        ctx.in_shape = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        n, c, h, w = ctx.in_shape
        dy = dy if dy.is_contiguous(memory_format=torch.channels_last) else dy.contiguous(memory_format=torch.channels_last)
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dx = torch.empty((n, c, h, w), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
        L.check(L.lib().dir_maxpool3x3s2_bwd(L.ptr(dy), L.ptr(idx), L.ptr(dx), n, h, w, c, L.stream_ptr(dy.device)),
                "dir_maxpool3x3s2_bwd")
        return dx


def maxpool3x3s2(x, module):
    """``module(x)`` for the registered ``nn.MaxPool2d(3, 2, 1)`` on a channels_last device tensor."""
    geom = (module.kernel_size in (3, (3, 3)) and module.stride in (2, (2, 2)) and module.padding in (1, (1, 1))
            and module.dilation in (1, (1, 1)) and not module.ceil_mode)
    if not x.is_cuda:
        raise L.DirHipError(f"maxpool: input on {x.device}; the pools run only as HIP kernels (no CPU fallback)")
    if not geom:
        raise L.DirHipError(f"maxpool: only MaxPool2d(3, 2, 1) is implemented, got {module}")
    if x.dtype == torch.bfloat16 and x.shape[1] % 8 == 0:
        return _MaxPoolFn.apply(x)
    from .conv_f32 import maxpool3x3s2_f32
    y = maxpool3x3s2_f32(x.float())
    return y if x.dtype == torch.float32 else y.to(x.dtype)


class _GlobalAvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
        n, c, h, w = x.shape
        y = torch.empty((n, c), dtype=torch.float32, device=x.device)
        L.check(L.lib().dir_avgpool_fwd(L.ptr(x), L.ptr(y), n, h * w, c, L.stream_ptr(x.device)), "dir_avgpool_fwd")
        ctx.in_shape = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w = ctx.in_shape
        dy = dy.contiguous()
        if dy.dtype != torch.float32:
            dy = dy.float()
        dx = torch.empty((n, c, h, w), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
        L.check(L.lib().dir_avgpool_bwd(L.ptr(dy), L.ptr(dx), n, h * w, c, L.stream_ptr(dy.device)), "dir_avgpool_bwd")
        return dx


def global_avgpool_flat(x, module):
    """``module(x).view(N, -1)`` of ``resnet.py:136-137`` (``module`` = the registered nn.AvgPool2d(7)) as float32
    ``[N, C]``: the pool window must be the whole map (it is, for the reference's 224-px inputs)."""
    k = module.kernel_size if isinstance(module.kernel_size, tuple) else (module.kernel_size, module.kernel_size)
    if not x.is_cuda:
        raise L.DirHipError(f"avgpool: input on {x.device}; the pools run only as HIP kernels (no CPU fallback)")
    if x.dim() != 4 or tuple(x.shape[2:]) != tuple(k) or module.padding not in (0, (0, 0)):
        raise L.DirHipError(f"avgpool: only the global pool (window = whole {tuple(k)} map) is implemented, got a "
                            f"{tuple(x.shape[2:])} map — the reference's network needs img_size 224")
    if x.dtype == torch.bfloat16 and x.shape[1] % 8 == 0:
        return _GlobalAvgPoolFn.apply(x)
    from .conv_f32 import global_avgpool_flat_f32
    return global_avgpool_flat_f32(x.float())


class _BnReluMaxPoolFn(torch.autograd.Function):
    """Stem tail ``maxpool(relu(bn(x)))`` (training mode) on the BatchNorm input: statistics pass + finalize
    (``dir_bn_prepare_train``), then one fused normalise / ReLU / 3x3-s2 max pass; backward = BatchNorm reductions over
    the pooled gradient + one fused pool-backward / BatchNorm-apply pass (``dir_bn_relu_maxpool_*``)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, partial=None, want_xmax=True):
        x = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
        n, c, h, w = x.shape
        m = n * h * w
        dev = x.device
        stream = L.stream_ptr(dev)
        f32 = torch.float32
        mean, rstd = torch.empty(c, dtype=f32, device=dev), torch.empty(c, dtype=f32, device=dev)
        coef = torch.empty(2, c, dtype=f32, device=dev)
        nbytes = L.lib().dir_bn_workspace(L.DIR_BF16, m, c)
        if nbytes == 0:
            raise L.DirHipError(f"dir_bn: unsupported shape M={m} C={c}")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        L.check(L.lib().dir_bn_prepare_train(L.ptr(x), L.DIR_BF16, m, c, L.ptr(partial), 0 if partial is None else partial.shape[0],
                                             L.ptr(gamma), L.ptr(beta), L.ptr(running_mean),
                                             L.ptr(running_var), float(momentum), float(eps), L.ptr(mean), L.ptr(rstd), L.ptr(coef),
                                             L.ptr(ws), ws.numel(), stream), "dir_bn_prepare_train")
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = torch.empty((n, c, ho, wo), dtype=torch.bfloat16, device=dev, memory_format=torch.channels_last)
        idx = torch.empty((n, c, ho, wo), dtype=torch.uint8, device=dev, memory_format=torch.channels_last)
        # x at the argmax, for the backward's sum g * x (pooled size: 1/4 of x); None = the older pair (gather reduction, per-pixel apply)
        xmax = torch.empty_like(y) if (_STEM_TAIL_XMAX[0] and want_xmax) else None   # (no backward, e.g. the no-grad epoch-tail forward: not written at all)
        L.check(L.lib().dir_bn_relu_maxpool_fwd_xmax(L.ptr(x), L.ptr(coef), L.ptr(y), L.ptr(idx), L.ptr(xmax), n, h, w, c, stream),
                "dir_bn_relu_maxpool_fwd_xmax")
        ctx.save_for_backward(x, gamma, mean, rstd, idx, xmax)
        ctx.beta_sink = gradsink.lookup(beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd, idx, xmax = ctx.saved_tensors
        n, c, h, w = x.shape
        dy = dy if dy.is_contiguous(memory_format=torch.channels_last) else dy.contiguous(memory_format=torch.channels_last)
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dev = x.device
        dx = torch.empty_like(x)
        dgamma = gradsink.out_for(gamma, (c,), dev)
        dbeta = gradsink.out_for(ctx.beta_sink, (c,), dev)
        ws = torch.empty(L.lib().dir_bn_relu_maxpool_bwd_workspace(c), dtype=torch.uint8, device=dev)
        L.check(L.lib().dir_bn_relu_maxpool_bwd_xmax(L.ptr(dy), L.ptr(idx), L.ptr(x), L.ptr(xmax), L.ptr(dx), n, h, w, c, L.ptr(gamma), L.ptr(mean),
                                                     L.ptr(rstd), L.ptr(dgamma), L.ptr(dbeta), L.ptr(ws), ws.numel(), L.stream_ptr(dev)),
                "dir_bn_relu_maxpool_bwd_xmax")
        return dx, dgamma, dbeta, None, None, None, None, None, None


# keep the forward's x-at-argmax for the stem tail's backward (the product's form) or recompute it (the C-ABI has no mode switch: the backward's
# form follows from the xmax argument being given or NULL). A constant of the product; tools/variant_switches.py flips it for A/B runs and tests.
_STEM_TAIL_XMAX = [True]


def bn_relu_maxpool(x, bn, pool, partial=None):
    """``pool(relu(bn(x)))`` for the stem (``bn`` = nn.BatchNorm2d, ``pool`` = nn.MaxPool2d(3, 2, 1)). Fused when
    training on a bf16 CUDA map; otherwise the fused BatchNorm node followed by the pool. ``partial``: statistics of x
    from the producing convolution's epilogue (``conv.stem_conv``) or None."""
    from .bn import _count_batch, bn_act
    c = x.shape[1]
    ok = (x.is_cuda and x.dtype == torch.bfloat16 and bn.training and bn.track_running_stats and bn.momentum is not None
          and c % 8 == 0 and c <= 128 and 256 % (c // 8) == 0 and pool.kernel_size in (3, (3, 3)) and pool.stride in (2, (2, 2))
          and pool.padding in (1, (1, 1)) and not pool.ceil_mode)
    if not ok:
        return maxpool3x3s2(bn_act(x, bn, relu=True, partial=partial), pool)
    _count_batch(bn)
    return _BnReluMaxPoolFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, partial, torch.is_grad_enabled())

