"""3x3 / stride-2 / pad-1 max pooling for bf16 channels_last activations (``dir_maxpool3x3s2_*``): forward keeps one
argmax byte per element, backward gathers (no atomics). Replaces ``nn.MaxPool2d(3, 2, 1)`` of ``resnet.py:82`` on the
bf16 path; other dtypes / geometries keep using the module."""
import torch

from . import _lib as L


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
    """``module`` = the registered nn.MaxPool2d (used for anything that is not bf16 CUDA 3x3/s2/p1)."""
    ok = (x.is_cuda and x.dtype == torch.bfloat16 and x.shape[1] % 8 == 0 and module.kernel_size in (3, (3, 3))
          and module.stride in (2, (2, 2)) and module.padding in (1, (1, 1)) and not module.ceil_mode)
    return _MaxPoolFn.apply(x) if ok else module(x)


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
    """``module(x).view(N, -1)`` of ``resnet.py:136-137`` (``module`` = the registered nn.AvgPool2d). When the pool
    window is the whole bf16 CUDA map the mean is formed by ``dir_avgpool_fwd`` and returned as float32 ``[N, C]``;
    anything else uses the module."""
    k = module.kernel_size if isinstance(module.kernel_size, tuple) else (module.kernel_size, module.kernel_size)
    if (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and tuple(x.shape[2:]) == tuple(k) and x.shape[1] % 8 == 0
            and module.padding in (0, (0, 0))):
        return _GlobalAvgPoolFn.apply(x)
    y = module(x)
    return y.view(y.size(0), -1)

