"""Fused BatchNorm2d (+ residual add) (+ ReLU) for channels_last activations — one autograd node per BN layer,
three HIP launches forward (statistics, finalize, apply) and three backward (reduce, finalize, apply) instead of
MIOpen BatchNorm + separate add / ReLU / ReLU-backward kernels (``dir_bn_*`` in ``include/dir_hip.h``).

``bn_act(x, bn, relu, residual)`` reads its parameters and running statistics from the ``nn.BatchNorm2d`` module
``bn`` that ``resnet.py`` registers (so state_dict keys and checkpoint compatibility are untouched) and implements
exactly ``relu(bn(x) + residual)`` with torch's BatchNorm semantics (biased variance for normalisation, unbiased
for ``running_var``, ``momentum`` blend, ``num_batches_tracked += 1``).
"""
import torch

from . import _lib as L

_DT = {torch.bfloat16: L.DIR_BF16, torch.float32: L.DIR_F32}


def _nhwc(t):
    return t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)


def _ws(dtype_code, m, c, device):
    n = L.lib().dir_bn_workspace(dtype_code, m, c)
    if n == 0:
        raise L.DirHipError(f"dir_bn: unsupported shape M={m} C={c}")
    return torch.empty(n, dtype=torch.uint8, device=device)


class _BNActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, momentum, eps, relu, training, partial,
                deferred=None):
        if not x.is_cuda:
            raise L.DirHipError(f"bn_act: input on {x.device}; the fused BatchNorm runs only as HIP kernels (no CPU fallback)")
        x = _nhwc(x)
        if residual is not None:
            residual = _nhwc(residual)
            if residual.dtype != x.dtype:
                residual = residual.to(x.dtype)
        n, c, h, w = x.shape
        m = n * h * w
        code = _DT[x.dtype]
        y = torch.empty_like(x)
        ws = _ws(code, m, c, x.device)
        stream = L.stream_ptr(x.device)
        if training:
            mean = torch.empty(c, dtype=torch.float32, device=x.device)
            rstd = torch.empty(c, dtype=torch.float32, device=x.device)
            if partial is not None:                  # statistics fused into the producing convolution's epilogue
                L.check(L.lib().dir_bn_fwd_train_partials(L.ptr(x), L.ptr(residual), L.ptr(y), code, m, c, L.ptr(partial),
                                                          partial.shape[0], L.ptr(gamma), L.ptr(beta), L.ptr(running_mean),
                                                          L.ptr(running_var), float(momentum), float(eps), int(relu),
                                                          L.ptr(mean), L.ptr(rstd), L.ptr(ws), ws.numel(), stream),
                        "dir_bn_fwd_train_partials")
            else:
                L.check(L.lib().dir_bn_fwd_train(L.ptr(x), L.ptr(residual), L.ptr(y), code, m, c, L.ptr(gamma), L.ptr(beta),
                                                 L.ptr(running_mean), L.ptr(running_var), float(momentum), float(eps),
                                                 int(relu), L.ptr(mean), L.ptr(rstd), L.ptr(ws), ws.numel(), stream),
                        "dir_bn_fwd_train")
            # the saved output is only needed as the ReLU mask when a residual was added; otherwise the backward
            # recomputes the mask from x (one tensor read less per pass)
            ctx.save_for_backward(x, gamma, beta, y if (relu and residual is not None) else None, mean, rstd)
            ctx.relu = bool(relu)
            ctx.has_res = residual is not None
            ctx.deferred = deferred
        else:
            L.check(L.lib().dir_bn_fwd_eval(L.ptr(x), L.ptr(residual), L.ptr(y), code, m, c, L.ptr(gamma), L.ptr(beta),
                                            L.ptr(running_mean), L.ptr(running_var), float(eps), int(relu), L.ptr(ws),
                                            ws.numel(), stream), "dir_bn_fwd_eval")
            ctx.eval_mode = True
        return y

    @staticmethod
    def backward(ctx, dout):
        if getattr(ctx, "eval_mode", False):
            raise NotImplementedError("bn_act backward in eval mode is not implemented (the reference never needs it)")
        x, gamma, beta, y, mean, rstd = ctx.saved_tensors
        dout = _nhwc(dout)
        if dout.dtype != x.dtype:
            dout = dout.to(x.dtype)
        n, c, h, w = x.shape
        m = n * h * w
        code = _DT[x.dtype]
        dx = torch.empty_like(x)
        relu = ctx.relu
        if ctx.deferred is not None and ctx.deferred[0]:
            # the consumer's data-gradient kernel already applied this node's ReLU backward (conv._ConvFn, relu_input):
            # dout IS the masked gradient, which is also the shortcut's gradient as it stands
            relu, y = False, None
            dres = None
        else:
            dres = torch.empty_like(x) if ctx.has_res else None
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
        ws = _ws(code, m, c, x.device)
        L.check(L.lib().dir_bn_bwd(L.ptr(dout), L.ptr(x), L.ptr(y), L.ptr(dx), L.ptr(dres), code, m, c, L.ptr(gamma),
                                   L.ptr(beta), L.ptr(mean), L.ptr(rstd), L.ptr(dgamma), L.ptr(dbeta), int(relu),
                                   L.ptr(ws), ws.numel(), L.stream_ptr(x.device)), "dir_bn_bwd")
        if dres is None and ctx.has_res:
            dres = dout
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None, None


def bn_act(x, bn, relu=True, residual=None, partial=None, defer_relu_grad=False):
    """``relu(bn(x) + residual)`` for a channels_last tensor with ``bn`` an ``nn.BatchNorm2d``. ``partial`` =
    the ``[rows][2][C]`` statistics partials emitted by ``conv.conv_bn_input`` for this very ``x`` (training only).
    ``defer_relu_grad``: the result carries a flag (``._dir_relu_flag``) that the ONE consumer of the result may claim
    (``conv.conv_bn_input(..., relu_flag=)``), promising to deliver the gradient with this node's ReLU backward already
    applied; the backward here then skips the mask (and the read of the saved output, and the shortcut-gradient write)."""
    training = bn.training or (bn.running_mean is None)
    if training and bn.track_running_stats and bn.num_batches_tracked is not None:
        counter = getattr(bn, "_dir_step_counter", None)
        if counter is not None:
            counter.pending += 1                 # one fused increment for all layers (flush_batch_counters)
        else:
            bn.num_batches_tracked.add_(1)
    if bn.momentum is None:
        raise NotImplementedError("cumulative-average BatchNorm (momentum=None) is not implemented")
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    deferred = [False] if (defer_relu_grad and relu and training and torch.is_grad_enabled()) else None
    y = _BNActFn.apply(x, bn.weight, bn.bias, residual, rm, rv, bn.momentum, bn.eps, relu, training,
                       partial if training else None, deferred)
    if deferred is not None:
        y._dir_relu_flag = deferred
    return y


class BatchCounters:
    """All ``num_batches_tracked`` buffers of a network as views of ONE int64 tensor: 53 one-element add kernels per
    forward become one. ``link(model)`` re-points the buffers (state_dict keys / values unchanged); ``flush()`` adds
    the forwards seen since the last flush. Re-link after ``model.to(device)``."""

    def __init__(self):
        self.flat = None
        self.pending = 0
        self.n_layers = 0

    def link(self, model):
        bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d) and m.num_batches_tracked is not None]
        if not bns:
            return self
        dev = bns[0].num_batches_tracked.device
        self.flat = torch.stack([m.num_batches_tracked.detach().to(dev) for m in bns]).contiguous()
        for i, m in enumerate(bns):
            m.num_batches_tracked = self.flat[i]
            m._dir_step_counter = self
        self.n_layers = len(bns)
        self.pending = 0
        return self

    def flush(self):
        if self.flat is not None and self.pending:
            self.flat.add_(self.pending // self.n_layers)
            self.pending = 0
