"""Fused BatchNorm2d (+ residual add) (+ ReLU) for channels_last activations — one autograd node per BN layer,
two to three HIP launches forward (statistics — or none when the producing convolution's epilogue supplied them —,
finalize, apply) and three backward (reduce, finalize, apply) instead of the library BatchNorm + separate add / ReLU /
ReLU-backward kernels (``dir_bn_*`` in ``include/dir_hip.h``).

Also here: ``bn_join`` — the projection-shortcut join ``relu(bn(x) + bn_r(r))`` with both normalisations applied in one
pass — and the hand-over that lets the consumer of a ``relu(bn(x) + shortcut)`` output apply that ReLU's backward inside
its own data-gradient kernel (``defer_relu_grad``).

``bn_act(x, bn, relu, residual)`` reads its parameters and running statistics from the ``nn.BatchNorm2d`` module
``bn`` that ``resnet.py`` registers (so state_dict keys and checkpoint compatibility are untouched) and implements
exactly ``relu(bn(x) + residual)`` with torch's BatchNorm semantics (biased variance for normalisation, unbiased
for ``running_var``, ``momentum`` blend, ``num_batches_tracked += 1``).
"""
import torch

from . import _lib as L
from . import gradsink

_DT = {torch.bfloat16: L.DIR_BF16, torch.float32: L.DIR_F32}


def _nhwc(t):
    return t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)


def _ws(dtype_code, m, c, device):
    n = L.lib().dir_bn_workspace(dtype_code, m, c)
    if n == 0:
        raise L.DirHipError(f"dir_bn: unsupported shape M={m} C={c}")
    return torch.empty(n, dtype=torch.uint8, device=device)


class BwdLink:
    """Hand-over from a BatchNorm node to the ONE convolution node that consumes its output (``bn_act(fuse_bwd_stats=True)``
    -> ``conv.conv_bn_input`` / ``conv.projection_pair``): that node's data-gradient kernel produces this BatchNorm's ``dout``,
    so it also forms the per-channel sums of ``dout`` and ``dout * x`` in its store loop (``dir_conv_dgrad_bnstats``) and
    leaves them in ``partial``; the BatchNorm backward then runs without its reduction pass (``dir_bn_bwd_partials``).
    ``needs_relu_claim``: the BatchNorm output went through ``relu(. + residual)``, so the sums are only right when the
    consumer also applies that ReLU's backward (``defer_relu_grad`` claimed).
    CONTRACT (checked where it is cheap): the BatchNorm output has exactly ONE consumer and that consumer's backward runs first; it
    records the gradient tensor it wrote (``note_dout``: the tensor itself, its storage address and its version counter) next to
    ``partial``, and the BatchNorm backward takes the sums only if the ``dout`` autograd hands it IS that tensor, unmodified — if
    anything else contributed to the gradient (an auxiliary head, a hook, another wiring of the block) autograd has summed into a
    new tensor and the reduction pass runs as usual. Holding the tensor here is what makes that reliable in either arrival order:
    autograd's input buffer accumulates IN PLACE into a first-arrived gradient nobody else references (same address, so an address
    test alone would pass); a second reference forces the out-of-place sum, and the version counter catches any other in-place edit."""
    __slots__ = ("x", "gamma", "beta", "mean", "rstd", "recompute_mask", "needs_relu_claim", "partial", "dout_ptr", "dout_ref", "dout_version")

    def __init__(self):
        self.x = self.gamma = self.beta = self.mean = self.rstd = self.partial = None
        self.recompute_mask = self.needs_relu_claim = False
        self.dout_ptr = self.dout_ref = self.dout_version = None

    def note_dout(self, t):
        self.dout_ref, self.dout_ptr, self.dout_version = t, t.data_ptr(), t._version

    def is_dout(self, t):
        ok = self.dout_ref is not None and self.dout_ptr == t.data_ptr() and self.dout_version == t._version and t.shape == self.dout_ref.shape
        self.dout_ref = None
        return ok


class _BNActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, momentum, eps, relu, training, partial,
                deferred=None, link=None, bits=None):
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
            if bits is not None:                     # + the ReLU's backward mask, one bit per element (deferred ReLU backward)
                L.check(L.lib().dir_bn_fwd_train_bits(L.ptr(x), L.ptr(residual), L.ptr(y), m, c, L.ptr(partial),
                                                      0 if partial is None else partial.shape[0], L.ptr(gamma), L.ptr(beta),
                                                      L.ptr(running_mean), L.ptr(running_var), float(momentum), float(eps),
                                                      L.ptr(mean), L.ptr(rstd), L.ptr(bits), L.ptr(ws), ws.numel(), stream),
                        "dir_bn_fwd_train_bits")
            elif partial is not None:                # statistics fused into the producing convolution's epilogue
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
            ctx.link = link
            if link is not None:
                link.x, link.gamma, link.beta, link.mean, link.rstd = x, gamma, beta, mean, rstd
                link.recompute_mask = bool(relu) and residual is None
                link.needs_relu_claim = bool(relu) and residual is not None
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
        dgamma = gradsink.out_for(gamma, (c,), x.device)     # (the data-parallel engine's bucket slot when there is one)
        dbeta = gradsink.out_for(beta, (c,), x.device)
        ws = _ws(code, m, c, x.device)
        link = getattr(ctx, "link", None)
        part = None
        if link is not None:
            part, same = link.partial, link.is_dout(dout)    # (is_dout also drops the link's reference to the gradient)
            if part is not None and not same:
                part, link.partial = None, None                  # the sums are of another tensor than the gradient we were handed
        if part is not None and dres is None and y is None:
            # the consumer's data-gradient kernel already summed dout and dout * x per channel (BwdLink): finalize + apply only
            link.partial = None
            L.check(L.lib().dir_bn_bwd_partials(L.ptr(dout), L.ptr(x), L.ptr(dx), code, m, c, L.ptr(gamma), L.ptr(beta), L.ptr(mean),
                                                L.ptr(rstd), L.ptr(dgamma), L.ptr(dbeta), int(relu), L.ptr(part), part.shape[0],
                                                L.ptr(ws), ws.numel(), L.stream_ptr(x.device)), "dir_bn_bwd_partials")
        else:
            L.check(L.lib().dir_bn_bwd(L.ptr(dout), L.ptr(x), L.ptr(y), L.ptr(dx), L.ptr(dres), code, m, c, L.ptr(gamma),
                                       L.ptr(beta), L.ptr(mean), L.ptr(rstd), L.ptr(dgamma), L.ptr(dbeta), int(relu),
                                       L.ptr(ws), ws.numel(), L.stream_ptr(x.device)), "dir_bn_bwd")
        if dres is None and ctx.has_res:
            dres = dout
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None, None, None, None


class _BNJoinFn(torch.autograd.Function):
    """``relu(bn(x) + bn_r(r))`` — the projection-shortcut join — as one node: both BatchNorms are prepared, one apply
    pass normalises both inputs and adds them (``dir_bn_prepare_train`` x2 + ``dir_bn_apply``); the backward is the two
    BatchNorm backwards on the shared (masked) gradient. Training mode, channels_last only."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, momentum, eps, partial, r, gamma_r, beta_r, rm_r, rv_r, momentum_r, eps_r,
                partial_r, relu, deferred, bits=None):
        x, r = _nhwc(x), _nhwc(r)
        if r.dtype != x.dtype:
            r = r.to(x.dtype)
        n, c, h, w = x.shape
        m = n * h * w
        code = _DT[x.dtype]
        dev = x.device
        stream = L.stream_ptr(dev)
        y = torch.empty_like(x)
        ws = _ws(code, m, c, dev)
        f32 = torch.float32
        mean, rstd, mean_r, rstd_r = (torch.empty(c, dtype=f32, device=dev) for _ in range(4))
        coef, coef_r = torch.empty(2, c, dtype=f32, device=dev), torch.empty(2, c, dtype=f32, device=dev)
        for (t, part, g_, b_, rm_, rv_, mom, ep, mu, rs, cf) in ((x, partial, gamma, beta, rm, rv, momentum, eps, mean, rstd, coef),
                                                                 (r, partial_r, gamma_r, beta_r, rm_r, rv_r, momentum_r, eps_r,
                                                                  mean_r, rstd_r, coef_r)):
            L.check(L.lib().dir_bn_prepare_train(L.ptr(t), code, m, c, L.ptr(part), 0 if part is None else part.shape[0],
                                                 L.ptr(g_), L.ptr(b_), L.ptr(rm_), L.ptr(rv_), float(mom), float(ep), L.ptr(mu),
                                                 L.ptr(rs), L.ptr(cf), L.ptr(ws), ws.numel(), stream), "dir_bn_prepare_train")
        if bits is not None and relu:
            L.check(L.lib().dir_bn_apply_bits(L.ptr(x), L.ptr(r), L.ptr(coef_r), L.ptr(y), m, c, L.ptr(coef), L.ptr(bits), stream),
                    "dir_bn_apply_bits")
        else:
            L.check(L.lib().dir_bn_apply(L.ptr(x), L.ptr(r), L.ptr(coef_r), L.ptr(y), code, m, c, L.ptr(coef), int(relu), stream),
                    "dir_bn_apply")
        ctx.save_for_backward(x, gamma, beta, mean, rstd, r, gamma_r, beta_r, mean_r, rstd_r, y if relu else None)
        ctx.relu = bool(relu)
        ctx.deferred = deferred
        return y

    @staticmethod
    def backward(ctx, dout):
        x, gamma, beta, mean, rstd, r, gamma_r, beta_r, mean_r, rstd_r, y = ctx.saved_tensors
        dout = _nhwc(dout)
        if dout.dtype != x.dtype:
            dout = dout.to(x.dtype)
        n, c, h, w = x.shape
        m = n * h * w
        code = _DT[x.dtype]
        dev = x.device
        stream = L.stream_ptr(dev)
        f32 = torch.float32
        dx, dr = torch.empty_like(x), torch.empty_like(r)
        dgamma, dbeta, dgamma_r, dbeta_r = (gradsink.out_for(t, (c,), dev) for t in (gamma, beta, gamma_r, beta_r))
        link = ctx.deferred
        if (not ctx.relu or (link is not None and link[0])) and _JOIN_BWD[0]:
            # no ReLU, or its backward was applied by the consumer (bn_act doc): dout is the gradient of both BatchNorms as it stands —
            # one reduction pass and one apply pass for the pair
            n_ws = 2 * L.lib().dir_bn_workspace(code, m, c)
            ws2 = torch.empty(n_ws, dtype=torch.uint8, device=dev)
            L.check(L.lib().dir_bn_bwd_join(L.ptr(dout), L.ptr(x), L.ptr(r), L.ptr(dx), L.ptr(dr), code, m, c, L.ptr(gamma), L.ptr(mean),
                                            L.ptr(rstd), L.ptr(gamma_r), L.ptr(mean_r), L.ptr(rstd_r), L.ptr(dgamma), L.ptr(dbeta),
                                            L.ptr(dgamma_r), L.ptr(dbeta_r), L.ptr(ws2), n_ws, stream), "dir_bn_bwd_join")
            return (dx, dgamma, dbeta, None, None, None, None, None, dr, dgamma_r, dbeta_r, None, None, None, None, None, None, None, None)
        ws = _ws(code, m, c, dev)
        if not ctx.relu or (link is not None and link[0]):
            g = dout
            L.check(L.lib().dir_bn_bwd(L.ptr(g), L.ptr(x), None, L.ptr(dx), None, code, m, c, L.ptr(gamma), L.ptr(beta),
                                       L.ptr(mean), L.ptr(rstd), L.ptr(dgamma), L.ptr(dbeta), 0, L.ptr(ws), ws.numel(), stream),
                    "dir_bn_bwd")
        else:
            g = torch.empty_like(x)                          # masked gradient = the shortcut branch's gradient
            L.check(L.lib().dir_bn_bwd(L.ptr(dout), L.ptr(x), L.ptr(y), L.ptr(dx), L.ptr(g), code, m, c, L.ptr(gamma), L.ptr(beta),
                                       L.ptr(mean), L.ptr(rstd), L.ptr(dgamma), L.ptr(dbeta), 1, L.ptr(ws), ws.numel(), stream),
                    "dir_bn_bwd")
        L.check(L.lib().dir_bn_bwd(L.ptr(g), L.ptr(r), None, L.ptr(dr), None, code, m, c, L.ptr(gamma_r), L.ptr(beta_r),
                                   L.ptr(mean_r), L.ptr(rstd_r), L.ptr(dgamma_r), L.ptr(dbeta_r), 0, L.ptr(ws), ws.numel(), stream),
                "dir_bn_bwd")
        return (dx, dgamma, dbeta, None, None, None, None, None, dr, dgamma_r, dbeta_r, None, None, None, None, None, None, None, None)


# The join's two BatchNorm backwards as one reduction + one apply pass (dir_bn_bwd_join). Off = two dir_bn_bwd calls (tests compare).
_JOIN_BWD = [True]


# The deferred ReLU backward reads its mask as one bit per element (emitted by the forward) instead of the bf16 tensor itself.
# Off = the tensor (tests and tools compare the two).
_RELU_BITS = [True]


def _relu_bits_for(x):
    """Buffer for the one-bit-per-element ReLU mask of a bf16 NHWC result shaped like ``x`` (``[N*H*W][C/8]`` bytes), or None
    where the kernels do not emit it (float32 parity mode, channel counts that are not a multiple of 8)."""
    if not _RELU_BITS[0] or x.dtype != torch.bfloat16 or not x.is_cuda or x.shape[1] % 8:
        return None
    n, c, h, w = x.shape
    return torch.empty((n * h * w, c // 8), dtype=torch.uint8, device=x.device)


def _count_batch(bn):
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        counter = getattr(bn, "_dir_step_counter", None)
        if counter is not None:
            counter.pending += 1                 # one fused increment for all layers (BatchCounters.flush)
        else:
            bn.num_batches_tracked.add_(1)


def bn_join(x, bn, partial, r, bn_r, partial_r, relu=True, defer_relu_grad=False):
    """``relu(bn(x) + bn_r(r))`` in training mode with both normalisations applied in one pass (see ``_BNJoinFn``).
    ``partial`` / ``partial_r``: conv-epilogue statistics of x / r or None."""
    assert bn.training and bn_r.training and bn.momentum is not None and bn_r.momentum is not None
    _count_batch(bn)
    _count_batch(bn_r)
    deferred = [False] if (defer_relu_grad and relu and torch.is_grad_enabled()) else None
    trs, trs_r = bn.track_running_stats, bn_r.track_running_stats
    bits = _relu_bits_for(x) if deferred is not None else None
    y = _BNJoinFn.apply(x, bn.weight, bn.bias, bn.running_mean if trs else None, bn.running_var if trs else None, bn.momentum,
                        bn.eps, partial, r, bn_r.weight, bn_r.bias, bn_r.running_mean if trs_r else None,
                        bn_r.running_var if trs_r else None, bn_r.momentum, bn_r.eps, partial_r, relu, deferred, bits)
    if deferred is not None:
        y._dir_relu_flag = deferred
        if bits is not None:
            y._dir_relu_bits = bits
    return y


def bn_act(x, bn, relu=True, residual=None, partial=None, defer_relu_grad=False, fuse_bwd_stats=False):
    """``relu(bn(x) + residual)`` for a channels_last tensor with ``bn`` an ``nn.BatchNorm2d``. ``partial`` =
    the ``[rows][2][C]`` statistics partials emitted by ``conv.conv_bn_input`` for this very ``x`` (training only).
    ``defer_relu_grad``: the result carries a flag (``._dir_relu_flag``) that the ONE consumer of the result may claim
    (``conv.conv_bn_input(..., relu_flag=)``), promising to deliver the gradient with this node's ReLU backward already
    applied; the backward here then skips the mask (and the read of the saved output, and the shortcut-gradient write).
    ``fuse_bwd_stats``: the caller guarantees that the result has exactly ONE consumer and that it is a convolution node of
    ``conv.py``; the result then carries a ``BwdLink`` (``._dir_bn_link``) through which that node's data-gradient kernel
    delivers this BatchNorm's backward reduction (bf16 training only)."""
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
    bits = _relu_bits_for(x) if deferred is not None else None
    link = None
    if fuse_bwd_stats and training and torch.is_grad_enabled() and x.dtype == torch.bfloat16 and x.is_cuda \
            and (residual is None or (relu and deferred is not None)):
        link = BwdLink()
    y = _BNActFn.apply(x, bn.weight, bn.bias, residual, rm, rv, bn.momentum, bn.eps, relu, training,
                       partial if training else None, deferred, link, bits)
    if deferred is not None:
        y._dir_relu_flag = deferred
        if bits is not None:
            y._dir_relu_bits = bits
    if link is not None:
        y._dir_bn_link = link
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
