"""Weighted regression losses — drop-in for ``imdb-wiki-dir/loss.py`` (= ``agedb-dir/loss.py``).

Same five names and signatures (``train.py:255`` resolves them through ``globals()``), same values
and gradients; each call is ONE hand-written HIP kernel (``dir_weighted_loss``) that emits the scalar
loss and d loss / d inputs together, instead of 3-6 element-wise torch kernels forward and as many
in the autograd backward. No CPU fallback.
"""
import torch
import torch.nn.functional as F  # noqa: F401  (re-exported: reference modules rely on star-imports)

from . import _lib as L
from . import ops


class _WeightedLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, targets, weights, kind, beta, gamma, activate):
        need_grad = inputs.requires_grad
        loss, dx_unit = ops.weighted_loss(kind, inputs, targets, weights, beta, gamma, activate, need_grad)
        if need_grad:
            ctx.save_for_backward(dx_unit)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        (dx_unit,) = ctx.saved_tensors
        grad_out = grad_out.contiguous().to(torch.float32)
        return ops.scale_by_device_scalar(dx_unit, grad_out), None, None, None, None, None, None


def _flat_f32(t, shape, name):
    if t.shape != shape:
        t = t.expand(shape)
    if t.dtype != torch.float32:
        t = t.float()
    return L.require_device_tensor(t.contiguous(), torch.float32, name).view(-1)


def _weighted(kind, inputs, targets, weights, beta=0.0, gamma=1.0, activate='sigmoid'):
    if not inputs.is_cuda:
        raise L.DirHipError(f"weighted_{kind}_loss: inputs on {inputs.device}; the loss runs only as a HIP kernel "
                            f"on an AMD GPU (no CPU fallback)")
    if targets.requires_grad or (weights is not None and weights.requires_grad):
        raise NotImplementedError("gradients w.r.t. targets/weights are not produced (the reference never uses them)")
    shape = torch.broadcast_shapes(inputs.shape, targets.shape)
    x = inputs if inputs.shape == shape else inputs.expand(shape)
    x = (x if x.dtype == torch.float32 else x.float()).contiguous().view(-1)
    y = _flat_f32(targets, shape, "targets")
    w = None if weights is None else _flat_f32(weights, shape, "weights")      # weights.expand_as(loss)
    return _WeightedLossFn.apply(x, y, w, kind, float(beta), float(gamma), activate)


def weighted_mse_loss(inputs, targets, weights=None):
    return _weighted('mse', inputs, targets, weights)


def weighted_l1_loss(inputs, targets, weights=None):
    return _weighted('l1', inputs, targets, weights)


def weighted_focal_mse_loss(inputs, targets, weights=None, activate='sigmoid', beta=.2, gamma=1):
    return _weighted('focal_mse', inputs, targets, weights, beta, gamma, activate)


def weighted_focal_l1_loss(inputs, targets, weights=None, activate='sigmoid', beta=.2, gamma=1):
    return _weighted('focal_l1', inputs, targets, weights, beta, gamma, activate)


def weighted_huber_loss(inputs, targets, weights=None, beta=1.):
    return _weighted('huber', inputs, targets, weights, beta)
