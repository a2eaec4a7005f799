"""Where a parameter's gradient is WRITTEN. ``parallel.DataParallelEngine`` keeps every gradient in a few flat float32
buckets (one RCCL all-reduce each); instead of letting the backward kernels write freshly allocated tensors that a hook
then copies into the bucket (161 copy kernels per step), the engine registers, per parameter, a factory of views into the
parameter's slot of its bucket, and the gradient-producing nodes (weight gradients of the convolutions, BatchNorm
dgamma / dbeta, the tail's linear layer) allocate their outputs through ``out_for``: the kernels write straight into the
bucket. Every call hands out a FRESH view object, so autograd's AccumulateGrad takes it over as ``param.grad`` without a copy.
Single process (no engine buckets): ``out_for`` is ``torch.empty``.
"""
import weakref

import torch

_SINKS = {}                       # parameter storage address -> zero-argument callable returning a fresh view of its bucket slot
_SLOT_OWNER = {}                  # bucket-slot address -> weak reference to the parameter whose gradient lives there


def register(param, factory, slot_ptr=None):
    _SINKS[param.data_ptr()] = factory
    if slot_ptr is not None:
        _SLOT_OWNER[int(slot_ptr)] = weakref.ref(param)


def unregister(params):
    for p in params:
        _SINKS.pop(p.data_ptr(), None)
    dead = [k for k, r in _SLOT_OWNER.items() if r() is None or any(r() is p for p in params)]
    for k in dead:
        _SLOT_OWNER.pop(k, None)


def owner_of_slot(slot_ptr):
    """The parameter whose gradient-bucket slot starts at ``slot_ptr`` (None: not a registered slot, or the parameter is gone)."""
    r = _SLOT_OWNER.get(slot_ptr)
    return r() if r is not None else None


def lookup(param):
    """The factory for ``param`` (any tensor object that shares the parameter's storage address), or None."""
    return _SINKS.get(param.data_ptr()) if _SINKS else None


def out_for_ex(param_or_factory, shape, device, memory_format=torch.contiguous_format):
    """``(tensor, from_sink)``: the float32 output tensor of ``shape`` for the gradient of ``param`` — its bucket slot when an engine
    registered one (and the geometry matches; ``from_sink`` True: persistent storage that autograd adopts as ``param.grad``), else a new tensor."""
    f = param_or_factory if (param_or_factory is None or callable(param_or_factory)) else lookup(param_or_factory)
    if f is not None:
        v = f()
        if v is not None and v.device == device:
            if tuple(v.shape) == tuple(shape):
                if memory_format == torch.contiguous_format or v.is_contiguous(memory_format=memory_format):
                    return v, True
            elif v.numel() == int(torch.Size(shape).numel()) and v.is_contiguous():
                return v.view(shape), True
    return torch.empty(shape, dtype=torch.float32, device=device, memory_format=memory_format), False


def out_for(param_or_factory, shape, device, memory_format=torch.contiguous_format):
    """Float32 output tensor of ``shape`` for the gradient of ``param``: its bucket slot when an engine registered one (and the
    geometry matches), else a new tensor."""
    return out_for_ex(param_or_factory, shape, device, memory_format)[0]
