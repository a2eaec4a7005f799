"""``Adam`` — drop-in for the ``torch.optim.Adam`` that ``imdb-wiki-dir/train.py:161-162`` builds (same constructor keywords, same
``state_dict`` layout: ``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter, so reference optimizer checkpoints load and save).

``step()`` runs ONE hand-written HIP launch for all parameter tensors (``dir_adam_step``: torch's single-tensor Adam arithmetic) and
rewrites, in the same pass, the bf16 operands of every convolution weight the MFMA kernels will read in the next forward / data
gradient — the work of torch's multi-tensor fused Adam (5 launches) + ``dir_conv_prep_weights_batched`` before. Anything the
kernel does not take (amsgrad, maximize, CPU / non-float32 / non-dense parameters, a differentiable step) goes to
``torch.optim.Adam``'s own ``step`` — there is no silent change of arithmetic.
"""
import torch

from . import _lib as L
from . import conv as _conv


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        kw.pop("fused", None)
        kw.pop("foreach", None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, foreach=False, fused=False, **kw)
        self._tables = {}                  # group index -> (key, device table)

    def _ours(self, group, params, relay):
        """Does the kernel take this group? Pure: state tensors that need the parameter's strides are only LISTED in ``relay``
        (``(state dict, key, parameter)``); ``step`` re-lays them out once every group has passed (no state change before the
        decision to fall back to torch's own step is final)."""
        if group.get("amsgrad") or group.get("maximize") or group.get("differentiable") or group.get("capturable"):
            return False
        if isinstance(group["lr"], torch.Tensor):
            return False
        dev = params[0].device
        for p in params:
            g = p.grad
            if not (p.is_cuda and p.device == dev and p.dtype == torch.float32 and g.dtype == torch.float32 and not g.is_sparse):
                return False
            if p.stride() != g.stride() or not (p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))):
                return False
            st = self.state.get(p)
            if st:                                                          # the kernel walks param, grad and both moments with ONE linear index
                for k in ("exp_avg", "exp_avg_sq"):
                    t = st.get(k)
                    if not (isinstance(t, torch.Tensor) and t.dtype == torch.float32 and t.device == dev and t.shape == p.shape):
                        return False
                    if t.stride() != p.stride():
                        # e.g. a reference (NCHW-contiguous) optimizer checkpoint loaded next to channels_last parameters:
                        # load_state_dict keeps the loaded strides. Re-lay the moment out once, same values, the parameter's strides
                        relay.append((st, k, p))
        return True

    def _drop_tables(self):
        self._tables = {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._drop_tables()                                                 # the cached table holds addresses of the replaced state tensors

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if hasattr(self, "_tables"):
            self._drop_tables()

    def __setstate__(self, state):
        super().__setstate__(state)
        self._tables = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        relay = []
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            if not self._ours(group, params, relay):
                super().step(None)                                          # torch's own step for every group (never a mix)
                return loss
        for st, k, p in relay:
            st[k] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(st[k])
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            step = None
            for p in params:
                st = self.state[p]
                if len(st) == 0:                                            # torch's lazy state initialisation (adam.py: _init_group)
                    st["step"] = torch.tensor(0.0, dtype=torch.get_default_dtype())
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                s = int(st["step"])
                if step is None:
                    step = s
                elif s != step:
                    raise L.DirHipError("dirhip.optim.Adam: parameters of one group with different step counts")
            dev = params[0].device
            key = (_conv.registry_size(dev),) + tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr()) for p in params)
            cached = self._tables.get(gi)
            if cached is None or cached[0] != key:
                prepared = _conv.prepared_operands(dev)                     # master-weight address -> the layer's bf16 operand buffers
                rows = []
                for p in params:
                    st = self.state[p]
                    pw = prepared.get(p.data_ptr())
                    if pw is not None and not p.is_contiguous(memory_format=torch.channels_last):
                        pw = None
                    if pw is None:
                        rows.append([p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), 0, 0, 0, 0, 0, 0, 0])
                    else:
                        cout, rs, cin = pw.shape
                        rows.append([p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), pw.w16.data_ptr(),
                                     0 if pw.w16_rot is None else pw.w16_rot.data_ptr(), cout, rs, cin, pw.rot_mode, 0])
                cached = (key, torch.tensor(rows, dtype=torch.int64).to(dev), [prepared.get(p.data_ptr()) for p in params])
                self._tables[gi] = cached
            b1, b2 = group["betas"]
            L.check(L.lib().dir_adam_step(L.ptr(cached[1]), len(params), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                          float(group["weight_decay"]), step, L.stream_ptr(dev)), "dir_adam_step")
            _conv.mark_prepared_after_step([pw for pw in cached[2] if pw is not None])
        return loss


class SGD(torch.optim.SGD):
    """Drop-in for the ``torch.optim.SGD`` of ``imdb-wiki-dir/train.py:163-164`` (``--optimizer sgd``: lr, momentum, weight decay; same
    ``state_dict`` layout: ``momentum_buffer`` per parameter). ``step()`` = ONE HIP launch for all parameter tensors (``dir_sgd_step``:
    torch's single-tensor SGD arithmetic) that also rewrites the bf16 operands of the convolution weights — as ``Adam`` above. What the
    kernel does not take (maximize, differentiable, CPU / non-float32 / non-dense parameters, parameters of one group in different
    momentum-buffer states) goes to ``torch.optim.SGD``'s own step."""

    def __init__(self, params, lr=1e-3, momentum=0, dampening=0, weight_decay=0, nesterov=False, **kw):
        kw.pop("fused", None)
        kw.pop("foreach", None)
        super().__init__(params, lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov, foreach=False, fused=False, **kw)
        self._tables = {}

    def _ours(self, group, params, relay):
        """As ``Adam._ours``: pure, re-layouts only listed."""
        if group.get("maximize") or group.get("differentiable") or isinstance(group["lr"], torch.Tensor):
            return False
        dev = params[0].device
        have = None
        for p in params:
            g = p.grad
            if not (p.is_cuda and p.device == dev and p.dtype == torch.float32 and g.dtype == torch.float32 and not g.is_sparse):
                return False
            if p.stride() != g.stride() or not (p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))):
                return False
            buf = self.state.get(p, {}).get("momentum_buffer")
            if have is None:
                have = buf is not None
            elif have != (buf is not None):
                return False                                                # mixed first / later steps inside one group: torch's loop handles it
            if buf is not None:
                if not (buf.dtype == torch.float32 and buf.device == dev and buf.shape == p.shape):
                    return False
                if buf.stride() != p.stride():                              # e.g. a reference checkpoint's NCHW buffers next to channels_last parameters
                    relay.append((self.state[p], "momentum_buffer", p))
        return True

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if hasattr(self, "_tables"):
            self._tables = {}

    def __setstate__(self, state):
        super().__setstate__(state)
        self._tables = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        relay = []
        for group in self.param_groups:
            params = [p for p in group["params"] if p.grad is not None]
            if params and not self._ours(group, params, relay):
                super().step(None)                                          # torch's own step for every group (never a mix)
                return loss
        for st, k, p in relay:
            st[k] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(st[k])
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            mom = float(group["momentum"])
            first = 0
            new_bufs = None
            if mom != 0.0:
                first = int(self.state[params[0]].get("momentum_buffer") is None)
                if first:
                    # filled by the kernel (= clone(grad)); they enter the optimizer state only once the launch was accepted: a failed
                    # launch must not leave uninitialised buffers behind that the next step would treat as momentum
                    new_bufs = [torch.empty_like(p, memory_format=torch.preserve_format) for p in params]
            dev = params[0].device
            bufs = new_bufs if new_bufs is not None else [self.state[p]["momentum_buffer"] if mom != 0.0 else None for p in params]
            key = (_conv.registry_size(dev),) + tuple((p.data_ptr(), p.grad.data_ptr(), 0 if b is None else b.data_ptr()) for p, b in zip(params, bufs))
            cached = self._tables.get(gi)
            if cached is None or cached[0] != key:
                prepared = _conv.prepared_operands(dev)
                rows = []
                for p, b in zip(params, bufs):
                    pw = prepared.get(p.data_ptr())
                    if pw is not None and not p.is_contiguous(memory_format=torch.channels_last):
                        pw = None
                    head = [p.data_ptr(), p.grad.data_ptr(), 0 if b is None else b.data_ptr(), 0, p.numel()]
                    if pw is None:
                        rows.append(head + [0, 0, 0, 0, 0, 0, 0])
                    else:
                        cout, rs, cin = pw.shape
                        rows.append(head + [pw.w16.data_ptr(), 0 if pw.w16_rot is None else pw.w16_rot.data_ptr(), cout, rs, cin, pw.rot_mode, 0])
                cached = (key, torch.tensor(rows, dtype=torch.int64).to(dev), [prepared.get(p.data_ptr()) for p in params])
                self._tables[gi] = cached
            L.check(L.lib().dir_sgd_step(L.ptr(cached[1]), len(params), float(group["lr"]), mom, float(group["dampening"]), float(group["weight_decay"]),
                                         int(bool(group["nesterov"])), first, L.stream_ptr(dev)), "dir_sgd_step")
            if new_bufs is not None:
                for p, b in zip(params, new_bufs):
                    self.state[p]["momentum_buffer"] = b
            _conv.mark_prepared_after_step([pw for pw in cached[2] if pw is not None])
        return loss
