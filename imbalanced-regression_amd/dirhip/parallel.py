"""One-process-per-GPU data parallelism for the ResNet-regression + FDS training loop.

Replaces the reference's single-process ``torch.nn.DataParallel`` wrap (``train.py:143``): no per-step
scatter of the batch, no per-forward broadcast of 23.5 M parameters + all 8 FDS buffers, no gather
of predictions to GPU 0, no GPU-0 optimizer. Each rank owns one MI355X, its own
batch and a full replica; what is exchanged is

  * per step  — the gradients, summed with RCCL all-reduces over a few large flat float32 buckets
                (23 510 081 elements = 94 MB), launched from autograd hooks as soon as a bucket is
                complete so that they overlap the rest of the backward pass. xGMI is point to point
                (7 links x ~153 GB/s per GPU): a ring is bound by one link, so buckets are few and large
                (default 32 MB -> 3 collectives), not NVSwitch-sized 25 MB defaults tuned elsewhere;
  * per epoch — the FDS (count, mean, M2) statistics (3.3 MB float64, ``fds.merge_stats_across_ranks``).

BatchNorm statistics stay per rank (the reference's replicas also normalise over their own chunk and
never synchronise). ``.module`` exposes the wrapped network like ``DataParallel`` does, so
``model.module.FDS.update_running_stats`` (``train.py:280-281``) and ``module.``-prefixed checkpoints keep working.

Backend: ``"nccl"`` (= RCCL on ROCm) on GPUs; the logic is backend-agnostic and is tested with gloo on CPU.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import gradsink


def init_distributed(backend=None):
    """Initialise the default process group from torchrun's environment (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*).
    Returns (rank, world_size, local_rank). No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


class _Bucket:
    __slots__ = ("flat", "params", "views", "offsets", "dense", "pending", "work", "ev0", "ev1")

    ALIGN = 64                          # elements: every slot starts on a 256-byte boundary (the gradient kernels store 16-byte vectors)

    def __init__(self, params, device):
        n = sum((p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN for p in params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.params = params
        self.views, self.offsets, self.dense = [], [], []
        off = 0
        for p in params:
            # same memory layout as the parameter (e.g. channels_last conv weights): fused / multi-tensor optimizer
            # kernels walk parameter and gradient storage with one linear index, so the layouts must agree
            dense = p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last) if p.dim() == 4 else p.is_contiguous()
            self.offsets.append(off)
            self.dense.append(bool(dense))
            self.views.append(self.fresh_view(len(self.offsets) - 1))
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.pending = len(params)
        self.work = None
        self.ev0 = self.ev1 = None

    def fresh_view(self, pi):
        """A NEW tensor object over parameter pi's slot (shape and strides of the parameter). The gradient kernels write into it
        (``gradsink``) and autograd's AccumulateGrad adopts it as ``param.grad`` without a copy — which it only does for a tensor
        nobody else references, hence a new object per call."""
        p = self.params[pi]
        seg = self.flat[self.offsets[pi]:self.offsets[pi] + p.numel()]
        return seg.as_strided(p.shape, p.stride()) if self.dense[pi] else seg.view_as(p)


def _conv_batching():
    from . import conv as _conv
    return _conv._WGRAD_BATCH["on"] and _conv._COLLECTIVES_LIVE[0] == 0


def _release_collectives_guard():
    from . import conv as _conv
    _conv._COLLECTIVES_LIVE[0] = max(0, _conv._COLLECTIVES_LIVE[0] - 1)


class DataParallelEngine(nn.Module):
    """Wrap ``module`` for one-process-per-GPU training. Plain ``loss.backward()`` + ``optimizer.step()`` work:
    the gradient exchange is driven by autograd hooks and completes before ``backward()`` returns."""

    def __init__(self, module, process_group=None, bucket_mb=32, amp_dtype=None, channels_last=False,
                 broadcast_from_rank0=True, force_collectives=False, f32_arith="exact"):
        super().__init__()
        self.module = module
        self.process_group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force_collectives: run the N > 1 machinery (parameter broadcast, bucket hooks, the all-reduces, the FDS statistic merge) in a
        # process group of ONE rank too — every collective then is the identity, which is how the RCCL path is executed and checked
        # on a one-GPU box (tests/test_hip_rccl.py); never set by the training entry points
        self._comm = bool(self.world > 1 or (force_collectives and dist.is_initialized()))
        if self._comm:
            # the bucket collectives start from per-parameter hooks, i.e. before the end of the backward pass: every weight gradient must be
            # complete when its node returns — the one-launch-per-pass split-K reduction of conv.py is for the one-process loop
            from . import conv as _conv
            if _conv._WGRAD_BATCH["on"]:
                _conv._WGRAD_BATCH["on"] = False
                _conv.wgrad_flush()
            _conv._COLLECTIVES_LIVE[0] += 1                 # guards the point of use too: turning the switch back on later changes nothing
            import weakref
            weakref.finalize(self, _release_collectives_guard)
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.amp_dtype = amp_dtype
        # arithmetic of the float32 conv stack (amp_dtype None): "exact" = v_mfma_f32_32x32x2_f32, the parity mode; "x3" / "x2" = split-bf16 on the
        # bf16 matrix pipe, float32-grade and 1.4x / 2x faster (conv_f32.arithmetic; `train.py --amp fp32x3`)
        assert f32_arith in ("exact", "x3", "x2"), f32_arith
        self.f32_arith = f32_arith
        self.channels_last = channels_last
        self._buckets = None
        self._bucket_of = {}
        self._signature = None
        self._hooks = []
        self._callback_queued = False
        # counters of avoidable per-step work (tests assert they stay 0 on the product graph) and communication timing
        self.stats = {"grad_copies": 0, "bucket_scale_kernels": 0, "steps": 0, "exposed_comm_ms": 0.0}
        self.measure_comm = False                       # bench.py: HIP events around the tail of the backward pass
        # mean over ranks: RCCL reduces with AVG itself; other backends (gloo: tests) get gradients pre-scaled by 1 / world at the
        # network's output (one [B, 1] kernel; exact for power-of-two world sizes) — never a pass over the 94 MB of buckets
        # — EXCEPT when a backward pass starts with gradients already held (accumulation over several backward passes without
        # zero_grad): the bucket then holds the previous, already averaged gradient, and SUM over ranks of (mean_1 + local_2 / world)
        # would count mean_1 `world` times. That pass sums the unscaled gradients and scales the buckets afterwards instead
        # (SUM(mean_1 + local_2) / world = mean_1 + mean_2, what AVG gives on RCCL); `bucket_scale_kernels` counts those passes.
        self._native_avg = bool(self._comm and dist.get_backend(process_group) == "nccl")
        self._accumulating = False
        self._decided = False
        if channels_last:
            self.module.to(memory_format=torch.channels_last)
        if self._comm and broadcast_from_rank0:
            with torch.no_grad():
                for t in list(self.module.parameters()) + list(self.module.buffers()):
                    dist.broadcast(t, src=dist.get_global_rank(process_group, 0) if process_group else 0,
                                   group=process_group)
        fds = getattr(self.module, "FDS", None)
        if fds is not None:
            fds.process_group = process_group
            if force_collectives:
                fds.force_collectives = True

    def set_amp_dtype(self, amp_dtype, f32_arith=None):
        """Switch the conv stack's arithmetic between runs of the SAME graph: ``torch.bfloat16`` (bf16 MFMA kernels under autocast) or None
        (float32 activations and weights on the float32 kernels, in the arithmetic ``f32_arith``: "exact" / "x3" / "x2"; None keeps the current one).
        Takes effect at the next forward pass; master weights, optimizer state, BatchNorm / FDS buffers are shared (``train.py --amp_switch_epoch``)."""
        assert amp_dtype in (None, torch.bfloat16)
        assert f32_arith in (None, "exact", "x3", "x2"), f32_arith
        self.amp_dtype = amp_dtype
        if f32_arith is not None:
            self.f32_arith = f32_arith

    # ---- forward ----------------------------------------------------------------------------------
    def forward(self, inputs, *args, **kwargs):
        if self.training and torch.is_grad_enabled() and (self._comm or inputs.is_cuda):
            self._prepare_buckets()         # (one process: the buckets are just persistent gradient storage — static addresses for the
                                            #  optimizer kernel's table, no allocation per gradient and step; no hooks, no collective)
        if self.channels_last and inputs.dim() == 4:
            inputs = inputs.contiguous(memory_format=torch.channels_last)
        if self.amp_dtype is not None:
            with torch.autocast(device_type=inputs.device.type, dtype=self.amp_dtype):
                out = self.module(inputs, *args, **kwargs)
        elif self.f32_arith != "exact":
            from .conv_f32 import arithmetic
            with arithmetic(self.f32_arith):
                out = self.module(inputs, *args, **kwargs)
        else:
            out = self.module(inputs, *args, **kwargs)
        if self._comm and not self._native_avg and self.training and torch.is_grad_enabled():
            for t in (out if isinstance(out, (tuple, list)) else (out,)):
                if isinstance(t, torch.Tensor) and t.requires_grad:
                    t.register_hook(self._scale_output_grad)                                  # (the prediction [B, 1]: every parameter gradient flows through it)
        return out

    def _scale_output_grad(self, g):
        """Fires when the backward pass reaches the network output, i.e. before any parameter gradient of this pass exists: the moment
        to see whether the parameters still HOLD gradients (train loops call zero_grad() between forward and backward, so the forward
        cannot know)."""
        if g is None:
            return None
        if not self._decided:
            self._decided = True
            # (held = "not None": a loop that keeps zero-valued gradients — zero_grad(set_to_none=False) — is classed as accumulating on every
            #  step; the result is the same, it only pays one scaling kernel per bucket and step. Looking at the values would cost a device
            #  sync per step; such loops should use set_to_none=True, torch's default.)
            self._accumulating = any(p.grad is not None for b in (self._buckets or []) for p in b.params)
        return None if self._accumulating else g * (1.0 / self.world)

    # ---- gradient buckets -------------------------------------------------------------------------
    def _prepare_buckets(self):
        params = [p for p in self.module.parameters() if p.requires_grad]
        sig = tuple(id(p) for p in params)
        if sig != self._signature:
            for h in self._hooks:
                h.remove()
            if self._buckets:
                gradsink.unregister([p for b in self._buckets for p in b.params])
            self._hooks, self._buckets, self._bucket_of = [], [], {}
            cur, cur_bytes = [], 0
            for p in reversed(params):                       # gradients become ready roughly in reverse order
                cur.append(p)
                cur_bytes += p.numel() * 4
                if cur_bytes >= self.bucket_bytes:
                    self._buckets.append(_Bucket(cur, p.device))
                    cur, cur_bytes = [], 0
            if cur:
                self._buckets.append(_Bucket(cur, cur[0].device))
            for bi, b in enumerate(self._buckets):
                for pi, p in enumerate(b.params):
                    self._bucket_of[id(p)] = (bi, pi)
                    if self._comm:
                        self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad_ready))
                    # the gradient kernels write into the bucket — unless the parameter still HOLDS a gradient (accumulation over
                    # several backward passes, zero_grad(set_to_none=False)): the slot is that gradient, so the kernel gets a new tensor
                    # and autograd accumulates as usual
                    gradsink.register(p, lambda b=b, pi=pi, p=p: b.fresh_view(pi) if p.grad is None else None, slot_ptr=b.views[pi].data_ptr())
            self._signature = sig
        for b in self._buckets:
            b.pending = len(b.params)
            b.work = None
        self._callback_queued = False
        self._accumulating = False          # decided when the backward pass starts (_scale_output_grad)
        self._decided = False

    def _on_grad_ready(self, p):
        assert not _conv_batching(), "live collectives with the batched weight-gradient reduction on: the bucket would be reduced over ranks before its split-K sum"
        if not self._callback_queued:
            torch.autograd.Variable._execution_engine.queue_callback(self._finish)
            self._callback_queued = True
        bi, pi = self._bucket_of[id(p)]
        b = self._buckets[bi]
        view = b.views[pi]
        if p.grad.data_ptr() != view.data_ptr():
            view.copy_(p.grad)                               # a producer that did not write into the bucket (gradsink): copy
            p.grad = view
            self.stats["grad_copies"] += 1
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b):
        if b.flat.is_cuda:
            from . import conv as _conv
            _conv.wgrad_join(b.flat.device)                  # weight gradients written on the side stream (tools/variant_switches.py; off in the product)
        if self.measure_comm and b.flat.is_cuda:
            b.ev0 = torch.cuda.Event(enable_timing=True)
            b.ev0.record()
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.AVG if self._native_avg else dist.ReduceOp.SUM,
                                 group=self.process_group, async_op=True)

    def _finish(self):
        for b in self._buckets:
            if b.work is None:                               # bucket with parameters that got no gradient
                for p, v in zip(b.params, b.views):
                    if p.grad is None:
                        v.zero_()
                        p.grad = v
                    elif p.grad.data_ptr() != v.data_ptr():
                        v.copy_(p.grad)
                        p.grad = v
                        self.stats["grad_copies"] += 1
                self._launch(b)
        ev_a = ev_b = None
        if self.measure_comm and self._buckets[0].flat.is_cuda:
            ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev_a.record()                                    # the compute stream has everything of the backward pass queued
        for b in self._buckets:
            b.work.wait()                                    # (the compute stream waits for the collective; the host does not)
            if self._accumulating and not self._native_avg:
                b.flat.mul_(1.0 / self.world)
                self.stats["bucket_scale_kernels"] += 1
        if ev_b is not None:
            ev_b.record()
            self._pending_events = (ev_a, ev_b)
        self.stats["steps"] += 1
        self._callback_queued = False
        self._decided = False               # every backward pass decides for itself (forward, forward, backward, backward: the second
                                            # pass finds the first one's averaged gradients held and must take the accumulation form)
        for b in self._buckets:             # ... and counts its own gradients: a second backward pass without a forward in between
            b.pending = len(b.params)       # (which is what re-arms the buckets otherwise) would never launch its collectives
            b.work = None

    def comm_report(self):
        """Observability of the N > 1 path (bench.py): ranks, bucket sizes, and — when ``measure_comm`` was on — the time the compute
        stream spent between the last backward kernel and the last collective (= communication NOT hidden behind the backward)."""
        rep = {"ranks": self.world, "backend": dist.get_backend(self.process_group) if self._comm else None,
               "reduce_op": "avg (in the collective)" if self._native_avg else "sum of gradients pre-scaled by 1/ranks at the network output",
               "buckets_MB": [round(b.flat.numel() * 4 / 2 ** 20, 2) for b in (self._buckets or [])],
               "grad_copies": self.stats["grad_copies"], "bucket_scale_kernels": self.stats["bucket_scale_kernels"], "steps": self.stats["steps"]}
        ev = getattr(self, "_pending_events", None)
        if ev is not None:
            torch.cuda.synchronize()
            rep["exposed_comm_ms_last_step"] = ev[0].elapsed_time(ev[1])
        return rep


def shard_indices(n, rank, world, epoch_seed=None, with_valid=False):
    """Indices of this rank's shard of an n-sample epoch (shuffled identically on every rank when a seed is
    given; padded by wrap-around so every rank runs the same number of steps). ``with_valid=True`` also returns a bool
    mask that is False at the padded (duplicated) positions: they are trained on like any sample (as with a
    DistributedSampler) but must not enter the FDS epoch statistics, which are statistics of the dataset."""
    if epoch_seed is None:
        order = torch.arange(n)
    else:
        order = torch.randperm(n, generator=torch.Generator().manual_seed(int(epoch_seed)))
    per = (n + world - 1) // world
    pad = per * world - n
    valid = torch.ones(n + pad, dtype=torch.bool)
    if pad:
        order = torch.cat([order, order[:pad]])
        valid[n:] = False
    if with_valid:
        return order[rank::world], valid[rank::world]
    return order[rank::world]
