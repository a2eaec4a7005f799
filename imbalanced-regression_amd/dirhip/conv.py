"""MFMA convolutions (``dir_conv_*``, ``dir_stem_conv_*``) — tensor-level wrappers and autograd nodes.

``conv2d_igemm(x, w, stride, padding, want_stats)``: x ``[N, Cin, H, W]`` and w ``[Cout, Cin, R, S]``, both bf16 and
channels_last (so their memory is NHWC / ``[Cout][R][S][Cin]``); returns y (channels_last bf16) and, optionally,
the per-tile BatchNorm partial statistics ``[rows][2][Cout]`` float32. The same kernel computes stride-1 data
gradients (rotated weights), with the gradient accumulation of a fan-out (``addend`` / ``addend_s2``) and a ReLU backward
(``relu_mask``) fused into its store loop. ``conv2d_wgrad``: deterministic split-K weight gradient.
``conv_bn_input`` / ``projection_pair`` / ``stem_conv``: the autograd nodes ``resnet.py`` is built from; the bf16 operands
of all layers are re-made by one launch per optimizer step.
"""
import torch

from . import _lib as L
from . import gradsink


def supported(cin, cout, x=None):
    """Does the bf16 MFMA implicit GEMM take this layer? Channel counts must be multiples of 64 and — when the input
    ``x`` [N, C, H, W] is given — every tensor of the layer's forward / data-gradient launches must stay inside the
    kernel's 32-bit byte offsets (``dir_conv_fwd``: N*H*W*C < 2^30 elements, N*H*W < 2^24 rows; reached at a per-GPU
    batch of ~1300 at 224 px). Anything else runs on the float32 kernels of ``conv_f32``."""
    if cin % 64 or cout % 64:
        return False
    if x is not None:
        return rows_supported(cin, cout, x.shape[0] * x.shape[2] * x.shape[3])
    return True


def rows_supported(cin, cout, rows):
    """The 32-bit-offset limits of ``dir_conv_fwd`` / its data gradient for a layer whose LARGER side (input or output map) has
    ``rows`` = N*H*W pixels: rows < 2^24 and rows * max(Cin, Cout) < 2^30 elements (the kernel itself refuses at N*H*W*Cin >= 2^30 /
    M*Cout >= 2^31 with DIR_EUNSUPPORTED; this is the conservative host-side twin that selects the float32 fallback first)."""
    return rows < (1 << 24) and rows * max(cin, cout) < (1 << 30)


def _bn_link_args(link):
    g = link.recompute_mask
    return (L.ptr(link.x), L.ptr(link.gamma) if g else None, L.ptr(link.beta) if g else None, L.ptr(link.mean) if g else None,
            L.ptr(link.rstd) if g else None)


def conv2d_igemm(x, w, stride=1, padding=0, want_stats=False, addend=None, relu_mask=None, addend_s2=None, bn_link=None,
                 relu_bits=None, variant=L.CONV_AUTO):
    """``variant``: ``L.CONV_*`` (``DIR_CONV_*`` of the C-ABI) — the kernel of THIS launch; 0 = the product heuristic. The statistics /
    BatchNorm-partial lists are sized by ``dir_conv_plan_rows`` for that same kernel and the launch checks the number."""
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
    stats, rows = None, 0
    if want_stats or bn_link is not None:
        rows = L.lib().dir_conv_plan_rows(n, h, wd, cin, cout, r, s, stride, padding, int(addend is not None and addend_s2 is not None), variant)
        if rows == 0:
            raise L.DirHipError(f"conv2d_igemm: kernel variant {variant} does not take this geometry ({tuple(x.shape)} * {tuple(w.shape)}, stride {stride})")
    if want_stats:
        stats = torch.empty((rows, 2, cout), dtype=torch.float32, device=x.device)
    if addend is not None:
        assert addend.shape == y.shape and addend.dtype == torch.bfloat16 and not want_stats
        if not addend.is_contiguous(memory_format=torch.channels_last):
            addend = addend.contiguous(memory_format=torch.channels_last)
    if relu_mask is not None:
        assert relu_mask.shape == y.shape and relu_mask.dtype == torch.bfloat16 and not want_stats
        if not relu_mask.is_contiguous(memory_format=torch.channels_last):
            relu_mask = relu_mask.contiguous(memory_format=torch.channels_last)
    if addend_s2 is not None:
        # compact data gradient of a stride-2 1x1 convolution of the same input: added at the even pixels only
        assert addend_s2.shape == (n, cout, ho // 2, wo // 2) and addend_s2.dtype == torch.bfloat16 and not want_stats and stride == 1
        if not addend_s2.is_contiguous(memory_format=torch.channels_last):
            addend_s2 = addend_s2.contiguous(memory_format=torch.channels_last)
    if bn_link is not None or relu_bits is not None:
        # data gradient whose result is the `dout` of the BatchNorm behind bn_link: that node's backward reduction is formed in
        # the store loop (bn.BwdLink); relu_bits: the ReLU mask as the bit-per-element buffer the forward emitted for it
        assert stride == 1 and not want_stats and (relu_bits is None or relu_mask is None)
        part, bn_args = None, (None, None, None, None, None)
        if bn_link is not None:
            assert bn_link.x.shape == y.shape and bn_link.x.dtype == torch.bfloat16
            part = torch.empty((rows, 2, cout), dtype=torch.float32, device=x.device)
            bn_args = _bn_link_args(bn_link)
        if relu_bits is not None:
            assert relu_bits.dtype == torch.uint8 and relu_bits.numel() * 8 == y.numel()
        L.check(L.lib().dir_conv_dgrad_ex(L.ptr(x), L.ptr(w), L.ptr(addend), L.ptr(addend_s2), L.ptr(relu_mask), L.ptr(relu_bits), L.ptr(y),
                                          n, h, wd, cin, cout, r, s, padding, *bn_args, L.ptr(part), rows if part is not None else 0, variant,
                                          L.stream_ptr(x.device)),
                "dir_conv_dgrad_ex")
        if bn_link is not None:
            bn_link.partial = part
            bn_link.note_dout(y)
        return y
    if addend_s2 is not None and variant == L.CONV_AUTO:
        L.check(L.lib().dir_conv_dgrad_join(L.ptr(x), L.ptr(w), L.ptr(addend), L.ptr(addend_s2), L.ptr(relu_mask), L.ptr(y), n, h, wd,
                                            cin, cout, r, s, padding, L.stream_ptr(x.device)), "dir_conv_dgrad_join")
        return y
    if variant != L.CONV_AUTO:
        if addend is None and relu_mask is None and addend_s2 is None:
            L.check(L.lib().dir_conv_fwd_variant(L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(stats), rows if want_stats else 0, n, h, wd, cin, cout, r, s,
                                                 stride, padding, variant, L.stream_ptr(x.device)), "dir_conv_fwd_variant")
        else:
            assert stride == 1 and not want_stats
            L.check(L.lib().dir_conv_dgrad_ex(L.ptr(x), L.ptr(w), L.ptr(addend), L.ptr(addend_s2), L.ptr(relu_mask), None, L.ptr(y), n, h, wd, cin, cout,
                                              r, s, padding, None, None, None, None, None, None, 0, variant, L.stream_ptr(x.device)), "dir_conv_dgrad_ex")
        return (y, stats) if want_stats else y
    L.check(L.lib().dir_conv_fwd_fused(L.ptr(x), L.ptr(w), L.ptr(addend), L.ptr(relu_mask), L.ptr(y), L.ptr(stats), rows if want_stats else 0, n, h, wd,
                                       cin, cout, r, s, stride, padding, L.stream_ptr(x.device)), "dir_conv_fwd")
    return (y, stats) if want_stats else y


# 3x3 / stride-1 weight gradients: all nine taps in one pass over dY and X (dir_conv_wgrad3x3). Off = the per-tap kernel
# (tests and tools compare the two).
_WGRAD3_ALL_TAPS = [True]


# Weight gradients on a SIDE STREAM (tools/variant_switches.py: set_wgrad_side_stream; off in the product): nothing in the backward pass reads dW, so the split-K kernels need
# not sit in the chain  BatchNorm backward -> data gradient -> BatchNorm backward ...; on their own HIP stream they fill the gaps of
# that chain (its ~5 us fold / finalize launches run on an otherwise idle chip, and every kernel has a tail). The side stream waits
# for the producer of dY, the operands are marked as in use on it (the caching allocator must not hand their memory out while the
# kernel runs), and the calling stream waits for the side stream once per backward pass — in an autograd end-of-pass callback, and
# in front of every gradient-bucket collective (parallel.py) — so `.grad` is ready on the caller's stream when backward() returns.
_WGRAD_SIDE = {"on": False, "streams": {}, "dirty": set(), "task": None}


def wgrad_join(device=None):
    """The current stream of ``device`` (default: every device with weight gradients in flight) waits for the side stream."""
    dirty = _WGRAD_SIDE["dirty"]
    for idx in list(dirty):
        if device is not None and torch.device(device).index not in (None, idx):
            continue
        torch.cuda.current_stream(idx).wait_stream(_WGRAD_SIDE["streams"][idx])
        dirty.discard(idx)


def _wgrad_join_callback():
    _WGRAD_SIDE["task"] = None
    wgrad_join()


def _wgrad_side(device):
    """The side stream for this call, made to wait for everything queued on the current stream; None = launch in place."""
    if not _WGRAD_SIDE["on"] or device.type != "cuda":
        return None
    idx = device.index if device.index is not None else torch.cuda.current_device()
    side = _WGRAD_SIDE["streams"].get(idx)
    if side is None:
        side = _WGRAD_SIDE["streams"][idx] = torch.cuda.Stream(idx)
    side.wait_stream(torch.cuda.current_stream(idx))
    return side


def _wgrad_side_done(device, side, *operands):
    for t in operands:
        t.record_stream(side)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    _WGRAD_SIDE["dirty"].add(idx)
    task = torch._C._current_graph_task_id()
    if task == -1:
        wgrad_join(device)                                   # called outside a backward pass: plain stream semantics
    elif _WGRAD_SIDE["task"] != task:
        torch.autograd.Variable._execution_engine.queue_callback(_wgrad_join_callback)
        _WGRAD_SIDE["task"] = task


# ONE split-K reduction launch per backward pass (round 5; OPT-IN: tools/variant_switches.py: ``set_wgrad_batched_reduce(True)``): inside a backward pass a weight-gradient
# node whose output is a gradient-bucket slot (gradsink: persistent storage that autograd adopts as ``param.grad`` by reference) runs only the
# split-K GEMM, leaves its partials in a workspace that persists per layer, and the reduction of ALL such layers is one launch in an autograd
# end-of-pass callback (``dir_conv_wgrad_reduce_batched``: same order per element, bit-identical): 52 launches of 5-8 us per ResNet-50 step
# become one, and ``.grad`` is complete when ``backward()`` returns, as before. Measured NEUTRAL on the device-bound one-GPU loop (same-box A/B,
# 18.435 vs 18.408 ms per train step, profiles/r05_ab_in_process.txt: the small launches already hid in the tails of their neighbours, the
# one large reduction does not), so it is off by default — it removes 51 launches and 52 allocations per step from the HOST side, which is
# what matters when several ranks share a box's cores. Never taken — the per-layer reduction runs at once — outside a backward pass, for
# ordinary (not bucket) outputs (autograd may add them into a held gradient immediately), on the side stream, and when the data-parallel
# engine's collectives are live (their hooks fire per parameter, before the end of the pass).
_WGRAD_BATCH = {"on": False, "pending": {}, "task": None, "ws": {}, "tables": {}}
_COLLECTIVES_LIVE = [0]          # engines whose per-parameter collectives are live (parallel.DataParallelEngine): no batching while > 0


def wgrad_flush():
    """Reduce the pending split-K partials of every device (one launch each) on the device's current stream."""
    st = _WGRAD_BATCH
    st["task"] = None
    for idx, rows in st["pending"].items():
        if not rows:
            continue
        key = tuple(rows)
        tab = st["tables"].get(idx)
        if tab is None or tab[0] != key:                          # static after the first step: persistent workspaces and bucket slots
            tab = (key, torch.tensor(rows, dtype=torch.int64).to(torch.device("cuda", idx)))
            st["tables"][idx] = tab
        rows.clear()
        L.check(L.lib().dir_conv_wgrad_reduce_batched(L.ptr(tab[1]), len(key), L.stream_ptr(torch.device("cuda", idx))), "dir_conv_wgrad_reduce_batched")
        # the reduction wrote the bucket slots: correct only if AccumulateGrad ADOPTED each slot as ``param.grad`` (it clones instead when
        # something else holds the view): a clone taken before this launch holds unreduced memory — fail loudly, never train on it
        for row in key:
            p = gradsink.owner_of_slot(row[3])
            if p is None or p.grad is None or p.grad.data_ptr() != row[3]:
                raise RuntimeError("batched weight-gradient reduction: autograd did not adopt the gradient-bucket slot as .grad for a layer "
                                   "(the gradient it holds was copied before the end-of-pass reduction and is invalid); "
                                   "do not batch this graph (tools/variant_switches.py: set_wgrad_batched_reduce)")


def _wgrad_batch_slot(dw, from_sink, nbytes):
    """The persistent workspace for this layer when its reduction can wait for the end of the backward pass, else None."""
    st = _WGRAD_BATCH
    if not (st["on"] and from_sink and dw.is_cuda and not _WGRAD_SIDE["on"]) or _COLLECTIVES_LIVE[0] > 0:
        return None
    task = torch._C._current_graph_task_id()
    if task == -1 or torch.is_grad_enabled():                    # (grad mode on INSIDE a backward pass = create_graph: AccumulateGrad clones)
        return None
    owner = gradsink.owner_of_slot(dw.data_ptr())
    if owner is None or owner._backward_hooks or getattr(owner, "_post_accumulate_grad_hooks", None):
        return None                                               # hooks see (or replace) the gradient before the end of the pass
    idx = dw.device.index if dw.device.index is not None else torch.cuda.current_device()
    rows = st["pending"].setdefault(idx, [])
    if any(r[3] == dw.data_ptr() for r in rows):                 # the same gradient twice in one pass (shared weights): not batched
        return None
    if st["task"] != task:
        if st["task"] is not None and any(st["pending"].values()):
            wgrad_flush()                                         # (a pass that ended without its callback: nothing may linger)
        torch.autograd.Variable._execution_engine.queue_callback(wgrad_flush)
        st["task"] = task
    key = (idx, dw.data_ptr())
    ws = st["ws"].get(key)
    if ws is None or ws.numel() < nbytes:
        ws = st["ws"][key] = torch.empty(nbytes, dtype=torch.uint8, device=dw.device)
    return ws


def conv2d_wgrad(dy, x, kernel_size, stride=1, padding=0, sink=None, form=L.WGRAD_AUTO):
    """Weight gradient (float32, channels_last ``[Cout, Cin, R, S]``) from bf16 channels_last ``dy`` and ``x``. ``sink``: the
    parameter's gradient-bucket factory (``gradsink.lookup``): the kernel then writes into the data-parallel bucket. ``form``:
    ``L.WGRAD_*`` (``DIR_WGRAD_*``) — the kernel form of THIS launch for the per-tap path; 0 = the product's choice."""
    assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16
    if not dy.is_contiguous(memory_format=torch.channels_last):
        dy = dy.contiguous(memory_format=torch.channels_last)
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    cout, cin = dy.shape[1], x.shape[1]
    dw, from_sink = gradsink.out_for_ex(sink, (cout, cin, kernel_size, kernel_size), x.device, torch.channels_last)
    side = _wgrad_side(x.device)
    if side is None:
        _wgrad_launch(dy, x, dw, kernel_size, stride, padding, form, from_sink)
    else:
        with torch.cuda.stream(side):
            _wgrad_launch(dy, x, dw, kernel_size, stride, padding, form, False)
        _wgrad_side_done(x.device, side, dy, x)
    return dw


def _wgrad_launch(dy, x, dw, kernel_size, stride, padding, form=L.WGRAD_AUTO, from_sink=False):
    import ctypes
    n, cin, h, w = x.shape
    cout = dy.shape[1]
    r = s = kernel_size
    lib = L.lib()

    def pend(ws, splits):
        idx = dw.device.index if dw.device.index is not None else torch.cuda.current_device()
        _WGRAD_BATCH["pending"][idx].append((ws.data_ptr(), int(splits.value), cout * r * s * cin, dw.data_ptr()))
    if kernel_size == 3 and stride == 1 and padding == 1 and _WGRAD3_ALL_TAPS[0]:
        nbytes = lib.dir_conv_wgrad3x3_workspace(n, h, w, cin, cout)
        if nbytes:
            ws = _wgrad_batch_slot(dw, from_sink, nbytes)
            if ws is not None:
                splits = ctypes.c_int(0)
                L.check(lib.dir_conv_wgrad3x3_partials(L.ptr(dy), L.ptr(x), ctypes.addressof(splits), n, h, w, cin, cout, L.ptr(ws), ws.numel(),
                                                       L.stream_ptr(x.device)), "dir_conv_wgrad3x3_partials")
                pend(ws, splits)
                return
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            L.check(lib.dir_conv_wgrad3x3(L.ptr(dy), L.ptr(x), L.ptr(dw), n, h, w, cin, cout, L.ptr(ws), ws.numel(),
                                          L.stream_ptr(x.device)), "dir_conv_wgrad3x3")
            return
    nbytes = lib.dir_conv_wgrad_workspace(n, h, w, cin, cout, r, s, stride, padding, form)
    if nbytes == 0:
        raise L.DirHipError(f"dir_conv_wgrad: unsupported shape Cin={cin} Cout={cout} (form {form})")
    ws = _wgrad_batch_slot(dw, from_sink, nbytes)
    if ws is not None:
        splits = ctypes.c_int(0)
        L.check(lib.dir_conv_wgrad_partials(L.ptr(dy), L.ptr(x), ctypes.addressof(splits), n, h, w, cin, cout, r, s, stride, padding, form, L.ptr(ws),
                                            ws.numel(), L.stream_ptr(x.device)), "dir_conv_wgrad_partials")
        pend(ws, splits)
        return
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    L.check(lib.dir_conv_wgrad(L.ptr(dy), L.ptr(x), L.ptr(dw), n, h, w, cin, cout, r, s, stride, padding, form, L.ptr(ws),
                               ws.numel(), L.stream_ptr(x.device)), "dir_conv_wgrad")


class _ConvFn(torch.autograd.Function):
    """Forward = hand-written MFMA implicit GEMM (+ BatchNorm partial statistics in the epilogue); data gradient of
    stride-1 layers = the same kernel on dY with rotated/transposed weights; weight gradient = the MFMA split-K
    kernel ``dir_conv_wgrad``; 3x3 stride-2 data gradients = four stride-1 launches by output-pixel parity class
    (``dir_conv_dgrad_s2``); the stride-2 1x1 data gradients live in ``_ProjectionPairFn``. Shapes none of these take
    (odd-sized maps) use the general float32 data gradient of ``conv_f32`` — never a library kernel."""

    @staticmethod
    def forward(ctx, x, weight, w16, w16_rot, stride, padding, want_stats, alias_input, relu_input, bn_link=None, relu_bits=None):
        ctx.stride, ctx.padding = stride, padding
        ctx.wsink = gradsink.lookup(weight)
        ctx.alias_input = alias_input
        ctx.bn_link = bn_link                    # bn.BwdLink of the BatchNorm that produced x (its backward reduction is ours)
        ctx.relu_bits = relu_bits if relu_input else None      # x > 0 as one bit per element (bn._relu_bits_for), if the forward made it
        # relu_input: x is the output of a relu(bn(.) + shortcut) node that was promised its gradient with the ReLU
        # backward already applied (bn.bn_act(defer_relu_grad=True)): the data gradient is masked with x > 0 on store
        ctx.relu_input = relu_input
        # the statistics output never carries a gradient: without this autograd would zero-fill a [rows][2][Cout]
        # float tensor for it on every backward (one fill kernel per layer and step)
        ctx.set_materialize_grads(False)
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
        if dy is None:                                                           # only the alias output was used
            assert not ctx.relu_input
            return dalias, None, None, None, None, None, None, None, None, None, None
        link = ctx.bn_link
        if dalias is not None and dalias.dtype != torch.bfloat16:
            dalias = dalias.to(torch.bfloat16)
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dy = dy.contiguous(memory_format=torch.channels_last)
        need_dx = ctx.needs_input_grad[0]
        dx = None
        if need_dx and w16_rot is not None and ctx.stride == 1:
            # data gradient of a stride-1 convolution = the SAME implicit GEMM on dY with the 180-degree rotated,
            # in/out-transposed weights and padding R-1-pad
            bits = ctx.relu_bits
            dx = conv2d_igemm(dy, w16_rot, 1, w16.shape[2] - 1 - ctx.padding, addend=dalias,
                              relu_mask=x if (ctx.relu_input and bits is None) else None, bn_link=link, relu_bits=bits)
            dalias = None
            need_dx = False
        dw = conv2d_wgrad(dy, x, w16.shape[2], ctx.stride, ctx.padding, sink=ctx.wsink)   # float32, deterministic split-K
        if need_dx and w16_rot is not None and w16_rot.dim() == 1 and ctx.stride == 2 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0:
            # 3x3 / stride 2 / pad 1: four stride-1 launches, one per output-pixel parity class (dir_conv_dgrad_s2)
            n_, cin_, h_, w_ = x.shape
            dx = torch.empty((n_, cin_, h_, w_), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
            if link is not None and dalias is None:
                part = torch.empty((4 * L.lib().dir_conv_stats_rows(n_, h_ // 2, w_ // 2), 2, cin_), dtype=torch.float32, device=x.device)
                L.check(L.lib().dir_conv_dgrad_s2_bnstats(L.ptr(dy), L.ptr(w16_rot), L.ptr(dx), n_, h_ // 2, w_ // 2, dy.shape[1], cin_,
                                                          *_bn_link_args(link), L.ptr(part), part.shape[0], L.stream_ptr(x.device)),
                        "dir_conv_dgrad_s2_bnstats")
                link.partial = part
                link.note_dout(dx)
            else:
                L.check(L.lib().dir_conv_dgrad_s2(L.ptr(dy), L.ptr(w16_rot), L.ptr(dx), n_, h_ // 2, w_ // 2, dy.shape[1], cin_,
                                                  L.stream_ptr(x.device)), "dir_conv_dgrad_s2")
            need_dx = False
        if need_dx:
            # strided data gradients the bf16 kernels do not take (odd-sized maps, a stride-2 1x1 outside a projection
            # pair): the general float32 MFMA data gradient, on the bf16 weights
            from .conv_f32 import conv2d_f32_dgrad
            dx = conv2d_f32_dgrad(dy.float(), w16.float().contiguous(memory_format=torch.channels_last), x.shape[2:],
                                  ctx.stride, ctx.padding).to(torch.bfloat16)
        if dalias is not None:                                                   # strided layer: eager accumulation
            dx = dalias if dx is None else dx + dalias
            if link is not None:
                link.partial = None                                              # the sums were of an incomplete gradient
        return dx, dw, None, None, None, None, None, None, None, None, None


class _ProjectionPairFn(torch.autograd.Function):
    """The two 1x1 convolutions that read a projection block's input (``conv1`` stride 1 and ``downsample[0]`` stride
    1 or 2, resnet.py:57-68) as ONE node, so that their data gradients meet inside one kernel: the downsample conv's
    data gradient is computed in its compact form (a plain 1x1 GEMM on its dY) and added — at the even pixels when the
    stride is 2 — inside conv1's data-gradient store loop (``dir_conv_dgrad_join``), together with the ReLU backward
    of the previous block's output. No strided transposed convolution, no zero fill, no gradient-add kernel."""

    @staticmethod
    def forward(ctx, x, w1, w1_16, w1_rot, wd, wd_16, wd_rot, stride_d, want_stats, relu_input, bn_link=None, relu_bits=None):
        ctx.set_materialize_grads(False)
        ctx.wsinks = (gradsink.lookup(w1), gradsink.lookup(wd))
        ctx.stride_d, ctx.relu_input = stride_d, relu_input
        ctx.bn_link = bn_link
        ctx.relu_bits = relu_bits if relu_input else None
        if want_stats:
            y1, s1 = conv2d_igemm(x, w1_16, 1, 0, want_stats=True)
            yd, sd = conv2d_igemm(x, wd_16, stride_d, 0, want_stats=True)
            ctx.mark_non_differentiable(s1, sd)
        else:
            y1, s1 = conv2d_igemm(x, w1_16, 1, 0), None
            yd, sd = conv2d_igemm(x, wd_16, stride_d, 0), None
        ctx.save_for_backward(x, w1_rot, wd_rot)
        return y1, s1, yd, sd

    @staticmethod
    def backward(ctx, dy1, _ds1, dyd, _dsd):
        x, w1_rot, wd_rot = ctx.saved_tensors

        def prep(t):
            if t is None:
                return None
            if t.dtype != torch.bfloat16:
                t = t.to(torch.bfloat16)
            return t.contiguous(memory_format=torch.channels_last)
        dy1, dyd = prep(dy1), prep(dyd)
        bits = ctx.relu_bits
        mask = x if (ctx.relu_input and bits is None) else None
        compact = conv2d_igemm(dyd, wd_rot, 1, 0) if dyd is not None else None
        if dy1 is None:
            raise L.DirHipError("projection pair: conv1's output received no gradient")
        link = ctx.bn_link
        if compact is None:
            dx = conv2d_igemm(dy1, w1_rot, 1, 0, relu_mask=mask, bn_link=link, relu_bits=bits)
        elif ctx.stride_d == 1:
            dx = conv2d_igemm(dy1, w1_rot, 1, 0, addend=compact, relu_mask=mask, bn_link=link, relu_bits=bits)
        else:
            dx = conv2d_igemm(dy1, w1_rot, 1, 0, addend_s2=compact, relu_mask=mask, bn_link=link, relu_bits=bits)
        dw1 = conv2d_wgrad(dy1, x, 1, 1, 0, sink=ctx.wsinks[0])
        dwd = conv2d_wgrad(dyd, x, 1, ctx.stride_d, 0, sink=ctx.wsinks[1]) if dyd is not None else None
        return dx if ctx.needs_input_grad[0] else None, dw1, None, None, dwd, None, None, None, None, None, None, None


# Generation counter of "the weights may have changed": fused / multi-tensor optimizer kernels update parameters
# WITHOUT bumping Tensor._version (measured: torch 2.10 fused Adam leaves _version untouched), so the bf16 weight
# cache is keyed on (generation, _version, data_ptr) and every optimizer.step() anywhere bumps the generation.
_GENERATION = [0]


def invalidate_weight_cache(*_a, **_k):
    _GENERATION[0] += 1


from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post_hook  # noqa: E402
_reg_post_hook(invalidate_weight_cache)


class _PreparedWeights:
    """bf16 operands of one conv layer (persistent buffers) + the key of the master weight they were made from."""
    __slots__ = ("key", "w16", "w16_rot", "conv_ref", "shape", "rot_mode")

    def __init__(self, conv):
        import weakref
        w = conv.weight
        cout, cin, r, s = w.shape
        self.shape = (cout, r * s, cin)
        self.w16 = torch.empty((cout, cin, r, s), dtype=torch.bfloat16, device=w.device, memory_format=torch.channels_last)
        self.w16_rot = None
        self.rot_mode = 0
        if conv.stride[0] == 2 and (r, s) == (3, 3) and conv.padding[0] == 1 and supported(cout, cin):
            # the four parity-class weights of the stride-2 data gradient, packed (dir_conv_dgrad_s2)
            self.w16_rot = torch.empty(cin * 9 * cout, dtype=torch.bfloat16, device=w.device)
            self.rot_mode = 1
        elif (conv.stride[0] == 1 or (r == 1 and s == 1)) and supported(cout, cin):
            # [Cin][R][S][Cout], taps rotated by 180 degrees: the weight of the data-gradient convolution
            self.w16_rot = torch.empty((cin, cout, r, s), dtype=torch.bfloat16, device=w.device, memory_format=torch.channels_last)
        self.key = None
        self.conv_ref = weakref.ref(conv)


def _weight_key(w):
    return (_GENERATION[0], w._version, w.data_ptr())


_REGISTRY = {}      # device -> list of _PreparedWeights (every conv layer that went through conv_bn_input there)
_TABLES = {}        # device -> (tuple of master-weight pointers, device table [n][6] int64, list of entries)


def _refresh_all(device):
    """Re-make the bf16 operands of every registered layer on ``device`` in ONE launch (after an optimizer step all of
    them are stale at once). Layers whose module died, moved or is not a dense channels_last float32 weight are skipped
    here and fall back to the single-layer path."""
    entries = [st for st in _REGISTRY.get(device, []) if st.conv_ref() is not None]
    _REGISTRY[device] = entries
    live = []
    for st in entries:
        conv = st.conv_ref()
        w = conv.weight
        if w.device == device and w.dtype == torch.float32 and w.is_contiguous(memory_format=torch.channels_last) \
                and tuple(w.shape) == (st.shape[0], st.shape[2]) + tuple(conv.kernel_size):
            live.append((st, w))
    if not live:
        return
    ptrs = tuple(w.data_ptr() for _, w in live)
    cached = _TABLES.get(device)
    if cached is None or cached[0] != ptrs:
        rows = [[w.data_ptr(), st.w16.data_ptr(), 0 if st.w16_rot is None else st.w16_rot.data_ptr(),
                 st.shape[0], st.shape[1], st.shape[2], st.rot_mode] for st, w in live]
        cached = (ptrs, torch.tensor(rows, dtype=torch.int64).to(device))
        _TABLES[device] = cached
    L.check(L.lib().dir_conv_prep_weights_batched(L.ptr(cached[1]), len(live), L.stream_ptr(device)),
            "dir_conv_prep_weights_batched")
    for st, w in live:
        st.key = _weight_key(w)


def registry_size(device):
    """Number of conv layers with bf16 operand buffers on ``device`` (grows when a float32-mode run switches to bf16)."""
    return len(_REGISTRY.get(device, []))


def prepared_operands(device):
    """{master weight address: _PreparedWeights} of the live conv layers on ``device`` whose bf16 operands the optimizer kernel may
    rewrite itself (``dirhip.optim.Adam``): float32 channels_last weights of the registered layers."""
    out = {}
    for st in _REGISTRY.get(device, []):
        conv = st.conv_ref()
        if conv is None:
            continue
        w = conv.weight
        if w.device == device and w.dtype == torch.float32 and w.is_contiguous(memory_format=torch.channels_last) \
                and tuple(w.shape) == (st.shape[0], st.shape[2]) + tuple(conv.kernel_size):
            out[w.data_ptr()] = st
    return out


def mark_prepared_after_step(entries):
    """The optimizer kernel has just rewritten these layers' bf16 operands from the updated weights: they are valid for the generation
    the step's post hook (``invalidate_weight_cache``) is about to open."""
    for st in entries:
        conv = st.conv_ref()
        if conv is not None:
            w = conv.weight
            st.key = (_GENERATION[0] + 1, w._version, w.data_ptr())


def _prepared(conv):
    """bf16 operands of ``conv.weight``, valid for the current optimizer generation."""
    w = conv.weight
    st = getattr(conv, "_dir_w16", None)
    if st is None or st.w16.device != w.device:
        st = _PreparedWeights(conv)
        object.__setattr__(conv, "_dir_w16", st)
        _REGISTRY.setdefault(w.device, []).append(st)
    key = _weight_key(w)
    if st.key != key and st.key is not None:
        _refresh_all(w.device)                     # everything went stale together: one launch for the whole network
    if st.key != key:                              # first use, or a layer the batched path skipped
        wd = w.detach()
        if wd.dtype != torch.float32:
            wd = wd.float()
        if not wd.is_contiguous(memory_format=torch.channels_last):
            wd = wd.contiguous(memory_format=torch.channels_last)
        cout, rs, cin = st.shape
        r = conv.kernel_size[0]
        L.check(L.lib().dir_conv_prep_weights_ex(L.ptr(wd), cout, r, rs // r, cin, L.ptr(st.w16), L.ptr(st.w16_rot), st.rot_mode,
                                                 L.stream_ptr(wd.device)), "dir_conv_prep_weights")
        st.key = key
    return st.w16, st.w16_rot


def conv_bn_input(x, conv, want_stats, alias_input=False, relu_flag=None):
    """Apply ``conv`` (an ``nn.Conv2d`` with bias=False, Cin/Cout multiples of 64) to a bf16 channels_last tensor with
    the MFMA kernel. Returns ``(y, partial_stats or None)``. The bf16 copy of the fp32 master weight is cached until
    the next optimizer step / in-place edit (one cast per step instead of one per use: train forward, epoch-tail
    forward and the backward share it), and all layers of a network are re-cast by one launch. The operands live in
    persistent buffers that are rewritten in place: a backward pass must run before the optimizer step that follows
    its forward pass (as in any training loop)."""
    if x.dtype == torch.float32:
        # float32 parity mode: the same node contract on the exact-float32 kernels (round 6: the tile kernels form the BatchNorm statistics in
        # their store loop too; geometries on the gather kernel return None and the BatchNorm node counts itself)
        from .conv_f32 import conv_graph_f32
        grad_mode = torch.is_grad_enabled()
        aliasing = bool(alias_input and grad_mode and x.requires_grad)
        relu_input = bool(relu_flag is not None and aliasing and conv.stride[0] == 1 and not relu_flag[0])
        if relu_input:
            relu_flag[0] = True
        y, alias, stats = conv_graph_f32(x, conv, aliasing, relu_input, want_stats=bool(want_stats))
        return (y, stats, alias if alias is not None else x) if alias_input else (y, stats)
    w = conv.weight
    w16, w16_rot = _prepared(conv)
    if x.dtype != torch.bfloat16:
        x = x.to(torch.bfloat16)
        relu_flag = None
    grad_mode = torch.is_grad_enabled()
    aliasing = bool(alias_input and grad_mode and x.requires_grad)
    # relu_flag (a one-element list made by bn_act(defer_relu_grad=True) for the tensor x): claimed only when EVERY use
    # of x goes through this node (alias mode) and the data gradient is our own kernel, which can mask on store
    relu_input = bool(relu_flag is not None and aliasing and w16_rot is not None and conv.stride[0] == 1 and not relu_flag[0])
    if relu_input:
        relu_flag[0] = True
    # BatchNorm that produced x and asked for its backward reduction (bn.BwdLink): ours when this node's data gradient is the
    # whole gradient of x (sole consumer; behind a relu(. + residual) only together with that ReLU's backward)
    link = getattr(x, "_dir_bn_link", None)
    if link is not None and not (grad_mode and x.requires_grad and w16_rot is not None and link.x is not None
                                 and (relu_input if link.needs_relu_claim else not alias_input)):
        link = None
    y, stats, alias = _ConvFn.apply(x, w, w16, w16_rot if grad_mode else None, conv.stride[0], conv.padding[0],
                                    want_stats, aliasing, relu_input, link, getattr(x, "_dir_relu_bits", None) if relu_input else None)
    if alias_input:
        return y, stats, (alias if alias is not None else x)
    return y, stats


def projection_pair(x, conv1, conv_d, want_stats, relu_flag=None):
    """``conv1(x)`` and ``conv_d(x)`` (both 1x1, bias-free, padding 0; ``conv1`` stride 1) as one node — see
    ``_ProjectionPairFn``. Returns ``(y1, stats1, yd, statsd)``. ``relu_flag``: as in ``conv_bn_input`` (x is only
    used here, so the flag can always be claimed when gradients flow)."""
    if x.dtype == torch.float32:
        from .conv_f32 import projection_pair_f32
        relu_input = bool(relu_flag is not None and torch.is_grad_enabled() and x.requires_grad and not relu_flag[0])
        if relu_input:
            relu_flag[0] = True
        y1, yd, s1, sd = projection_pair_f32(x, conv1, conv_d, relu_input, want_stats=bool(want_stats))
        return y1, s1, yd, sd
    w1_16, w1_rot = _prepared(conv1)
    wd_16, wd_rot = _prepared(conv_d)
    assert w1_rot is not None and wd_rot is not None
    if x.dtype != torch.bfloat16:
        x = x.to(torch.bfloat16)
        relu_flag = None
    grad_mode = torch.is_grad_enabled()
    relu_input = bool(relu_flag is not None and grad_mode and x.requires_grad and not relu_flag[0])
    if relu_input:
        relu_flag[0] = True
    link = getattr(x, "_dir_bn_link", None)
    if link is not None and not (grad_mode and x.requires_grad and link.x is not None
                                 and (relu_input or not link.needs_relu_claim)):
        link = None
    return _ProjectionPairFn.apply(x, conv1.weight, w1_16, w1_rot, conv_d.weight, wd_16, wd_rot, conv_d.stride[0], want_stats,
                                   relu_input, link, getattr(x, "_dir_relu_bits", None) if relu_input else None)


def projection_pair_ok(conv1, conv_d, x=None):
    def one(c, strides):
        return (c.kernel_size == (1, 1) and c.padding == (0, 0) and c.stride[0] in strides and c.stride[0] == c.stride[1]
                and c.bias is None and c.groups == 1 and supported(c.in_channels, c.out_channels, x))
    return one(conv1, (1,)) and one(conv_d, (1, 2))


class _StemConvFn(torch.autograd.Function):
    """7x7 / stride-2 stem convolution on a bf16 channels_last image: forward = ``dir_stem_conv_fwd`` (+ BatchNorm partial
    statistics), weight gradient = ``dir_stem_conv_wgrad``; the image needs no gradient."""

    @staticmethod
    def forward(ctx, x16, weight, wpack, want_stats):
        ctx.set_materialize_grads(False)
        n, c, h, w = x16.shape
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = torch.empty((n, 64, ho, wo), dtype=torch.bfloat16, device=x16.device, memory_format=torch.channels_last)
        stats = None
        if want_stats:
            stats = torch.empty((L.lib().dir_stem_conv_stats_rows(n, h), 2, 64), dtype=torch.float32, device=x16.device)
            ctx.mark_non_differentiable(stats)
        L.check(L.lib().dir_stem_conv_fwd(L.ptr(x16), L.ptr(wpack), L.ptr(y), L.ptr(stats), n, h, w, L.stream_ptr(x16.device)),
                "dir_stem_conv_fwd")
        ctx.save_for_backward(x16, weight)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        x16, weight = ctx.saved_tensors
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dy = dy.contiguous(memory_format=torch.channels_last)
        n, _, h, w = x16.shape
        dw = gradsink.out_for(weight, (64, 3, 7, 7), x16.device, torch.channels_last)
        ws = torch.empty(L.lib().dir_stem_conv_wgrad_workspace(n, h), dtype=torch.uint8, device=x16.device)
        L.check(L.lib().dir_stem_conv_wgrad(L.ptr(dy), L.ptr(x16), L.ptr(dw), n, h, w, L.ptr(ws), ws.numel(),
                                            L.stream_ptr(x16.device)), "dir_stem_conv_wgrad")
        return None, dw.to(weight.dtype), None, None


def stem_conv_ok(x, conv):
    return (x.is_cuda and x.dim() == 4 and x.shape[1] == 3 and conv.in_channels == 3 and conv.out_channels == 64
            and conv.kernel_size == (7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3) and conv.bias is None
            and conv.dilation == (1, 1) and conv.groups == 1 and x.shape[3] % 8 == 0 and x.shape[3] <= 256
            and conv.weight.dtype == torch.float32)


def stem_conv(x, conv, want_stats):
    """``conv(x)`` for the 7x7/2 stem in bf16 (what autocast computes) with the hand-written kernel; returns ``(y, partial
    statistics or None)``. ``x``: float32 or bf16 image batch, any memory format."""
    w = conv.weight
    key = _weight_key(w)
    cache = getattr(conv, "_dir_wpack", None)
    if cache is None or cache[0] != key or cache[1].device != w.device:
        wd = w.detach()
        if not wd.is_contiguous(memory_format=torch.channels_last):
            wd = wd.contiguous(memory_format=torch.channels_last)
        wpack = cache[1] if cache is not None and cache[1].device == w.device else torch.empty((64, 176), dtype=torch.bfloat16, device=w.device)
        L.check(L.lib().dir_stem_conv_prep_weights(L.ptr(wd), L.ptr(wpack), L.stream_ptr(w.device)), "dir_stem_conv_prep_weights")
        object.__setattr__(conv, "_dir_wpack", (key, wpack))
    else:
        wpack = cache[1]
    x16 = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
    x16 = x16.contiguous(memory_format=torch.channels_last)
    return _StemConvFn.apply(x16, w, wpack, want_stats)

