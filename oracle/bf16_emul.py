"""TEST INFRASTRUCTURE ONLY — bf16-rounding emulation of the product path on torch-CPU modules.

The bf16 product graph (``dirhip.resnet`` under autocast) stores bfloat16 at fixed points and computes in float32 in
between. ``bf16_points(model)`` installs a round-to-nearest-even-bfloat16 at exactly those points on a reference-architecture
torch module (the live reference's ``resnet50`` in ``tests/golden/gen_golden_r3.py``, or ``oracle.torch_oracle``'s port on the
GPU box), so that the module — run in float64 — computes the product path's arithmetic with exact accumulation:

  * every convolution weight (``dir_conv_prep_weights``: float32 master -> bf16 operand),
  * every convolution output (the MFMA kernels' epilogue: float32 accumulator -> bf16; the BatchNorm statistics are those of
    the ROUNDED outputs, as in the product),
  * relu(bn1(.)) / relu(bn2(.)) of every block and of the stem (``dir_bn_*``: float32 arithmetic -> bf16; rounding commutes
    with ReLU and max-pool),
  * every block output relu(bn3(.) + shortcut) (the join kernel adds bn3 and the shortcut — identity, or the unrounded
    downsample BatchNorm — in float32 and rounds once),
  * gradients at the same tensors (activation gradients are stored as bf16).
The caller rounds the input image. The pool / FDS / linear / loss tail is float32 in the product and stays unrounded.
"""
import torch
import torch.nn as nn


class RoundBF16(torch.autograd.Function):
    """value -> nearest bfloat16 (ties to even), in both directions of the graph."""

    @staticmethod
    def forward(ctx, x):
        return x.float().bfloat16().to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.float().bfloat16().to(g.dtype)


def bf16_points(model):
    """Install the rounding points on a ResNet-50 (or a single Bottleneck) of the reference architecture; rounds the
    convolution weights in place. Returns the hook handles."""
    handles = []
    hook = lambda m, i, o: RoundBF16.apply(o)       # noqa: E731
    for name, m in model.named_modules():
        leaf = name.rsplit(".", 1)[-1]
        if isinstance(m, nn.Conv2d):
            m.weight.data = m.weight.data.float().bfloat16().to(m.weight.dtype)
            handles.append(m.register_forward_hook(hook))
        elif isinstance(m, nn.BatchNorm2d) and leaf in ("bn1", "bn2"):
            handles.append(m.register_forward_hook(hook))
        elif type(m).__name__.endswith("Bottleneck"):
            handles.append(m.register_forward_hook(hook))
    return handles
