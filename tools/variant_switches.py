"""Test / A-B switches of the product's kernel-path constants (VERDICT r5 hygiene: the product package exposes no process-wide setters; the alternatives
exist so that tests can hold two forms of the same computation against each other bit for bit, and tools/ab_train_step.py can time them). Every function
returns the previous setting.   from variant_switches import set_graph_fusion, ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))


def _flip(cell, key, value):
    prev = cell[key]
    cell[key] = bool(value)
    return prev


def set_wgrad3_all_taps(enabled):
    """3x3 / stride-1 weight gradients: all nine taps in one pass (product) or the per-tap kernel."""
    from dirhip import conv
    return _flip(conv._WGRAD3_ALL_TAPS, 0, enabled)


def set_wgrad_side_stream(enabled):
    """Weight gradients on a side stream (off in the product). Turning it off joins whatever is still in flight."""
    from dirhip import conv
    prev = _flip(conv._WGRAD_SIDE, "on", enabled)
    if prev and not enabled:
        conv.wgrad_join()
    return prev


def set_wgrad_batched_reduce(enabled):
    """One split-K reduction launch per backward pass (off in the product: measured neutral). Turning it off reduces whatever is still pending."""
    from dirhip import conv
    prev = _flip(conv._WGRAD_BATCH, "on", enabled)
    if prev and not enabled:
        conv.wgrad_flush()
    return prev


def set_join_bwd(enabled):
    """The join's two BatchNorm backwards as one reduction + one apply pass (product) or two dir_bn_bwd calls."""
    from dirhip import bn
    return _flip(bn._JOIN_BWD, 0, enabled)


def set_relu_bits(enabled):
    """Deferred ReLU backward from the one-bit-per-element mask (product) or from the bf16 tensor."""
    from dirhip import bn
    return _flip(bn._RELU_BITS, 0, enabled)


def set_stem_tail_xmax(enabled):
    """Keep the forward's x-at-argmax for the stem tail's backward (product) or recompute it."""
    from dirhip import pool
    return _flip(pool._STEM_TAIL_XMAX, 0, enabled)


def set_graph_fusion(enabled):
    """The fused wiring of a bottleneck block (product) or the plain composition of the same kernels (the tests' oracle for the wiring)."""
    from dirhip import resnet
    return _flip(resnet._FUSED_GRAPH, 0, enabled)


def set_bn_bwd_fusion(enabled):
    """BatchNorm-backward sums inside the producing data-gradient kernel (product) or the plain three-pass dir_bn_bwd."""
    from dirhip import resnet
    return _flip(resnet._FUSE_BN_BWD, 0, enabled)
