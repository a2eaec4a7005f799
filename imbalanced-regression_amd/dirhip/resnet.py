"""ResNet-50 age regressor — drop-in for ``imdb-wiki-dir/resnet.py`` (= ``agedb-dir/resnet.py``).

Same constructor keywords (``resnet50(fds=, bucket_num=, bucket_start=, start_update=, start_smooth=,
kernel=, ks=, sigma=, momentum=[, dropout=])``), same sub-module names (-> state_dict keys, so
reference checkpoints load), same initialisation stream, same forward contract
(``(pred, encoding)`` when training with FDS, else ``pred``; the returned ``encoding`` is the tensor
``FDS.smooth`` calibrated in place — SURVEY A.2).

Backbone, bf16 product path (channels_last, under ``dirhip.parallel.DataParallelEngine``'s autocast): the 7x7 stem
(``dir_stem_conv_*``), every other convolution as the hand-written MFMA implicit GEMM (``dirhip.conv`` -> ``dir_conv_*``:
forward, stride-1 / stride-2 data gradients, weight gradient; BatchNorm statistics in the epilogue), every BatchNorm
(+ residual add) (+ ReLU) as ONE fused HIP node (``dirhip.bn`` -> ``dir_bn_*``), fused stem tail and pools
(``dirhip.pool``). Parity mode (no autocast, float32 activations): the SAME graph with the exact-float32 MFMA convolutions
and pools of ``dirhip.conv_f32`` (``dir_conv_f32_*``) in place of the bf16 ones. No library convolution, BatchNorm or
pooling call exists on either path; shapes the bf16 kernels do not take fall back to the float32 kernels, never to a library.
The pool -> FDS calibrate -> linear -> weighted-loss tail is hand-written HIP and always fp32. GPU only (no CPU fallback).
"""
import logging
import math

import torch
import torch.nn as nn

from .bn import BatchCounters, bn_act, bn_join
from .conv import conv_bn_input, projection_pair, projection_pair_ok, rows_supported, stem_conv, stem_conv_ok, supported as _igemm_ok
from .conv_f32 import conv_f32, conv_ok
from .fds import FDS
from .pool import bn_relu_maxpool, global_avgpool_flat, maxpool3x3s2
from .tail import fusable as tail_fusable, tail_forward

print = logging.info

# The fused wiring of a bottleneck block (shared block-input gradient accumulation inside conv1's data-gradient kernel,
# ReLU backward of relu(bn3 + shortcut) deferred into the consumer, projection pair, two-BatchNorm join) is ONE graph for
# both activation dtypes. The graph-fusion constant below set to False (tools/variant_switches.py: tests only) builds the plain composition of the very same kernels instead
# (conv -> BatchNorm(+residual)(+ReLU) nodes, autograd's own gradient accumulation): the tests use it as the oracle for
# the fusion wiring — both graphs must produce the same forward bit for bit and the same gradients.
_FUSED_GRAPH = [True]


# BatchNorm backward: form the per-channel sums inside the data-gradient kernel that produces the BatchNorm's `dout`
# (bn.BwdLink) instead of in a reduction pass of their own. Off = the plain three-pass dir_bn_bwd (tests compare the two).
_FUSE_BN_BWD = [True]


def _fusable(x):
    return _FUSED_GRAPH[0] and x.dtype in (torch.bfloat16, torch.float32)


def _own_conv(x, conv):
    """Does this layer run as a graph node of ``dirhip.conv`` (bf16 MFMA kernels, or their float32 twins in parity mode)?"""
    if x.dtype == torch.float32:
        return conv_ok(conv)
    return x.dtype == torch.bfloat16 and _igemm_ok(conv.in_channels, conv.out_channels, x)


def _conv_bn(x, conv, bn, relu, residual=None, defer_relu_grad=False, fuse_bwd_stats=False):
    """conv -> BatchNorm (+ residual) (+ ReLU). bf16 activations: hand-written MFMA implicit-GEMM convolution whose
    epilogue already produced the BatchNorm statistics, then ONE fused normalise/add/ReLU pass. float32 activations
    (parity mode): the exact-float32 MFMA convolution + the same fused BatchNorm node. bf16 layers whose channel counts
    the MFMA kernel does not take are computed by the float32 kernels (casts around them)."""
    if _own_conv(x, conv):
        y, partial = conv_bn_input(x, conv, want_stats=bn.training)
        return bn_act(y, bn, relu=relu, residual=residual, partial=partial, defer_relu_grad=defer_relu_grad,
                      fuse_bwd_stats=fuse_bwd_stats)
    return bn_act(conv_f32(x, conv), bn, relu=relu, residual=residual)


class Bottleneck(nn.Module):
    """1x1 reduce -> 3x3 (carries the stride) -> 1x1 expand, BN after each, residual add (resnet.py:41-70)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        width_out = planes * self.expansion
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, width_out, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(width_out)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        fused = _fusable(x)
        if fused and self.downsample is not None and self.bn3.training and self.downsample[1].training \
                and projection_pair_ok(self.conv1, self.downsample[0], x) \
                and _igemm_ok(self.conv3.in_channels, self.conv3.out_channels) and (x.shape[2] | x.shape[3]) % 2 == 0 \
                and rows_supported(self.conv3.in_channels, self.conv3.out_channels,
                                   x.shape[0] * (x.shape[2] // self.stride) * (x.shape[3] // self.stride)):
            # (conv3 is called through conv_bn_input without the fallback of _conv_bn: its 4x wider output — and the gradient its data
            # gradient reads — must fit the kernel's 32-bit offsets too, else the block takes the general path below)
            # projection block: conv1 and the downsample conv are one node (their data gradients and the previous
            # block's ReLU backward meet inside one kernel, the stride-2 gradient in compact form), and
            # relu(bn3(conv3(.)) + bn_d(conv_d(x))) is one join with both normalisations in ONE apply pass
            y, partial, r, partial_r = projection_pair(x, self.conv1, self.downsample[0], want_stats=True,
                                                       relu_flag=getattr(x, "_dir_relu_flag", None))
            # (bn1 / bn2 outputs have one consumer, the next convolution: its data-gradient kernel also forms their backward sums)
            y = bn_act(y, self.bn1, relu=True, partial=partial, fuse_bwd_stats=_FUSE_BN_BWD[0])
            y = _conv_bn(y, self.conv2, self.bn2, relu=True, fuse_bwd_stats=_FUSE_BN_BWD[0])
            y3, partial3 = conv_bn_input(y, self.conv3, want_stats=True)
            return bn_join(y3, self.bn3, partial3, r, self.downsample[1], partial_r, relu=True, defer_relu_grad=True)
        if fused and _own_conv(x, self.conv1) and _igemm_ok(self.conv1.in_channels, self.conv1.out_channels, x):
            # conv1's node also hands back the block input, and the shortcut branch (identity or projection) reads THAT:
            # the two gradients that meet at the block input are then summed inside conv1's data-gradient kernel
            # (its `addend`) instead of by an eager add kernel
            # (and, when x is the previous block's relu(bn3 + shortcut), that node's ReLU backward is applied there too)
            y, partial, xin = conv_bn_input(x, self.conv1, want_stats=self.bn1.training, alias_input=True,
                                            relu_flag=getattr(x, "_dir_relu_flag", None))
            y = bn_act(y, self.bn1, relu=True, partial=partial, fuse_bwd_stats=_FUSE_BN_BWD[0])
            shortcut = xin if self.downsample is None else _conv_bn(xin, self.downsample[0], self.downsample[1], relu=False)
        else:
            shortcut = x if self.downsample is None else _conv_bn(x, self.downsample[0], self.downsample[1], relu=False)
            y = _conv_bn(x, self.conv1, self.bn1, relu=True, fuse_bwd_stats=fused and _FUSE_BN_BWD[0])
        y = _conv_bn(y, self.conv2, self.bn2, relu=True, fuse_bwd_stats=fused and _FUSE_BN_BWD[0])
        # relu(bn3(conv3(.)) + shortcut); the next block's conv1 may take over the ReLU backward of this node — and with it
        # the reduction pass of bn3's backward (the block output is read by that conv1 node alone: identity shortcuts go through
        # its alias output)
        return _conv_bn(y, self.conv3, self.bn3, relu=True, residual=shortcut, defer_relu_grad=fused,
                        fuse_bwd_stats=fused and _FUSE_BN_BWD[0])


class ResNet(nn.Module):

    def __init__(self, block, layers, fds, bucket_num, bucket_start, start_update, start_smooth,
                 kernel, ks, sigma, momentum, dropout=None):
        self.inplanes = 64
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.linear = nn.Linear(512 * block.expansion, 1)

        if fds:
            self.FDS = FDS(feature_dim=512 * block.expansion, bucket_num=bucket_num, bucket_start=bucket_start,
                           start_update=start_update, start_smooth=start_smooth, kernel=kernel, ks=ks,
                           sigma=sigma, momentum=momentum)
        self.fds = fds
        self.start_smooth = start_smooth

        self.use_dropout = True if dropout else False
        if self.use_dropout:
            print(f'Using dropout: {dropout}')
            self.dropout = nn.Dropout(p=dropout)

        # resnet.py:103-109: He-normal on fan-out for every conv, BN gamma 1 / beta 0
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / fan_out))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion),
            )
        stages = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        stages += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*stages)

    def feature_map(self, x):
        """The conv stack up to the last stage's output [B, 2048, 7, 7] (resnet.py:128-135)."""
        partial = None
        amp_bf16 = torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16
        if amp_bf16 and stem_conv_ok(x, self.conv1):
            x, partial = stem_conv(x, self.conv1, want_stats=self.bn1.training)   # hand-written MFMA stem, statistics in its epilogue
        else:
            x = conv_f32(x, self.conv1)                   # parity mode (float32) / image widths the bf16 stem does not take
            if amp_bf16:
                x = x.to(torch.bfloat16)
        x = bn_relu_maxpool(x, self.bn1, self.maxpool, partial=partial)
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))

    def features(self, x):
        """conv stack + global 7x7 average pool -> [B, 2048] (resnet.py:128-138)."""
        return global_avgpool_flat(self.feature_map(x), self.avgpool)

    def _batch_counters(self):
        bc = getattr(self, "_dir_counters", None)
        if bc is None or bc.flat is None or bc.flat.device != self.bn1.weight.device or \
                self.bn1.num_batches_tracked.data_ptr() != bc.flat.data_ptr():
            bc = BatchCounters().link(self)
            object.__setattr__(self, "_dir_counters", bc)
        return bc

    def forward(self, x, targets=None, epoch=None):
        counters = self._batch_counters() if self.training else None
        fmap = self.feature_map(x)
        if counters is not None:
            counters.flush()
        smooth = bool(self.training and self.fds and epoch >= self.start_smooth)
        # the pool / FDS / linear / loss tail is fp32 whatever precision the conv stack ran in
        with torch.autocast(device_type=fmap.device.type, enabled=False):
            if not self.use_dropout and tail_fusable(fmap, self.avgpool, self.linear):
                # pool -> FDS.smooth -> linear as ONE kernel; `encoding` is the calibrated tensor (in the reference smooth()
                # is in place and the same object is returned, A.2)
                x, encoding = tail_forward(fmap, self.linear, self.FDS if smooth else None, targets)
            else:
                encoding = global_avgpool_flat(fmap, self.avgpool)
                if encoding.dtype != torch.float32:
                    encoding = encoding.float()
                encoding_s = encoding
                if smooth:
                    encoding_s = self.FDS.smooth(encoding_s, targets, epoch)     # in place (A.2)
                if self.use_dropout:
                    encoding_s = self.dropout(encoding_s)
                x = self.linear(encoding_s)

        if self.training and self.fds:
            return x, encoding
        else:
            return x


def resnet50(**kwargs):
    return ResNet(Bottleneck, [3, 4, 6, 3], **kwargs)
