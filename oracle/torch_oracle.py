"""TEST INFRASTRUCTURE ONLY — torch-CPU port of the reference's training step, used (a) as the
``cpu_baseline`` leg of ``bench.py`` ("kind": "port") on the GPU box, where /root/reference does not exist,
and (b) as the differentiable checker for end-to-end parity tests of the GPU stack.

It restates, with the reference's own cost structure (eager torch ops, per-unique-label host loops):
  * ``resnet.py:41-157``  -> ``RefResNet50`` (same layer names, same init stream)
  * ``fds.py:14-144``     -> ``RefFDS``      (per-label loops, ``.item()``-style host branching)
  * ``utils.py:97-107``   -> ``ref_calibrate_mean_var``
  * ``loss.py:5-48``      -> ``ref_weighted_loss``
  * ``train.py:246-262`` / ``:269-281`` -> ``train_step`` / ``epoch_tail``
Pinned in ``tests/test_torch_oracle.py`` against the live reference (build container): identical loss
trajectory and FDS buffers on identical inputs.  Never imported by the product package.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .fds_oracle import fds_kernel_window


def ref_calibrate_mean_var(matrix, m1, v1, m2, v2, clip_min=0.1, clip_max=10):     # utils.py:97-107
    if torch.sum(v1) < 1e-10:
        return matrix
    if (v1 == 0.).any():
        ok = (v1 != 0.)
        factor = torch.clamp(v2[ok] / v1[ok], clip_min, clip_max)
        matrix[:, ok] = (matrix[:, ok] - m1[ok]) * torch.sqrt(factor) + m2[ok]
        return matrix
    factor = torch.clamp(v2 / v1, clip_min, clip_max)
    return (matrix - m1) * torch.sqrt(factor) + m2


class RefFDS(nn.Module):
    """fds.py:14-144 restated (buffers, aliasing and loops as in the reference)."""

    def __init__(self, feature_dim, bucket_num=100, bucket_start=0, start_update=0, start_smooth=1,
                 kernel='gaussian', ks=5, sigma=2, momentum=0.9):
        super().__init__()
        self.feature_dim, self.bucket_num, self.bucket_start = feature_dim, bucket_num, bucket_start
        self.kernel_window = torch.tensor(fds_kernel_window(kernel, ks, sigma))
        self.half_ks = (ks - 1) // 2
        self.momentum, self.start_update, self.start_smooth = momentum, start_update, start_smooth
        nb = bucket_num - bucket_start
        self.register_buffer('epoch', torch.zeros(1).fill_(start_update))
        self.register_buffer('running_mean', torch.zeros(nb, feature_dim))
        self.register_buffer('running_var', torch.ones(nb, feature_dim))
        self.register_buffer('running_mean_last_epoch', torch.zeros(nb, feature_dim))
        self.register_buffer('running_var_last_epoch', torch.ones(nb, feature_dim))
        self.register_buffer('smoothed_mean_last_epoch', torch.zeros(nb, feature_dim))
        self.register_buffer('smoothed_var_last_epoch', torch.ones(nb, feature_dim))
        self.register_buffer('num_samples_tracked', torch.zeros(nb))

    def _conv_bins(self, table):                                                   # fds.py:58-67
        x = F.pad(table.unsqueeze(1).permute(2, 1, 0), pad=(self.half_ks, self.half_ks), mode='reflect')
        return F.conv1d(input=x, weight=self.kernel_window.view(1, 1, -1), padding=0).permute(2, 1, 0).squeeze(1)

    def update_last_epoch_stats(self, epoch):                                       # fds.py:78-82
        if epoch == self.epoch + 1:
            self.epoch += 1
            self.running_mean_last_epoch = self.running_mean
            self.running_var_last_epoch = self.running_var
            self.smoothed_mean_last_epoch = self._conv_bins(self.running_mean_last_epoch)
            self.smoothed_var_last_epoch = self._conv_bins(self.running_var_last_epoch)

    def _groups(self, labels):                                                      # fds.py:91-99 / :120-137
        for label in torch.unique(labels):
            if label > self.bucket_num - 1 or label < self.bucket_start:
                continue
            elif label == self.bucket_start:
                yield label, labels <= label
            elif label == self.bucket_num - 1:
                yield label, labels >= label
            else:
                yield label, labels == label

    def update_running_stats(self, features, labels, epoch):                        # fds.py:84-113
        if epoch < self.epoch:
            return
        for label, rows in self._groups(labels):
            cur = features[rows]
            n = cur.size(0)
            mean = torch.mean(cur, 0)
            var = torch.var(cur, 0, unbiased=True if n != 1 else False)
            b = int(label - self.bucket_start)
            self.num_samples_tracked[b] += n
            factor = self.momentum if self.momentum is not None else (1 - n / float(self.num_samples_tracked[b]))
            factor = 0 if epoch == self.start_update else factor
            self.running_mean[b] = (1 - factor) * mean + factor * self.running_mean[b]
            self.running_var[b] = (1 - factor) * var + factor * self.running_var[b]

    def smooth(self, features, labels, epoch):                                      # fds.py:115-144
        if epoch < self.start_smooth:
            return features
        labels = labels.squeeze(1)
        for label, rows in self._groups(labels):
            b = int(label - self.bucket_start)
            features[rows] = ref_calibrate_mean_var(
                features[rows], self.running_mean_last_epoch[b], self.running_var_last_epoch[b],
                self.smoothed_mean_last_epoch[b], self.smoothed_var_last_epoch[b])
        return features


def ref_weighted_loss(kind, inputs, targets, weights=None, activate='sigmoid', beta=None, gamma=1):   # loss.py:5-48
    err = inputs - targets
    if kind == 'mse':
        loss = err ** 2
    elif kind == 'l1':
        loss = F.l1_loss(inputs, targets, reduction='none')
    elif kind in ('focal_mse', 'focal_l1'):
        beta = .2 if beta is None else beta
        loss = err ** 2 if kind == 'focal_mse' else F.l1_loss(inputs, targets, reduction='none')
        loss = loss * ((torch.tanh(beta * torch.abs(err))) ** gamma if activate == 'tanh' else
                       (2 * torch.sigmoid(beta * torch.abs(err)) - 1) ** gamma)
    elif kind == 'huber':
        beta = 1. if beta is None else beta
        a = torch.abs(err)
        loss = torch.where(a < beta, 0.5 * a ** 2 / beta, a - 0.5 * beta)
    else:
        raise ValueError(kind)
    if weights is not None:
        loss = loss * weights.expand_as(loss)
    return torch.mean(loss)


class _Bottleneck(nn.Module):                                                       # resnet.py:41-70
    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        r = x if self.downsample is None else self.downsample(x)
        o = self.relu(self.bn1(self.conv1(x)))
        o = self.relu(self.bn2(self.conv2(o)))
        o = self.bn3(self.conv3(o))
        o += r
        return self.relu(o)


class RefResNet50(nn.Module):                                                       # resnet.py:73-157
    def __init__(self, fds=False, **fds_kw):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for i, (planes, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), 1):
            ds = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
            layer = [_Bottleneck(cin, planes, stride, ds)]
            cin = planes * 4
            layer += [_Bottleneck(cin, planes) for _ in range(1, blocks)]
            setattr(self, f'layer{i}', nn.Sequential(*layer))
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.linear = nn.Linear(2048, 1)
        self.fds = fds
        if fds:
            self.FDS = RefFDS(feature_dim=2048, **fds_kw)
            self.start_smooth = self.FDS.start_smooth
        for m in self.modules():                                                    # resnet.py:103-109
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, math.sqrt(2. / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def forward(self, x, targets=None, epoch=None):                                 # resnet.py:127-153
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        enc = self.avgpool(x).view(x.size(0), -1)
        enc_s = enc
        if self.training and self.fds and epoch >= self.start_smooth:
            enc_s = self.FDS.smooth(enc_s, targets, epoch)
        out = self.linear(enc_s)
        return (out, enc) if (self.training and self.fds) else out


def train_step(model, optimizer, inputs, targets, weights, epoch, loss_kind='l1'):
    """train.py:246-262 on CPU tensors. Returns the loss value."""
    model.train()
    outputs = model(inputs, targets, epoch)
    if model.fds:
        outputs = outputs[0]
    loss = ref_weighted_loss(loss_kind, outputs, targets, weights)
    assert not (np.isnan(loss.item()) or loss.item() > 1e6)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.item()


def epoch_tail(model, batches, epoch):
    """train.py:269-281: second pass in train mode, host round trip of the features, FDS update."""
    encodings, labels = [], []
    with torch.no_grad():
        for inputs, targets in batches:
            _, feature = model(inputs, targets, epoch)
            encodings.extend(feature.data.squeeze().cpu().numpy())
            labels.extend(targets.data.squeeze().cpu().numpy())
    encodings, labels = torch.from_numpy(np.vstack(encodings)), torch.from_numpy(np.hstack(labels))
    model.FDS.update_last_epoch_stats(epoch)
    model.FDS.update_running_stats(encodings, labels, epoch)
