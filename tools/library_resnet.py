"""Measurement tools only (never imported by the product): the reference's ResNet-50 (imdb-wiki-dir/resnet.py:41-157) as plain torch
modules on the vendor LIBRARY kernels (MIOpen convolutions / BatchNorm), with the product's FDS module, for the attribution arm of
tools/valmae_proxy.py: the same network, the same initial weights, bf16 through ``torch.autocast`` instead of through this repo's graph."""
import math

import torch
import torch.nn as nn


class Bottleneck(nn.Module):
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
        return self.relu(o + r)


class LibraryResNet50(nn.Module):
    """Same sub-module names as dirhip.resnet.ResNet (state_dicts interchange); ``fds``: a dirhip.fds.FDS instance or None."""

    def __init__(self, fds=None):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for i, (planes, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), 1):
            ds = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
            layer = [Bottleneck(cin, planes, stride, ds)]
            cin = planes * 4
            layer += [Bottleneck(cin, planes) for _ in range(1, blocks)]
            setattr(self, f"layer{i}", nn.Sequential(*layer))
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.linear = nn.Linear(2048, 1)
        self.fds = fds is not None
        if fds is not None:
            self.FDS = fds
            self.start_smooth = fds.start_smooth
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, math.sqrt(2. / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def forward(self, x, targets=None, epoch=None):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        with torch.autocast(device_type="cuda", enabled=False):
            enc = self.avgpool(x.float()).reshape(x.size(0), -1).contiguous()
            enc_s = enc
            if self.training and self.fds and epoch >= self.start_smooth:
                enc_s = self.FDS.smooth(enc_s, targets, epoch)          # in place, returns the same tensor (A.2)
            out = self.linear(enc_s)
        return (out, enc) if (self.training and self.fds) else out


class AutocastModel(nn.Module):
    """The wrapper interface the product's train_step / epoch_tail / validate use (``.module``, call under autocast)."""

    def __init__(self, module, amp_dtype=torch.bfloat16):
        super().__init__()
        self.module = module
        self.amp_dtype = amp_dtype

    def forward(self, x, *a, **k):
        x = x.contiguous(memory_format=torch.channels_last)
        if self.amp_dtype is None:
            return self.module(x, *a, **k)
        with torch.autocast(device_type="cuda", dtype=self.amp_dtype):
            return self.module(x, *a, **k)


if __name__ == "__main__":
    import sys
    import time
    dev = torch.device("cuda", 0)
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    which = sys.argv[2] if len(sys.argv) > 2 else "both"              # bf16 | f32 | both
    for amp in [a for a, tag in ((torch.bfloat16, "bf16"), (None, "f32")) if which in (tag, "both")]:
        m = AutocastModel(LibraryResNet50().to(dev).to(memory_format=torch.channels_last), amp)
        m.train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        x = torch.randn(b, 3, 224, 224, device=dev)
        y = torch.randn(b, 1, device=dev)
        for it in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            loss = (m(x) - y).abs().mean()
            opt.zero_grad(); loss.backward(); opt.step()
            torch.cuda.synchronize()
            print(f"amp={amp} B={b} step {it}: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
