"""GPU probe: the launches of dir_fds_scatter_stats on the NYUD2 dense map's shape (N = 32 * 114 * 152 rows, C = 128, 93 buckets) — which of them
the 0.37-of-HBM-peak figure of bench.py's kernel_rooflines is made of.   python tools/probe_scatter_narrow.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
b, c, h, w = 32, 128, 114, 152
depth = torch.rand(b * h * w, device=dev, generator=g) * 9.3 + 0.7
rows = torch.rand(b * h * w, c, device=dev, generator=g)
bins = ops.bin_scaled(depth, 10.0, 7, 100)
for _ in range(3):
    ops.scatter_stats(rows, bins, 93)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(10):
        ops.scatter_stats(rows, bins, 93)
    torch.cuda.synchronize()
tot = 0.0
for e in sorted(prof.key_averages(), key=lambda e: -(getattr(e, "device_time_total", 0) or 0)):
    t = (getattr(e, "device_time_total", None) or getattr(e, "cuda_time_total", 0.0)) / 10
    if t:
        tot += t
        print(f"{e.key[:90]:90s} {e.count / 10:5.1f}/call {t:8.1f} us")
print(f"busy {tot:.1f} us per call; {rows.numel() * 4 / 1e6:.0f} MB of rows")
