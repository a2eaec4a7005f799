"""GPU probe of the FDS kernels at their full-epoch sizes (run under `rocprofv3 --kernel-trace --stats` for the per-kernel split):
dir_fds_scatter_stats at N = 191 509 x 2048 (and the NYUD2 narrow-row shape), dir_fds_smooth_fwd / dir_fds_calibrate_bwd at B = 65 536."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip import ops  # noqa: E402


def ev(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(3)
n, c, nb = 191509, 2048, 100
lab = torch.as_tensor(np.clip(np.round(np.abs(np.random.default_rng(3).normal(0, 18, n)) + 20), 0, 120).astype(np.float32), device=dev)
feats = torch.randn(n, c, device=dev, generator=g).abs_()
bins, _ = ops.bin_index(lab, 0, 100)
ms = ev(lambda: ops.scatter_stats(feats, bins, nb))
print(f"scatter_stats N={n} C={c}: {ms * 1e3:.1f} us  {(n * c * 4 + n * 4) / ms / 1e6:.0f} GB/s")
del feats
b = 65536
x = torch.randn(b, c, device=dev, generator=g)
labb = lab[:b].contiguous()
m1, sc, m2 = torch.randn(nb, c, device=dev, generator=g), torch.rand(nb, c, device=dev, generator=g) + 0.5, torch.randn(nb, c, device=dev, generator=g)
ms = ev(lambda: ops.smooth_fwd_(x, labb, 0, 100, m1, sc, m2))
print(f"smooth_fwd B={b}: {ms * 1e3:.1f} us  {2 * b * c * 4 / ms / 1e6:.0f} GB/s (x only)")
bb, _ = ops.bin_index(labb, 0, 100)
dy = torch.randn(b, c, device=dev, generator=g)
ms = ev(lambda: ops.calibrate_bwd(dy, bb, sc))
print(f"calibrate_bwd B={b}: {ms * 1e3:.1f} us  {2 * b * c * 4 / ms / 1e6:.0f} GB/s (dy, dx only)")
del x, dy
bq, cq, h, w = 32, 128, 114, 152
depth = torch.rand(bq * h * w, device=dev, generator=g) * 9.3 + 0.7
rows = torch.rand(bq * h * w, cq, device=dev, generator=g)
binsq = ops.bin_scaled(depth, 10.0, 7, 100)
ms = ev(lambda: ops.scatter_stats(rows, binsq, 93), 5)
print(f"scatter_stats narrow N={rows.shape[0]} C={cq}: {ms * 1e3:.1f} us  {rows.numel() * 4 / ms / 1e6:.0f} GB/s")
