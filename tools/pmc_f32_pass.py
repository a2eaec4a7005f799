"""A few float32 tile-kernel launches for a rocprofv3 --pmc pass.   python tools/pmc_f32_pass.py [variant: 2 = exact tile (default), 3 = split x3, 4 = split x2]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip.conv_f32 import TILE, conv2d_f32_dgrad, conv2d_f32_fwd, conv2d_f32_wgrad  # noqa: E402

B = 256
V = int(sys.argv[1]) if len(sys.argv) > 1 else TILE
for cin, cout, k, st, h in ((64, 64, 3, 1, 56), (256, 256, 3, 1, 14), (1024, 512, 1, 1, 14), (256, 64, 1, 1, 56)):
    pad = k // 2
    ho = (h + 2 * pad - k) // st + 1
    x = torch.randn(B, cin, h, h, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, cout, ho, ho, device="cuda").contiguous(memory_format=torch.channels_last)
    for _ in range(2):
        conv2d_f32_fwd(x, w, st, pad, variant=V)
        conv2d_f32_dgrad(dy, w, (h, h), st, pad, variant=V)
        conv2d_f32_wgrad(dy, x, (k, k), st, pad, variant=V)
    torch.cuda.synchronize()
