"""A few conv configurations through the round-2 kernels (ring off) and the persistent ring kernel (ring on), two launches each,
for `rocprofv3 --pmc SQ_*` passes (tools/gpu_pmc_ring.sh): where do the wavefronts of a K loop spend their cycles."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip import _lib as L
from dirhip.conv import conv2d_igemm
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SH = [(512, 512, 3, 1, 7), (1024, 256, 1, 1, 14), (256, 1024, 1, 1, 14), (256, 256, 3, 1, 14)]
for ci, co, k, st, h in SH:
    x = torch.randn(B, ci, h, h, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, k, k, device="cuda") * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for mode in (0, 1):
        L.lib().dir_conv_set_ring(mode)
        for _ in range(2):
            conv2d_igemm(x, w, st, k // 2, want_stats=True)
        torch.cuda.synchronize()
    L.lib().dir_conv_set_ring(1)
