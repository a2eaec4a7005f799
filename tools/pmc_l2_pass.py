"""Representative conv launches (128-row tile kernels forced through `variant`, so that the 3x3 layers go through the implicit-GEMM K loop)
for `rocprofv3 --pmc TCC_* / TCP_*` passes (tools/gpu_pmc_l2.sh): does the L2 -> LDS stream hit in the L2, and at what rate?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip import _lib as L
from dirhip.conv import conv2d_igemm
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SH = [(256, 256, 3, 1, 14), (512, 512, 3, 1, 7), (128, 128, 3, 1, 28), (256, 256, 3, 2, 28), (1024, 256, 1, 1, 14), (256, 1024, 1, 1, 14),
      (512, 2048, 1, 1, 7), (64, 256, 1, 1, 56)]
for ci, co, k, st, h in SH:
    x = torch.randn(B, ci, h, h, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, k, k, device="cuda") * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for _ in range(3):
        conv2d_igemm(x, w, st, k // 2, want_stats=True, variant=L.CONV_TILE_DMA if ci * k * k // 64 >= 32 else L.CONV_TILE_REG)
    torch.cuda.synchronize()
