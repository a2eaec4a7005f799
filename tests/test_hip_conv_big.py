"""The 256 x 256 CU-tile convolution kernel (conv_igemm_big_kernel; the product heuristic takes it for launches with >= 16 K-steps and
>= 150 tiles) and the kernel-selection contract of the C-ABI (no process-wide switches: `variant` per launch, `stats_rows` checked).
-m gpu: tools/check_conv_variants.py compares every launch form the product uses (forward + statistics, fused data gradients, stride-2
parity classes) BIT FOR BIT between two kernels forced through `variant`, and the forward against fp32 torch. CPU: the statistics-row
count the host sizes buffers with comes from the same planning function the launcher uses."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["big", "tiles"])
def test_conv_kernels_bit_identical_across_variants(what):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_conv_variants.py"), what], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "FAILS: []" in p.stdout, (p.stdout + p.stderr)[-3000:]


@pytest.mark.gpu
def test_launch_refuses_a_statistics_list_sized_for_another_kernel():
    """the tiling (and so the rows of `stats`) depends on the kernel; a list sized for another tiling must be refused,
    not overrun. The launcher re-derives the plan and returns DIR_EINVAL (-1) on a mismatch, DIR_EUNSUPPORTED (-2) for a forced kernel
    that does not take the geometry."""
    import torch
    from dirhip import _lib as L
    lib = L.lib()
    n, h, c = 4, 14, 256
    x = torch.randn(n, c, h, h, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(c, c, 3, 3, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = torch.empty_like(x)
    rows_auto = lib.dir_conv_plan_rows(n, h, h, c, c, 3, 3, 1, 1, 0, L.CONV_AUTO)          # patch-staged here: 2 chunks per image
    rows_tile = lib.dir_conv_plan_rows(n, h, h, c, c, 3, 3, 1, 1, 0, L.CONV_TILE_DMA)      # 128-row tiles
    assert rows_auto == 2 * n and rows_tile == lib.dir_conv_stats_rows(n, h, h) == 7 and rows_auto != rows_tile
    st = torch.zeros(max(rows_auto, rows_tile), 2, c, device="cuda")
    s = L.stream_ptr(x.device)
    assert lib.dir_conv_fwd(L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(st), rows_tile, n, h, h, c, c, 3, 3, 1, 1, s) == -1
    assert lib.dir_conv_fwd(L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(st), rows_auto, n, h, h, c, c, 3, 3, 1, 1, s) == 0
    assert lib.dir_conv_fwd_variant(L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(st), rows_auto, n, h, h, c, c, 3, 3, 1, 1, L.CONV_TILE_DMA, s) == -1
    assert lib.dir_conv_fwd_variant(L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(st), rows_tile, n, h, h, c, c, 3, 3, 1, 1, L.CONV_TILE_DMA, s) == 0
    assert lib.dir_conv_fwd_variant(L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(st), rows_tile, n, h, h, c, c, 3, 3, 1, 1, L.CONV_BIG, s) == L.DIR_EUNSUPPORTED   # M = 784: not a multiple of 256
    assert lib.dir_conv_fwd_variant(L.ptr(x), L.ptr(w), L.ptr(y), None, 0, n, h, h, c, c, 3, 3, 1, 1, 4, s) == -1      # not a variant
    torch.cuda.synchronize()


def test_statistics_rows_follow_the_kernel_selection():
    sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
    from dirhip import _lib as L
    lib = L.lib()
    A = L.CONV_AUTO
    # 3x3 256 -> 256 on 14^2 at B=256: 36 K-steps, 196 tiles of 256 x 256 -> the CU-tile kernel, one row per 128 pixels
    assert lib.dir_conv_plan_rows(256, 14, 14, 256, 256, 3, 3, 1, 1, 0, A) == lib.dir_conv_stats_rows(256, 14, 14) == 392
    # ... unless the launch carries both fused addends (the CU-tile epilogue takes one): patch-staged, rows = image-row chunks
    assert lib.dir_conv_plan_rows(256, 14, 14, 256, 256, 3, 3, 1, 1, 1, A) == 512
    # the same layer at B=64 (49 tiles) and the 128-channel layer stay on the patch-staged kernel
    assert lib.dir_conv_plan_rows(64, 14, 14, 256, 256, 3, 3, 1, 1, 0, A) == 128
    assert lib.dir_conv_plan_rows(256, 28, 28, 128, 128, 3, 3, 1, 1, 0, A) == 256 * 7
    assert lib.dir_conv_plan_rows(256, 56, 56, 64, 64, 3, 3, 1, 1, 0, A) == 256 * 28
    # short K loops, strided layers and the 7^2 layers: 128-row tiles
    assert lib.dir_conv_plan_rows(256, 14, 14, 256, 1024, 1, 1, 1, 0, 0, A) == lib.dir_conv_stats_rows(256, 14, 14)
    assert lib.dir_conv_plan_rows(256, 7, 7, 512, 512, 3, 3, 1, 1, 0, A) == 98
    assert lib.dir_conv_plan_rows(256, 56, 56, 128, 128, 3, 3, 2, 1, 0, A) == lib.dir_conv_stats_rows(256, 28, 28) == 1568
    assert lib.dir_conv_plan_rows(3, 56, 28, 64, 64, 3, 3, 1, 1, 0, A) == lib.dir_conv_stats_rows(3, 56, 28)      # not square
    # forced kernels
    assert lib.dir_conv_plan_rows(256, 14, 14, 256, 256, 3, 3, 1, 1, 0, L.CONV_TILE_REG) == 392
    assert lib.dir_conv_plan_rows(256, 14, 14, 256, 256, 3, 3, 1, 1, 0, L.CONV_PATCH3) == 512
    assert lib.dir_conv_plan_rows(256, 7, 7, 512, 512, 3, 3, 1, 1, 0, L.CONV_PATCH3) == 0                           # not a patch geometry
    assert lib.dir_conv_plan_rows(64, 14, 14, 256, 320, 1, 1, 1, 0, 0, L.CONV_BIG) == 0                             # Cout % 256
    assert lib.dir_conv_plan_rows(0, 56, 56, 64, 64, 3, 3, 1, 1, 0, A) == 0 and lib.dir_conv_plan_rows(4, 8, 8, 64, 64, 1, 1, 1, 0, 0, 4) == 0
