"""-m gpu: the 256 x 256 CU-tile convolution kernel (conv_igemm_big_kernel; default for launches with >= 16 K-steps and >= 150 tiles,
dir_conv_set_big): tools/check_big.py compares every launch form it takes (forward + statistics, fused data gradients, stride-2 parity
classes) BIT FOR BIT with the 128 x 128 tile kernels and the forward against fp32 torch. CPU: the statistics-row count the host side
sizes buffers with follows the same selection rule."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_big_tile_kernel_bit_identical_to_tile_kernels():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_big.py")], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "FAILS: []" in p.stdout, (p.stdout + p.stderr)[-3000:]


def test_statistics_rows_follow_the_kernel_selection():
    sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
    from dirhip import _lib as L
    lib = L.lib()
    prev = lib.dir_conv_set_big(1)
    try:
        # 3x3 256 -> 256 on 14^2 at B=256: 36 K-steps, 196 tiles of 256 x 256 -> the big kernel, one row per 128 pixels
        assert lib.dir_conv_tile_rows_ex(256, 14, 14, 256, 256, 3, 3, 1, 1) == lib.dir_conv_stats_rows(256, 14, 14) == 392
        # the same layer at B=64 (49 tiles) and the 128-channel layer stay on the patch-staged kernel: rows = image-row chunks
        assert lib.dir_conv_tile_rows_ex(64, 14, 14, 256, 256, 3, 3, 1, 1) == lib.dir_conv_tile_rows(64, 14, 14, 3, 3, 1, 1) == 128
        assert lib.dir_conv_tile_rows_ex(256, 28, 28, 128, 128, 3, 3, 1, 1) == lib.dir_conv_tile_rows(256, 28, 28, 3, 3, 1, 1)
        # short K loops and the 7^2 layers: unchanged
        assert lib.dir_conv_tile_rows_ex(256, 14, 14, 256, 1024, 1, 1, 1, 0) == lib.dir_conv_stats_rows(256, 14, 14)
        assert lib.dir_conv_tile_rows_ex(256, 7, 7, 512, 512, 3, 3, 1, 1) == lib.dir_conv_tile_rows(256, 7, 7, 3, 3, 1, 1)
        lib.dir_conv_set_big(0)
        assert lib.dir_conv_tile_rows_ex(256, 14, 14, 256, 256, 3, 3, 1, 1) == lib.dir_conv_tile_rows(256, 14, 14, 3, 3, 1, 1) == 512
    finally:
        lib.dir_conv_set_big(prev)
