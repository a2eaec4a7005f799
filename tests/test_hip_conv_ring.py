"""-m gpu: the persistent ring convolution kernel (csrc/dir_conv_ring.hip; off by default, dir_conv_set_ring) stays correct:
tools/check_ring.py compares every launch form (forward + statistics, fused data gradients, stride-2 parity classes; ragged and
whole M; K loops of 1 ... 72 steps) BIT FOR BIT with the one-tile-per-workgroup kernels and the forward against fp32 torch."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_ring_kernel_bit_identical_to_tile_kernels():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_ring.py"), "quick"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "FAILS: []" in p.stdout, (p.stdout + p.stderr)[-3000:]
