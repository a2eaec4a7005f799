"""GPU check of the 256 x 256 CU-tile convolution kernel (conv_igemm_big_kernel, dir_conv_set_big) against the 128 x 128 tile kernels:
plain forward with statistics, fused data gradients (shortcut addend / compact stride-2 addend / ReLU bits / BatchNorm-backward sums),
stride-2 parity classes — outputs and partial-sum lists BIT-IDENTICAL (same K order per element, same epilogue), plus fp32 torch.
    python tools/check_big.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_ring as R  # noqa: E402
from check_ring import L  # noqa: E402


MODES = (0, 4) if (len(sys.argv) > 1 and sys.argv[1] == "tall") else (0, 2)      # tall: the 256 x 128 single-stage form (K loops <= 18 steps)


def both(fn):
    out = []
    for mode in MODES:
        prev = L.lib().dir_conv_set_big(mode)
        try:
            out.append(fn())
        finally:
            L.lib().dir_conv_set_big(prev)
    torch.cuda.synchronize()
    return out


R.both = both


def main():
    fwd = [(64, 1024, 256, 1, 1, 14), (256, 512, 2048, 1, 1, 7), (4, 64, 256, 1, 1, 56), (64, 512, 1024, 1, 2, 28), (256, 512, 512, 3, 1, 7),
           (16, 256, 512, 1, 2, 56), (64, 256, 256, 3, 2, 28), (64, 128, 256, 1, 1, 14)]
    for i, c in enumerate(fwd):
        R.fwd_case(*c, seed=100 + i)
    dg = [  # n, cin, cout, k, h, addend, s2, bits, bn, recompute
        (64, 1024, 256, 1, 14, True, False, True, True, False), (64, 256, 1024, 1, 14, True, False, True, True, True),
        (64, 512, 256, 1, 14, False, True, True, True, False), (256, 2048, 512, 1, 7, False, False, False, True, True),
        (256, 512, 512, 3, 7, False, False, False, True, True), (64, 256, 256, 1, 14, True, False, False, False, False)]
    for i, c in enumerate(dg):
        R.dgrad_case(c[0], c[1], c[2], c[3], c[4], 200 + i, *c[5:])
    for i, c in enumerate([(64, 256, 256, 14), (256, 512, 512, 7)]):
        R.s2_case(*c, seed=300 + i)
    print("FAILS:", R.FAILS)
    sys.exit(1 if R.FAILS else 0)


if __name__ == "__main__":
    main()
