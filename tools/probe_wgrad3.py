"""A/B of the two 3x3 / stride-1 weight-gradient forms at ResNet-50's shapes, batch 256: per-tap tiles (dir_conv_wgrad) vs all
taps in one pass (dir_conv_wgrad3x3). HIP events on the launch stream, inputs rotated over > 256 MB so nothing is cache resident."""
import os
import sys

import torch
import variant_switches as VS  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip import conv as C  # noqa: E402


def ev(fn, iters, warm=2):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    g = torch.Generator(device="cuda").manual_seed(0)
    for c, hw in ((64, 56), (128, 28), (256, 14), (512, 7)):
        nbuf = max(2, int(600e6 / (2 * n * c * hw * hw * 2)) + 1)
        xs = [torch.randn(n, c, hw, hw, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(nbuf)]
        dys = [torch.randn(n, c, hw, hw, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(nbuf)]
        res = {}
        for name, flag in (("per_tap", False), ("all_taps", True)):
            prev = VS.set_wgrad3_all_taps(flag)
            try:
                res[name] = ev(lambda i: C.conv2d_wgrad(dys[i % nbuf], xs[i % nbuf], 3, 1, 1), 20)
            finally:
                VS.set_wgrad3_all_taps(prev)
        flop = 2.0 * n * hw * hw * c * c * 9
        print(f"{c:4d}->{c:4d} k3 H{hw:2d} N={n}: per-tap {res['per_tap']:7.1f} us ({flop / res['per_tap'] / 1e6:6.0f} TF)   "
              f"all-taps {res['all_taps']:7.1f} us ({flop / res['all_taps'] / 1e6:6.0f} TF)   x{res['per_tap'] / res['all_taps']:.2f}", flush=True)


if __name__ == "__main__":
    main()
