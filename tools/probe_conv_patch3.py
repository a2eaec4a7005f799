"""A/B of the 3x3 / stride-1 forward kernels at ResNet-50's shapes (and the data-gradient configuration = same shapes), batch
256: per-tap implicit GEMM (variants 1 = register-staged, 2 = LDS-DMA) vs the patch-staged kernel (3). Rotating buffers."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip import _lib as L  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda")
    for c, hw in ((64, 56), (128, 28), (256, 14)):
        nbuf = max(2, int(600e6 / (2 * n * c * hw * hw * 2)) + 1)
        xs = [torch.randn(n, c, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(nbuf)]
        ys = [torch.empty(n, c, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(nbuf)]
        w = (torch.randn(c, c, 3, 3, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        rows = {v: L.lib().dir_conv_plan_rows(n, hw, hw, c, c, 3, 3, 1, 1, 0, v) for v in (1, 2, 3)}
        st = torch.empty(max(rows.values()), 2, c, dtype=torch.float32, device=dev)
        res = {}
        for v in (1, 2, 3):
            def run(i):
                L.check(L.lib().dir_conv_fwd_variant(L.ptr(xs[i % nbuf]), L.ptr(w), L.ptr(ys[i % nbuf]), L.ptr(st), rows[v], n, hw, hw, c, c, 3, 3, 1, 1, v,
                                                     L.stream_ptr(dev)), "variant")
            for i in range(3):
                run(i)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(20):
                run(i)
            b.record()
            torch.cuda.synchronize()
            res[v] = a.elapsed_time(b) / 20 * 1e3
        flop = 2.0 * n * hw * hw * c * c * 9
        print(f"{c:4d}->{c:4d} k3 H{hw:2d} N={n}: " + "  ".join(f"v{v} {res[v]:6.1f} us ({flop / res[v] / 1e6:5.0f} TF)" for v in (1, 2, 3)), flush=True)


if __name__ == "__main__":
    main()
