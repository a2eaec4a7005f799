"""Do an HBM-bound BatchNorm backward and an MFMA/LDS-bound weight gradient overlap when issued on two streams?
Wall time of N x (A; B) on one stream vs N x A on stream 1 || N x B on stream 2."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip import _lib as L  # noqa: E402
from dirhip import conv as C  # noqa: E402


def main():
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    cl = torch.channels_last
    n = 256
    cases = {"wgrad3 256ch 14x14": (256, 14, 3), "wgrad3 64ch 56x56": (64, 56, 3), "wgrad 1x1 1024->256 14x14": ((1024, 256), 14, 1)}
    # A: BatchNorm backward (three-pass) on [256, 1024, 14, 14] bf16 (103 MB tensors)
    c, hw = 1024, 14
    m = n * hw * hw
    x = torch.randn(n, c, hw, hw, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=cl)
    dout = torch.randn(n, c, hw, hw, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=cl)
    dx = torch.empty_like(x)
    gamma, beta, mean, rstd = (torch.rand(c, device=dev) + 0.5 for _ in range(4))
    dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
    ws = torch.empty(L.lib().dir_bn_workspace(L.DIR_BF16, m, c), dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def A(stream):
        L.check(L.lib().dir_bn_bwd(L.ptr(dout), L.ptr(x), None, L.ptr(dx), None, L.DIR_BF16, m, c, L.ptr(gamma), L.ptr(beta), L.ptr(mean),
                                   L.ptr(rstd), L.ptr(dg), L.ptr(db), 0, L.ptr(ws), ws.numel(), stream.cuda_stream), "bn")

    for name, (ch, h, k) in cases.items():
        cin, cout = (ch, ch) if isinstance(ch, int) else ch
        xx = torch.randn(n, cin, h, h, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=cl)
        dy = torch.randn(n, cout, h, h, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=cl)

        def B(stream):
            with torch.cuda.stream(stream):
                C.conv2d_wgrad(dy, xx, k, 1, k // 2)

        def wall(fn, reps=30):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e6
        ta = wall(lambda: A(s1))
        tb = wall(lambda: B(s1))
        ts = wall(lambda: (A(s1), B(s1)))
        tc = wall(lambda: (A(s1), B(s2)))
        print(f"{name}: A {ta:6.1f} us  B {tb:6.1f} us  serial {ts:6.1f} us  two streams {tc:6.1f} us  (sum {ta + tb:6.1f})", flush=True)


if __name__ == "__main__":
    main()
