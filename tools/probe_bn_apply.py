"""GPU probe: the BatchNorm apply passes alone (no statistics pass) at ResNet-50's B=256 shapes through the C-ABI, buffers rotated
over > 256 MB: dir_bn_apply (forward: read x [+ residual], write y) and dir_bn_bwd_partials (finalize of a short partial list + the
backward apply: read dout, x, write dx). Microseconds and GB/s of the tensor passes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip import _lib as L  # noqa: E402


def ev(fn, it=12):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    best = 1e9
    for r in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(it):
            fn(i)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / it)
    return best


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda")
    lib = L.lib()
    st = L.stream_ptr(dev)
    for c, hw in [(64, 56), (256, 56), (128, 28), (512, 28), (256, 14), (1024, 14), (512, 7), (2048, 7)]:
        m = B * hw * hw
        nbytes = m * c * 2
        nbuf = max(2, min(8, int(700e6 // (3 * nbytes)) + 1))
        xs = [torch.randn(B, c, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(nbuf)]
        rs = [torch.randn(B, c, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(nbuf)]
        ys = [torch.empty_like(x) for x in xs]
        coef = torch.randn(2, c, device=dev)
        bits = torch.empty(m, c // 8, dtype=torch.uint8, device=dev)
        t_plain = ev(lambda i: L.check(lib.dir_bn_apply(L.ptr(xs[i % nbuf]), None, None, L.ptr(ys[i % nbuf]), L.DIR_BF16, m, c, L.ptr(coef), 1, st), "apply"))
        t_bits = ev(lambda i: L.check(lib.dir_bn_apply_bits(L.ptr(xs[i % nbuf]), None, None, L.ptr(ys[i % nbuf]), m, c, L.ptr(coef), L.ptr(bits), st), "apply_bits"))
        t_res = ev(lambda i: L.check(lib.dir_bn_apply_bits(L.ptr(xs[i % nbuf]), L.ptr(rs[i % nbuf]), None, L.ptr(ys[i % nbuf]), m, c, L.ptr(coef), L.ptr(bits), st), "apply_res"))
        gamma, beta, mean, rstd = (torch.rand(c, device=dev) + 0.5 for _ in range(4))
        dgamma, dbeta = torch.empty(c, device=dev), torch.empty(c, device=dev)
        part = torch.randn(8, 2, c, device=dev)
        nws = lib.dir_bn_workspace(L.DIR_BF16, m, c)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        t_bwd = ev(lambda i: L.check(lib.dir_bn_bwd_partials(L.ptr(rs[i % nbuf]), L.ptr(xs[i % nbuf]), L.ptr(ys[i % nbuf]), L.DIR_BF16, m, c, L.ptr(gamma), L.ptr(beta),
                                                               L.ptr(mean), L.ptr(rstd), L.ptr(dgamma), L.ptr(dbeta), 0, L.ptr(part), 8, L.ptr(ws), nws, st), "bwd"))
        print(f"C={c:5d} HW={hw:3d} tensor {nbytes / 1e6:6.1f} MB | apply {t_plain * 1e3:6.1f} us {2 * nbytes / t_plain / 1e6:5.0f} GB/s | +bits {t_bits * 1e3:6.1f} us "
              f"{2.0625 * nbytes / t_bits / 1e6:5.0f} | +residual {t_res * 1e3:6.1f} us {3.0625 * nbytes / t_res / 1e6:5.0f} | finalize + bwd apply {t_bwd * 1e3:6.1f} us "
              f"{3 * nbytes / t_bwd / 1e6:5.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
