"""GPU check of one convolution kernel against another through the explicit `variant` argument of the C-ABI (DIR_CONV_*; there are
no process-wide kernel switches): every launch form the product uses — plain forward with statistics; data gradients with shortcut
addend / compact stride-2 addend / ReLU bit mask / fused BatchNorm-backward sums with and without mask recompute; the parity classes
of the stride-2 data gradient — outputs and partial-sum lists must be BIT-IDENTICAL (same K order per element, same epilogue, same
summation order), plus an fp32 reference check of the forward.
    python tools/check_conv_variants.py big     # the 256 x 256 CU-tile kernel (5) against the 128-row LDS-DMA tile kernel (2)
    python tools/check_conv_variants.py tiles   # LDS-DMA (2) against register-staged (1) 128-row tiles, ragged M included"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip import _lib as L  # noqa: E402
from dirhip import bn as B  # noqa: E402
from dirhip import conv as C  # noqa: E402

dev = torch.device("cuda")
FAILS = []
PAIR = (L.CONV_TILE_DMA, L.CONV_BIG)


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def both(fn):
    out = [fn(v) for v in PAIR]
    torch.cuda.synchronize()
    return out


def same(name, a, b):
    if isinstance(a, (tuple, list)):
        for i, (x, y) in enumerate(zip(a, b)):
            same(f"{name}[{i}]", x, y)
        return
    if a is None:
        return
    if a.dtype == torch.bfloat16:
        eq = torch.equal(a.view(torch.int16), b.view(torch.int16))
    else:
        eq = torch.equal(a, b)
    if not eq:
        d = (a.float() - b.float()).abs()
        FAILS.append(name)
        print(f"  MISMATCH {name}: max abs diff {d.max().item():.4g}, {int((d > 0).sum())} of {d.numel()} elements, nan {int(torch.isnan(a.float()).sum())}/{int(torch.isnan(b.float()).sum())}")


def fwd_case(n, cin, cout, k, stride, h, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = cl(torch.randn(n, cin, h, h, device=dev, generator=g).to(torch.bfloat16))
    w = cl((torch.randn(cout, cin, k, k, device=dev, generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(torch.bfloat16))
    pad = k // 2
    (y0, s0), (y1, s1) = both(lambda v: C.conv2d_igemm(x, w, stride=stride, padding=pad, want_stats=True, variant=v))
    name = f"fwd n{n} {cin}->{cout} k{k} s{stride} H{h}"
    same(name + " y", y0, y1)
    same(name + " stats", s0, s1)
    ref = F.conv2d(x.float(), w.float(), stride=stride, padding=pad)
    err = ((y1.float() - ref).norm() / ref.norm()).item()
    ssum = y1.float().sum((0, 2, 3))
    serr = ((s1[:, 0].sum(0) - ssum).abs().max() / ssum.abs().max()).item()
    ok = err < 4e-3 and serr < 1e-3
    if not ok:
        FAILS.append(name + " vs fp32")
    print(f"{name}: variant {PAIR[1]} vs fp32 rel {err:.2e}, stats {serr:.1e} {'ok' if ok else 'FAIL'}", flush=True)


def make_link(yshape, cout, seed, recompute):
    gen = lambda k: torch.Generator(device="cuda").manual_seed(seed + k)      # noqa: E731
    link = B.BwdLink()
    link.x = cl(torch.randn(yshape, device=dev, generator=gen(1)).to(torch.bfloat16))
    link.gamma = torch.rand(cout, device=dev, generator=gen(2)) + 0.5
    link.beta = torch.randn(cout, device=dev, generator=gen(3)) * 0.3
    link.mean = torch.randn(cout, device=dev, generator=gen(4)) * 0.2
    link.rstd = torch.rand(cout, device=dev, generator=gen(5)) + 0.7
    link.recompute_mask = recompute
    return link


def dgrad_case(n, cin, cout, k, h, seed, addend, s2, bits, bn, recompute):
    """x = dY [n, cin, h, h], w = rotated weights [cout, cin, k, k] (whatever: both kernels see the same), stride 1."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = cl(torch.randn(n, cin, h, h, device=dev, generator=g).to(torch.bfloat16))
    w = cl((torch.randn(cout, cin, k, k, device=dev, generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(torch.bfloat16))
    pad = k // 2
    yshape = (n, cout, h, h)
    add = cl(torch.randn(yshape, device=dev, generator=g).to(torch.bfloat16)) if addend else None
    add2 = cl(torch.randn(n, cout, h // 2, h // 2, device=dev, generator=g).to(torch.bfloat16)) if s2 else None
    rb = torch.randint(0, 256, (n * h * h * cout // 8,), device=dev, generator=g, dtype=torch.uint8) if bits else None

    def run(v):
        link = make_link(yshape, cout, seed, recompute) if bn else None
        y = C.conv2d_igemm(x, w, stride=1, padding=pad, addend=add, addend_s2=add2, bn_link=link, relu_bits=rb, variant=v)
        return y, (link.partial if link is not None else None)
    r0, r1 = both(run)
    name = f"dgrad n{n} {cin}->{cout} k{k} H{h} add={int(addend)} s2={int(s2)} bits={int(bits)} bn={int(bn)}/{int(recompute)}"
    nf = len(FAILS)
    same(name, r0, r1)
    print(f"{name}: {'bit-identical' if len(FAILS) == nf else 'FAIL'}", flush=True)


def s2_case(n, cy, cx, ho, seed, bn=False):
    g = torch.Generator(device="cuda").manual_seed(seed)
    dy = cl(torch.randn(n, cy, ho, ho, device=dev, generator=g).to(torch.bfloat16))
    wcls = (torch.randn(9 * cx * cy, device=dev, generator=g) * 0.05).to(torch.bfloat16)

    def run(v):
        dx = torch.empty((n, cx, 2 * ho, 2 * ho), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
        part, args, rows = None, (None,) * 5, 0
        if bn:
            link = make_link(tuple(dx.shape), cx, seed, True)
            rows = 4 * L.lib().dir_conv_stats_rows(n, ho, ho)
            part = torch.empty((rows, 2, cx), dtype=torch.float32, device=dev)
            args = C._bn_link_args(link)
        L.check(L.lib().dir_conv_dgrad_s2_ex(L.ptr(dy), L.ptr(wcls), L.ptr(dx), n, ho, ho, cy, cx, *args, L.ptr(part), rows, v, L.stream_ptr(dev)),
                "dir_conv_dgrad_s2_ex")
        return dx, part
    r0, r1 = both(run)
    name = f"dgrad_s2 n{n} {cy}->{cx} Ho{ho} bn={int(bn)}"
    nf = len(FAILS)
    same(name, r0, r1)
    print(f"{name}: {'bit-identical' if len(FAILS) == nf else 'FAIL'}", flush=True)


def main():
    global PAIR
    what = sys.argv[1] if len(sys.argv) > 1 else "big"
    if what == "big":
        PAIR = (L.CONV_TILE_DMA, L.CONV_BIG)
        fwd = [(64, 1024, 256, 1, 1, 14), (256, 512, 2048, 1, 1, 7), (4, 64, 256, 1, 1, 56), (64, 512, 1024, 1, 2, 28), (256, 512, 512, 3, 1, 7),
               (16, 256, 512, 1, 2, 56), (64, 256, 256, 3, 2, 28), (64, 128, 256, 1, 1, 14), (64, 256, 256, 3, 1, 14)]
        dg = [  # n, cin, cout, k, h, addend, s2, bits, bn, recompute
            (64, 1024, 256, 1, 14, True, False, True, True, False), (64, 256, 1024, 1, 14, True, False, True, True, True),
            (64, 512, 256, 1, 14, False, True, True, True, False), (256, 2048, 512, 1, 7, False, False, False, True, True),
            (256, 512, 512, 3, 7, False, False, False, True, True), (64, 256, 256, 1, 14, True, False, False, False, False)]
        s2 = [(64, 256, 256, 14, False), (256, 512, 512, 7, True)]
    else:
        PAIR = (L.CONV_TILE_REG, L.CONV_TILE_DMA)
        fwd = [(3, 64, 256, 1, 1, 56), (3, 128, 512, 1, 1, 28), (5, 256, 1024, 1, 1, 14), (5, 1024, 256, 1, 1, 14), (3, 512, 2048, 1, 1, 7),
               (2, 128, 128, 3, 2, 56), (3, 512, 512, 3, 1, 7), (2, 256, 512, 1, 2, 56), (1, 64, 128, 1, 1, 8), (16, 2048, 512, 1, 1, 7)]
        dg = [(3, 64, 256, 1, 56, True, False, True, True, False), (3, 256, 1024, 1, 14, True, False, True, True, True),
              (5, 1024, 256, 1, 14, False, False, False, True, True), (2, 128, 512, 1, 28, False, True, True, True, False),
              (3, 512, 512, 3, 7, False, False, False, True, True), (2, 256, 256, 1, 14, True, False, False, False, False)]
        s2 = [(2, 128, 128, 28, False), (3, 256, 256, 14, True)]
    for i, c in enumerate(fwd):
        fwd_case(*c, seed=100 + i)
    for i, c in enumerate(dg):
        dgrad_case(c[0], c[1], c[2], c[3], c[4], 200 + i, *c[5:])
    for i, c in enumerate(s2):
        s2_case(*c[:4], seed=300 + i, bn=c[4])
    print("FAILS:", FAILS)
    sys.exit(1 if FAILS else 0)


if __name__ == "__main__":
    main()
