"""GPU check of the persistent ring convolution kernel (csrc/dir_conv_ring.hip) against the one-tile-per-workgroup kernels:
every launch form (plain forward with statistics; data gradients with shortcut addend / compact stride-2 addend / ReLU bit
mask / fused BatchNorm-backward sums with and without mask recompute; the parity classes of the stride-2 data gradient), ragged
and whole M, K loops of 1, 2, 4, 8, 16, 18, 72 steps — outputs and partial-sum lists must be BIT-IDENTICAL (same MFMA order, same
summation order), plus an fp32 reference check of the forward.
    python tools/check_ring.py [quick]"""
import os
import sys
import types

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip import _lib as L  # noqa: E402
from dirhip import conv as C  # noqa: E402

dev = torch.device("cuda")
FAILS = []


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def both(fn):
    """fn() with the ring kernel off, then on."""
    out = []
    for mode in (0, 1):
        prev = L.lib().dir_conv_set_ring(mode)
        try:
            out.append(fn())
        finally:
            L.lib().dir_conv_set_ring(prev)
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
    (y0, s0), (y1, s1) = both(lambda: C.conv2d_igemm(x, w, stride=stride, padding=pad, want_stats=True))
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
    print(f"{name}: ring vs fp32 rel {err:.2e}, stats {serr:.1e} {'ok' if ok else 'FAIL'}", flush=True)


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

    def run():
        link = None
        if bn:
            link = types.SimpleNamespace(x=cl(torch.randn(yshape, device=dev, generator=torch.Generator(device="cuda").manual_seed(seed + 1)).to(torch.bfloat16)),
                                         gamma=torch.rand(cout, device=dev, generator=torch.Generator(device="cuda").manual_seed(seed + 2)) + 0.5,
                                         beta=torch.randn(cout, device=dev, generator=torch.Generator(device="cuda").manual_seed(seed + 3)) * 0.3,
                                         mean=torch.randn(cout, device=dev, generator=torch.Generator(device="cuda").manual_seed(seed + 4)) * 0.2,
                                         rstd=torch.rand(cout, device=dev, generator=torch.Generator(device="cuda").manual_seed(seed + 5)) + 0.7,
                                         recompute_mask=recompute, partial=None)
        y = C.conv2d_igemm(x, w, stride=1, padding=pad, addend=add, addend_s2=add2, bn_link=link, relu_bits=rb)
        return y, (link.partial if link is not None else None)
    r0, r1 = both(run)
    name = f"dgrad n{n} {cin}->{cout} k{k} H{h} add={int(addend)} s2={int(s2)} bits={int(bits)} bn={int(bn)}/{int(recompute)}"
    nf = len(FAILS)
    same(name, r0, r1)
    print(f"{name}: {'bit-identical' if len(FAILS) == nf else 'FAIL'}", flush=True)


def s2_case(n, cy, cx, ho, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    dy = cl(torch.randn(n, cy, ho, ho, device=dev, generator=g).to(torch.bfloat16))
    wcls = (torch.randn(9 * cx * cy, device=dev, generator=g) * 0.05).to(torch.bfloat16)

    def run():
        dx = torch.empty((n, cx, 2 * ho, 2 * ho), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
        L.check(L.lib().dir_conv_dgrad_s2(L.ptr(dy), L.ptr(wcls), L.ptr(dx), n, ho, ho, cy, cx, L.stream_ptr(dev)), "dir_conv_dgrad_s2")
        return dx
    r0, r1 = both(run)
    name = f"dgrad_s2 n{n} {cy}->{cx} Ho{ho}"
    nf = len(FAILS)
    same(name, r0, r1)
    print(f"{name}: {'bit-identical' if len(FAILS) == nf else 'FAIL'}", flush=True)


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    fwd = [(3, 64, 256, 1, 1, 56), (3, 128, 512, 1, 1, 28), (5, 256, 1024, 1, 1, 14), (5, 1024, 256, 1, 1, 14), (3, 512, 2048, 1, 1, 7),
           (2, 128, 128, 3, 2, 56), (3, 512, 512, 3, 1, 7), (2, 256, 512, 1, 2, 56), (1, 64, 128, 1, 1, 8), (16, 2048, 512, 1, 1, 7)]
    if not quick:
        fwd += [(256, 256, 1024, 1, 1, 14), (256, 64, 256, 1, 1, 56), (64, 1024, 2048, 1, 2, 14)]
    for i, c in enumerate(fwd):
        fwd_case(*c, seed=100 + i)
    dg = [  # n, cin, cout, k, h, addend, s2, bits, bn, recompute
        (3, 64, 256, 1, 56, True, False, True, True, False), (3, 256, 1024, 1, 14, True, False, True, True, True),
        (5, 1024, 256, 1, 14, False, False, False, True, True), (2, 128, 512, 1, 28, False, True, True, True, False),
        (3, 512, 128, 1, 28, False, False, False, True, True), (3, 512, 512, 3, 7, False, False, False, True, True),
        (2, 256, 256, 1, 14, True, False, False, False, False), (4, 2048, 512, 1, 7, False, False, True, False, False)]
    if not quick:
        dg += [(256, 64, 256, 1, 56, True, False, True, True, False), (256, 1024, 256, 1, 14, False, False, False, True, True)]
    for i, c in enumerate(dg):
        dgrad_case(c[0], c[1], c[2], c[3], c[4], 200 + i, *c[5:])
    for i, c in enumerate([(2, 128, 128, 28), (3, 256, 256, 14), (2, 512, 512, 7)]):
        s2_case(*c, seed=300 + i)
    print("FAILS:", FAILS)
    sys.exit(1 if FAILS else 0)


if __name__ == "__main__":
    main()
