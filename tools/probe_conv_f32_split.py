"""GPU probe: the float32 tile kernels in their three arithmetics — exact (v_mfma_f32_32x32x2_f32), split-bf16 x3 (6 products) and x2 (3 products) —
on every ResNet-50 conv shape at batch B: time of forward / data gradient / weight gradient, and the error of each against the EXACT kernel's result
relative to the result's RMS (x3 is expected at float32 rounding level, ~1e-7; x2 at ~2^-17; a bf16 product at ~2^-9).
    python tools/probe_conv_f32_split.py [B] [library to load instead of the product's]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip import _lib as L  # noqa: E402
if len(sys.argv) > 2:
    L.LIB_PATH = os.path.abspath(sys.argv[2])
import bench  # noqa: E402
from dirhip.conv_f32 import TILE, conv2d_f32_dgrad, conv2d_f32_fwd, conv2d_f32_wgrad  # noqa: E402

X3, X2 = 3, 4


def relerr(a, ref):
    return float((a.double() - ref.double()).pow(2).mean().sqrt() / ref.double().pow(2).mean().sqrt())


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    names = {TILE: "exact", X3: "x3", X2: "x2"}
    tot = {v: [0.0, 0.0, 0.0] for v in names}
    worst = {v: [0.0, 0.0, 0.0] for v in (X3, X2)}
    for cin, cout, k, st, h, cnt in bench.RESNET50_CONVS:
        if cin % 16 or cout % 16:
            continue                                    # (the 7x7 stem stays on the gather kernel)
        pad = k // 2
        ho = (h + 2 * pad - k) // st + 1
        x = torch.randn(B, cin, h, h, device="cuda").contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(B, cout, ho, ho, device="cuda").contiguous(memory_format=torch.channels_last)
        flop = 2.0 * B * ho * ho * cout * cin * k * k
        row = f"{cin:5d}->{cout:5d} k{k} s{st} H{h:3d} x{cnt}"
        ref = None
        for v in (TILE, X3, X2):
            outs = (conv2d_f32_fwd(x, w, st, pad, variant=v), conv2d_f32_dgrad(dy, w, (h, h), st, pad, variant=v), conv2d_f32_wgrad(dy, x, (k, k), st, pad, variant=v))
            f = bench.event_time_ms(lambda i: conv2d_f32_fwd(x, w, st, pad, variant=v), 3, warm=1)
            d = bench.event_time_ms(lambda i: conv2d_f32_dgrad(dy, w, (h, h), st, pad, variant=v), 3, warm=1)
            g = bench.event_time_ms(lambda i: conv2d_f32_wgrad(dy, x, (k, k), st, pad, variant=v), 3, warm=1)
            for j, t in enumerate((f, d, g)):
                tot[v][j] += t * cnt
            row += f" | {names[v]:5s} fwd {f * 1e3:6.0f}us {flop / f / 1e9:4.0f}TF dgrad {d * 1e3:6.0f} {flop / d / 1e9:4.0f}TF wgrad {g * 1e3:6.0f} {flop / g / 1e9:4.0f}TF"
            if v == TILE:
                ref = outs
            else:
                errs = [relerr(o, r) for o, r in zip(outs, ref)]
                worst[v] = [max(a, b) for a, b in zip(worst[v], errs)]
                row += " err " + "/".join(f"{e:.1e}" for e in errs)
        print(row, flush=True)
        del x, w, dy, ref
    for v in names:
        print(f"sum ms (fwd, dgrad, wgrad) {names[v]:5s}: {tot[v][0]:.2f} {tot[v][1]:.2f} {tot[v][2]:.2f}  total {sum(tot[v]):.2f}")
    for v in (X3, X2):
        print(f"worst error vs exact, relative to the result's RMS (fwd, dgrad, wgrad) {names[v]}: " + " ".join(f"{e:.2e}" for e in worst[v]))


if __name__ == "__main__":
    main()
