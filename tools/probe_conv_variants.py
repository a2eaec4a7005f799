"""GPU probe: every ResNet-50 conv shape (forward configuration, and the stride-1 data-gradient configuration = the same
kernel with Cin/Cout swapped) at batch B through dir_conv_fwd_variant, K-loop variants side by side (1 = register-staged,
2 = LDS-DMA), interleaved in one process, inputs rotated over enough distinct buffers that the working set exceeds the
256 MB Infinity Cache. Prints per-layer microseconds, TFLOP/s and the per-layer roofline max(FLOP / 2.5 PF, bytes / 8 TB/s).
    python tools/probe_conv_variants.py [B] [variants, e.g. 1,2] [library to load instead of the product's, or -] [rows = only the row-resident kernel's launches]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip import _lib as L  # noqa: E402

SH = [(64, 64, 1, 1, 56, 1), (64, 64, 3, 1, 56, 3), (64, 256, 1, 1, 56, 4), (256, 64, 1, 1, 56, 2), (256, 128, 1, 1, 56, 1), (128, 128, 3, 2, 56, 1),
      (128, 512, 1, 1, 28, 4), (256, 512, 1, 2, 56, 1), (512, 128, 1, 1, 28, 3), (128, 128, 3, 1, 28, 3), (512, 256, 1, 1, 28, 1), (256, 256, 3, 2, 28, 1),
      (256, 1024, 1, 1, 14, 6), (512, 1024, 1, 2, 28, 1), (1024, 256, 1, 1, 14, 5), (256, 256, 3, 1, 14, 5), (1024, 512, 1, 1, 14, 1),
      (512, 512, 3, 2, 14, 1), (512, 2048, 1, 1, 7, 3), (1024, 2048, 1, 2, 14, 1), (2048, 512, 1, 1, 7, 2), (512, 512, 3, 1, 7, 2)]


def run(x, w, y, st, n, h, cin, cout, k, stride, pad, variant):
    rows = L.lib().dir_conv_plan_rows(n, h, h, cin, cout, k, k, stride, pad, 0, variant) if st is not None else 0
    L.check(L.lib().dir_conv_fwd_variant(L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(st), rows, n, h, h, cin, cout, k, k, stride, pad, variant,
                                         L.stream_ptr(x.device)), "dir_conv_fwd_variant")


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2]
    if len(sys.argv) > 3 and sys.argv[3] != "-":
        L.LIB_PATH = os.path.abspath(sys.argv[3])
    only_rows = len(sys.argv) > 4 and sys.argv[4] == "rows"        # only the launches the row-resident 1x1 kernel takes
    dev = torch.device("cuda")
    tot = {v: [0.0, 0.0] for v in variants}
    tot_roof = [0.0, 0.0]
    print(f"B={B}  columns per variant: us (TFLOP/s, fraction of the per-layer roofline max(FLOP/2.5PF, bytes/8TB/s))")
    for cin, cout, k, st, h, cnt in SH:
        pad = k // 2
        ho = (h + 2 * pad - k) // st + 1
        for kind, ci, co, hh, s_ in (("fwd", cin, cout, h, st),) + ((("dgrad", cout, cin, ho, 1),) if st == 1 else ()):
            hout = (hh + 2 * pad - k) // s_ + 1
            if only_rows and not (k == 1 and ci <= 256 and co >= 256):
                continue
            nbytes = (B * hh * hh * ci + B * hout * hout * co) * 2
            nbuf = max(2, min(8, int(600e6 // nbytes) + 1))
            xs = [torch.randn(B, ci, hh, hh, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(nbuf)]
            ys = [torch.empty(B, co, hout, hout, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(nbuf)]
            w = (torch.randn(co, ci, k, k, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            rows = max([L.lib().dir_conv_plan_rows(B, hh, hh, ci, co, k, k, s_, pad, 0, v) for v in variants] + [1])
            stt = torch.empty(rows, 2, co, dtype=torch.float32, device=dev) if kind == "fwd" else None
            flop = 2.0 * B * hout * hout * co * ci * k * k
            roof_us = max(flop / 2.5e15, nbytes / 8e12) * 1e6
            res = {}
            iters = 12
            ok = []
            for v in variants:                         # warm (a variant that does not take the geometry is skipped)
                try:
                    run(xs[0], w, ys[0], stt, B, hh, ci, co, k, s_, pad, v)
                    ok.append(v)
                except L.DirHipError:
                    res[v] = float("nan")
            for rnd in range(2):
                for v in ok:                           # interleaved rounds
                    torch.cuda.synchronize()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for i in range(iters):
                        run(xs[i % nbuf], w, ys[i % nbuf], stt, B, hh, ci, co, k, s_, pad, v)
                    b.record()
                    torch.cuda.synchronize()
                    us = a.elapsed_time(b) / iters * 1e3
                    res[v] = min(res.get(v, 1e9), us)
            line = f"{ci:5d}->{co:5d} k{k} s{s_} H{hh:3d} x{cnt} {kind:5s} KT={k*k*ci//64:3d} roof {roof_us:6.1f}us |"
            for v in variants:
                line += f" v{v}: {res[v]:7.1f}us ({flop / res[v] / 1e6:6.0f} TF, {roof_us / res[v]:4.2f})"
                tot[v][0 if kind == "fwd" else 1] += (res[v] if res[v] == res[v] else min(r for r in res.values() if r == r)) * cnt
            tot_roof[0 if kind == "fwd" else 1] += roof_us * cnt
            print(line, flush=True)
            del xs, ys, w
    for v in variants:
        print(f"variant {v}: fwd {tot[v][0] / 1e3:.3f} ms  dgrad(stride-1) {tot[v][1] / 1e3:.3f} ms")
    print(f"per-layer roofline: fwd {tot_roof[0] / 1e3:.3f} ms  dgrad {tot_roof[1] / 1e3:.3f} ms")


if __name__ == "__main__":
    main()
