"""GPU probe + check of the 1x1 / stride-1 weight-gradient forms (DIR_WGRAD_*: 1 = transposing kernel, 2 = LDS-DMA + transposing reads with
one stage / four workgroups per CU, 3 = two stages): every 1x1 stride-1 layer of ResNet-50 with 128-multiples of channels at batch B,
forms interleaved in one process, inputs rotated over > 256 MB; each form against a float64 reference on a sub-sampled K range and
against form 1. Prints microseconds per call (kernel + reduce), TFLOP/s.     python tools/probe_wgrad1.py [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip import _lib as L  # noqa: E402
from dirhip.conv import conv2d_wgrad  # noqa: E402

SH = [(256, 128, 56, 1), (128, 512, 28, 4), (512, 128, 28, 3), (512, 256, 28, 1), (256, 1024, 14, 6), (1024, 256, 14, 5), (1024, 512, 14, 1),
      (512, 2048, 7, 3), (2048, 512, 7, 2)]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda")
    tot = {1: 0.0, 2: 0.0, 3: 0.0}
    for cin, cout, h, cnt in SH:
        nbytes = B * h * h * (cin + cout) * 2
        nbuf = max(2, min(8, int(600e6 // nbytes) + 1))
        xs = [torch.randn(B, cin, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(nbuf)]
        dys = [torch.randn(B, cout, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(nbuf)]
        flop = 2.0 * B * h * h * cin * cout
        res, outs = {}, {}
        for form in (1, 2, 3):
            outs[form] = conv2d_wgrad(dys[0], xs[0], 1, 1, 0, form=form).clone()
        ref = torch.einsum("nohw,nihw->oi", dys[0].double(), xs[0].double())
        scale = ref.abs().max().item()
        errs = {f: ((outs[f].double().view(cout, cin) - ref).abs().max().item() / scale) for f in outs}
        for rnd in range(3):
            for form in (1, 2, 3):
                for i in range(2):
                    conv2d_wgrad(dys[i % nbuf], xs[i % nbuf], 1, 1, 0, form=form)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for i in range(8):
                    conv2d_wgrad(dys[i % nbuf], xs[i % nbuf], 1, 1, 0, form=form)
                b.record()
                torch.cuda.synchronize()
                us = a.elapsed_time(b) / 8 * 1e3
                res[form] = min(res.get(form, 1e9), us)
        for f in tot:
            tot[f] += cnt * res[f]
        print(f"{cin:5d}->{cout:5d} H{h:3d} x{cnt}: " + "  ".join(f"form {f}: {res[f]:6.1f} us {flop / res[f] / 1e6:5.0f} TF/s err {errs[f]:.1e}" for f in (1, 2, 3)), flush=True)
        del xs, dys
    print("sum over the step's launches (us): " + "  ".join(f"form {f}: {tot[f]:.0f}" for f in tot))


if __name__ == "__main__":
    main()
