"""GPU probe: the launches the product heuristic gives to the 256 x 256 CU-tile kernel (and a few candidates), product library vs an alternative
build (tools/build_alt_lib.py), alternating child processes. Forward configuration with statistics, inputs rotated over > 256 MB.
    python tools/probe_conv_big.py [alt.so]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip import _lib as L
if sys.argv[2] != "-":
    L.LIB_PATH = sys.argv[2]
from dirhip.conv import conv2d_igemm
SH = [(256, 256, 3, 1, 14), (1024, 256, 1, 1, 14), (1024, 512, 1, 1, 14), (256, 256, 3, 2, 28), (512, 512, 3, 1, 7), (2048, 512, 1, 1, 7)]
B = 256
dev = torch.device("cuda")
out = []
for cin, cout, k, st, h in SH:
    pad = k // 2
    ho = (h + 2 * pad - k) // st + 1
    nbytes = (B * h * h * cin + B * ho * ho * cout) * 2
    nbuf = max(2, min(8, int(600e6 // nbytes) + 1))
    xs = [torch.randn(B, cin, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(nbuf)]
    w = (torch.randn(cout, cin, k, k, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    var = L.CONV_BIG if L.lib().dir_conv_plan_rows(B, h, h, cin, cout, k, k, st, pad, 0, L.CONV_BIG) else L.CONV_AUTO
    best = 1e9
    for rnd in range(3):
        for i in range(2): conv2d_igemm(xs[i % nbuf], w, st, pad, want_stats=True, variant=var)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(8): conv2d_igemm(xs[i % nbuf], w, st, pad, want_stats=True, variant=var)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 8 * 1e3)
    out.append(best)
    del xs, w
print(" ".join(f"{v:.2f}" for v in out))
'''
SH = ["256->256 k3 H14", "1024->256 k1 H14", "1024->512 k1 H14", "256->256 k3 s2 H28", "512->512 k3 H7 (98 tiles)", "2048->512 k1 H7 (98 tiles)"]
FLOP = [2 * 256 * 196 * 256 * 256 * 9, 2 * 256 * 196 * 1024 * 256, 2 * 256 * 196 * 1024 * 512, 2 * 256 * 196 * 256 * 256 * 9, 2 * 256 * 49 * 512 * 512 * 9, 2 * 256 * 49 * 2048 * 512]


def main():
    alt = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else None
    res = {"product": [], "alt": []}
    for r in range(2):
        for name, path in (("product", "-"),) + ((("alt", alt),) if alt else ()):
            o = subprocess.run([sys.executable, "-c", CHILD, ROOT, path], capture_output=True, text=True, timeout=600)
            if o.returncode != 0:
                print(o.stderr[-1500:])
                continue
            res[name].append([float(v) for v in o.stdout.strip().splitlines()[-1].split()])
    for i, s in enumerate(SH):
        p = min(r[i] for r in res["product"])
        line = f"{s:28s} forced CU tile: product {p:6.1f} us = {FLOP[i] / p / 1e6:5.0f} TF/s"
        if res["alt"]:
            a = min(r[i] for r in res["alt"])
            line += f"   alt {a:6.1f} us = {FLOP[i] / a / 1e6:5.0f} TF/s   alt/product {a / p:.3f}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
