"""Phase ablation of the float32 tile kernels (GPU box): builds of csrc/dir_conv_f32.hip with FT_ABLATE = 1 (no DMA), 2 (no epilogue
stores), 3 (no MFMA), 4 (no barriers / DMA waits) next to the product build, each timed on a few layers in a child process.
    python tools/ablate_f32.py [variant: 2 = exact tile (default), 3 = split x3, 4 = split x2]       Results of the ablated builds are wrong on purpose; only the times mean anything."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, os, torch, ctypes
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "imbalanced-regression_amd"))
from dirhip import _lib as L
L.LIB_PATH = %(lib)r
from dirhip.conv_f32 import TILE, conv2d_f32_dgrad, conv2d_f32_fwd, conv2d_f32_wgrad
import bench
B = 256
out = []
for cin, cout, k, st, h in ((64, 64, 3, 1, 56), (256, 256, 3, 1, 14), (1024, 256, 1, 1, 14), (64, 256, 1, 1, 56), (512, 2048, 1, 1, 7)):
    pad = k // 2; ho = (h + 2 * pad - k) // st + 1
    x = torch.randn(B, cin, h, h, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, cout, ho, ho, device="cuda").contiguous(memory_format=torch.channels_last)
    f = bench.event_time_ms(lambda i: conv2d_f32_fwd(x, w, st, pad, variant=%(variant)d), 3, warm=1)
    d = bench.event_time_ms(lambda i: conv2d_f32_dgrad(dy, w, (h, h), st, pad, variant=%(variant)d), 3, warm=1)
    g = bench.event_time_ms(lambda i: conv2d_f32_wgrad(dy, x, (k, k), st, pad, variant=%(variant)d), 3, warm=1)
    out.append(f"{cin}->{cout} k{k} H{h}: fwd {f*1e3:6.0f} dgrad {d*1e3:6.0f} wgrad {g*1e3:6.0f} us")
print(%(tag)r.ljust(12), " | ".join(out))
"""


def main():
    libs = [("product", os.path.join(ROOT, "imbalanced-regression_amd", "dirhip", "libdir_hip.so"))]
    variant = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    for n, tag in ((1, "no_dma"), (2, "no_epilogue"), (3, "no_mfma"), (4, "no_barrier")) + (((5, "no_split"),) if variant > 2 else ()):
        out = os.path.join(ROOT, "build_ablate", f"libdir_hip_f32_{tag}.so")
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "build_alt_lib.py"), "dir_conv_f32.hip", "constexpr int FT_ABLATE = 0;",
                        f"constexpr int FT_ABLATE = {n};", out], check=True, stdout=subprocess.DEVNULL)
        libs.append((tag, out))
    for tag, lib in libs:
        subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "lib": lib, "tag": tag, "variant": variant}], check=True)


if __name__ == "__main__":
    main()
