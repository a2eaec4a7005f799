"""L2 -> CU delivery rate with the latency covered (dir_probe_l2_read): region size x loads in flight per lane x workgroups per CU."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip import _lib as L  # noqa: E402


def main():
    dev = torch.device("cuda")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import toolslib
    lib = toolslib.lib()
    src = torch.empty(64 << 20, dtype=torch.uint8, device=dev).random_(0, 255)
    out = torch.empty(1 << 16, dtype=torch.float32, device=dev)
    st = L.stream_ptr(dev)
    for region_mb in (2, 8, 32):
        for wgs_per_cu in (1, 2, 4, 8):
            for depth in (4, 8, 16):
                wgs, passes = 256 * wgs_per_cu, max(1, 64 // region_mb // wgs_per_cu * 2)
                region = region_mb << 20
                L.check(lib.dir_probe_l2_read(L.ptr(src), L.ptr(out), region, wgs, 1, depth, st), "warm")
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                L.check(lib.dir_probe_l2_read(L.ptr(src), L.ptr(out), region, wgs, passes, depth, st), "probe")
                b.record()
                torch.cuda.synchronize()
                ms = a.elapsed_time(b)
                tb = wgs * passes * region / ms / 1e9
                print(f"region {region_mb:2d} MB  {wgs_per_cu} WG/CU  depth {depth:2d} ({wgs_per_cu * depth * 4:3d} KB in flight per CU): {tb:6.1f} TB/s = "
                      f"{tb * 1e12 / 256 / 2.4e9:5.1f} B/clk/CU", flush=True)


if __name__ == "__main__":
    main()
