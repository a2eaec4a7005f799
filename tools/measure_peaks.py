"""Measured peaks of THIS box: STREAM-style HBM copy / read bandwidth and the bf16 / f32 MFMA issue peak,
through the dir_probe_* entry points of tools/lib/libdir_hip_tools.so, timed with HIP events on the launch stream. Prints one JSON object;
`python tools/measure_peaks.py > gpurun_out/peaks.json` (committed as profiles/rNN_peaks.json together with rocminfo)."""
import ctypes
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip import _lib as L  # noqa: E402


def ev(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def measure(device=None):
    dev = device or torch.device("cuda", torch.cuda.current_device())
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import toolslib
    lib = toolslib.lib()
    out = {"device": torch.cuda.get_device_name(dev), "nominal": {"hbm_GBs": 8000.0, "bf16_mfma_TFs": 2500.0, "f32_mfma_TFs": 157.3}}
    nbytes = 2 << 30                                   # 2 GiB src + 2 GiB dst: far beyond the 256 MB Infinity Cache
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev).random_(0, 255)
    dst = torch.empty_like(src)
    red = torch.empty(8192, dtype=torch.float32, device=dev)
    st = L.stream_ptr(dev)
    ms = ev(lambda: L.check(lib.dir_probe_stream_copy(L.ptr(src), L.ptr(dst), nbytes, st), "copy"), 10)
    out["stream_copy_GBs"] = 2 * nbytes / ms / 1e6
    ms = ev(lambda: L.check(lib.dir_probe_stream_read(L.ptr(src), L.ptr(red), nbytes, st), "read"), 10)
    out["stream_read_GBs"] = nbytes / ms / 1e6
    ms = ev(lambda: L.check(lib.dir_probe_stream_write(L.ptr(dst), nbytes, st), "write"), 10)
    out["stream_write_GBs"] = nbytes / ms / 1e6
    del src, dst
    wgs = 256 * 8                                     # 8 workgroups of 4 wavefronts per CU = 8 wavefronts per SIMD (2 resident rounds)
    buf = torch.empty(wgs * 256, dtype=torch.float32, device=dev)
    fl = ctypes.c_double(0.0)
    for name, fn, iters in (("bf16_mfma_TFs", lib.dir_probe_mfma_bf16, 4000), ("f32_mfma_TFs", lib.dir_probe_mfma_f32, 1000)):
        ms = ev(lambda: L.check(fn(wgs, iters, L.ptr(buf), ctypes.byref(fl), st), name), 5)
        out[name] = fl.value / ms / 1e9
    return out


if __name__ == "__main__":
    res = measure()
    try:
        info = subprocess.run(["/opt/rocm/bin/rocminfo"], capture_output=True, text=True, timeout=60).stdout
        keep = [ln.strip() for ln in info.splitlines() if any(k in ln for k in ("Marketing Name", "Compute Unit", "Max Clock", "gfx", "Wavefront Size", "Cacheline", "L2:", "L3:"))]
        res["rocminfo_excerpt"] = keep[:60]
    except Exception as e:                              # noqa: BLE001
        res["rocminfo_excerpt"] = [f"rocminfo failed: {e}"]
    print(json.dumps(res, indent=1))
