"""ctypes binding of tools/lib/libdir_hip_tools.so (tools/csrc/dir_hip_tools.h): the box-calibration and lane-mapping probes.
Not part of the product package — only bench.py's `peaks` leg, tools/ and tests/ load it. Built by __graft_entry__.build()."""
import ctypes
import os

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libdir_hip_tools.so")
_c = ctypes
SIGNATURES = {
    "dir_probe_stream_copy": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "dir_probe_stream_read": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "dir_probe_stream_write": (_c.c_int, [_c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "dir_probe_mfma_bf16": (_c.c_int, [_c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "dir_probe_mfma_f32": (_c.c_int, [_c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "dir_probe_l2_read": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
    "dir_probe_tr16": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p]),
}
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (or `make -C tools/csrc`)")
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib
