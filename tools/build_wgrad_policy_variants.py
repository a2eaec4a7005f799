"""Alternative builds of libdir_hip.so whose 1x1 weight-gradient kernel loads dY and / or X NON-TEMPORAL (aux = 2 on the buffer loads),
for tools/ab_two_libs.py:   python tools/build_wgrad_policy_variants.py   ->  dirhip/libdir_hip_wgnt_{dy,x,both}.so"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "imbalanced-regression_amd", "csrc")
OUTD = os.path.join(ROOT, "imbalanced-regression_amd", "dirhip")
BUILD = os.path.join(ROOT, "build_ablate")
os.makedirs(BUILD, exist_ok=True)
subprocess.check_call(["make", "-C", CSRC, "-j8"])
src = open(os.path.join(CSRC, "dir_conv_wgrad.hip")).read()
DEF = "#define WG_BL(rs, vo, so) __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0)"
assert src.count(DEF) == 1
others = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".o") and f != "dir_conv_wgrad.o"]
for name, (a, b) in {"dy": (2, 0), "x": (0, 2), "both": (2, 2)}.items():
    s = src.replace(DEF, f"#define WG_BLA(rs, vo, so) __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, {a})\n"
                         f"#define WG_BLB(rs, vo, so) __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, {b})")
    s = s.replace("WG_BL(rs_dy,", "WG_BLA(rs_dy,").replace("WG_BL(rs_x,", "WG_BLB(rs_x,")
    assert "WG_BL(" not in s
    p = os.path.join(BUILD, f"dir_conv_wgrad_nt_{name}.hip")
    open(p, "w").write(s)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}",
                           "-c", p, "-o", p[:-4] + ".o"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUTD, f"libdir_hip_wgnt_{name}.so"), p[:-4] + ".o"] + others)
    print("built", name, flush=True)
