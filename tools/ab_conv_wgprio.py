"""A/B builds (GPU box): wavefront PRIORITY experiments on the bf16 conv tile kernels (never tried in rounds 1-5; the phase ablations say the short-K
launches cost the SUM of their phases although four independent workgroups share a CU — i.e. the co-resident workgroups run their phases in
lockstep and share every pipe round-robin instead of overlapping one's DMA wait with another's MFMAs).
    python tools/ab_conv_wgprio.py <variant>[,<variant>...] [B] [probe|step|both]
variants (each one library under build_ablate/):
  wgprio    static priority per WORKGROUP, (blockIdx.x >> 8) & 3, set once at kernel entry (the workgroups that share a CU differ by 256 in block id)
  wgprio2   the same with two levels, (blockIdx.x >> 8) & 1
  mfmaprio  s_setprio 1 around every MFMA step, 0 after it
  epiprio   s_setprio 3 from the start of the epilogue (the workgroup that is about to free its slot goes first)
  wgprio+epiprio, ... : combinations with '+'
Prints tools/probe_conv_variants.py (heuristic kernels, every layer isolated) for the product library and each variant, and tools/ab_two_libs.py
(whole train steps / tail forwards, alternating child processes)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "imbalanced-regression_amd", "csrc")
BUILD = os.path.join(ROOT, "build_ablate")

SMEM_DECL = "    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];\n"
WGPRIO = ("    { const int pr_ = (blockIdx.x >> 8) & 3; if (pr_ == 1) __builtin_amdgcn_s_setprio(1); else if (pr_ == 2) __builtin_amdgcn_s_setprio(2); "
          "else if (pr_ == 3) __builtin_amdgcn_s_setprio(3); }\n")
WGPRIO2 = "    { if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(1); }\n"


def macro_span(src, name, start=0):
    i = src.index("#define " + name, start)
    j = i
    while True:
        e = src.index("\n", j)
        if not src[j:e].rstrip().endswith("\\"):
            return i, e
        j = e + 1


def patch_conv(src, parts):
    out = src
    if "wgprio" in parts or "wgprio2" in parts:
        ins = WGPRIO if "wgprio" in parts else WGPRIO2
        for kern in ("conv_igemm_kernel(ConvP p) {", "conv_igemm_dma_kernel(ConvP p) {", "conv3x3_patch_kernel(ConvP p) {"):
            k = out.index(kern)
            d = out.index(SMEM_DECL, k)
            out = out[:d + len(SMEM_DECL)] + ins + out[d + len(SMEM_DECL):]
    if "mfmaprio" in parts:
        pos = 0
        for name in ("CV_MFMA_STEP(buf)", "CV_MFMA_STEP(stage)", "CVB_MFMA(stage)", "CP_MFMA_STEP(r_, s_, ps, bs)"):
            i, e = macro_span(out, name, pos)
            body = out[i:e]
            lines = body.split("\n")
            # line 0: #define ... \ ; line 1: "    {   \" ; last line: "    }"
            assert lines[1].strip().startswith("{") and lines[-1].strip() == "}", (name, lines[1], lines[-1])
            pad = " " * 100
            lines.insert(2, "        __builtin_amdgcn_s_setprio(1);" + pad + "\\")
            lines.insert(len(lines) - 1, "        __builtin_amdgcn_s_setprio(0);" + pad + "\\")
            out = out[:i] + "\n".join(lines) + out[e:]
            pos = i + 10
    return out


def patch_epilogue(src, parts):
    if "epiprio" not in parts:
        return src
    k = src.index("__device__ __forceinline__ void cv_epilogue(const ConvP& p")
    b = src.index("{\n", k)
    return src[:b + 2] + "    __builtin_amdgcn_s_setprio(3);\n" + src[b + 2:]


def build(variant):
    parts = variant.split("+")
    d = os.path.join(BUILD, "prio_" + variant.replace("+", "_"))
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "dir_conv.hip"), "w").write(patch_conv(open(os.path.join(CSRC, "dir_conv.hip")).read(), parts))
    open(os.path.join(d, "dir_conv_epilogue.h"), "w").write(patch_epilogue(open(os.path.join(CSRC, "dir_conv_epilogue.h")).read(), parts))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    obj = os.path.join(d, "dir_conv.o")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function", f"-I{d}", f"-I{ROOT}/include", f"-I{CSRC}",
                    "-c", os.path.join(d, "dir_conv.hip"), "-o", obj], check=True)
    out = os.path.join(BUILD, f"libdir_hip_prio_{variant.replace('+', '_')}.so")
    others = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".o") and f != "dir_conv.o"]
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj] + others, check=True)
    return out


def main():
    variants = sys.argv[1].split(",")
    b = sys.argv[2] if len(sys.argv) > 2 else "256"
    what = sys.argv[3] if len(sys.argv) > 3 else "both"
    libs = [(v, build(v)) for v in variants]
    if what == "compile-only":
        return
    if what in ("probe", "both"):
        for name, lib in [("product", "-")] + libs:
            print("=== probe, library:", name, flush=True)
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe_conv_variants.py"), b, "0", lib], check=True)
    if what in ("step", "both"):
        for name, lib in libs:
            print("=== whole steps, library:", name, flush=True)
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab_two_libs.py"), lib, "2", "16"], check=True)


if __name__ == "__main__":
    main()
