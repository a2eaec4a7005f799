"""A/B build (GPU box): the MFMA steps of the bf16 conv K loops with the NEXT sub-step's fragments requested before the current sub-step's MFMAs
(sched_barrier keeps the order; the compiler's own schedule issues the four reads of sub-step kk + 1 only after the MFMAs of kk: in-order
wavefronts then expose the LDS latency once per sub-step unless another wavefront covers it).    python tools/ab_conv_frag_prefetch.py [B]
Builds build_ablate/libdir_hip_fragpf.so from a patched copy of csrc/dir_conv.hip and runs tools/probe_conv_variants.py for both libraries."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "imbalanced-regression_amd", "csrc")
OUT = os.path.join(ROOT, "build_ablate", "libdir_hip_fragpf.so")

NEW_STEP = r'''    {                                                                                                           \
        bf16x8 a[2][MI], b[2][NI];                                                                              \
        _Pragma("unroll")                                                                                       \
        for (int mi = 0; mi < MI; ++mi) a[0][mi] = *reinterpret_cast<const bf16x8*>(SBASE + AOFF(mi, 0));        \
        _Pragma("unroll")                                                                                       \
        for (int ni = 0; ni < NI; ++ni) b[0][ni] = *reinterpret_cast<const bf16x8*>(SBASE + BOFF(ni, 0));        \
        _Pragma("unroll")                                                                                       \
        for (int kk = 0; kk < 4; ++kk) {                                                                        \
            if (kk < 3) {                                                                                       \
                _Pragma("unroll")                                                                               \
                for (int mi = 0; mi < MI; ++mi) a[(kk + 1) & 1][mi] = *reinterpret_cast<const bf16x8*>(SBASE + AOFF(mi, kk + 1)); \
                _Pragma("unroll")                                                                               \
                for (int ni = 0; ni < NI; ++ni) b[(kk + 1) & 1][ni] = *reinterpret_cast<const bf16x8*>(SBASE + BOFF(ni, kk + 1)); \
            }                                                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
            _Pragma("unroll")                                                                                   \
            for (int mi = 0; mi < MI; ++mi)                                                                     \
                _Pragma("unroll")                                                                               \
                for (int ni = 0; ni < NI; ++ni)                                                                 \
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[kk & 1][ni], a[kk & 1][mi], acc[mi][ni], 0, 0, 0); \
        }                                                                                                       \
    }
'''


def patch(src):
    # the DMA tile kernel's and the CU-tile kernel's MFMA steps (both read fragments at smem + stage * STAGE + af / bf [..][kk])
    out = src
    for name, stage_expr in (("CV_MFMA_STEP(stage)", "(stage) * STAGE"), ("CVB_MFMA(stage)", "(stage) * G::STAGE")):
        i = out.index("#define " + name)
        if name.startswith("CV_MFMA_STEP"):
            i = out.index("#define " + name, out.index("conv_igemm_dma_kernel(ConvP p)"))
        j = i                                                       # end of the macro: the first line that does not end with a backslash
        while True:
            e = out.index("\n", j)
            if not out[j:e].rstrip().endswith("\\"):
                j = e
                break
            j = e + 1
        body = NEW_STEP.replace("SBASE", f"smem + {stage_expr}").replace("AOFF(mi, 0)", "af[mi][0]").replace("BOFF(ni, 0)", "bf[ni][0]") \
                       .replace("AOFF(mi, kk + 1)", "af[mi][kk + 1]").replace("BOFF(ni, kk + 1)", "bf[ni][kk + 1]")
        out = out[:i] + "#define " + name + " " * 20 + "\\\n" + body.rstrip("\n") + out[j:]
    return out


def main():
    b = sys.argv[1] if len(sys.argv) > 1 else "256"
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    src = open(os.path.join(CSRC, "dir_conv.hip")).read()
    tmp = os.path.join(os.path.dirname(OUT), "alt_dir_conv.hip")
    open(tmp, "w").write(patch(src))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    obj = tmp.replace(".hip", ".o")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function", f"-I{ROOT}/include", f"-I{CSRC}",
                    "-c", tmp, "-o", obj], check=True)
    if len(sys.argv) > 2 and sys.argv[2] == "compile-only":
        return
    others = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".o") and f != "dir_conv.o"]
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, obj] + others, check=True)
    for lib in ("-", OUT):
        print("=== library:", "product" if lib == "-" else lib, flush=True)
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe_conv_variants.py"), b, "0,2,5", lib], check=True)


if __name__ == "__main__":
    main()
