"""Build a variant of libdir_hip.so with ONE translation unit textually patched (A/B of two builds of a kernel with tools/ab_two_libs.py):
    python tools/build_alt_lib.py dir_conv_wgrad.hip "constexpr int W1_FORM = 2;" "constexpr int W1_FORM = 0;" build_ablate/libdir_hip_w1form0.so
The other objects are the product's (imbalanced-regression_amd/csrc/*.o: run `make` there first). Output under build_ablate/ (git-ignored,
gpurun-ignored: build on the box, i.e. inside the gpurun command)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "imbalanced-regression_amd", "csrc")


def main():
    unit, old, new, out = sys.argv[1:5]
    out = os.path.abspath(out)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    src = open(os.path.join(CSRC, unit)).read()
    assert src.count(old) == 1, f"{old!r} occurs {src.count(old)} times in {unit}"
    tmp_src = os.path.join(os.path.dirname(out), "alt_" + unit)
    open(tmp_src, "w").write(src.replace(old, new))
    obj = tmp_src.replace(".hip", ".o")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function", f"-I{ROOT}/include", f"-I{CSRC}",
                    "-c", tmp_src, "-o", obj], check=True)
    others = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".o") and f != unit.replace(".hip", ".o")]
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj] + others, check=True)
    print("built", out)


if __name__ == "__main__":
    main()
