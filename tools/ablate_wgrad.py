"""Where does the 1x1 weight-gradient kernel (conv_wgrad_kernel<128, 128, SIMPLE, PF = 2>) spend its time?  Builds of a patched COPY of
csrc/dir_conv_wgrad.hip with one phase removed (as tools/ablate_conv.py; results are wrong on purpose):
    full / nomfma (fragment reads kept) / nofrag (no fragment reads, no MFMA) / noload (no global loads) / nostore (no transposing LDS
    writes) / loadonly (global loads + barriers only) / qpart, nopart (a quarter / none of the partial-tile stores of the epilogue)
    python tools/ablate_wgrad.py build ;  python tools/ablate_wgrad.py run [B]     (GPU box)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "imbalanced-regression_amd", "csrc")
OUTD = os.path.join(ROOT, "imbalanced-regression_amd", "dirhip")
BUILD = os.path.join(ROOT, "build_ablate")
NAMES = ["full", "nomfma", "nofrag", "noload", "nostore", "loadonly", "qpart", "nopart"]
MFMA = "acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], bb[ni], acc[mi][ni], 0, 0, 0);   \\"
LOADA = "S##a0 = WG_BL(rs_dy,"


def patched(src, name):
    assert src.count(MFMA) == 1
    if name in ("nomfma",):
        src = src.replace(MFMA, 'asm volatile("" :: "v"(a[mi]), "v"(bb[ni]));   \\')
    if name in ("nofrag", "loadonly"):
        src = src.replace("#define WG_MFMA_STEP(buf)                                                                                         \\\n    {",
                          "#define WG_MFMA_STEP(buf)                                                                                         \\\n    if (false) {")
    if name == "noload":
        src = src.replace("#define WG_BL(rs, vo, so) __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0)",
                          "#define WG_BL(rs, vo, so) (u32x4){(uint32_t)(vo), (uint32_t)(so), 1u, 2u}")
    if name in ("nostore", "loadonly"):
        # keep the loaded registers alive (one never-true store) but skip the transposing LDS writes
        src = src.replace("        if (doA) wg_transpose_store(As + (buf) * A_BYTES, aw, S##a0, S##a1, S##a2, S##a3);                        \\",
                          "        if ((S##a0.x ^ S##a1.y ^ S##a2.z ^ S##a3.w ^ S##b0.x ^ S##b1.y ^ S##b2.z ^ S##b3.w) == 0x12345u) As[t] = 1;         \\")
        src = src.replace("        if (doB) wg_transpose_store(Bs + (buf) * B_BYTES, bw, S##b0, S##b1, S##b2, S##b3);                        \\", "        \\")
    if name in ("qpart", "nopart"):
        # the partial-tile stores of the epilogue (4-byte stores, 64 per lane): a quarter of them / none
        old = "                ob[toff] = acc[mi][ni][e];"
        assert src.count(old) == 1
        src = src.replace(old, "                if ((e & 3) == 0) ob[toff] = acc[mi][ni][e];" if name == "qpart" else
                          "                if (acc[mi][ni][e] == 12345.678f) ob[toff] = acc[mi][ni][e];")
    return src


def build():
    os.makedirs(BUILD, exist_ok=True)
    subprocess.check_call(["make", "-C", CSRC, "-j8"])
    src = open(os.path.join(CSRC, "dir_conv_wgrad.hip")).read()
    others = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".o") and f != "dir_conv_wgrad.o"]
    procs = []
    for n in NAMES:
        p = os.path.join(BUILD, f"dir_conv_wgrad_{n}.hip")
        open(p, "w").write(patched(src, n))
        procs.append((n, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function",
                                           "-Wno-unused-variable", "-Wno-unused-but-set-variable", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}", "-c", p, "-o", p[:-4] + ".o"])))
    for n, pr in procs:
        assert pr.wait() == 0, n
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUTD, f"libdir_hip_ablw_{n}.so"),
                               os.path.join(BUILD, f"dir_conv_wgrad_{n}.o")] + others)
        print("built", n, flush=True)


CHILD = r'''
import os, sys, torch
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip import _lib as L
L.LIB_PATH = sys.argv[2]
B = int(sys.argv[3])
SH = [(64, 256, 56), (256, 64, 56), (128, 512, 28), (512, 128, 28), (256, 1024, 14), (1024, 256, 14), (512, 2048, 7), (2048, 512, 7)]
dev = torch.device("cuda")
lib = L.lib()
out = []
for ci, co, h in SH:
    nbytes = B * h * h * (ci + co) * 2
    nbuf = max(2, min(8, int(600e6 // nbytes) + 1))
    xs = [torch.randn(B, ci, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(nbuf)]
    dys = [torch.randn(B, co, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(nbuf)]
    dw = torch.empty(co, ci, device=dev)
    nws = lib.dir_conv_wgrad_workspace(B, h, h, ci, co, 1, 1, 1, 0, 1)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    def run(i):
        L.check(lib.dir_conv_wgrad(L.ptr(dys[i % nbuf]), L.ptr(xs[i % nbuf]), L.ptr(dw), B, h, h, ci, co, 1, 1, 1, 0, 1, L.ptr(ws), nws, L.stream_ptr(dev)), "wgrad")
    for i in range(3): run(i)
    best = 1e9
    for r in range(3):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(16): run(i)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 16 * 1e3)
    out.append(best)
    del xs, dys
print(" ".join(f"{v:.1f}" for v in out))
'''
SHN = ["64->256@56", "256->64@56", "128->512@28", "512->128@28", "256->1024@14", "1024->256@14", "512->2048@7", "2048->512@7"]


def run(B):
    res = {}
    for rnd in range(2):
        for n in NAMES:
            lib = os.path.join(OUTD, f"libdir_hip_ablw_{n}.so")
            o = subprocess.run([sys.executable, "-c", CHILD, ROOT, lib, str(B)], capture_output=True, text=True, timeout=600)
            if o.returncode != 0:
                print(n, o.stderr[-1500:])
                continue
            v = [float(x) for x in o.stdout.strip().splitlines()[-1].split()]
            res[n] = [min(a, b) for a, b in zip(res.get(n, v), v)]
    print(f"B={B}: microseconds per dir_conv_wgrad call (1x1 / stride 1: split-K kernel + reduce), per build")
    print(f"{'layer':>14s} " + " ".join(f"{n:>9s}" for n in NAMES))
    for i, s in enumerate(SHN):
        print(f"{s:>14s} " + " ".join(f"{res[n][i]:9.1f}" if n in res else "      n/a" for n in NAMES))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 256)
