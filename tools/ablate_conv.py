"""Where does a short-K 1x1 conv launch spend its time?  Ablation builds of conv_igemm_dma_kernel<128, 1, *> (the single-stage
LDS-DMA kernel of every K <= 1024 1x1 layer), each with one phase removed from a patched COPY of csrc/dir_conv.hip (the product
source is not touched; the copies and their objects live under build_ablate/, the libraries next to the product's, git-ignored):
    full      the product kernel
    nomfma    fragment reads kept, v_mfma removed
    nofrag    neither fragment reads nor v_mfma (DMA + barriers + epilogue)
    noload    no DMA (MFMAs on whatever LDS holds) + epilogue
    noepi     DMA + MFMA, no epilogue (one never-taken store keeps the accumulators alive)
    loadonly  DMA + barriers only
    epionly   no DMA, no fragment reads, no v_mfma: the store loop (of zeros) + statistics
    epinostats  epionly without the statistics
    stagger<n>  the product kernel, but the first 1024 workgroups (the ones resident together at launch) start (slot) x n x 64 clocks
                late, slot = (blockIdx / 8 / 32) & 3: are the co-resident workgroups of a CU marching in lockstep through load / MFMA / store?
    timed     the product kernel with per-workgroup phase clocks (s_memrealtime, 100 MHz): load (DMA issue -> barrier), MFMA step,
              epilogue — NOTE the final clock read makes the compiler wait for the tile's stores (vmcnt(0)), so "epilogue" includes
              the store acknowledgement the product kernel pays at s_endpgm
    python tools/ablate_conv.py build          (here: hipcc cross-compiles; ABLATE_ONLY=a,b limits the set)
    python tools/ablate_conv.py timed [B]      (GPU box)
    python tools/ablate_conv.py run [B]        (GPU box: one child process per library, the 1x1 stride-1 shapes, forward + dgrad)
Results are wrong on purpose; only the times mean anything."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "imbalanced-regression_amd", "csrc")
OUTD = os.path.join(ROOT, "imbalanced-regression_amd", "dirhip")
BUILD = os.path.join(ROOT, "build_ablate")
NAMES = ["full", "nomfma", "nofrag", "noload", "noepi", "loadonly", "epionly", "epinostats"]
if os.environ.get("ABLATE_ONLY"):
    NAMES = os.environ["ABLATE_ONLY"].split(",")

LOOP = """            CV_ISSUE_TILE(0);
            __syncthreads();                                        // (drains the DMA: vmcnt(0) before the barrier)
            CV_MFMA_STEP(0);
            __syncthreads();"""
EPI = "    cv_epilogue<BN, LEAN, (NST == 1 && !LEAN) ? 4 : 2>(p, acc, smem, t, m0, n0, mt);"
MFMA = "acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);  /* D'[channel][pixel] */  \\"
NOEPI = """    { float s_ = 0.0f;
      for (int mi = 0; mi < MI; ++mi) for (int ni = 0; ni < NI; ++ni) for (int e = 0; e < 16; ++e) s_ += acc[mi][ni][e];
      if (s_ == 1234.5f) p.y[t] = 1; }"""


def patched(src, name):
    assert src.count(LOOP) == 1 and src.count(EPI) == 1 and src.count(MFMA) == 2
    # (the accumulators keep their zero start: a run-time start value costs the product kernel 70 spilled registers. Without MFMAs the
    # compiler therefore folds the epilogue's arithmetic; its stores and barriers stay)
    head, tail = src.split("conv_igemm_dma_kernel(ConvP p) {", 1)
    if name in ("nomfma", "loadonly"):
        tail = tail.replace(MFMA, 'asm volatile("" :: "v"(a[mi]), "v"(b[ni]));  \\', 1)
    src = head + "conv_igemm_dma_kernel(ConvP p) {" + tail
    if name == "nofrag" or name == "loadonly":
        src = src.replace(LOOP, LOOP.replace("            CV_MFMA_STEP(0);\n", ""))
    if name.startswith("stagger"):
        n = int(name[7:])
        mark = "    const int mt = lin / p.ntn, nt = lin - mt * p.ntn;\n    const int m0 = mt * CV_BM, n0 = nt * BN;\n    const int t = threadIdx.x, lane = t & 63;\n    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);      // wave-uniform by construction; tell the compiler (LDS base -> M0)"
        assert tail.count(mark) >= 1
        tail = tail.replace(mark, mark + "\n    if (NST == 1 && blockIdx.x < 1024) { const int slot_ = (blockIdx.x >> 8) & 3; for (int d_ = 0; d_ < slot_; ++d_) __builtin_amdgcn_s_sleep(%d); }" % n, 1)
        src = head + "conv_igemm_dma_kernel(ConvP p) {" + tail
    if name == "timed":
        # per-workgroup phase clocks (100 MHz real-time counter): [start, sum(load: issue -> barrier), sum(MFMA step + barrier), epilogue, end]
        tl = LOOP.replace("            CV_ISSUE_TILE(0);", "            const unsigned long long ta_ = wall_clock64();\n            CV_ISSUE_TILE(0);")
        tl = tl.replace("            CV_MFMA_STEP(0);", "            const unsigned long long tb_ = wall_clock64();\n            CV_MFMA_STEP(0);")
        tl += "\n            { const unsigned long long tc_ = wall_clock64(); tload_ += tb_ - ta_; tmfma_ += tc_ - tb_; }"
        src = src.replace(LOOP, tl)
        src = src.replace("    if (NST == 1) {\n        for (int kt = 0; kt < p.KT; ++kt) {\n            const unsigned long long ta_",
                          "    unsigned long long tload_ = 0, tmfma_ = 0; const unsigned long long ts_ = wall_clock64();\n    if (NST == 1) {\n        for (int kt = 0; kt < p.KT; ++kt) {\n            const unsigned long long ta_")
        assert "tload_ = 0" in src
        src = src.replace(EPI, "    const unsigned long long td_ = wall_clock64();\n" + EPI + "\n    if (NST == 1 && t == 0 && g_dbg_) { const unsigned long long te_ = wall_clock64(); unsigned long long* o_ = g_dbg_ + (size_t)lin * 5; o_[0] = ts_; o_[1] = tload_; o_[2] = tmfma_; o_[3] = te_ - td_; o_[4] = te_; }")
        src = src.replace("namespace {", "__device__ unsigned long long* g_dbg_ = nullptr;\nextern \"C\" int dir_dbg_set(unsigned long long* p_) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_), &p_, sizeof(p_)); }\nnamespace {", 1)
    if name == "noload":
        src = src.replace(LOOP, LOOP.replace("CV_ISSUE_TILE(0);", "++ld_k;"))
    if name in ("epionly", "epinostats"):
        src = src.replace(LOOP, LOOP.replace("CV_ISSUE_TILE(0);", "++ld_k;").replace("            CV_MFMA_STEP(0);\n", ""))
    if name == "epinostats":
        src = src.replace(EPI, "    { ConvP pq = p; pq.stats = nullptr;\n" + EPI.replace("(p, acc", "(pq, acc") + " }")
    if name in ("noepi", "loadonly"):
        src = src.replace(EPI, NOEPI)
    return src


def build():
    os.makedirs(BUILD, exist_ok=True)
    subprocess.check_call(["make", "-C", CSRC, "-j8"])
    src = open(os.path.join(CSRC, "dir_conv.hip")).read()
    others = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".o") and f != "dir_conv.o"]
    procs = []
    for n in NAMES:
        p = os.path.join(BUILD, f"dir_conv_{n}.hip")
        open(p, "w").write(patched(src, n))
        procs.append((n, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                                           "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable",
                                           f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}", "-c", p, "-o", p[:-4] + ".o"])))
    for n, pr in procs:
        assert pr.wait() == 0, n
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUTD, f"libdir_hip_abl_{n}.so"),
                               os.path.join(BUILD, f"dir_conv_{n}.o")] + others)
        print("built", n, flush=True)


CHILD = r'''
import os, sys, torch
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip import _lib as L
L.LIB_PATH = sys.argv[2]
B = int(sys.argv[3])
SH = [(64, 256, 56), (256, 64, 56), (256, 128, 56), (128, 512, 28), (512, 128, 28), (512, 256, 28), (256, 1024, 14), (1024, 256, 14), (1024, 512, 14),
      (512, 2048, 7), (2048, 512, 7), (64, 64, 56)]
dev = torch.device("cuda")
out = []
for ci, co, h in SH:
    nbytes = B * h * h * (ci + co) * 2
    nbuf = max(2, min(8, int(600e6 // nbytes) + 1))
    xs = [torch.randn(B, ci, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(nbuf)]
    ys = [torch.empty(B, co, h, h, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(nbuf)]
    w = (torch.randn(co, ci, 1, 1, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    st = torch.empty(L.lib().dir_conv_stats_rows(B, h, h), 2, co, dtype=torch.float32, device=dev)
    def run(i):
        L.check(L.lib().dir_conv_fwd_variant(L.ptr(xs[i % nbuf]), L.ptr(w), L.ptr(ys[i % nbuf]), L.ptr(st), st.shape[0], B, h, h, ci, co, 1, 1, 1, 0, 0,
                                             L.stream_ptr(dev)), "conv")
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
    del xs, ys
print(" ".join(f"{v:.1f}" for v in out))
'''
SHN = ["64->256@56", "256->64@56", "256->128@56", "128->512@28", "512->128@28", "512->256@28", "256->1024@14", "1024->256@14", "1024->512@14",
       "512->2048@7", "2048->512@7", "64->64@56"]


def run(B):
    res = {}
    for rnd in range(2):
        for n in NAMES:
            lib = os.path.join(OUTD, f"libdir_hip_abl_{n}.so")
            o = subprocess.run([sys.executable, "-c", CHILD, ROOT, lib, str(B)], capture_output=True, text=True, timeout=600)
            if o.returncode != 0:
                print(n, o.stderr[-1500:])
                continue
            v = [float(x) for x in o.stdout.strip().splitlines()[-1].split()]
            res[n] = [min(a, b) for a, b in zip(res.get(n, v), v)]
    print(f"B={B}: microseconds per launch, forward configuration, stats epilogue (lean), K loop variant per build")
    print(f"{'layer':>14s} " + " ".join(f"{n:>9s}" for n in NAMES))
    for i, s in enumerate(SHN):
        print(f"{s:>14s} " + " ".join(f"{res[n][i]:9.1f}" if n in res else "      n/a" for n in NAMES))


TIMED = r'''
import ctypes, os, sys, torch
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip import _lib as L
L.LIB_PATH = sys.argv[2]
B = int(sys.argv[3])
dev = torch.device("cuda")
lib = L.lib()
raw = ctypes.CDLL(sys.argv[2])
raw.dir_dbg_set.argtypes = [ctypes.c_void_p]
for ci, co, h in [(64, 256, 56), (128, 512, 28), (512, 128, 28), (256, 1024, 14), (1024, 256, 14), (1024, 512, 14), (512, 2048, 7)]:
    x = torch.randn(B, ci, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = torch.empty(B, co, h, h, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 1, 1, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    st = torch.empty(lib.dir_conv_stats_rows(B, h, h), 2, co, dtype=torch.float32, device=dev)
    nb = ((B * h * h + 127) // 128) * (co // 128)
    dbg = torch.zeros(nb, 5, dtype=torch.int64, device=dev)
    assert raw.dir_dbg_set(dbg.data_ptr()) == 0
    for i in range(3):
        L.check(lib.dir_conv_fwd_variant(L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(st), st.shape[0], B, h, h, ci, co, 1, 1, 1, 0, 0, L.stream_ptr(dev)), "conv")
    torch.cuda.synchronize()
    d = dbg.cpu().double()
    t0 = d[:, 0].min()
    start, end = (d[:, 0] - t0) / 100.0, (d[:, 4] - t0) / 100.0          # microseconds
    life = end - start
    first = start < 0.5 * life.mean()
    print(f"{ci}->{co}@{h}: {nb} workgroups, launch span {end.max():.1f} us; per workgroup (us): life {life.mean():.2f}  load {d[:, 1].mean() / 100:.2f}  "
          f"mfma {d[:, 2].mean() / 100:.2f}  epilogue {d[:, 3].mean() / 100:.2f} | first-resident workgroups ({int(first.sum())}): life {life[first].mean():.2f}  "
          f"load {d[first, 1].mean() / 100:.2f} mfma {d[first, 2].mean() / 100:.2f} epilogue {d[first, 3].mean() / 100:.2f} | "
          f"resident on average {life.sum() / end.max():.0f}")
    qs = torch.tensor([0.1, 0.5, 0.9], dtype=torch.float64)
    print("    deciles (10/50/90 %): load", [round(v, 2) for v in (torch.quantile(d[:, 1], qs) / 100).tolist()], " mfma", [round(v, 2) for v in (torch.quantile(d[:, 2], qs) / 100).tolist()],
          " epilogue", [round(v, 2) for v in (torch.quantile(d[:, 3], qs) / 100).tolist()])
'''

if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "timed":
        o = subprocess.run([sys.executable, "-c", TIMED, ROOT, os.path.join(OUTD, "libdir_hip_abl_timed.so"), sys.argv[2] if len(sys.argv) > 2 else "256"],
                           capture_output=True, text=True, timeout=600)
        print(o.stdout, o.stderr[-3000:])
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 256)
