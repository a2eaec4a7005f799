#!/usr/bin/env python
"""bench.py — images/sec of the ResNet-50 + LDS + FDS training hot path on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md §8d config 2, per GPU): IMDB-WIKI-DIR shapes — ResNet-50, bf16 conv stack,
B=256 synthetic 224x224 batches, LDS weights (sqrt_inv, gaussian 5/2) from a synthetic 191 509-label long-tailed set, FDS
(bucket_num=100, bucket_start=0, ks=5, sigma=2, momentum 0.9) with tables populated by two update rounds and the run at
epoch >= 2 so calibration is non-trivial (A.4), loss l1, Adam 1e-3.

A "step" is one optimisation step (train.py:246-262). Nothing of the hot path is skipped: every --epoch-len steps the timed
region also runs the reference's epoch tail (train.py:269-281) over the same number of batches — the no-grad train-mode
feature pass, FDS.update_last_epoch_stats and FDS.update_running_stats (with the cross-rank statistic all-reduce when
N > 1). `value` counts trained images only (K * B * N / time), so it is the throughput of the whole loop, tail included;
`train_only_images_per_sec` is the same clock without the tail.

Rank 0 prints ONE JSON line of at most 6 KB (a 20 KB line was not parsed by the driver) and writes the untrimmed result to
gpurun_out/bench_detail.json. After the timed region, rank 0 at N=1 also measures (none of it inside `value`):
  * `roofline` — the dominant kernel family, the MFMA implicit-GEMM convolution (conv_igemm_*: 52 forward + 49 stride-1 /
    compact data-gradient + 12 parity-class launches per step), IN SITU: per-kernel device times of whole training steps
    from the profiler's kernel trace (the same numbers `rocprofv3 --kernel-trace` of this command reports;
    profiles/rNN_train_step_breakdown.txt), against the algorithmic FLOPs of those launches; `traffic` from two live
    rocprofv3 --pmc child passes;
  * `kernel_rooflines` — six rows: weight gradient and BatchNorm family in situ, the FDS epoch scatter, calibrate forward / backward at
    an HBM-resident size, the NYUD2 narrow-row scatter, each with algorithmic bytes / FLOPs from SURVEY.md §8d;
  * `roofline_step` — the per-step floor (sum of per-kernel rooflines) at nominal and measured peaks;
  * `cpu_baseline` — the reference's own loop (or its pinned torch-CPU port where /root/reference is absent) on the host cores at B=8.
`--extras` (minutes; tools/bench_extras.py) adds to the detail file: every conv shape alone (`conv_layers`), all FDS shapes, the float32 modes (exact / split-bf16 x3 / x2),
the real-file input pipeline, the vendor-library step, the CPU micro-baselines.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))

FLOP_FWD_BWD = 24.287e9       # per 224^2 image, SURVEY.md §8d (FlopCounterMode on the reference resnet50)
FLOP_FWD = 8.174e9
PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0         # HBM3E spec, MI355X_MICROARCH.md (achievable ~6300)
N_TRAIN = 191509              # IMDB-WIKI-DIR train-set size (paper; the csv is not vendored)
_T0 = time.time()

# ResNet-50 convolutions on the MFMA implicit-GEMM kernels (every layer but the 7x7 stem):
# (Cin, Cout, k, stride, Hin, count) — SURVEY.md Appendix B
RESNET50_CONVS = [(64, 64, 1, 1, 56, 1), (64, 64, 3, 1, 56, 3), (64, 256, 1, 1, 56, 4), (256, 64, 1, 1, 56, 2), (256, 128, 1, 1, 56, 1),
                  (128, 128, 3, 2, 56, 1), (128, 512, 1, 1, 28, 4), (256, 512, 1, 2, 56, 1), (512, 128, 1, 1, 28, 3), (128, 128, 3, 1, 28, 3),
                  (512, 256, 1, 1, 28, 1), (256, 256, 3, 2, 28, 1), (256, 1024, 1, 1, 14, 6), (512, 1024, 1, 2, 28, 1), (1024, 256, 1, 1, 14, 5),
                  (256, 256, 3, 1, 14, 5), (1024, 512, 1, 1, 14, 1), (512, 512, 3, 2, 14, 1), (512, 2048, 1, 1, 7, 3), (1024, 2048, 1, 2, 14, 1),
                  (2048, 512, 1, 1, 7, 2), (512, 512, 3, 1, 7, 2)]


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.time() - _T0:7.1f}s]", *a, file=sys.stderr, flush=True)


def long_tail_labels(rng, n):
    """Fixed long-tailed pmf over integer ages 0..120: round(clip(|N(0,18)| + 20, 0, 120))."""
    return np.clip(np.round(np.abs(rng.normal(0, 18, n)) + 20), 0, 120).astype(np.float32)


def conv_flops_per_image():
    """Forward FLOPs per image of the 52 implicit-GEMM layers (2 * Ho*Wo * Cout * Cin * k*k each)."""
    tot = 0.0
    for cin, cout, k, st, h, cnt in RESNET50_CONVS:
        ho = (h + 2 * (k // 2) - k) // st + 1
        tot += cnt * 2.0 * ho * ho * cout * cin * k * k
    return tot


def build(args, device, rank, amp_dtype=torch.bfloat16, force_collectives=False):
    from dirhip import lds
    from dirhip.parallel import DataParallelEngine
    from dirhip.resnet import resnet50
    torch.manual_seed(0)
    model = resnet50(fds=True, bucket_num=100, bucket_start=0, start_update=0, start_smooth=1,
                     kernel="gaussian", ks=5, sigma=2, momentum=0.9).to(device)
    engine = DataParallelEngine(model, amp_dtype=amp_dtype, channels_last=True, force_collectives=force_collectives)
    engine.train()
    from dirhip.optim import Adam
    optimizer = Adam(engine.parameters(), lr=1e-3)          # torch.optim.Adam's arithmetic and state, one HIP launch (+ the bf16 weight operands)
    # ---- LDS weights: native host routine on the synthetic train-label set (identical on every rank)
    rng_all = np.random.default_rng(1)
    all_labels = long_tail_labels(rng_all, N_TRAIN)
    w_all = np.asarray(lds.prepare_weights(all_labels, "sqrt_inv", lds=True, lds_kernel="gaussian", lds_ks=5, lds_sigma=2))
    w_by_age = np.zeros(121, np.float32)
    w_by_age[all_labels.astype(np.int64)] = w_all
    # ---- per-rank synthetic batches, resident in HBM
    g = torch.Generator(device=device).manual_seed(1000 + rank)
    rng = np.random.default_rng(1000 + rank)
    batches = []
    for _ in range(args.epoch_len):
        x = torch.randn(args.batch, 3, 224, 224, device=device, generator=g).contiguous(memory_format=torch.channels_last)
        lab = long_tail_labels(rng, args.batch)
        y = torch.as_tensor(lab, device=device).view(-1, 1)
        w = torch.as_tensor(w_by_age[lab.astype(np.int64)], device=device).view(-1, 1)
        batches.append((x, y, w))
    # ---- populate the FDS tables: two synthetic update rounds (epochs 0 and 1)
    fds = model.FDS
    fds.sync_across_ranks = False                  # identical synthetic rounds on every rank
    for ep in range(2):
        lab = torch.as_tensor(long_tail_labels(np.random.default_rng(50 + ep), 20000), device=device)
        feats = torch.randn(20000, 2048, device=device, generator=torch.Generator(device=device).manual_seed(60 + ep)).abs_() * 0.5 \
            + 0.01 * lab[:, None]
        fds.update_last_epoch_stats(ep)
        fds.update_running_stats(feats, lab, ep)
    fds.sync_across_ranks = True
    return model, engine, optimizer, batches


def run_steps(engine, optimizer, batches, store, n_steps, epoch_len, epoch0, loss_fn, with_tail=True):
    from dirhip.train_loop import epoch_tail, train_step
    epoch = epoch0
    loss = None
    for s in range(n_steps):
        x, y, w = batches[s % len(batches)]
        loss = train_step(engine, optimizer, x, y, w, epoch, loss_fn)
        if with_tail and (s + 1) % epoch_len == 0:
            epoch_tail(engine, ((bx, by) for bx, by, _ in batches), epoch, store)
            epoch += 1
    rem = n_steps % epoch_len
    if with_tail and rem:
        # a last, shorter epoch: the reference's second pass touches every trained batch exactly once (train.py:269-281), so the
        # timed region holds ONE tail forward per trained batch for any --steps (not only multiples of --epoch-len)
        epoch_tail(engine, ((bx, by) for bx, by, _ in batches[:rem]), epoch, store)
        epoch += 1
    return loss, epoch


def timed(fn, device, world):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, out


def event_time_ms(fn, iters, warm=3):
    """Average duration of fn(i) over `iters` launches with HIP events on torch's current stream
    (= the stream the C-ABI launches on)."""
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


FAMILIES = (("conv_igemm", "conv_igemm"), ("conv3x3_patch", "conv_igemm"), ("conv_wgrad", "conv_wgrad"), ("stem_", "stem"), ("bn_relu_maxpool", "stem_tail"),
            ("bn_", "batchnorm"), ("tail_", "tail"), ("fds_", "fds"), ("loss_", "loss"), ("scale_by_scalar", "loss"),
            ("conv_prep_weights", "weight_prep"), ("FusedAdam", "optimizer"), ("adam_step", "optimizer"), ("Cijk_", "library_gemm"), ("miopen", "library_miopen"),
            ("ncclDevKernel", "rccl_collective"), ("nccl", "rccl_collective"), ("rccl", "rccl_collective"))


def in_situ_breakdown(engine, optimizer, batches, loss_fn, epoch, steps=4):
    """Device time of every kernel of `steps` whole training steps (profiler kernel trace = roctracer, the data rocprofv3
    --kernel-trace records), per kernel family and per step."""
    from torch.profiler import ProfilerActivity, profile
    from dirhip.train_loop import train_step
    for s in range(2):
        train_step(engine, optimizer, *batches[s % len(batches)], epoch, loss_fn)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for s in range(steps):
            train_step(engine, optimizer, *batches[s % len(batches)], epoch, loss_fn)
        torch.cuda.synchronize()
    fam = {}
    for e in prof.key_averages():
        tot = getattr(e, "device_time_total", None)
        if tot is None:
            tot = getattr(e, "cuda_time_total", 0.0)
        if not tot:
            continue
        name = e.key
        tag = "other"
        for needle, t in FAMILIES:
            if needle in name:
                tag = t
                break
        f = fam.setdefault(tag, {"us_per_step": 0.0, "launches_per_step": 0.0})
        f["us_per_step"] += tot / steps
        f["launches_per_step"] += e.count / steps
    return fam


def fds_kernel_rooflines(device, full=False):
    """Algorithmic bytes / measured duration for the hand-written FDS kernels at their full sizes. Default: the rows the
    driver's line carries (epoch scatter, calibrate forward / backward at an HBM-resident B, the NYUD2 narrow-row scatter); `full` adds the
    B = 256 launch-bound pair, the NYUD2 calibrate forms and the STS-B shapes (bench_detail.json)."""
    from dirhip import ops
    out = []
    g = torch.Generator(device=device).manual_seed(3)

    def row(kernel, bound, shape, ms, alg):
        return {"kernel": kernel, "bound": bound, "shape": shape, "ms": ms, "algorithmic_bytes": alg, "achieved": alg / ms / 1e6,
                "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": alg / ms / 1e6 / PEAK_HBM_GBS}
    # --- dir_fds_scatter_stats over the whole epoch's features: N*C*4 + N*4 bytes read
    n, c, nb = N_TRAIN, 2048, 100
    lab = torch.as_tensor(long_tail_labels(np.random.default_rng(3), n), device=device)
    feats = torch.randn(n, c, device=device, generator=g).abs_()
    bins, _ = ops.bin_index(lab, 0, 100)
    ms = event_time_ms(lambda i: ops.scatter_stats(feats, bins, nb), 10)
    out.append(row("dir_fds_scatter_stats", "hbm", f"N={n} C={c} Nb={nb}", ms, n * c * 4 + n * 4))
    del feats
    # --- calibrate forward: 2*B*C*4 + T*U*C*4 + B*4, T = 3 tables (m1, scale, m2; scale precomputed per epoch)
    for b in ((256, 65536) if full else (65536,)):
        x = torch.randn(b, c, device=device, generator=g)
        labb = torch.as_tensor(long_tail_labels(np.random.default_rng(4), b), device=device)
        m1 = torch.randn(nb, c, device=device, generator=g)
        sc = torch.rand(nb, c, device=device, generator=g) + 0.5
        m2 = torch.randn(nb, c, device=device, generator=g)
        u = int(torch.unique(labb.clamp(max=99)).numel())
        ms = event_time_ms(lambda i: ops.smooth_fwd_(x, labb, 0, 100, m1, sc, m2), 50 if b == 256 else 10)
        out.append(row("dir_fds_smooth_fwd", "hbm" if b > 2048 else "launch", f"B={b} C={c} U={u} T=3", ms, 2 * b * c * 4 + 3 * u * c * 4 + b * 4))
        bins_b, _ = ops.bin_index(labb, 0, 100)
        dy = torch.randn(b, c, device=device, generator=g)
        ms = event_time_ms(lambda i: ops.calibrate_bwd(dy, bins_b, sc), 50 if b == 256 else 10)
        out.append(row("dir_fds_calibrate_bwd", "hbm" if b > 2048 else "launch", f"B={b} C={c} U={u}", ms, 2 * b * c * 4 + u * c * 4 + b * 4))
        del x, dy
    # --- "next" rows: NYUD2 dense map [32,128,114,152] (per-pixel buckets, narrow-row kernels)
    b, c, h, w = 32, 128, 114, 152
    depth = torch.rand(b * h * w, device=device, generator=g) * 9.3 + 0.7
    rows = torch.rand(b * h * w, c, device=device, generator=g)
    bins = ops.bin_scaled(depth, 10.0, 7, 100)
    t1, sc, t2 = (torch.rand(93, c, device=device, generator=g) + 0.5 for _ in range(3))
    ms = event_time_ms(lambda i: ops.scatter_stats(rows, bins, 93), 5)
    out.append(row("dir_fds_scatter_stats (NYUD2 narrow rows)", "hbm", f"N={rows.shape[0]} C={c} Nb=93", ms, rows.numel() * 4 + rows.shape[0] * 4))
    if not full:
        return out
    ms = event_time_ms(lambda i: ops.calibrate_fwd_(rows, bins, t1, sc, t2), 10)
    out.append(row("dir_fds_calibrate_fwd (NYUD2 narrow rows)", "hbm", f"[{b},{c},{h},{w}] 93 buckets", ms,
                   2 * rows.numel() * 4 + 3 * 93 * c * 4 + rows.shape[0] * 4))
    fmap = torch.rand(b, c, h, w, device=device, generator=g)
    outm = torch.empty_like(fmap)
    ms = event_time_ms(lambda i: ops.calibrate_nchw(fmap, bins, t1, sc, t2, out=outm), 10)
    out.append(row("dir_fds_calibrate_fwd_nchw (NYUD2 map, NCHW)", "hbm", f"[{b},{c},{h},{w}] 93 buckets", ms,
                   2 * fmap.numel() * 4 + 3 * 93 * c * 4 + rows.shape[0] * 4))
    del fmap, outm, rows
    # --- STS-B-DIR (BASELINE configs[4]; sts-b-dir/fds.py:96-143, util.py:63-73): C = 12000 sentence features, 50 histogram buckets on [0, 5]
    c, nb = 12000, 50
    edges = torch.tensor(np.histogram(np.array([], np.float32), bins=nb, range=(0., 5.))[1].astype(np.float32), device=device)
    t1, sc, t2 = (torch.rand(nb, c, device=device, generator=g) + 0.5 for _ in range(3))
    for b in (128, 16384):
        lab = torch.rand(b, device=device, generator=g) * 5.0
        bins_b = ops.bin_edges(lab, edges, 0, nb)
        u = int(torch.unique(bins_b).numel())
        x = torch.randn(b, c, device=device, generator=g)
        ms = event_time_ms(lambda i: ops.calibrate_fwd_(x, bins_b, t1, sc, t2), 50 if b == 128 else 10)
        out.append(row("dir_fds_calibrate_fwd (STS-B)", "hbm" if b > 2048 else "launch", f"B={b} C={c} U={u} T=3", ms, 2 * b * c * 4 + 3 * u * c * 4 + b * 4))
        dy = torch.randn(b, c, device=device, generator=g)
        ms = event_time_ms(lambda i: ops.calibrate_bwd(dy, bins_b, sc), 50 if b == 128 else 10)
        out.append(row("dir_fds_calibrate_bwd (STS-B)", "hbm" if b > 2048 else "launch", f"B={b} C={c} U={u}", ms, 2 * b * c * 4 + u * c * 4 + b * 4))
        if b > 2048:
            ms = event_time_ms(lambda i: ops.scatter_stats(x, bins_b, nb), 10)
            out.append(row("dir_fds_scatter_stats (STS-B)", "hbm", f"N={b} C={c} Nb={nb}", ms, b * c * 4 + b * 4))
        del x, dy
    return out


def pmc_traffic(batch):
    """HBM bytes of the conv forward / data-gradient launches from the hardware counters: two rocprofv3 --pmc passes (FETCH_SIZE,
    WRITE_SIZE — separate runs, counter collection only next to --kernel-trace) over tools/pmc_conv_pass.py as child processes,
    parsed by tools/pmc_conv_parse.py (units and the gfx950 FETCH_SIZE correction as MI355X_MICROARCH.md prescribes)."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.isfile(exe):
        raise FileNotFoundError("rocprofv3")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_conv_parse
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        env = dict(os.environ, TMPDIR="/tmp")
        for name in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--pmc", name, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, name), "-o", name, "--",
                   sys.executable, os.path.join(ROOT, "tools", "pmc_conv_pass.py"), str(batch)]
            p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            if p.returncode != 0:
                raise RuntimeError(f"rocprofv3 --pmc {name}: rc {p.returncode}: {p.stderr[-300:]}")
        return pmc_conv_parse.parse(tmp, batch)


def cpu_baseline(seconds_budget=20.0, full=False):
    """The reference's training loop on the host cores, next to the GPU number (a reported baseline, not the target):
      * the loop of BASELINE configs[1] at the CPU-runnable batch of configs[0]: ResNet-50 + FDS + LDS weights + l1 + Adam, B=8, an
        epoch tail (second train-mode pass + FDS update) every 4 steps;
      * configs[0] itself: AgeDB-DIR, LDS-only (fds=False), B=8;
      * the SURVEY §8d micro-baselines of the reference's FDS / loss code path.
    "kind": "reference" = the reference's OWN modules (resnet.py / fds.py / loss.py imported from /root/reference through
    oracle/refshim.py) when that tree exists (build container); on the GPU box it does not, and the same loop runs on
    oracle/torch_oracle.py, the torch-CPU port that tests/test_torch_oracle.py pins to the live reference (loss trajectory 1e-6,
    FDS buffers) — "kind": "port"."""
    from contextlib import nullcontext
    from oracle import refshim, torch_oracle
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    fds_kw = dict(bucket_num=100, bucket_start=0, start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=2, momentum=0.9)
    live = refshim.available()
    if live:
        ref = refshim.load("imdb-wiki-dir")
        ctx = refshim.cuda_identity
        make = lambda **kw: refshim.make_resnet50("imdb-wiki-dir", **kw)        # noqa: E731
        loss_l1 = ref.loss.weighted_l1_loss
        source = "the reference's own modules (imdb-wiki-dir/{resnet,fds,loss}.py via oracle/refshim.py)"
    else:
        ctx = nullcontext
        make = lambda **kw: torch_oracle.RefResNet50(fds=kw.pop("fds"), **kw)       # noqa: E731
        loss_l1 = lambda o, t, w: torch_oracle.ref_weighted_loss("l1", o, t, w)      # noqa: E731
        source = "oracle/torch_oracle.py (torch-CPU port of resnet.py / fds.py / loss.py / train.py:246-281, pinned to the live reference at 1e-6)"

    def step(model, opt, x, y, w, epoch, fds):                     # train.py:246-262
        model.train()
        with ctx():
            out = model(x, y, epoch)
        out = out[0] if fds else out
        loss = loss_l1(out, y, w)
        assert np.isfinite(loss.item()) and loss.item() < 1e6
        opt.zero_grad()
        loss.backward()
        opt.step()

    def tail(model, batches, epoch):                                # train.py:269-281
        enc, labs = [], []
        with torch.no_grad(), ctx():
            for x, y, _ in batches:
                _, f = model(x, y, epoch)
                enc.extend(f.data.squeeze().cpu().numpy())
                labs.extend(y.data.squeeze().cpu().numpy())
            model.FDS.update_last_epoch_stats(epoch)
            model.FDS.update_running_stats(torch.from_numpy(np.vstack(enc)), torch.from_numpy(np.hstack(labs)), epoch)
    rng = np.random.default_rng(0)
    b, epoch_len = 8, 4
    g = torch.Generator().manual_seed(0)
    batches = []
    for _ in range(epoch_len):
        lab = long_tail_labels(rng, b)
        batches.append((torch.randn(b, 3, 224, 224, generator=g), torch.as_tensor(lab).view(-1, 1), torch.rand(b, 1, generator=g) + 0.5))
    # ---- leg 1: IMDB-WIKI loop with LDS + FDS
    with ctx():
        model = make(fds=True, **fds_kw)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    epoch = 0
    for _ in range(2):                                  # warm-up epochs also populate the FDS tables (epochs 0, 1)
        for x, y, w in batches[:2]:
            step(model, opt, x, y, w, epoch, True)
        tail(model, batches[:2], epoch)
        epoch += 1
    t0 = time.perf_counter()
    steps = 0
    while True:
        for x, y, w in batches:
            step(model, opt, x, y, w, epoch, True)
            steps += 1
        tail(model, batches, epoch)
        epoch += 1
        if time.perf_counter() - t0 > seconds_budget * 0.45 or steps >= 32:
            break
    dt = time.perf_counter() - t0
    res = {"value": steps * b / dt, "unit": "images/sec", "cores": cores, "kind": "reference" if live else "port", "batch": b,
           "batch_note": f"B={b} (BASELINE configs[0]'s CPU-runnable batch), not the benchmarked B=256: a B=256 float32 step of the reference needs "
                         "~60 GB of activations and minutes per step on the host",
           "sample": f"{steps} steps of B={b} 224x224 fp32 + {steps // epoch_len} epoch tails ({epoch_len} fwd passes + FDS update each), "
                     f"{source}, {dt:.1f} s"}
    del model, opt
    if not full:                                         # the driver's line carries leg 1 only; --extras adds configs[0] and the micro-baselines
        return res
    # ---- leg 2: BASELINE configs[0] = BASELINE.md §3 config 1: AgeDB-DIR ResNet-50, LDS-only (no FDS), l1, Adam, B=8
    with ctx():
        m0 = make(fds=False, **dict(fds_kw, bucket_start=3))
    o0 = torch.optim.Adam(m0.parameters(), lr=1e-3)
    ages = [(x, torch.clamp(y, 1, 101), w) for x, y, w in batches]
    step(m0, o0, *ages[0], 0, False)
    t0 = time.perf_counter()
    n0 = 0
    while time.perf_counter() - t0 < seconds_budget * 0.2 and n0 < 16:
        step(m0, o0, *ages[n0 % len(ages)], 0, False)
        n0 += 1
    dt0 = time.perf_counter() - t0
    res["config0_agedb_lds_only"] = {"value": n0 * b / dt0, "unit": "images/sec", "sample": f"{n0} steps of B={b} (AgeDB-DIR, LDS weights, fds=False), {dt0:.1f} s"}
    del m0, o0
    # ---- micro-baselines: the reference's per-label host loops on the CPU
    def med(fn, n):
        ts = []
        for _ in range(n):
            t = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t)
        return float(np.median(ts))
    micro = []
    fds = torch_oracle.RefFDS(2048, bucket_num=100, bucket_start=0, start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=2, momentum=0.9)
    lab_n = torch.as_tensor(long_tail_labels(np.random.default_rng(5), 50000))
    feats_n = torch.randn(50000, 2048, generator=g).abs_() * 0.5 + 0.01 * lab_n[:, None]
    for ep in range(2):
        fds.update_last_epoch_stats(ep)
        fds.update_running_stats(feats_n[:20000], lab_n[:20000], ep)
    t = med(lambda: fds.update_running_stats(feats_n, lab_n, 2), 3)
    micro.append({"what": "FDS.update_running_stats N=50000 C=2048 (fds.py:84-113)", "ms": t * 1e3, "GB/s": 50000 * 2048 * 4 / t / 1e9})
    lab_b = torch.as_tensor(long_tail_labels(np.random.default_rng(6), 256)).view(-1, 1)

    def smooth_fb():
        xb = torch.randn(256, 2048, generator=g).requires_grad_(True)
        fds.smooth(xb * 1.0, lab_b, 2).sum().backward()
    smooth_fb()
    t = med(smooth_fb, 5)
    micro.append({"what": "FDS.smooth forward + backward B=256 C=2048 (fds.py:115-144)", "ms": t * 1e3})
    xo, yo, wo = torch.randn(256, 1, generator=g) * 10 + 40, torch.as_tensor(long_tail_labels(rng, 256)).view(-1, 1), torch.rand(256, 1, generator=g) + 0.5
    for kind in ("mse", "l1", "focal_mse", "focal_l1", "huber"):
        def lfb(kind=kind):
            xi = xo.clone().requires_grad_(True)
            torch_oracle.ref_weighted_loss(kind, xi, yo, wo).backward()
        lfb()
        micro.append({"what": f"weighted_{kind}_loss forward + backward B=256 (loss.py)", "ms": med(lfb, 20) * 1e3})
    res["micro"] = micro
    return res


def self_spawn(args):
    """`python bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment): re-execute this very command line
    under torch.distributed.run with N ranks on this node (one process per GPU, rendezvous on 127.0.0.1) and return its exit code.
    The driver's own `python -m torch.distributed.run ... bench.py --gpus N` form arrives with WORLD_SIZE set and is not touched."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"--gpus {args.gpus} without a launcher: spawning {args.gpus} ranks: {' '.join(cmd[1:8])} ...")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    return subprocess.run(cmd, env=env).returncode


def compact(o, sig=5):
    """Floats to `sig` significant digits, recursively (the line the driver parses stays short; the detail file keeps full precision)."""
    if isinstance(o, float):
        return float(f"{o:.{sig}g}") if np.isfinite(o) else None
    if isinstance(o, dict):
        return {k: compact(v, sig) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [compact(v, sig) for v in o]
    return o


LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
             "train_only_images_per_sec", "roofline", "roofline_step", "kernel_rooflines", "cpu_baseline", "comm", "detail")
MAX_LINE_BYTES = 6144


def driver_line(result):
    """The ONE line rank 0 prints: the contract's keys + roofline + cpu_baseline + six kernel rows, everything else by reference to the detail file.
    Trimmed field by field (never truncated mid-JSON) so that it stays under MAX_LINE_BYTES whatever the legs returned."""
    line = {k: result[k] for k in LINE_KEYS if k in result}
    if "roofline" in line:
        keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_of_measured_peak", "traffic", "traffic_unit", "algorithmic_bytes_per_launch", "traffic_source",
                "mfma_busy", "launches_per_step", "avg_launch_us", "ms_per_step_in_this_kernel", "algorithmic_flop_per_step")
        line["roofline"] = {k: line["roofline"][k] for k in keep if k in line["roofline"]}
    if "roofline_step" in line:
        rs = line["roofline_step"]
        line["roofline_step"] = {"achieved_ms": rs["achieved_ms"], "floor_nominal_ms": rs["floor"]["nominal"]["total_ms"],
                                 "floor_measured_peaks_ms": rs["floor"]["measured_on_this_box"]["total_ms"],
                                 "frac_of_nominal_floor": rs["frac_of_nominal_floor"], "frac_of_measured_peak_floor": rs["frac_of_measured_peak_floor"]}
    if "kernel_rooflines" in line:
        line["kernel_rooflines"] = [{k: r[k] for k in ("kernel", "bound", "shape", "ms", "achieved", "peak", "unit", "frac") if r.get(k) is not None}
                                    for r in line["kernel_rooflines"][:7]]
    if "cpu_baseline" in line:
        line["cpu_baseline"] = {k: line["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind", "batch", "sample") if k in line["cpu_baseline"]}
    if "comm" in line:
        c = line["comm"]
        line["comm"] = {k: c[k] for k in ("rccl_ranks", "backend", "reduce_op", "allreduce_ms_per_step_if_serial", "exposed_comm_ms_per_step") if k in c}
        line["comm"]["bucket_MB"] = [r["MB"] for r in c.get("buckets", [])]
        line["comm"]["bucket_bus_GBs"] = [r["bus_GBs"] for r in c.get("buckets", [])]
    s = json.dumps(compact(line))
    for drop in ("comm", "kernel_rooflines", "roofline_step"):           # (cannot happen with the fields above; a guard, not a path)
        if len(s) <= MAX_LINE_BYTES:
            break
        line.pop(drop, None)
        s = json.dumps(compact(line))
    assert len(s) <= MAX_LINE_BYTES, len(s)
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--epoch-len", type=int, default=8, help="steps per bench epoch (one FDS epoch tail each)")
    ap.add_argument("--backend", default=None, choices=[None, "nccl", "gloo"], help="torch.distributed backend (default: nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true", help="TEST ONLY: every rank uses cuda:0 (with --backend gloo), to run the N > 1 "
                    "control flow on a one-GPU box; the throughput it prints is meaningless")
    ap.add_argument("--extras", action="store_true", help="rank 0 also runs tools/bench_extras.py (every conv layer alone, all FDS shapes, float32 modes (exact / split x3 / x2), "
                    " input pipeline, vendor-library step, CPU micro-baselines: minutes) and writes them to the detail file")
    ap.add_argument("--detail", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"), help="where the untrimmed result goes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-rooflines", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child passes (roofline.traffic: the last committed pass)")
    ap.add_argument("--force-collectives", action="store_true", help="N = 1 only: run the loop through the engine's N > 1 branch in a ONE-rank nccl (RCCL) "
                    "group — gradient hooks, bucket all-reduces, FDS statistic merge — to price that machinery on one GPU (`comm` object in the line)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_spawn(args))
    sys.modules.setdefault("bench", sys.modules[__name__])        # tools/ import `bench`: this module, not a second copy of it
    # stdout carries ONE JSON line and nothing else: libraries that write to file descriptor 1 on their own (RCCL prints a five-line version
    # banner there when it initialises) are pointed at stderr for the life of the process; the line goes to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    args.epoch_len = max(1, min(args.epoch_len, args.steps))      # at least one epoch tail inside the timed region
    from dirhip.parallel import init_distributed
    rank, world, local_rank = init_distributed(backend=args.backend)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU (the hot path has no CPU fallback)")
    device = torch.device("cuda", 0 if args.share_gpu else local_rank)
    torch.cuda.set_device(device)
    forced = bool(args.force_collectives and world == 1)
    if forced:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(so.getsockname()[1]))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)

    from dirhip.train_loop import EpochFeatures, epoch_tail, resolve_loss, train_step
    model, engine, optimizer, batches = build(args, device, rank, force_collectives=forced)
    store = EpochFeatures(args.epoch_len * args.batch, 2048, device)
    loss_fn = resolve_loss("l1")
    log("model + data built")
    # set-up, not warm-up: one step + one tail forward (first-use allocations, weight preparation), untimed
    train_step(engine, optimizer, *batches[0], 2, loss_fn)
    epoch_tail(engine, [(batches[0][0], batches[0][1])], 2, store)
    torch.cuda.synchronize(device)
    log("set-up step done")
    _, epoch = run_steps(engine, optimizer, batches, store, args.warmup, args.epoch_len, 3, loss_fn)
    torch.cuda.synchronize(device)
    log("warmup done")
    dt, (loss, epoch) = timed(lambda: run_steps(engine, optimizer, batches, store, args.steps, args.epoch_len, epoch, loss_fn),
                              device, world)
    loss_val = float(loss.item())
    log(f"timed region done: {dt:.2f}s for {args.steps} steps")
    assert np.isfinite(loss_val) and loss_val < 1e6, f"Loss explosion: {loss_val}"
    # same steps without the tail (for the plain-ResNet comparison and the MFMA fraction of the step itself)
    dt_train, _ = timed(lambda: run_steps(engine, optimizer, batches, store, args.steps, args.epoch_len, epoch, loss_fn, with_tail=False),
                        device, world)

    images = args.steps * args.batch * world
    flops = args.steps * args.batch * FLOP_FWD_BWD + args.steps * args.batch * FLOP_FWD   # per GPU: one tail forward per trained batch
    result = {
        "metric": "images/sec ResNet-50+FDS IMDB-WIKI 224x224 (train loop incl. FDS epoch tail)",
        "value": images / dt, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic" if not args.share_gpu else "synthetic (TEST RUN: all ranks share one GPU, number meaningless)",
        "config": {"workload": "BASELINE configs[1]: IMDB-WIKI-DIR ResNet-50 + LDS + FDS (ks=5, sigma=2), bf16 MFMA conv stack, fp32 BatchNorm statistics / FDS tail / "
                               "loss, batch 256 per MI355X, l1, Adam 1e-3; one epoch-tail forward per trained batch + FDS update per bench epoch inside the timed region"
                               + ("" if world == 1 else f"; configs[2] form: {world} ranks, RCCL gradient all-reduce per step + FDS statistic all-reduce per epoch tail"),
                   "per_gpu_batch": args.batch, "global_batch": args.batch * world, "epoch_len_steps": args.epoch_len,
                   "tail_forward_batches_in_timed_region": args.steps, "parallelism": f"dp{world}", "final_loss": loss_val},
        "train_only_images_per_sec": images / dt_train,
        # whole-loop MFMA fraction (24.287 GFLOP per trained image + 8.174 per tail-forward image over the wall clock)
        "roofline_loop": {"bound": "mfma", "kernel": "whole train loop per GPU: ResNet-50 fwd+bwd (+ fwd-only epoch tail)",
                          "achieved": flops / dt / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                          "frac": flops / dt / 1e12 / PEAK_BF16_TFLOPS,
                          "train_only_frac": args.steps * args.batch * FLOP_FWD_BWD / dt_train / 1e12 / PEAK_BF16_TFLOPS},
    }
    result["roofline"] = dict(result["roofline_loop"], traffic=None)      # replaced below by the dominant kernel's when measured
    # ---- measurements that still need every rank (collectives inside the steps): communication report, in-situ kernel times
    if world > 1 or forced:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_extras
        result["comm"] = bench_extras.comm_probe(engine, optimizer, batches, loss_fn, epoch, device, world)
        if forced:
            result["comm"]["note_forced"] = ("ONE-rank nccl group on one GPU (--force-collectives): every collective is the identity; what this line prices is the "
                                             "machinery — 161 post-accumulate hooks, 3 bucket all-reduce launches per step, the FDS statistic merge per epoch tail")
    fam = None
    if not args.no_kernel_rooflines:
        fam = in_situ_breakdown(engine, optimizer, batches, loss_fn, epoch)
        log("in-situ kernel breakdown done")
    if world > 1 or forced:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    # ---- rank 0 alone from here on (the other ranks have left; nothing below communicates)
    del engine, optimizer, batches, store, model
    torch.cuda.empty_cache()
    extras = args.extras and world == 1
    if fam is not None:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_extras
        peaks = bench_extras.measured_peaks(device)
        log("measured peaks done")
        conv_fwd_flop = conv_flops_per_image() * args.batch
        busy = sum(f["us_per_step"] for k, f in fam.items() if k != "rccl_collective")
        ig = fam.get("conv_igemm", {"us_per_step": float("nan"), "launches_per_step": 0})
        alg_flop = 2.0 * conv_fwd_flop                                      # forward + data gradient of the 52 layers
        ach = alg_flop / (ig["us_per_step"] * 1e-6) / 1e12
        # HBM traffic per launch: the PMC counters cannot be read from inside this process, so two separate rocprofv3 --pmc passes
        # (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md) run as child processes over tools/pmc_conv_pass.py — every forward /
        # data-gradient configuration of the 52 layers once, isolated — after the timed region; if rocprofv3 is not usable here,
        # the last committed pass is quoted instead and `traffic` stays null
        traffic, alg_per_launch, traffic_detail = None, None, None
        if not args.no_pmc and world == 1:
            try:
                pm = pmc_traffic(args.batch)
                traffic = pm["traffic_bytes_per_launch"]
                alg_per_launch = pm["algorithmic_bytes_per_step"] / pm["launches_per_step"]
                traffic_detail = {"source": "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes over tools/pmc_conv_pass.py (isolated launches)",
                                  **{k: pm[k] for k in ("launches_per_step", "hbm_read_bytes_per_step", "hbm_write_bytes_per_step", "algorithmic_bytes_per_step")}}
                log("PMC traffic passes done")
            except Exception as e:                                      # noqa: BLE001
                log(f"PMC traffic passes failed ({type(e).__name__}: {e}); quoting the committed pass")
        if traffic_detail is None:
            for name in ("r06_conv_pmc_traffic.json", "r05_conv_pmc_traffic.json", "r04_conv_pmc_traffic.json"):
                tpath = os.path.join(ROOT, "profiles", name)
                if os.path.isfile(tpath):
                    tj = json.load(open(tpath))
                    alg_per_launch = tj.get("algorithmic_bytes_per_step", 0) / max(1, tj.get("launches_per_step", 1))
                    traffic_detail = {"source": f"profiles/{name} (committed pass, not live)", "traffic_bytes_per_launch": tj.get("traffic_bytes_per_launch")}
                    break
        mfma_busy = None
        for name in ("r06_conv_mfma_util.json", "r05_conv_mfma_util.json", "r04_conv_mfma_util.json"):
            mpath = os.path.join(ROOT, "profiles", name)
            if os.path.isfile(mpath):
                mfma_busy = {"source": f"profiles/{name}", "weighted_mfma_util": json.load(open(mpath)).get("weighted_mfma_util")}
                break
        result["roofline"] = {
            "bound": "mfma", "kernel": "conv_igemm_* / conv3x3_patch_* (MFMA implicit GEMM): all forward + data-gradient launches of the 52 conv layers of a training step, in situ",
            "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
            "frac_of_measured_peak": ach / peaks["bf16_mfma_TFs"], "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC)",
            "algorithmic_bytes_per_launch": alg_per_launch, "traffic_source": (traffic_detail or {}).get("source"), "traffic_detail": traffic_detail,
            "mfma_busy": (mfma_busy or {}).get("weighted_mfma_util"), "mfma_busy_source": (mfma_busy or {}).get("source"),
            "launches_per_step": ig["launches_per_step"], "avg_launch_us": ig["us_per_step"] / max(1.0, ig["launches_per_step"]),
            "algorithmic_flop_per_step": alg_flop, "ms_per_step_in_this_kernel": ig["us_per_step"] / 1e3,
            "method": "device time of every conv_igemm* / conv3x3_patch* launch over 4 whole training steps (profiler kernel trace), algorithmic "
                      "FLOPs = 2 x (forward FLOPs of the 52 layers) = forward + data gradient; SURVEY.md §8d"}
        result["step_breakdown_in_situ"] = {"busy_ms_per_step": busy / 1e3, "wall_ms_per_step_train_only": dt_train / args.steps * 1e3,
                                            "families": {k: {"ms_per_step": v["us_per_step"] / 1e3, "launches_per_step": v["launches_per_step"],
                                                             "share_of_busy": v["us_per_step"] / busy} for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["us_per_step"])}}
        kr = []
        if "conv_wgrad" in fam:
            a = conv_fwd_flop / (fam["conv_wgrad"]["us_per_step"] * 1e-6) / 1e12
            kr.append({"kernel": "conv_wgrad_* + reduce (52 layers), in situ", "bound": "mfma", "ms": fam["conv_wgrad"]["us_per_step"] / 1e3,
                       "algorithmic_flop": conv_fwd_flop, "achieved": a, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": a / PEAK_BF16_TFLOPS,
                       "frac_of_measured_peak": a / peaks["bf16_mfma_TFs"]})
        if "batchnorm" in fam:
            # BatchNorm family as launched: forward apply reads y and writes z (2 passes), backward apply reads g, x and writes dx (3)
            # over the 11.11 M BatchNorm-output elements per image in bf16; the backward reduction pass (reads g, x:
            # 2 passes) only where it is still a kernel of this family — the last block's bn3 and the four two-BatchNorm joins
            # (3.11 M elements per image); for the other 43 BatchNorms it runs inside the data-gradient epilogues (conv family)
            alg = (5 * 11.11e6 + 2 * 3.11e6) * args.batch * 2
            a = alg / (fam["batchnorm"]["us_per_step"] * 1e-6) / 1e9
            kr.append({"kernel": "dir_bn_* family, in situ", "bound": "hbm",
                       "ms": fam["batchnorm"]["us_per_step"] / 1e3, "algorithmic_bytes": alg, "achieved": a, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                       "frac": a / PEAK_HBM_GBS, "frac_of_measured_peak": a / peaks["stream_copy_GBs"]})
        tail_row = None
        if "tail" in fam:
            alg = 2 * args.batch * 49 * 2048 * 2 + 3 * args.batch * 2048 * 4
            a = alg / (fam["tail"]["us_per_step"] * 1e-6) / 1e9
            tail_row = {"kernel": "dir_tail_fwd + dir_tail_bwd (pool -> FDS calibrate -> linear), in situ", "bound": "launch", "ms": fam["tail"]["us_per_step"] / 1e3,
                        "algorithmic_bytes": alg, "achieved": a, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": a / PEAK_HBM_GBS}
        result["kernel_rooflines"] = kr + (fds_kernel_rooflines(device, full=extras) if world == 1 else []) + ([tail_row] if tail_row and extras else [])
        for r in result["kernel_rooflines"]:
            if r.get("unit") == "GB/s" and "frac_of_measured_peak" not in r:
                r["frac_of_measured_peak"] = r["achieved"] / peaks["stream_read_GBs"]
        log("kernel rooflines done")
        result["peaks"] = peaks
        # ---- what the training step could cost at best with THIS algorithm (training-mode BatchNorm = a grid-wide reduction between
        # every convolution and its consumer, so no kernel can be fused across it): the sum over its kernels of max(FLOP / MFMA peak,
        # algorithmic bytes / HBM peak), at the nominal peaks and at the peaks measured on this box
        def conv_floor(pf, bw):
            tot = 0.0
            for cin, cout, k, st, h, cnt in RESNET50_CONVS:
                ho = (h + 2 * (k // 2) - k) // st + 1
                flop = 2.0 * args.batch * ho * ho * cout * cin * k * k
                nbytes = (args.batch * h * h * cin + args.batch * ho * ho * cout) * 2
                tot += 3 * cnt * max(flop / pf, nbytes / bw)                 # forward, data gradient, weight gradient
            return tot * 1e3
        bn_bytes = (5 * 11.11e6 + 2 * 3.11e6) * args.batch * 2
        stem_bytes = args.batch * (224 * 224 * 3 * 2 * 2 + 112 * 112 * 64 * 2 * 6 + 56 * 56 * 64 * (2 * 3 + 1 * 3))   # image x2, stem map x6, pooled map / indices x3
        opt_bytes = 23510081 * 4 * 7 + 23454912 * 2 * 2
        floors = {}
        for tag, pf, bw in (("nominal", PEAK_BF16_TFLOPS * 1e12, PEAK_HBM_GBS * 1e9), ("measured_on_this_box", peaks["bf16_mfma_TFs"] * 1e12, peaks["stream_copy_GBs"] * 1e9)):
            parts = {"conv_fwd_dgrad_wgrad_ms": conv_floor(pf, bw), "batchnorm_ms": bn_bytes / bw * 1e3, "stem_ms": stem_bytes / bw * 1e3, "optimizer_ms": opt_bytes / bw * 1e3}
            floors[tag] = dict(parts, total_ms=sum(parts.values()))
        step_ms = dt_train / args.steps * 1e3
        result["roofline_step"] = {"achieved_ms": step_ms, "floor": floors, "frac_of_nominal_floor": floors["nominal"]["total_ms"] / step_ms,
                                   "frac_of_measured_peak_floor": floors["measured_on_this_box"]["total_ms"] / step_ms,
                                   "note": "floor = sum over the step's kernels of max(FLOP / bf16 MFMA peak, algorithmic bytes / HBM peak): per conv layer and "
                                           "direction, BatchNorm family bytes (SURVEY 8d), stem + stem tail, optimizer; batch-statistics BatchNorm forbids fusing across it"}
    if not args.no_cpu_baseline and world == 1:
        result["cpu_baseline"] = cpu_baseline(seconds_budget=24.0, full=extras)
        log("cpu baseline done")
    if extras:
        import bench_extras
        bench_extras.run_all(result, device, args, loss_fn, dt_train)
    # ---- the untrimmed result -> the detail file (gpurun_out/ is merged back by gpurun; --extras runs also refresh the committed copy's source)
    detail_rel = None
    try:
        os.makedirs(os.path.dirname(args.detail), exist_ok=True)
        with open(args.detail, "w") as f:
            json.dump(result, f, indent=1)
        detail_rel = os.path.relpath(args.detail, ROOT)
    except OSError as e:
        log(f"detail file not written: {e}")
    result["detail"] = f"{detail_rel} (this run); committed copy of an --extras run: profiles/r06_bench_detail.json"
    sys.stdout.flush()
    os.write(json_fd, (driver_line(result) + "\n").encode())


if __name__ == "__main__":
    main()
