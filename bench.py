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

Rank 0 prints ONE JSON line. After the timed region, rank 0 at N=1 also measures (none of it inside `value`):
  * `roofline` — the dominant kernel family, the MFMA implicit-GEMM convolution (conv_igemm_*: 52 forward + 49 stride-1 /
    compact data-gradient + 12 parity-class launches per step), IN SITU: per-kernel device times of whole training steps
    from the profiler's kernel trace (the same numbers `rocprofv3 --kernel-trace` of this command reports;
    profiles/rNN_train_step_breakdown.txt), against the algorithmic FLOPs of those launches;
  * `kernel_rooflines` — the other hand-written kernels in situ (weight gradient, BatchNorm family, tail) and the FDS
    kernels at their full-epoch sizes, each with algorithmic bytes / FLOPs from SURVEY.md §8d;
  * `conv_layers` — every conv shape alone (forward, data gradient, weight gradient) with inputs rotated over > 256 MB
    of distinct buffers (nothing Infinity-Cache resident) next to its own roofline max(FLOP / peak, bytes / HBM);
  * `peaks` — STREAM-style copy / read bandwidth and bf16 / f32 MFMA issue rate measured on THIS box (dir_probe_*);
    every fraction is quoted against the nominal peak (MI355X_MICROARCH.md) and against the measured one;
  * `cpu_baseline` — the reference loop's torch-CPU port (oracle/torch_oracle.py, pinned to the live reference) on the
    host cores: the whole loop at B=8, plus the SURVEY §8d micro-baselines (FDS.smooth, update_running_stats, the losses).
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


def comm_probe(engine, optimizer, batches, loss_fn, epoch, device, world):
    """N > 1 observability (rank 0 prints it): ranks and backend, the gradient buckets, each bucket's all-reduce alone (ms and bus
    bandwidth 2 (N - 1) / N x bytes / time — the figure to hold against one xGMI link), and the communication a training step does
    NOT hide behind its backward pass (HIP events on the compute stream: last backward kernel -> last collective done)."""
    from dirhip.train_loop import train_step
    engine.measure_comm = True
    exposed = []
    for i in range(4):
        train_step(engine, optimizer, *batches[i % len(batches)], epoch, loss_fn)
        exposed.append(engine.comm_report().get("exposed_comm_ms_last_step"))
    engine.measure_comm = False
    rep = engine.comm_report()
    rows = []
    for b in engine._buckets:
        t = torch.zeros_like(b.flat)
        for _ in range(2):
            dist.all_reduce(t)
        torch.cuda.synchronize(device)
        dist.barrier()
        t0 = time.perf_counter()
        iters = 5
        for _ in range(iters):
            dist.all_reduce(t)
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / iters
        nbytes = t.numel() * 4
        rows.append({"MB": round(nbytes / 2 ** 20, 2), "allreduce_ms": dt * 1e3, "bus_GBs": 2.0 * (world - 1) / world * nbytes / dt / 1e9})
        del t
    vals = [v for v in exposed[1:] if v is not None]
    return {"rccl_ranks": rep["ranks"], "backend": rep["backend"], "reduce_op": rep["reduce_op"], "buckets": rows,
            "allreduce_ms_per_step_if_serial": sum(r["allreduce_ms"] for r in rows),
            "exposed_comm_ms_per_step": (sum(vals) / len(vals)) if vals else None,
            "grad_copies_per_step": rep["grad_copies"] / max(1, rep["steps"]), "bucket_scale_kernels": rep["bucket_scale_kernels"],
            "note": "exposed = compute-stream time between the last backward kernel and the completion of the last bucket's all-reduce; "
                    "the per-bucket rows are isolated collectives (no overlap with compute)"}


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


def measured_peaks(device):
    """STREAM-style HBM bandwidth and MFMA issue peaks of this box (tools/csrc/dir_probe.hip -> tools/lib/libdir_hip_tools.so: the
    probes are not part of the product library), HIP events on the launch stream."""
    from dirhip import _lib as L
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import toolslib
    lib = toolslib.lib()
    st = L.stream_ptr(device)
    nbytes = 1 << 30
    src = torch.empty(nbytes, dtype=torch.uint8, device=device).random_(0, 255)
    dst = torch.empty_like(src)
    red = torch.empty(8192, dtype=torch.float32, device=device)
    out = {"nominal": {"hbm_GBs": PEAK_HBM_GBS, "bf16_mfma_TFs": PEAK_BF16_TFLOPS, "f32_mfma_TFs": 157.3}}
    ms = event_time_ms(lambda i: L.check(lib.dir_probe_stream_copy(L.ptr(src), L.ptr(dst), nbytes, st), "copy"), 10)
    out["stream_copy_GBs"] = 2 * nbytes / ms / 1e6
    ms = event_time_ms(lambda i: L.check(lib.dir_probe_stream_read(L.ptr(src), L.ptr(red), nbytes, st), "read"), 10)
    out["stream_read_GBs"] = nbytes / ms / 1e6
    ms = event_time_ms(lambda i: L.check(lib.dir_probe_stream_write(L.ptr(dst), nbytes, st), "write"), 10)
    out["stream_write_GBs"] = nbytes / ms / 1e6
    # on-chip re-read rates with the latency covered (4 workgroups per CU, 8 x 16 B in flight per lane): a 2 MB region lives in
    # every XCD's 4 MB L2, a 32 MB one only in the 256 MB Infinity Cache. The convolution K loops move ~12 TB/s from the same
    # levels: between the two, i.e. bound by bytes in flight x latency, not by the L2's bandwidth (HISTORY.md §4)
    for key, region, passes in (("l2_resident_read_GBs", 2 << 20, 16), ("mall_resident_read_GBs", 32 << 20, 1)):
        ms = event_time_ms(lambda i: L.check(lib.dir_probe_l2_read(L.ptr(src), L.ptr(red), region, 1024, passes, 8, st), key), 5)
        out[key] = 1024 * passes * region / ms / 1e6
    del src, dst
    wgs = 256 * 8
    buf = torch.empty(wgs * 256, dtype=torch.float32, device=device)
    fl = ctypes.c_double(0.0)
    for name, fn, iters in (("bf16_mfma_TFs", lib.dir_probe_mfma_bf16, 2000), ("f32_mfma_TFs", lib.dir_probe_mfma_f32, 500)):
        ms = event_time_ms(lambda i: L.check(fn(wgs, iters, L.ptr(buf), ctypes.byref(fl), st), name), 5)
        out[name] = fl.value / ms / 1e9
    return out


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


def conv_layer_probe(device, batch):
    """Every conv shape alone: forward (incl. BatchNorm statistics), data gradient (stride-1: the same kernel on dY;
    3x3 stride-2: four parity-class launches; 1x1 stride-2: the compact 1x1 GEMM on dY) and weight gradient, inputs rotated
    over > 256 MB of distinct buffers. Returns rows [cin, cout, k, stride, H, count, kind, us, roofline_us, launches]."""
    from dirhip import _lib as L
    from dirhip.conv import conv2d_igemm, conv2d_wgrad
    rows = []
    lib = L.lib()

    def bufs(shape, nbytes_pair):
        n = max(2, min(8, int(400e6 // nbytes_pair) + 1))
        return [torch.randn(shape, device=device).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(n)]
    for cin, cout, k, st, h, cnt in RESNET50_CONVS:
        pad = k // 2
        ho = (h + 2 * pad - k) // st + 1
        flop = 2.0 * batch * ho * ho * cout * cin * k * k
        nbytes = (batch * h * h * cin + batch * ho * ho * cout) * 2
        roof_us = max(flop / (PEAK_BF16_TFLOPS * 1e12), nbytes / (PEAK_HBM_GBS * 1e9)) * 1e6
        xs = bufs((batch, cin, h, h), nbytes)
        dys = bufs((batch, cout, ho, ho), nbytes)
        w = (torch.randn(cout, cin, k, k, device=device) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        nb = len(xs)
        ms = event_time_ms(lambda i: conv2d_igemm(xs[i % nb], w, st, pad, want_stats=True), 8, warm=2)
        rows.append([cin, cout, k, st, h, cnt, "fwd", ms * 1e3, roof_us, 1])
        if st == 1:
            wr = (torch.randn(cin, cout, k, k, device=device) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            ms = event_time_ms(lambda i: conv2d_igemm(dys[i % nb], wr, 1, pad), 8, warm=2)
            rows.append([cin, cout, k, st, h, cnt, "dgrad", ms * 1e3, roof_us, 1])
        elif k == 3:
            wf = torch.randn(cout, cin, 3, 3, device=device).contiguous(memory_format=torch.channels_last)
            w16 = torch.empty((cout, cin, 3, 3), dtype=torch.bfloat16, device=device).contiguous(memory_format=torch.channels_last)
            wcls = torch.empty(cin * 9 * cout, dtype=torch.bfloat16, device=device)
            L.check(lib.dir_conv_prep_weights_ex(L.ptr(wf), cout, 3, 3, cin, L.ptr(w16), L.ptr(wcls), 1, L.stream_ptr(device)), "prep")
            dxs = [torch.empty((batch, cin, h, h), dtype=torch.bfloat16, device=device).contiguous(memory_format=torch.channels_last) for _ in range(2)]
            ms = event_time_ms(lambda i: L.check(lib.dir_conv_dgrad_s2(L.ptr(dys[i % nb]), L.ptr(wcls), L.ptr(dxs[i % 2]), batch, ho, ho, cout, cin,
                                                                       L.stream_ptr(device)), "dgrad_s2"), 8, warm=2)
            rows.append([cin, cout, k, st, h, cnt, "dgrad(4 parity classes)", ms * 1e3, roof_us, 4])
        else:
            wr = (torch.randn(cin, cout, 1, 1, device=device) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            ms = event_time_ms(lambda i: conv2d_igemm(dys[i % nb], wr, 1, 0), 8, warm=2)
            cb = (batch * ho * ho * (cin + cout)) * 2
            rows.append([cin, cout, k, st, h, cnt, "dgrad(compact)", ms * 1e3, max(flop / (PEAK_BF16_TFLOPS * 1e12), cb / (PEAK_HBM_GBS * 1e9)) * 1e6, 1])
        ms = event_time_ms(lambda i: conv2d_wgrad(dys[i % nb], xs[i % nb], k, st, pad), 8, warm=2)
        rows.append([cin, cout, k, st, h, cnt, "wgrad", ms * 1e3, roof_us, 2])
        del xs, dys, w
    return rows


def fds_kernel_rooflines(device):
    """Algorithmic bytes / measured duration for the hand-written FDS kernels at their full sizes."""
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
    out.append(row("dir_fds_scatter_stats (grouping + piece sums + combine)", "hbm", f"N={n} C={c} Nb={nb} f32", ms, n * c * 4 + n * 4))
    del feats
    # --- calibrate forward: 2*B*C*4 + T*U*C*4 + B*4, T = 3 tables (m1, scale, m2; scale precomputed per epoch)
    for b in (256, 65536):
        x = torch.randn(b, c, device=device, generator=g)
        labb = torch.as_tensor(long_tail_labels(np.random.default_rng(4), b), device=device)
        m1 = torch.randn(nb, c, device=device, generator=g)
        sc = torch.rand(nb, c, device=device, generator=g) + 0.5
        m2 = torch.randn(nb, c, device=device, generator=g)
        u = int(torch.unique(labb.clamp(max=99)).numel())
        ms = event_time_ms(lambda i: ops.smooth_fwd_(x, labb, 0, 100, m1, sc, m2), 50 if b == 256 else 10)
        out.append(row("dir_fds_smooth_fwd (K1+K5 fused)" if b <= 2048 else "dir_fds_smooth_fwd (K1, K5 with the tables staged in LDS)", "hbm" if b > 2048 else "launch",
                       f"B={b} C={c} U={u} T=3 f32", ms, 2 * b * c * 4 + 3 * u * c * 4 + b * 4))
        bins_b, _ = ops.bin_index(labb, 0, 100)
        dy = torch.randn(b, c, device=device, generator=g)
        ms = event_time_ms(lambda i: ops.calibrate_bwd(dy, bins_b, sc), 50 if b == 256 else 10)
        out.append(row("dir_fds_calibrate_bwd", "hbm" if b > 2048 else "launch", f"B={b} C={c} U={u} f32", ms, 2 * b * c * 4 + u * c * 4 + b * 4))
        del x, dy
    # --- "next" rows: NYUD2 dense map [32,128,114,152] (per-pixel buckets, narrow-row kernels)
    b, c, h, w = 32, 128, 114, 152
    depth = torch.rand(b * h * w, device=device, generator=g) * 9.3 + 0.7
    rows = torch.rand(b * h * w, c, device=device, generator=g)
    bins = ops.bin_scaled(depth, 10.0, 7, 100)
    t1, sc, t2 = (torch.rand(93, c, device=device, generator=g) + 0.5 for _ in range(3))
    ms = event_time_ms(lambda i: ops.calibrate_fwd_(rows, bins, t1, sc, t2), 10)
    out.append(row("dir_fds_calibrate_fwd (narrow rows, NYUD2 dense map)", "hbm", f"[{b},{c},{h},{w}] f32, 93 buckets", ms,
                   2 * rows.numel() * 4 + 3 * 93 * c * 4 + rows.shape[0] * 4))
    fmap = torch.rand(b, c, h, w, device=device, generator=g)
    outm = torch.empty_like(fmap)
    ms = event_time_ms(lambda i: ops.calibrate_nchw(fmap, bins, t1, sc, t2, out=outm), 10)
    out.append(row("dir_fds_calibrate_fwd_nchw (NYUD2 map in its own NCHW layout: 16 KB plane chunks in address order, the channel's table column in LDS)", "hbm", f"[{b},{c},{h},{w}] f32, 93 buckets", ms,
                   2 * fmap.numel() * 4 + 3 * 93 * c * 4 + rows.shape[0] * 4))
    del fmap, outm
    ms = event_time_ms(lambda i: ops.scatter_stats(rows, bins, 93), 5)
    out.append(row("dir_fds_scatter_stats (narrow rows, NYUD2 dense map)", "hbm", f"N={rows.shape[0]} C={c} Nb=93 f32", ms,
                   rows.numel() * 4 + rows.shape[0] * 4))
    del rows
    # --- STS-B-DIR (BASELINE configs[4]; sts-b-dir/fds.py:96-143, util.py:63-73): C = 12000 sentence features, 50 histogram buckets on [0, 5],
    # clip [0.5, 2]: the batch call ([128, 12000], launch bound) and an HBM-resident size, and the epoch statistics at C = 12000
    c, nb = 12000, 50
    edges = torch.tensor(np.histogram(np.array([], np.float32), bins=nb, range=(0., 5.))[1].astype(np.float32), device=device)
    t1, sc, t2 = (torch.rand(nb, c, device=device, generator=g) + 0.5 for _ in range(3))
    for b in (128, 16384):
        lab = torch.rand(b, device=device, generator=g) * 5.0
        bins_b = ops.bin_edges(lab, edges, 0, nb)
        u = int(torch.unique(bins_b).numel())
        x = torch.randn(b, c, device=device, generator=g)
        ms = event_time_ms(lambda i: ops.calibrate_fwd_(x, bins_b, t1, sc, t2), 50 if b == 128 else 10)
        out.append(row("dir_fds_calibrate_fwd (STS-B: C = 12000, 50 buckets)", "hbm" if b > 2048 else "launch", f"B={b} C={c} U={u} T=3 f32", ms,
                       2 * b * c * 4 + 3 * u * c * 4 + b * 4))
        dy = torch.randn(b, c, device=device, generator=g)
        ms = event_time_ms(lambda i: ops.calibrate_bwd(dy, bins_b, sc), 50 if b == 128 else 10)
        out.append(row("dir_fds_calibrate_bwd (STS-B)", "hbm" if b > 2048 else "launch", f"B={b} C={c} U={u} f32", ms, 2 * b * c * 4 + u * c * 4 + b * 4))
        if b > 2048:
            ms = event_time_ms(lambda i: ops.scatter_stats(x, bins_b, nb), 10)
            out.append(row("dir_fds_scatter_stats (STS-B: C = 12000)", "hbm", f"N={b} C={c} Nb={nb} f32", ms, b * c * 4 + b * 4))
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


def input_pipeline_probe(device, consumer_img_s, batch=256, seconds=12.0, n_gpus_target=8, only_end_to_end=False, e2e_kw=None):
    """SURVEY §8f-4 measured: can the real-file input pipeline feed the GPU loop? Synthetic JPEG files on local disk ->
    dirhip.datasets.IMDBWIKI (PIL decode + bilinear Resize to 224, host) in DataLoader workers -> (a) raw uint8 batches + ONE
    dir_augment_u8 launch on the GPU (train.py --gpu_augment; uint8 over PCIe) or (b) the host float transform chain of the
    reference (datasets.py:38-53) -> device. Reports images/s of each against the consumer (the timed loop's images/s)."""
    import shutil
    import tempfile
    import pandas as pd
    from PIL import Image
    from torch.utils.data import DataLoader, RandomSampler
    from dirhip.datasets import IMDBWIKI, DeviceAugment
    tmp = tempfile.mkdtemp(dir="/tmp", prefix="dir_jpeg_")
    try:
        rng = np.random.default_rng(0)
        n_files, side = 192, 320
        yy, xx = np.mgrid[0:side, 0:side].astype(np.float32) / side
        rows = []
        for i in range(n_files):                          # smooth colour fields + texture noise: JPEGs of photographic entropy (~25-40 KB)
            base = np.stack([np.sin(6.3 * (xx * rng.uniform(0.5, 3) + yy * rng.uniform(0.5, 3)) + rng.uniform(0, 6)) for _ in range(3)], -1)
            arr = np.clip(128 + 90 * base + rng.normal(0, 12, (side, side, 3)), 0, 255).astype(np.uint8)
            Image.fromarray(arr).save(os.path.join(tmp, f"f{i}.jpg"), quality=90)
            rows.append({"path": f"f{i}.jpg", "age": float(rng.integers(1, 100)), "split": "train"})
        df = pd.DataFrame(rows)
        kb = sum(os.path.getsize(os.path.join(tmp, r["path"])) for r in rows) / n_files / 1024
        workers = max(1, min((os.cpu_count() or 2) - 2, 32))          # (32 decode-only workers = 60 k img/s of supply; more only lengthens start-up)
        out = {"files": f"{n_files} synthetic {side}x{side} JPEGs (quality 90, {kb:.0f} KB each) on local disk, sampled with replacement",
               "workers": workers, "batch": batch, "consumer_images_per_sec": consumer_img_s}

        def rate(raw, budget, batch=batch):
            from dirhip.datasets import DeviceResize, ragged_collate
            ds = IMDBWIKI(df, tmp, img_size=224, split="train", raw=raw)
            n_img = 5000 * batch
            # decode-only batches are ragged and 4x larger (file-size uint8: 78 MB per 256 images of 320 x 320): the loader's pinned pool would
            # be workers x prefetch x 78 MB = 15 GB, whose allocation alone takes tens of seconds — that leg hands over pageable batches
            # (H2D ~8 ms per batch) with two batches prefetched per worker
            decoded = raw == "decoded"
            dl = DataLoader(ds, batch_size=batch, sampler=RandomSampler(ds, replacement=True, num_samples=n_img), num_workers=workers,
                            pin_memory=not decoded, drop_last=True, prefetch_factor=2 if decoded else 4, persistent_workers=False,
                            collate_fn=ragged_collate if decoded else None)
            aug = DeviceAugment(224, train=True, dtype=torch.bfloat16) if raw else None
            rz = DeviceResize(224, device) if decoded else None
            it = iter(dl)
            for _ in range(3):                                # worker start-up + first batches (through the device path once: first-use costs)
                b = next(it)
                if rz is not None:
                    aug(rz(b[0], b[1]))
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            n = 0
            t_dev = 0.0
            while time.perf_counter() - t0 < budget:
                b = next(it)
                t1 = time.perf_counter()
                if rz is not None:
                    img, lab, w = rz(b[0], b[1]), b[2], b[3]
                else:
                    img, lab, w = b[0].to(device, non_blocking=True), b[1], b[2]
                x = aug(img) if raw else img.contiguous(memory_format=torch.channels_last)
                lab.to(device, non_blocking=True); w.to(device, non_blocking=True)
                torch.cuda.synchronize(device)
                t_dev += time.perf_counter() - t1
                n += 1
            dt = time.perf_counter() - t0
            del it, dl
            return n * batch / dt, t_dev / max(1, n) * 1e3, tuple(x.shape), str(x.dtype)
        if only_end_to_end:                                    # (tools/probe_input_pipeline.py e2e: iterate on that leg alone)
            out["end_to_end"] = end_to_end_from_files(device, df, tmp, workers, batch, consumer_img_s, **(e2e_kw or {}))
            return out
        r, ms, shp, dt_ = rate(True, seconds * 0.4)
        out["uint8_files_gpu_augment"] = {"images_per_sec": r, "h2d_plus_dir_augment_u8_ms_per_batch": ms, "network_input": f"{shp} {dt_} channels_last",
                                          "keeps_up_with_consumer": bool(r >= consumer_img_s)}
        r, ms, shp, dt_ = rate("decoded", seconds * 0.4)
        out["decode_only_workers_gpu_resize_augment"] = {"images_per_sec": r, "h2d_plus_dir_resize_u8_plus_dir_augment_u8_ms_per_batch": ms,
                                                         "network_input": f"{shp} {dt_} channels_last", "keeps_up_with_consumer": bool(r >= consumer_img_s),
                                                         "what": "workers: PIL decode only (file-size uint8, ragged batch); GPU: dir_resize_u8 (Pillow bilinear, bit-exact) + dir_augment_u8"}
        # ---- end to end: TRAIN from the files. The real DataLoader (decode-only workers, ragged batches) -> DeviceResize ->
        # DeviceAugment -> train_step, and the epoch-tail forward from a second pass over the loader (as train.py:269-281 re-reads the training
        # set), one tail batch per trained batch like the headline loop: images/s next to the synthetic `value`, plus the host time spent
        # blocked in the loader (a stall only matters once it exceeds the slack the device-bound loop leaves the host).
        try:
            out["end_to_end"] = end_to_end_from_files(device, df, tmp, workers, batch, consumer_img_s)
        except Exception as e:                                          # noqa: BLE001
            out["end_to_end"] = {"error": f"{type(e).__name__}: {e}"}
        # the reference's own host transform chain (float32 CHW out of __getitem__), per core, in this process: decode + Resize +
        # pad / crop / flip + ToTensor + Normalize
        ds_f = IMDBWIKI(df, tmp, img_size=224, split="train")
        ds_f[0]
        t0 = time.perf_counter()
        for i in range(96):
            ds_f[i]
        per_core = 96 / (time.perf_counter() - t0)
        ds_r = IMDBWIKI(df, tmp, img_size=224, split="train", raw=True)
        t0 = time.perf_counter()
        for i in range(96):
            ds_r[i]
        per_core_raw = 96 / (time.perf_counter() - t0)
        ds_d = IMDBWIKI(df, tmp, img_size=224, split="train", raw="decoded")
        t0 = time.perf_counter()
        for i in range(96):
            ds_d[i]
        per_core_dec = 96 / (time.perf_counter() - t0)
        out["per_core_images_per_sec"] = {"decode_only_uint8": per_core_dec, "decode_resize_uint8": per_core_raw, "decode_resize_host_float_chain": per_core,
                                          "cores_needed_for_consumer_uint8": consumer_img_s / per_core_raw,
                                          "cores_needed_for_consumer_decode_only": consumer_img_s / per_core_dec}
        host_cores = os.cpu_count() or 1
        out["host_cores"] = host_cores
        out["cores_needed_for_8_gpus"] = {"decode_only_gpu_resize": n_gpus_target * consumer_img_s / per_core_dec,
                                          "host_decode_plus_resize_uint8": n_gpus_target * consumer_img_s / per_core_raw,
                                          "reference_host_float_chain": n_gpus_target * consumer_img_s / per_core,
                                          "consumer_images_per_sec_per_gpu": consumer_img_s, "gpus": n_gpus_target,
                                          "verdict": "host-bound at 8 GPUs on this box" if n_gpus_target * consumer_img_s / per_core_dec > host_cores
                                                     else "the box's cores can feed 8 GPUs (decode-only workers, Resize + augmentation on the GPUs)"}
        out["note"] = ("decode is host PIL in loader workers (no GPU JPEG decoder in this image), Resize either there or on the GPU; the two loader rates are the "
                       "loaders' own; `end_to_end` trains from the files (loader, device transforms and the training loop running concurrently); `value` of this "
                       "bench uses HBM-resident synthetic batches")
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def end_to_end_from_files(device, df, data_dir, workers, batch, synthetic_img_s, steps=48, epoch_len=8, cold_steps=24, switch_interval=None, depth=2,
                          pinned=False, diagnose=None):
    """datasets.py:38-53 + train.py:246-250, 269-281 with the files as the source: the product's --gpu_resize --gpu_cache configuration
    (train_main.py). Two measurements of the same loop (train_step per batch + one epoch-tail forward per trained batch, like `value`):
    COLD — every batch of both passes comes through the loader (decode-only workers, ragged pageable batches, DevicePrefetcher: H2D,
    dir_resize_u8, dir_augment_u8 on a side stream) and its resized bytes are stored in the HBM cache; CACHED — what every pass after a
    sample's first one costs: gather from datasets.DeviceImageCache + a fresh dir_augment_u8 draw, no loader."""
    from torch.utils.data import DataLoader, Dataset, RandomSampler
    from dirhip.datasets import IMDBWIKI, DeviceAugment, DeviceImageCache, DevicePrefetcher, DeviceResize, PinnedStager, ragged_collate
    from dirhip.train_loop import EpochFeatures, epoch_tail, resolve_loss, train_step

    class A:
        pass
    a = A()
    a.batch, a.epoch_len, a.gpus = batch, 1, 1
    model, engine, optimizer, _ = build(a, device, 0)
    loss_fn = resolve_loss("l1")
    store = EpochFeatures(epoch_len * batch, 2048, device)
    bases = {raw: IMDBWIKI(df, data_dir, img_size=224, split="train", reweight="sqrt_inv", lds=True, lds_kernel="gaussian", lds_ks=5, lds_sigma=2, raw=raw)
             for raw in ("decoded", True)}

    class Indexed(Dataset):                                          # (+ the sample index: the key of the HBM cache, as train_main._ShardSubset)
        def __init__(self, base):
            self.base = base

        def __len__(self):
            return len(self.base)

        def __getitem__(self, i):
            return tuple(self.base[i]) + (int(i),)
    ds = Indexed(bases["decoded"])
    n_img = (2 * (cold_steps + 2 * epoch_len) + 8) * batch

    def loader(raw="decoded"):
        d = Indexed(bases[raw])
        if raw == "decoded":
            return iter(DataLoader(d, batch_size=batch, sampler=RandomSampler(d, replacement=True, num_samples=n_img), num_workers=workers, pin_memory=False,
                                   drop_last=True, prefetch_factor=2, persistent_workers=False, collate_fn=ragged_collate))
        return iter(DataLoader(d, batch_size=batch, sampler=RandomSampler(d, replacement=True, num_samples=n_img), num_workers=workers, pin_memory=True,
                               drop_last=True, prefetch_factor=4, persistent_workers=False))
    aug = DeviceAugment(224, train=True, dtype=torch.bfloat16)
    cache = DeviceImageCache(len(ds), 224, device)
    stall = [0.0, 0]
    fixed = [None]

    def device_half_of(rz):
        if diagnose == "host_only":                               # (diagnosis: loader workers + the producer threads, nothing touches the GPU)
            return lambda b: (b[0][:16], b[2], b[3])

        def half(b):
            if rz is None:                                        # workers decoded AND resized: fixed-size pinned uint8 batches
                u8, y, w, idx = b[0].to(device, non_blocking=True), b[1].to(device, non_blocking=True), b[2].to(device, non_blocking=True), b[3]
            else:
                u8, y, w, idx = rz(b[0], b[1]), b[2].to(device, non_blocking=True), b[3].to(device, non_blocking=True), b[4]
            cache.put(idx, u8, y, w)
            return aug(u8), y, w
        return half

    def fetch(it):
        t0 = time.perf_counter()
        b = next(it)
        stall[0] += time.perf_counter() - t0
        stall[1] += 1
        if diagnose in ("fixed_batch", "host_only"):           # (diagnosis: the pipeline runs, the loop trains on one resident batch)
            if fixed[0] is None:
                g = torch.Generator(device=device).manual_seed(7)
                fixed[0] = (torch.randn(batch, 3, 224, 224, device=device, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last),
                            torch.full((batch, 1), 30.0, device=device), torch.ones(batch, 1, device=device))
            return fixed[0]
        return b

    def run(n, it_train, it_tail, epoch):
        for s0 in range(0, n, epoch_len):
            k = min(epoch_len, n - s0)
            for _ in range(k):
                x, y, w = fetch(it_train)
                train_step(engine, optimizer, x, y, w, epoch, loss_fn)
            epoch_tail(engine, (fetch(it_tail)[:2] for _ in range(k)), epoch, store)
            epoch += 1
        return epoch

    def timed_run(n, it_a, it_b, epoch):
        torch.cuda.synchronize(device)
        stall[0], stall[1] = 0.0, 0
        t0 = time.perf_counter()
        epoch = run(n, it_a, it_b, epoch)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        return epoch, {"images_per_sec": n * batch / dt, "ms_per_step": dt / n * 1e3, "steps": n, "tail_forward_batches": n,
                       "ratio_to_synthetic": n * batch / dt / synthetic_img_s, "blocked_waiting_for_a_ready_batch_ms_per_fetch": stall[0] / max(1, stall[1]) * 1e3}
    old_si = sys.getswitchinterval()
    if switch_interval:
        sys.setswitchinterval(switch_interval)
    what = {"decoded": "JPEG files -> DataLoader workers (PIL decode only, ragged pageable batches of ~78 MB) -> datasets.DevicePrefetcher (thread + side stream: H2D -> "
                       "dir_resize_u8 -> store in the HBM cache -> dir_augment_u8, bf16 NHWC) -> train_step; the epoch-tail forward reads a second pass of the loader: two "
                       "decoded batches per trained batch. Bound by the loaders' hand-over of the ragged batches through shared memory; on the training stream instead of the "
                       "prefetcher the same device half measured 4 814 img/s (a pageable copy is stream-ordered and blocks the host behind the queued step)",
            True: "the same loop with the Resize in the workers (train.py --gpu_augment): half the bytes per batch, fixed-size pinned uint8 batches, 3.5 x the host work per image"}
    res = {"batch": batch, "synthetic_images_per_sec": synthetic_img_s, "prefetch_depth": depth, "pinned_staging": bool(pinned), **({"diagnose": diagnose} if diagnose else {})}
    epoch = 2
    for raw, tag in (("decoded", "cold_decode_only_workers"), (True, "cold_decode_and_resize_in_workers")):
        if diagnose and raw is True:
            continue
        pf_a, pf_b = (DevicePrefetcher(loader(raw), device, device_half_of(DeviceResize(224, device, stager=PinnedStager() if pinned else None) if raw == "decoded" else None),
                                       depth=depth) for _ in range(2))
        it_a, it_b = iter(pf_a), iter(pf_b)
        epoch = run(epoch_len, it_a, it_b, epoch)                          # worker start-up, first-use costs of the device path, one epoch tail
        epoch, cold = timed_run(cold_steps, it_a, it_b, epoch)
        pf_a.close(); pf_b.close()
        del it_a, it_b, pf_a, pf_b
        res[tag] = dict(cold, workers_per_loader=workers, loaders=2, what=what[raw])
    sys.setswitchinterval(old_si)
    cold = res["cold_decode_only_workers"]
    if not diagnose and cache.covers(np.arange(len(ds))):
        gen = torch.Generator().manual_seed(11)
        n_warm = epoch_len
        idx_all = torch.randint(0, len(ds), (2 * (steps + n_warm) * batch,), generator=gen)
        half = (steps + n_warm) * batch
        it_a = iter(cache.batches(idx_all[:half], batch, aug, shuffle=False))
        it_b = iter(cache.batches(idx_all[half:], batch, aug, shuffle=False))
        epoch = run(n_warm, it_a, it_b, epoch)
        epoch, hot = timed_run(steps, it_a, it_b, epoch)
        res["cached_every_later_pass"] = dict(hot, what="datasets.DeviceImageCache: resized uint8 images resident in HBM (28.8 GB for IMDB-WIKI's 191 509 training images at 224; "
                                                       f"here the bench's {len(ds)} distinct files) -> index_select -> dir_augment_u8 with a fresh draw -> train_step, and the same "
                                                       "for the epoch-tail forward: what the feature pass of every epoch and every epoch after the first cost — no decode, no loader")
        res["images_per_sec"] = hot["images_per_sec"]
        res["ratio_to_synthetic"] = hot["ratio_to_synthetic"]
    else:
        res["images_per_sec"] = cold["images_per_sec"]
        res["ratio_to_synthetic"] = cold["ratio_to_synthetic"]
    del engine, optimizer, model, store, cache
    torch.cuda.empty_cache()
    return res


def library_baseline(batch, timeout_s=240):
    """A second measured baseline: the reference's ResNet-50 as plain torch modules on the vendor library (MIOpen) under
    torch.autocast(bfloat16), same GPU, same batch — tools/library_resnet.py in a CHILD process (MIOpen's kernel search takes ~70 s on a fresh
    box; nothing of it is loaded into this process, whose training steps are asserted library-free by the tests)."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "tools", "library_resnet.py"), str(batch), "bf16"]
    t0 = time.perf_counter()
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
    steps = [float(l.split(":")[-1].split("ms")[0]) for l in p.stdout.splitlines() if l.startswith("amp=torch.bfloat16") and " step " in l]
    if p.returncode != 0 or len(steps) < 3:
        return {"error": (p.stderr or p.stdout)[-400:]}
    ms = min(steps[1:])
    return {"what": "imdb-wiki-dir/resnet.py as plain torch modules on the vendor library (MIOpen conv / BatchNorm), torch.autocast(bfloat16), torch.optim.Adam, "
                    "channels_last, L1 loss on the prediction; no FDS, no epoch-tail forward; child process",
            "batch": batch, "ms_per_train_step": ms, "images_per_sec": batch / ms * 1e3, "first_step_s": steps[0] / 1e3, "wall_s": time.perf_counter() - t0}


def cpu_baseline(seconds_budget=20.0):
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


def float32_mode_probe(device, args, loss_fn):
    """The parity-exact configuration next to the benchmarked one: the SAME loop with `amp_dtype=None` — the whole
    network on the exact-float32 MFMA kernels (v_mfma_f32_32x32x2_f32, peak 157 TFLOP/s = 1/16 of bf16), the mode that meets
    north_star's 1e-5 loss bar — timed over a few steps, and its step-0 loss at B=256 against the reference's own float32 CPU run
    (tests/golden/step0_b256.npz, written by tests/golden/gen_golden_r3.py from the reference; inputs regenerated from its seeds)."""
    from dirhip import resnet as R
    from dirhip.loss import weighted_l1_loss
    from dirhip.optim import Adam
    from dirhip.parallel import DataParallelEngine
    from dirhip.train_loop import EpochFeatures
    out = {"amp_dtype": None, "kernels": "dir_conv_f32_* (exact float32 MFMA), same fused autograd graph as the bf16 path"}
    gpath = os.path.join(ROOT, "tests", "golden", "step0_b256.npz")
    if os.path.isfile(gpath):
        g = np.load(gpath, allow_pickle=False)
        cfg = json.loads(str(g["config"]))
        lt = lambda rng, n: np.clip(np.round(np.abs(rng.normal(0, 18, n)) + 20), 0, 120).astype(np.float32)      # noqa: E731
        x = torch.randn(cfg["batch"], 3, 224, 224, generator=torch.Generator().manual_seed(cfg["seed_x"]))
        rng = np.random.default_rng(cfg["seed_lab"])
        y = torch.tensor(lt(rng, cfg["batch"])).view(-1, 1)
        w = torch.tensor(rng.uniform(0.5, 1.5, cfg["batch"]).astype(np.float32)).view(-1, 1)
        assert np.array_equal(y.numpy(), g["in_labels"]) and np.array_equal(w.numpy(), g["in_weights"])
        torch.manual_seed(cfg["seed_model"])
        model = R.resnet50(fds=True, bucket_num=cfg["bucket_num"], bucket_start=cfg["bucket_start"], start_update=cfg["start_update"],
                           start_smooth=cfg["start_smooth"], kernel=cfg["kernel"], ks=cfg["ks"], sigma=cfg["sigma"], momentum=cfg["momentum"]).to(device)
        eng = DataParallelEngine(model, amp_dtype=None, channels_last=True)
        eng.train()
        for ep in range(2):
            rr = np.random.default_rng(cfg["seed_fds"] + ep)
            lab = lt(rr, cfg["n_fds"])
            feats = (np.abs(rr.normal(0, 1, (cfg["n_fds"], 2048))) * 0.5 + 0.01 * lab[:, None]).astype(np.float32)
            model.FDS.update_last_epoch_stats(ep)
            model.FDS.update_running_stats(torch.tensor(feats).to(device), torch.tensor(lab).to(device), ep)
        pred, _ = eng(x.to(device), y.to(device), cfg["epoch"])
        loss = float(weighted_l1_loss(pred, y.to(device), w.to(device)).item())
        ref = float(g["ref_loss"])
        out["step0_loss"] = loss
        out["step0_loss_reference_float32_cpu"] = ref
        out["loss_rel_err_vs_golden"] = abs(loss - ref) / abs(ref)
        out["golden"] = f"tests/golden/step0_b256.npz (B={cfg['batch']}, epoch {cfg['epoch']}, FDS live; the reference's own modules on the CPU)"
        del model, eng, pred
    else:
        out["loss_rel_err_vs_golden"] = None
    # ---- throughput of the same loop (train steps + one tail forward per trained batch) in float32 mode
    saved = (args.epoch_len,)
    steps, epoch_len = 4, 2
    args.epoch_len = epoch_len
    try:
        model, engine, optimizer, batches = build(args, device, 0, amp_dtype=None)
    finally:
        args.epoch_len, = saved
    store = EpochFeatures(epoch_len * args.batch, 2048, device)
    run_steps(engine, optimizer, batches, store, 2, epoch_len, 2, loss_fn)           # set-up + warm-up
    dt, (loss, _) = timed(lambda: run_steps(engine, optimizer, batches, store, steps, epoch_len, 3, loss_fn), device, 1)
    dt_train, _ = timed(lambda: run_steps(engine, optimizer, batches, store, steps, epoch_len, 5, loss_fn, with_tail=False), device, 1)
    assert np.isfinite(float(loss.item()))
    out.update({"images_per_sec": steps * args.batch / dt, "ms_per_step": dt / steps * 1e3, "train_only_images_per_sec": steps * args.batch / dt_train,
                "train_only_ms_per_step": dt_train / steps * 1e3, "steps": steps, "batch": args.batch,
                "achieved_TFLOPs_train_only": steps * args.batch * FLOP_FWD_BWD / dt_train / 1e12, "peak_f32_mfma_TFLOPs": 157.3,
                "frac_of_f32_mfma_peak_train_only": steps * args.batch * FLOP_FWD_BWD / dt_train / 1e12 / 157.3})
    del model, engine, optimizer, batches, store
    torch.cuda.empty_cache()
    return out


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-library-baseline", action="store_true", help="skip the vendor-library (MIOpen, torch.autocast) train step of the same network in a child process (~80 s)")
    ap.add_argument("--no-input-pipeline", action="store_true", help="skip the real-file input pipeline probe (synthetic JPEGs through the DataLoader)")
    ap.add_argument("--no-kernel-rooflines", action="store_true")
    ap.add_argument("--no-float32-mode", action="store_true", help="skip the float32 (parity-exact) mode leg")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child passes (roofline.traffic stays null)")
    ap.add_argument("--force-collectives", action="store_true", help="N = 1 only: run the loop through the engine's N > 1 branch in a ONE-rank nccl (RCCL) "
                    "group — gradient hooks, bucket all-reduces, FDS statistic merge — to price that machinery on one GPU (`comm` object in the line)")
    ap.add_argument("--full-probes", action="store_true", help="N > 1 only: rank 0 also runs the per-layer / FDS-kernel / input-pipeline probes "
                    "that the N = 1 line carries (by default an N > 1 line carries roofline, step breakdown, peaks, comm, cpu_baseline)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_spawn(args))
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
    n_tails = -(-args.steps // args.epoch_len)
    flops = args.steps * args.batch * FLOP_FWD_BWD + args.steps * args.batch * FLOP_FWD   # per GPU: one tail forward per trained batch
    result = {
        "metric": "images/sec ResNet-50+FDS IMDB-WIKI 224x224 (train loop incl. FDS epoch tail)",
        "value": images / dt, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic" if not args.share_gpu else "synthetic (TEST RUN: all ranks share one GPU, number meaningless)",
        "config": {"workload": "BASELINE configs[1]: IMDB-WIKI-DIR ResNet-50 + LDS + FDS (ks=5, sigma=2), bf16 conv stack (own MFMA kernels: stem, "
                               "implicit-GEMM fwd/dgrad/wgrad) + fused HIP BatchNorm / join / pool nodes, fp32 fused pool-FDS-linear tail + loss, "
                               "batch=256 per MI355X, l1 loss, Adam 1e-3" + ("" if world == 1 else f"; BASELINE configs[2] form: {world} ranks, RCCL "
                               "gradient all-reduce per step + FDS statistic all-reduce per epoch tail"),
                   "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                   "epoch_len_steps": args.epoch_len, "epoch_tails_in_timed_region": n_tails, "tail_forward_batches_in_timed_region": args.steps,
                   "tail_batches_per_trained_batch": 1.0, "parallelism": f"dp{world}", "final_loss": loss_val},
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
        result["comm"] = comm_probe(engine, optimizer, batches, loss_fn, epoch, device, world)
        if forced:
            result["comm"]["note_forced"] = ("ONE-rank nccl group on one GPU (--force-collectives): every collective is the identity; what this line prices is the "
                                             "machinery — 161 post-accumulate hooks, 3 bucket all-reduce launches per step, the FDS statistic merge per epoch tail — "
                                             "against the same loop without it (the default N = 1 line)")
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
    full = world == 1 or args.full_probes
    if fam is not None:
        del engine, optimizer, batches, store
        torch.cuda.empty_cache()
        peaks = measured_peaks(device)
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
        traffic, traffic_note = None, None
        if not args.no_pmc and full:
            try:
                pm = pmc_traffic(args.batch)
                traffic = pm["traffic_bytes_per_launch"]
                traffic_note = {"source": "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes over tools/pmc_conv_pass.py (isolated launches, "
                                          "operands of the fused epilogues not included)", **{k: pm[k] for k in (
                                              "launches_per_step", "hbm_read_bytes_per_step", "hbm_write_bytes_per_step", "algorithmic_bytes_per_step")},
                                "algorithmic_bytes_per_launch": pm["algorithmic_bytes_per_step"] / pm["launches_per_step"]}
                log("PMC traffic passes done")
            except Exception as e:                                      # noqa: BLE001
                log(f"PMC traffic passes failed ({type(e).__name__}: {e}); quoting the committed pass")
        if traffic_note is None:
            for name in ("r05_conv_pmc_traffic.json", "r04_conv_pmc_traffic.json", "r02_conv_pmc_traffic.json", "r01_conv_pmc_traffic.json"):
                tpath = os.path.join(ROOT, "profiles", name)
                if os.path.isfile(tpath):
                    tj = json.load(open(tpath))
                    traffic_note = {"source": f"profiles/{name}", "traffic_bytes_per_launch": tj.get("traffic_bytes_per_launch"),
                                    "algorithmic_bytes_per_launch": tj.get("algorithmic_bytes_per_step", 0) / max(1, tj.get("launches_per_step", 1)),
                                    "launches_per_step_in_that_pass": tj.get("launches_per_step")}
                    break
        mfma_busy = None
        for name in ("r05_conv_mfma_util.json", "r04_conv_mfma_util.json", "r02_conv_mfma_util.json"):
            mpath = os.path.join(ROOT, "profiles", name)
            if os.path.isfile(mpath):
                mj = json.load(open(mpath))
                mfma_busy = {"source": f"profiles/{name} (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE, isolated launches)",
                             **{k: v for k, v in mj.items() if not isinstance(v, (list, dict))}}
                break
        result["roofline"] = {
            "bound": "mfma", "kernel": "conv_igemm_kernel / conv_igemm_dma_kernel / conv_igemm_big_kernel / conv3x3_patch_kernel (hand-written MFMA implicit "
                                       "GEMM): every forward and data-gradient launch of the 52 conv layers of one training step, in situ; their store "
                                       "loops also carry the shortcut-gradient adds, ReLU masks and 43 of the 52 BatchNorm backward reductions",
            "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
            "frac_of_measured_peak": ach / peaks["bf16_mfma_TFs"], "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_detail": traffic_note,
            "mfma_busy": mfma_busy,
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
            kr.append({"kernel": "conv_wgrad_* kernels + reduce (weight gradients of the 52 layers), in situ", "bound": "mfma", "ms": fam["conv_wgrad"]["us_per_step"] / 1e3,
                       "algorithmic_flop": conv_fwd_flop, "achieved": a, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": a / PEAK_BF16_TFLOPS,
                       "frac_of_measured_peak": a / peaks["bf16_mfma_TFs"]})
        if "batchnorm" in fam:
            # BatchNorm family as launched: forward apply reads y and writes z (2 passes), backward apply reads g, x and writes dx (3)
            # over the 11.11 M BatchNorm-output elements per image in bf16; the backward reduction pass (reads g, x:
            # 2 passes) only where it is still a kernel of this family — the last block's bn3 and the four two-BatchNorm joins
            # (3.11 M elements per image); for the other 43 BatchNorms it runs inside the data-gradient epilogues (conv family)
            alg = (5 * 11.11e6 + 2 * 3.11e6) * args.batch * 2
            a = alg / (fam["batchnorm"]["us_per_step"] * 1e-6) / 1e9
            kr.append({"kernel": "dir_bn_* family (apply / join / backward apply + the 5 remaining reduction passes (last bn3, one per join pair) + finalize), in situ", "bound": "hbm",
                       "ms": fam["batchnorm"]["us_per_step"] / 1e3, "algorithmic_bytes": alg, "achieved": a, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                       "frac": a / PEAK_HBM_GBS, "frac_of_measured_peak": a / peaks["stream_copy_GBs"]})
        if "tail" in fam:
            alg = 2 * args.batch * 49 * 2048 * 2 + 3 * args.batch * 2048 * 4
            a = alg / (fam["tail"]["us_per_step"] * 1e-6) / 1e9
            kr.append({"kernel": "dir_tail_fwd + dir_tail_bwd (pool -> FDS calibrate -> linear), in situ", "bound": "hbm", "ms": fam["tail"]["us_per_step"] / 1e3,
                       "algorithmic_bytes": alg, "achieved": a, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": a / PEAK_HBM_GBS})
        result["kernel_rooflines"] = kr + (fds_kernel_rooflines(device) if full else [])
        for r in result["kernel_rooflines"]:
            if r.get("unit") == "GB/s" and "frac_of_measured_peak" not in r:
                r["frac_of_measured_peak"] = r["achieved"] / peaks["stream_read_GBs"]
        log("kernel rooflines done")
        result["peaks"] = peaks
        if full:
            rows = conv_layer_probe(device, args.batch)
            result["conv_layers"] = {"columns": ["cin", "cout", "k", "stride", "H", "count", "kind", "us", "roofline_us", "launches"],
                                     "rows": [[*r[:7], round(r[7], 1), round(r[8], 1), r[9]] for r in rows],
                                     "sum_ms": {kind: sum(r[7] * r[5] for r in rows if r[6].startswith(kind)) / 1e3 for kind in ("fwd", "dgrad", "wgrad")},
                                     "sum_roofline_ms": {kind: sum(r[8] * r[5] for r in rows if r[6].startswith(kind)) / 1e3 for kind in ("fwd", "dgrad", "wgrad")},
                                     "note": "isolated launches, inputs rotated over > 256 MB of distinct buffers; roofline_us = max(FLOP / 2.5 PF, bytes / 8 TB/s)"}
            log("conv layer probe done")
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
    else:
        del engine, optimizer, batches, store
        torch.cuda.empty_cache()
    if not args.no_float32_mode:
        try:
            result["float32_mode"] = float32_mode_probe(device, args, loss_fn)
            log("float32 (parity-exact) mode leg done")
        except Exception as e:                                          # noqa: BLE001  (a measurement, never a reason to lose the bench line)
            result["float32_mode"] = {"error": f"{type(e).__name__}: {e}"}
    if full and not args.no_input_pipeline:
        try:
            result["input_pipeline"] = input_pipeline_probe(device, result["value"] / world, args.batch, n_gpus_target=8)
            log("input pipeline probe done")
        except Exception as e:                                          # noqa: BLE001  (a measurement, never a reason to lose the bench line)
            result["input_pipeline"] = {"error": f"{type(e).__name__}: {e}"}
    if full and not args.no_library_baseline:
        try:
            torch.cuda.empty_cache()
            lb = library_baseline(args.batch)
            if "ms_per_train_step" in lb:
                lb["this_build_train_step_ms"] = dt_train / args.steps * 1e3
                lb["speedup_of_the_train_step"] = lb["ms_per_train_step"] / lb["this_build_train_step_ms"]
            result["library_baseline"] = lb
            log("library baseline done")
        except Exception as e:                                          # noqa: BLE001  (a measurement, never a reason to lose the bench line)
            result["library_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline()
        log("cpu baseline done")
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(result) + "\n").encode())


if __name__ == "__main__":
    main()
