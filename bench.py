#!/usr/bin/env python
"""bench.py — images/sec of the ResNet-50 + LDS + FDS training hot path on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md §8d config 2, per GPU): IMDB-WIKI-DIR shapes — ResNet-50,
bf16 conv stack, B=256 synthetic 224x224 batches, LDS weights (sqrt_inv, gaussian 5/2) from a synthetic
191 509-label long-tailed set, FDS (bucket_num=100, bucket_start=0, ks=5, sigma=2, momentum 0.9) with tables
populated by two update rounds and the run at epoch >= 2 so calibration is non-trivial (A.4), loss l1, Adam 1e-3.

A "step" is one optimisation step (train.py:246-262). Nothing of the hot path is skipped: every
--epoch-len steps the timed region also runs the reference's epoch tail (train.py:269-281) over the same
number of batches — the no-grad train-mode feature pass, FDS.update_last_epoch_stats and
FDS.update_running_stats (with the cross-rank statistic all-reduce when N > 1). `value` counts trained
images only (K * B * N / time), so it is the throughput of the whole loop, tail included;
`train_only_images_per_sec` is the same clock without the tail, for comparison with plain ResNet-50 numbers.

Rank 0 prints ONE JSON line. Extra legs on rank 0 at N=1 after the timed region: hand-written-kernel
micro-rooflines (HIP events on the launch stream) and the CPU baseline (oracle/torch_oracle.py = a torch-CPU
port of the reference loop, timed on the host cores on a bounded sample).
"""
import argparse
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


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.time() - _T0:7.1f}s]", *a, file=sys.stderr, flush=True)


def long_tail_labels(rng, n):
    """Fixed long-tailed pmf over integer ages 0..120: round(clip(|N(0,18)| + 20, 0, 120))."""
    return np.clip(np.round(np.abs(rng.normal(0, 18, n)) + 20), 0, 120).astype(np.float32)


def build(args, device, rank):
    from dirhip import lds
    from dirhip.parallel import DataParallelEngine
    from dirhip.resnet import resnet50
    torch.manual_seed(0)
    model = resnet50(fds=True, bucket_num=100, bucket_start=0, start_update=0, start_smooth=1,
                     kernel="gaussian", ks=5, sigma=2, momentum=0.9).to(device)
    engine = DataParallelEngine(model, amp_dtype=torch.bfloat16, channels_last=True)
    engine.train()
    optimizer = torch.optim.Adam(engine.parameters(), lr=1e-3, fused=True)
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
    """Average duration of fn() over `iters` launches with HIP events on torch's current stream
    (= the stream the C-ABI launches on)."""
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


def kernel_rooflines(device):
    """Algorithmic bytes / measured duration for the hand-written FDS kernels."""
    from dirhip import ops
    out = []
    g = torch.Generator(device=device).manual_seed(3)
    # --- dir_fds_scatter_stats over the whole epoch's features: N*C*4 + N*4 bytes read
    n, c, nb = N_TRAIN, 2048, 100
    lab = torch.as_tensor(long_tail_labels(np.random.default_rng(3), n), device=device)
    feats = torch.randn(n, c, device=device, generator=g).abs_()
    bins, _ = ops.bin_index(lab, 0, 100)
    ms = event_time_ms(lambda: ops.scatter_stats(feats, bins, nb), 10)
    alg = n * c * 4 + n * 4
    out.append({"kernel": "dir_fds_scatter_stats (5 launches: group x3, piece sums, combine)", "bound": "hbm",
                "shape": f"N={n} C={c} Nb={nb} f32", "ms": ms, "algorithmic_bytes": alg,
                "achieved": alg / ms / 1e6, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": alg / ms / 1e6 / PEAK_HBM_GBS})
    del feats
    # --- calibrate forward: 2*B*C*4 + T*U*C*4 + B*4, T = 3 tables (m1, scale, m2; scale precomputed per epoch)
    for b in (256, 65536):
        x = torch.randn(b, c, device=device, generator=g)
        labb = torch.as_tensor(long_tail_labels(np.random.default_rng(4), b), device=device)
        m1 = torch.randn(nb, c, device=device, generator=g)
        sc = torch.rand(nb, c, device=device, generator=g) + 0.5
        m2 = torch.randn(nb, c, device=device, generator=g)
        u = int(torch.unique(labb.clamp(max=99)).numel())
        ms = event_time_ms(lambda: ops.smooth_fwd_(x, labb, 0, 100, m1, sc, m2), 50 if b == 256 else 10)
        alg = 2 * b * c * 4 + 3 * u * c * 4 + b * 4
        out.append({"kernel": "dir_fds_smooth_fwd (K1+K5 fused)" if b <= 2048 else "dir_fds_smooth_fwd (K1, K5)",
                    "bound": "hbm" if b > 2048 else "launch", "shape": f"B={b} C={c} U={u} T=3 f32", "ms": ms,
                    "algorithmic_bytes": alg, "achieved": alg / ms / 1e6, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": alg / ms / 1e6 / PEAK_HBM_GBS})
        bins_b, _ = ops.bin_index(labb, 0, 100)
        dy = torch.randn(b, c, device=device, generator=g)
        ms = event_time_ms(lambda: ops.calibrate_bwd(dy, bins_b, sc), 50 if b == 256 else 10)
        alg = 2 * b * c * 4 + u * c * 4 + b * 4
        out.append({"kernel": "dir_fds_calibrate_bwd", "bound": "hbm" if b > 2048 else "launch",
                    "shape": f"B={b} C={c} U={u} f32", "ms": ms, "algorithmic_bytes": alg,
                    "achieved": alg / ms / 1e6, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": alg / ms / 1e6 / PEAK_HBM_GBS})
        del x, dy
    # --- "next" rows: NYUD2 dense map [32,128,114,152] (per-pixel buckets, narrow-row kernels) and STS-B [128,12000]
    b, c, h, w = 32, 128, 114, 152
    depth = torch.rand(b * h * w, device=device, generator=g) * 9.3 + 0.7
    rows = torch.rand(b * h * w, c, device=device, generator=g)
    bins = ops.bin_scaled(depth, 10.0, 7, 100)
    t1, sc, t2 = (torch.rand(93, c, device=device, generator=g) + 0.5 for _ in range(3))
    ms = event_time_ms(lambda: ops.calibrate_fwd_(rows, bins, t1, sc, t2), 10)
    alg = 2 * rows.numel() * 4 + 3 * 93 * c * 4 + rows.shape[0] * 4
    out.append({"kernel": "dir_fds_calibrate_fwd (narrow rows, NYUD2 dense map)", "bound": "hbm", "shape": f"[{b},{c},{h},{w}] f32, 93 buckets",
                "ms": ms, "algorithmic_bytes": alg, "achieved": alg / ms / 1e6, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": alg / ms / 1e6 / PEAK_HBM_GBS})
    ms = event_time_ms(lambda: ops.scatter_stats(rows, bins, 93), 5)
    alg = rows.numel() * 4 + rows.shape[0] * 4
    out.append({"kernel": "dir_fds_scatter_stats (narrow rows, NYUD2 dense map)", "bound": "hbm", "shape": f"N={rows.shape[0]} C={c} Nb=93 f32",
                "ms": ms, "algorithmic_bytes": alg, "achieved": alg / ms / 1e6, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": alg / ms / 1e6 / PEAK_HBM_GBS})
    del rows
    return out


# ResNet-50 convolutions that run on the hand-written MFMA kernel (every layer but the 7x7 stem):
# (Cin, Cout, k, stride, Hin, count) — SURVEY.md Appendix B
RESNET50_CONVS = [(64, 64, 1, 1, 56, 1), (64, 64, 3, 1, 56, 3), (64, 256, 1, 1, 56, 4), (256, 64, 1, 1, 56, 2), (256, 128, 1, 1, 56, 1),
                  (128, 128, 3, 2, 56, 1), (128, 512, 1, 1, 28, 4), (256, 512, 1, 2, 56, 1), (512, 128, 1, 1, 28, 3), (128, 128, 3, 1, 28, 3),
                  (512, 256, 1, 1, 28, 1), (256, 256, 3, 2, 28, 1), (256, 1024, 1, 1, 14, 6), (512, 1024, 1, 2, 28, 1), (1024, 256, 1, 1, 14, 5),
                  (256, 256, 3, 1, 14, 5), (1024, 512, 1, 1, 14, 1), (512, 512, 3, 2, 14, 1), (512, 2048, 1, 1, 7, 3), (1024, 2048, 1, 2, 14, 1),
                  (2048, 512, 1, 1, 7, 2), (512, 512, 3, 1, 7, 2)]


def conv_roofline(device, batch):
    """Dominant kernel of the step = conv_igemm_kernel (forward + stride-1 data-gradient launches, ~1/3 of the GPU
    time). Algorithmic FLOPs per launch = 2*M*Cout*Cin*R*S (implicit GEMM, SURVEY.md §8d: 8.174 GFLOP/image forward);
    duration = HIP events on the launch stream, 5 launches per layer shape, every layer shape of the network in its
    forward and (stride-1 layers) data-gradient configuration."""
    from dirhip.conv import conv2d_igemm
    tot_flop = tot_ms = 0.0
    launches = 0
    per_kind = {}
    for cin, cout, k, st, h, cnt in RESNET50_CONVS:
        pad = k // 2
        ho = (h + 2 * pad - k) // st + 1
        cfgs = [("fwd", cin, cout, h, st)]
        if st == 1:
            cfgs.append(("dgrad", cout, cin, ho, 1))          # same kernel on dY with rotated weights
        for kind, ci, co, hh, s_ in cfgs:
            x = torch.randn(batch, ci, hh, hh, device=device).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            w = (torch.randn(co, ci, k, k, device=device) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            ms = event_time_ms(lambda: conv2d_igemm(x, w, s_, pad, want_stats=(kind == "fwd")), 5, warm=2)
            hout = (hh + 2 * pad - k) // s_ + 1
            flop = 2.0 * batch * hout * hout * co * ci * k * k
            tot_flop += flop * cnt
            tot_ms += ms * cnt
            launches += cnt
            a = per_kind.setdefault(kind, [0.0, 0.0])
            a[0] += flop * cnt
            a[1] += ms * cnt
            del x, w
    # HBM traffic per launch: PMC counters cannot be read from inside this process; they come from the separate
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the same 98 launches (tools/pmc_conv_pass.py ->
    # tools/pmc_conv_parse.py, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), committed under profiles/.
    traffic = traffic_alg = None
    tpath = os.path.join(ROOT, "profiles", "r01_conv_pmc_traffic.json")
    if os.path.isfile(tpath):
        tj = json.load(open(tpath))
        if tj.get("batch") == batch and tj.get("launches_per_step") == launches:
            traffic, traffic_alg = tj["traffic_bytes_per_launch"], tj["algorithmic_bytes_per_step"] / launches
    return {"bound": "mfma", "kernel": "conv_igemm_kernel<128|64> (hand-written MFMA implicit GEMM; all 52 conv layers fwd + 46 stride-1 dgrad)",
            "achieved": tot_flop / tot_ms / 1e9, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": tot_flop / tot_ms / 1e9 / PEAK_BF16_TFLOPS, "traffic": traffic, "algorithmic_bytes_per_launch": traffic_alg,
            "launches_per_step": launches, "avg_launch_us": tot_ms / launches * 1e3,
            "algorithmic_flop_per_step": tot_flop, "ms_per_step_in_this_kernel": tot_ms,
            "fwd_tflops": per_kind["fwd"][0] / per_kind["fwd"][1] / 1e9, "dgrad_tflops": per_kind["dgrad"][0] / per_kind["dgrad"][1] / 1e9}


def bn_roofline(device, batch):
    """Fused BatchNorm(+residual)(+ReLU) forward apply / backward on the widest layer-1 tensor (HBM bound)."""
    import torch.nn as nn
    from dirhip.bn import bn_act
    out = []
    c, hw = 256, 56
    x = torch.randn(batch, c, hw, hw, device=device).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r = torch.randn_like(x).requires_grad_(True)
    dy = torch.randn_like(x)
    bn = nn.BatchNorm2d(c).to(device)
    nbytes = x.numel() * 2
    tf = event_time_ms(lambda: bn_act(x, bn, True, r), 10)

    def fb():
        bn_act(x, bn, True, r).backward(dy)
    tb = event_time_ms(fb, 10) - tf
    for name, ms, passes in (("dir_bn_fwd_train (stats + finalize + apply, residual+ReLU)", tf, 4), ("dir_bn_bwd (reduce + finalize + apply, residual+ReLU)", tb, 8)):
        out.append({"kernel": name, "bound": "hbm", "shape": f"[{batch},{c},{hw},{hw}] bf16 NHWC", "ms": ms,
                    "algorithmic_bytes": passes * nbytes, "achieved": passes * nbytes / ms / 1e6, "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "frac": passes * nbytes / ms / 1e6 / PEAK_HBM_GBS})
    return out


def cpu_baseline(seconds_budget=25.0):
    """The oracle port (torch-CPU restatement of the reference loop) on the host cores: ResNet-50 + FDS + LDS
    weights + l1 + Adam, B=8 (BASELINE configs[0] batch), epoch tail included every 4 steps."""
    from oracle import torch_oracle
    cores = min(os.cpu_count() or 1, int(os.environ.get("DIR_CPU_BASELINE_THREADS", "64")))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = torch_oracle.RefResNet50(fds=True, bucket_num=100, bucket_start=0, start_update=0, start_smooth=1,
                                     kernel="gaussian", ks=5, sigma=2, momentum=0.9)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    rng = np.random.default_rng(0)
    b, epoch_len = 8, 4
    g = torch.Generator().manual_seed(0)
    batches = []
    for _ in range(epoch_len):
        lab = long_tail_labels(rng, b)
        batches.append((torch.randn(b, 3, 224, 224, generator=g), torch.as_tensor(lab).view(-1, 1), torch.rand(b, 1, generator=g) + 0.5))
    epoch = 0
    for _ in range(2):                                  # warm-up epochs also populate the FDS tables (epochs 0, 1)
        for x, y, w in batches[:2]:
            torch_oracle.train_step(model, opt, x, y, w, epoch, "l1")
        torch_oracle.epoch_tail(model, [(x, y) for x, y, _ in batches[:2]], epoch)
        epoch += 1
    t0 = time.perf_counter()
    steps = 0
    while True:
        for x, y, w in batches:
            torch_oracle.train_step(model, opt, x, y, w, epoch, "l1")
            steps += 1
        torch_oracle.epoch_tail(model, [(x, y) for x, y, _ in batches], epoch)
        epoch += 1
        if time.perf_counter() - t0 > seconds_budget * 0.6 or steps >= 32:
            break
    dt = time.perf_counter() - t0
    return {"value": steps * b / dt, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{steps} steps of B={b} 224x224 fp32 + {steps // epoch_len} epoch tails ({epoch_len} fwd passes + FDS update each), "
                      f"oracle/torch_oracle.py (torch-CPU port of train.py:246-281), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--epoch-len", type=int, default=8, help="steps per bench epoch (one FDS epoch tail each)")
    ap.add_argument("--miopen-find", action="store_true", help="MIOpen find mode (cudnn.benchmark=True like train.py:198)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-rooflines", action="store_true")
    args = ap.parse_args()

    args.epoch_len = max(1, min(args.epoch_len, args.steps))      # at least one epoch tail inside the timed region
    from dirhip.parallel import init_distributed
    rank, world, local_rank = init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU (the hot path has no CPU fallback)")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    torch.backends.cudnn.benchmark = bool(args.miopen_find)

    from dirhip.train_loop import EpochFeatures, resolve_loss
    model, engine, optimizer, batches = build(args, device, rank)
    store = EpochFeatures(args.epoch_len * args.batch, 2048, device)
    loss_fn = resolve_loss("l1")

    log("model + data built")
    # set-up, not warm-up: the first step makes the library pick kernels for the few layers still on it (stem conv,
    # stride-2 data gradients: ~17 s of solver search on a fresh box); one step + one tail forward, untimed
    from dirhip.train_loop import epoch_tail, train_step
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
    n_tails = args.steps // args.epoch_len
    flops = args.steps * args.batch * FLOP_FWD_BWD + n_tails * args.epoch_len * args.batch * FLOP_FWD   # per GPU
    result = {
        "metric": "images/sec ResNet-50+FDS IMDB-WIKI 224x224 (train loop incl. FDS epoch tail)",
        "value": images / dt, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: IMDB-WIKI-DIR ResNet-50 + LDS + FDS (ks=5, sigma=2), bf16 conv stack (own MFMA kernels: stem, implicit-GEMM fwd/dgrad/wgrad) + fused HIP BatchNorm / join / pool nodes, fp32 FDS+loss tail, "
                               "batch=256 per MI355X, l1 loss, Adam 1e-3", "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                   "epoch_len_steps": args.epoch_len, "epoch_tails_in_timed_region": n_tails, "parallelism": f"dp{world}",
                   "final_loss": loss_val},
        "train_only_images_per_sec": images / dt_train,
        # whole-loop MFMA fraction (24.287 GFLOP per trained image + 8.174 per tail-forward image over the wall clock)
        "roofline_loop": {"bound": "mfma", "kernel": "whole train loop per GPU: ResNet-50 fwd+bwd (+ fwd-only epoch tail)",
                          "achieved": flops / dt / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                          "frac": flops / dt / 1e12 / PEAK_BF16_TFLOPS,
                          "train_only_frac": args.steps * args.batch * FLOP_FWD_BWD / dt_train / 1e12 / PEAK_BF16_TFLOPS},
    }
    result["roofline"] = dict(result["roofline_loop"], traffic=None)      # replaced below by the dominant kernel's when measured
    if rank == 0 and world == 1:
        del engine, optimizer, batches, store
        torch.cuda.empty_cache()
        if not args.no_kernel_rooflines:
            result["roofline"] = conv_roofline(device, args.batch)
            log("conv roofline done")
            result["kernel_rooflines"] = kernel_rooflines(device) + bn_roofline(device, args.batch)
            log("kernel rooflines done")
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline()
            log("cpu baseline done")
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
