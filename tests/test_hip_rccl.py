"""-m gpu: the RCCL path, EXECUTED. The lease has one MI355X, so the process group has ONE rank (backend "nccl" = RCCL): every
collective is the identity, which makes the expected result exact — and RCCL still has to accept every (dtype, reduce-op, call form)
the data-parallel design uses (reference: the DataParallel wrap at imdb-wiki-dir/train.py:143 that this engine replaces):

  * init_process_group("nccl", device_id=cuda:0) as dirhip.parallel.init_distributed does it;
  * ReduceOp.AVG on float32 flat buckets, async_op + .wait() on the compute stream (gradient exchange, per step);
  * ReduceOp.SUM on float64 [Nb], [Nb, C] (FDS statistic merge, per epoch), ReduceOp.MAX on int32 (presence flags, A.3);
  * broadcast of parameters and buffers from rank 0, barrier, all_reduce MAX on a float64 scalar (bench.py's timing).

Then the engine's N > 1 branch itself (``force_collectives=True``): post-accumulate-grad hooks -> bucket all-reduces launched during
the backward pass -> optimizer step, for two steps of the real fused ResNet-50 + FDS graph, against the same steps without any
collective: bit-identical parameters; and FDS.update_running_stats through merge_stats_across_ranks.
Runs in a spawned child with a hard timeout, so that a wedged rendezvous cannot hang the suite."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FDS_KW = dict(bucket_num=100, bucket_start=0, start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=2, momentum=0.9)


def _long_tail(rng, n):
    return np.clip(np.round(np.abs(rng.normal(0, 18, n)) + 20), 0, 120).astype(np.float32)


def _steps(force, tmp):
    """Two train steps + one epoch tail; returns (parameters, fds buffers, engine report, kernel names of a third step)."""
    from torch.profiler import ProfilerActivity, profile
    from dirhip.optim import Adam
    from dirhip.parallel import DataParallelEngine
    from dirhip.resnet import resnet50
    from dirhip.train_loop import EpochFeatures, epoch_tail, resolve_loss, train_step
    torch.manual_seed(11)
    model = resnet50(fds=True, **FDS_KW).cuda()
    eng = DataParallelEngine(model, amp_dtype=torch.bfloat16, channels_last=True, bucket_mb=8, force_collectives=force)
    eng.train()
    opt = Adam(eng.parameters(), lr=1e-3)
    rng = np.random.default_rng(5)
    g = torch.Generator().manual_seed(6)
    batches = []
    for _ in range(2):
        lab = _long_tail(rng, 16)
        batches.append((torch.randn(16, 3, 224, 224, generator=g).cuda(), torch.tensor(lab).view(-1, 1).cuda(),
                        torch.tensor(rng.uniform(0.5, 1.5, 16).astype(np.float32)).view(-1, 1).cuda()))
    loss_fn = resolve_loss("l1")
    store = EpochFeatures(32, 2048, torch.device("cuda"))
    for ep in range(2):
        for x, y, w in batches:
            train_step(eng, opt, x, y, w, ep, loss_fn)
        epoch_tail(eng, ((x, y) for x, y, _ in batches), ep, store)
    torch.cuda.synchronize()
    params = {k: v.detach().cpu().clone() for k, v in model.named_parameters()}
    bufs = {k: v.detach().cpu().clone() for k, v in model.FDS.named_buffers()}
    eng.measure_comm = True
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        train_step(eng, opt, *batches[0], 2, loss_fn)
        torch.cuda.synchronize()
    names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    names += ["host:" + e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and ("c10d::" in e.name or "nccl:" in e.name)]
    rep = eng.comm_report()
    rep["n_buckets"] = len(eng._buckets)
    rep["grad_is_bucket_view"] = all(p.grad is not None and p.grad.data_ptr() == eng._buckets[eng._bucket_of[id(p)][0]].views[eng._bucket_of[id(p)][1]].data_ptr()
                                     for p in eng.parameters())
    return params, bufs, rep, names


def _worker(rank, port, tmp):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)          # the call form of parallel.init_distributed
    out = {"backend": dist.get_backend()}
    # ---- every (dtype, op, call form) of the design, on RCCL
    t = torch.randn(1 << 20, device=dev)
    ref = t.clone()
    wk = dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=True)
    wk.wait()
    out["avg_f32_identity"] = bool(torch.equal(t, ref))
    d64 = torch.randn(100, 2048, dtype=torch.float64, device=dev)
    r64 = d64.clone()
    dist.all_reduce(d64)
    c64 = torch.arange(100, dtype=torch.float64, device=dev)
    dist.all_reduce(c64)
    out["sum_f64_identity"] = bool(torch.equal(d64, r64) and torch.equal(c64, torch.arange(100, dtype=torch.float64, device=dev)))
    bits = torch.tensor([1, 0, 1, 0], dtype=torch.int32, device=dev)
    dist.all_reduce(bits, op=dist.ReduceOp.MAX)
    out["max_i32_identity"] = bits.tolist() == [1, 0, 1, 0]
    tt = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dist.barrier()
    b = torch.randn(1000, device=dev)
    rb = b.clone()
    dist.broadcast(b, src=0)
    out["broadcast_identity"] = bool(torch.equal(b, rb)) and float(tt.item()) == 1.25
    # ---- the engine's N > 1 branch over RCCL vs the same steps with no collective at all
    p1, f1, rep1, names1 = _steps(True, tmp)
    p0, f0, rep0, names0 = _steps(False, tmp)
    out["params_bit_identical"] = all(torch.equal(p0[k], p1[k]) for k in p0)
    worst = 0.0
    for k in f0:
        a, bb = f0[k].double(), f1[k].double()
        worst = max(worst, float((a - bb).abs().max() / max(float(a.abs().max()), 1e-30)))
    out["fds_buffers_max_rel"] = worst
    out["fds_counts_equal"] = bool(torch.equal(f0["num_samples_tracked"], f1["num_samples_tracked"]))
    out["report_forced"] = rep1
    out["report_plain"] = {k: rep0[k] for k in ("ranks", "backend", "n_buckets")}
    dev_k = lambda ns: [n for n in ns if not n.startswith("host:") and ("nccl" in n.lower() or "rccl" in n.lower())]      # noqa: E731
    out["allreduce_calls_in_one_step"] = sum(1 for n in names1 if n.startswith("host:c10d::allreduce_"))
    out["allreduce_calls_in_plain_step"] = sum(1 for n in names0 if n.startswith("host:c10d::allreduce_"))
    out["rccl_device_kernels_in_one_step"] = len(dev_k(names1))      # (informational: a one-rank communicator may not need a kernel)
    out["rccl_device_kernel_names"] = sorted(set(dev_k(names1)))[:4]
    # ---- non-integer labels through the N > 1 branch: labels all-gathered for the common value-group list, per-group statistics merged
    from dirhip.fds import FDS
    rng = np.random.default_rng(3)
    kw = dict(feature_dim=64, bucket_num=30, bucket_start=3, start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=2, momentum=None)
    Fa, Fb = FDS(**kw).cuda(), FDS(**kw).cuda()
    Fb.force_collectives = True
    for ep in range(3):
        lab = (np.round(rng.normal(15, 8, 400) * 2) / 2).astype(np.float32)
        lab[:4] = [3.0, 29.0, 2.5, 30.5]
        feats = (np.abs(rng.normal(0, 1, (400, 64))) * 0.5 + 0.02 * lab[:, None]).astype(np.float32)
        for Fx in (Fa, Fb):
            Fx.update_last_epoch_stats(ep)
            Fx.update_running_stats(torch.tensor(feats).cuda(), torch.tensor(lab).cuda(), ep)
    worst = 0.0
    for (k, a), (_, bb) in zip(Fa.named_buffers(), Fb.named_buffers()):
        worst = max(worst, float((a.double() - bb.double()).abs().max() / max(float(a.double().abs().max()), 1e-30)))
    out["frac_labels_forced_vs_plain_max_rel"] = worst
    out["frac_labels_counts_equal"] = bool(torch.equal(Fa.num_samples_tracked, Fb.num_samples_tracked))
    with open(os.path.join(tmp, "rccl.json"), "w") as f:
        json.dump(out, f, indent=1)
    dist.destroy_process_group()


def test_rccl_world1_every_collective_of_the_design_and_the_engine_branch(tmp_path):
    import torch.multiprocessing as mp
    port = 35000 + int(np.random.default_rng().integers(0, 2000))
    ctx = mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=1, join=False)
    ok = ctx.join(timeout=600)
    if not ok:
        for p in ctx.processes:
            p.kill()                                               # (our own child, by handle)
        pytest.fail("the RCCL world-size-1 worker did not finish within 600 s")
    out = json.load(open(tmp_path / "rccl.json"))
    from conftest import records_dir
    prof = records_dir()
    json.dump(out, open(os.path.join(prof, "rccl_world1.json"), "w"), indent=1)
    assert out["backend"] == "nccl"
    assert out["avg_f32_identity"] and out["sum_f64_identity"] and out["max_i32_identity"] and out["broadcast_identity"], out
    assert out["params_bit_identical"], "engine over RCCL (1 rank) != engine without collectives"
    assert out["fds_counts_equal"] and out["fds_buffers_max_rel"] <= 1e-6, out         # Chan merge of ONE triple: n*mean/n round trip in float64
    assert out["frac_labels_counts_equal"] and out["frac_labels_forced_vs_plain_max_rel"] <= 1e-6, out
    rep = out["report_forced"]
    assert rep["backend"] == "nccl" and rep["reduce_op"].startswith("avg") and rep["grad_copies"] == 0 and rep["bucket_scale_kernels"] == 0, rep
    assert rep["grad_is_bucket_view"] and rep["n_buckets"] >= 3
    assert out["allreduce_calls_in_one_step"] == rep["n_buckets"], out                    # one RCCL all-reduce per bucket, issued from the hooks
    assert out["allreduce_calls_in_plain_step"] == 0
