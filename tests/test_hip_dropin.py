"""-m gpu: the drop-in surface EXECUTED — the `train.py` command line of imbalanced-regression_amd/imdb-wiki-dir (train ->
resume -> evaluate in subprocesses, like a user would), reference-format checkpoints, `FDS.reset`, the epoch tail's padding
mask, and the data-parallel paths on DEVICE tensors with two processes sharing the one GPU over gloo (the logic RCCL
runs at N > 1: FDS statistic merge incl. the OR-reduced boundary flags, gradient buckets driven by the real fused nodes)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import PKG, ROOT, assert_close

pytestmark = pytest.mark.gpu

FDS_KW = dict(bucket_num=100, bucket_start=0, start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=2, momentum=0.9)
FDS_BUFFERS = ["epoch", "running_mean", "running_var", "running_mean_last_epoch", "running_var_last_epoch",
               "smoothed_mean_last_epoch", "smoothed_var_last_epoch", "num_samples_tracked"]


def _cli(args, log):
    cmd = [sys.executable, os.path.join(PKG, "imdb-wiki-dir", "train.py")] + args
    p = subprocess.run(cmd, cwd=os.path.join(PKG, "imdb-wiki-dir"), capture_output=True, text=True, timeout=900)
    out = p.stdout + p.stderr
    from conftest import records_dir
    with open(os.path.join(records_dir(), log), "w") as f:
        f.write(" ".join(cmd) + "\n" + out)
    assert p.returncode == 0, out[-3000:]
    return out


def test_train_py_cli_train_resume_evaluate(tmp_path):
    """python train.py --synthetic ... (3 epochs, FDS + LDS) ; --resume ; --evaluate — from the drop-in folder."""
    base = ["--synthetic", "2048", "--fds", "--lds", "--reweight", "sqrt_inv", "--lds_kernel", "gaussian", "--lds_ks", "5",
            "--lds_sigma", "2", "--fds_kernel", "gaussian", "--fds_ks", "5", "--fds_sigma", "2", "--batch_size", "64",
            "--store_root", str(tmp_path), "--print_freq", "8"]
    out = _cli(base + ["--epoch", "3"], "cli_train.log")
    name = "imdb_wiki_resnet50_lds_gau_5_2.0_fds_gau_5_2.0_0_1_0.9_adam_l1_0.001_64"           # train.py:78-93 naming
    store = tmp_path / name
    assert store.is_dir(), os.listdir(tmp_path)
    for needle in ("Using FDS: [GAUSSIAN] (5/2.0)", "Using re-weighting: [SQRT_INV]", "Using LDS: [GAUSSIAN] (5/2.0)",
                   "Create Epoch [0] features of all training data...", "Updated running statistics with Epoch [0] features!",
                   "Updated smoothed statistics on Epoch [1]!", "Updated running statistics with Epoch [2] features!",
                   " * Overall: MSE", " * Many: MSE", "Epoch #2: Train loss", "Test best model on testset...", "Test loss: MSE"):
        assert needle in out, needle
    ckpt = torch.load(store / "ckpt.pth.tar", map_location="cpu")
    assert set(ckpt) == {"epoch", "model", "best_loss", "state_dict", "optimizer"} and ckpt["epoch"] == 3 and ckpt["model"] == "resnet50"
    keys = list(ckpt["state_dict"])
    assert all(k.startswith("module.") for k in keys)                                          # DataParallel-style prefix (train.py:143)
    assert [k for k in keys if ".FDS." in k] == [f"module.FDS.{b}" for b in FDS_BUFFERS]
    assert float(ckpt["state_dict"]["module.FDS.epoch"]) == 2.0
    assert float(ckpt["state_dict"]["module.FDS.num_samples_tracked"].sum()) == 3 * 2048        # three epoch tails over the whole set
    assert (store / "ckpt.best.pth.tar").is_file() and (store / "training.log").is_file()
    # a relaunch of the same configuration without --resume / --overwrite must refuse, not delete
    p = subprocess.run([sys.executable, os.path.join(PKG, "imdb-wiki-dir", "train.py")] + base + ["--epoch", "3"],
                       cwd=os.path.join(PKG, "imdb-wiki-dir"), capture_output=True, text=True, timeout=300, stdin=subprocess.DEVNULL)
    assert p.returncode != 0 and "--overwrite" in (p.stdout + p.stderr) and (store / "ckpt.pth.tar").is_file()
    out = _cli(base + ["--epoch", "4", "--resume", str(store / "ckpt.pth.tar")], "cli_resume.log")
    assert "Loaded checkpoint" in out and "(Epoch [3])" in out and "Epoch #3: Train loss" in out and "Epoch #2: Train loss" not in out
    ckpt2 = torch.load(store / "ckpt.pth.tar", map_location="cpu")
    assert ckpt2["epoch"] == 4 and float(ckpt2["state_dict"]["module.FDS.epoch"]) == 3.0
    out = _cli(base + ["--evaluate", "--resume", str(store / "ckpt.best.pth.tar")], "cli_evaluate.log")
    assert "testing..." in out and "Test: " in out and " * Low: MSE" in out


def test_train_py_cli_precision_schedule(tmp_path):
    """train.py --amp_switch_epoch 1 (round 6): epoch 0 on the float32 conv stack, epoch 1 in bf16, one engine; a resume at epoch 1 starts in bf16."""
    base = ["--synthetic", "512", "--fds", "--lds", "--reweight", "sqrt_inv", "--batch_size", "64", "--store_root", str(tmp_path), "--print_freq", "4",
            "--amp_switch_epoch", "1"]
    out = _cli(base + ["--epoch", "2"], "cli_amp_switch.log")
    assert "Epoch [0]: conv stack in float32 (--amp_switch_epoch 1)" in out and "Epoch [1]: conv stack in bf16 (--amp_switch_epoch 1)" in out, out[-2000:]
    assert "Epoch #1: Train loss" in out and "Test loss: MSE" in out
    store = [d for d in os.listdir(tmp_path) if (tmp_path / d / "ckpt.pth.tar").is_file()]
    assert len(store) == 1
    out = _cli(base + ["--epoch", "3", "--resume", str(tmp_path / store[0] / "ckpt.pth.tar")], "cli_amp_switch_resume.log")
    assert "conv stack in float32" not in out and "Epoch [2]: conv stack in bf16" in out and "Epoch #2: Train loss" in out, out[-2000:]


def test_train_py_cli_split_bf16_arithmetic(tmp_path):
    """train.py --amp fp32x2 (whole run on the split-bf16 float32 kernels) and --amp_switch_epoch 1 --amp_early fp32x3 (split arithmetic first, bf16 after)."""
    common = ["--synthetic", "256", "--fds", "--lds", "--reweight", "sqrt_inv", "--batch_size", "64", "--print_freq", "4"]
    out = _cli(common + ["--store_root", str(tmp_path / "a"), "--amp", "fp32x2", "--epoch", "1"], "cli_fp32x2.log")
    assert "Epoch #0: Train loss" in out and "Test loss: MSE" in out and "conv stack in" not in out
    out = _cli(common + ["--store_root", str(tmp_path / "b"), "--amp_switch_epoch", "1", "--amp_early", "fp32x3", "--epoch", "2"], "cli_amp_early.log")
    assert "Epoch [0]: conv stack in float32 on split-bf16 x3 (--amp_switch_epoch 1)" in out and "Epoch [1]: conv stack in bf16 (--amp_switch_epoch 1)" in out, out[-2000:]
    assert "Epoch #1: Train loss" in out


def test_train_py_cli_on_image_files_host_and_gpu_augment(tmp_path):
    """The real-data path of the drop-in CLI, executed: an agedb-style csv + image files (written here), DataLoader workers, the
    host transform chain — the same run with --gpu_augment (uint8 batches, dir_augment_u8 on the GPU) — and with
    --gpu_resize (workers only decode: ragged uint8 batches, dir_resize_u8 + dir_augment_u8 on the GPU), and with --gpu_cache on top (the
    resized bytes stay in HBM: the feature pass of epoch 0 and all of epoch 1 run without the loader)."""
    import pandas as pd
    from PIL import Image
    rng = np.random.default_rng(0)
    data = tmp_path / "data"
    (data / "imgs").mkdir(parents=True)
    rows = []
    ages = [30] * 110 + [40] * 28 + [50] * 12                     # many- (> 100), median- and low-shot (< 20) labels, train.py:338-391
    splits = ["train"] * 150 + ["val"] * 12 + ["test"] * 12
    ages += [30, 40, 50] * 4 + [30, 40, 50] * 4                    # every shot class present in val and test (shot_metrics needs that)
    for k, (age, split) in enumerate(zip(ages, splits)):
        h, w = int(rng.integers(60, 120)), int(rng.integers(60, 120))
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(data / "imgs" / f"{k}.png")
        rows.append({"age": age, "path": f"imgs/{k}.png", "split": split})
    pd.DataFrame(rows).to_csv(data / "agedb.csv", index=False)
    base = ["--dataset", "agedb", "--data_dir", str(data), "--fds", "--lds", "--reweight", "sqrt_inv", "--batch_size", "32", "--epoch", "2",
            "--workers", "2", "--img_size", "224", "--print_freq", "1", "--bucket_start", "3"]
    losses = {}
    for tag, extra in (("host", []), ("gpu", ["--gpu_augment"]), ("gpu_resize", ["--gpu_resize"]), ("gpu_cache", ["--gpu_resize", "--gpu_cache"])):
        out = _cli(base + extra + ["--store_root", str(tmp_path / tag)], f"cli_files_{tag}.log")
        if tag == "gpu_cache":
            assert "Training images cached in HBM once decoded" in out
        for needle in ("Training data size: 150", "Validation data size: 12", "Create Epoch [1] features of all training data...",
                       "Updated smoothed statistics on Epoch [1]!", " * Overall: MSE", "Test loss: MSE"):
            assert needle in out, (tag, needle)
        line = [l for l in out.splitlines() if "Test loss: MSE" in l][-1]
        losses[tag] = float(line.split("L1 [")[1].split("]")[0])
        assert np.isfinite(losses[tag])
    # evaluation has no randomness: the two paths see identical pixels only if their weights agree, which two 2-epoch runs with
    # different augmentation draws do not — so only sanity here; pixel identity of the two paths is test_augment.py's job


def test_reference_format_checkpoint_roundtrip(golden, tmp_path):
    """A checkpoint with the reference's key set (tests/golden/resnet50_forward.npz `keys`, taken from the reference's
    resnet50.state_dict()) and DataParallel's `module.` prefix (train.py:143,209-215) loads strictly, predicts the same,
    and is saved back key for key; the backbone-only filter of --pretrained / --retrain_fc (train.py:174-181) works on it."""
    from dirhip.parallel import DataParallelEngine
    from dirhip.resnet import resnet50
    g = golden("resnet50_forward.npz")
    ref_keys = [f"module.{k}" for k in g["keys"]]
    torch.manual_seed(7)
    src = DataParallelEngine(resnet50(fds=True, **FDS_KW).cuda(), amp_dtype=torch.bfloat16, channels_last=True)
    sd = src.state_dict()
    assert list(sd) == ref_keys                                                               # same names, same ORDER
    for v, shp in zip(sd.values(), g["shapes"]):
        assert list(v.shape) == __import__("json").loads(str(shp))
    # make every tensor distinctive (incl. the FDS tables and BatchNorm statistics), as a trained reference model's would be
    gen = torch.Generator().manual_seed(8)
    ref_state = {k: (torch.rand(v.shape, generator=gen) + 0.5).to(v.dtype) if v.is_floating_point() else torch.full_like(v.cpu(), 5)
                 for k, v in sd.items()}
    torch.save({"epoch": 11, "model": "resnet50", "best_loss": 7.7, "state_dict": ref_state, "optimizer": {}}, tmp_path / "ref.pth.tar")
    torch.manual_seed(9)
    dst = DataParallelEngine(resnet50(fds=True, **FDS_KW).cuda(), amp_dtype=torch.bfloat16, channels_last=True)
    ck = torch.load(tmp_path / "ref.pth.tar", map_location="cuda")
    missing = dst.load_state_dict(ck["state_dict"], strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    back = dst.state_dict()
    assert list(back) == ref_keys
    for k in ref_keys:
        assert torch.equal(back[k].cpu(), ref_state[k]), k
    x = torch.randn(4, 3, 224, 224, device="cuda")
    dst.eval()
    with torch.no_grad():
        p1 = dst(x)
    src.load_state_dict(ck["state_dict"])
    src.eval()
    with torch.no_grad():
        assert torch.equal(p1, src(x))
    # --pretrained (train.py:174-181): everything but the regressor
    torch.manual_seed(10)
    third = DataParallelEngine(resnet50(fds=True, **FDS_KW).cuda(), amp_dtype=torch.bfloat16, channels_last=True)
    lin_before = third.module.linear.weight.detach().clone()
    backbone = {k: v for k, v in ck["state_dict"].items() if "linear" not in k and "fc" not in k}
    assert len(backbone) == len(ref_keys) - 2
    third.load_state_dict(backbone, strict=False)
    assert torch.equal(third.module.linear.weight, lin_before)
    assert torch.equal(third.module.layer3[2].conv2.weight.cpu(), ref_state["module.layer3.2.conv2.weight"])


def test_fds_reset_restores_initial_tables():
    """FDS.reset (fds.py:69-76): the 7 statistic buffers back to their initial values in place, `epoch` untouched."""
    from dirhip.fds import FDS
    f = FDS(64, bucket_num=30, bucket_start=3, start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=2, momentum=0.9).cuda()
    g = torch.Generator(device="cuda").manual_seed(2)
    lab = torch.randint(0, 34, (500,), device="cuda", generator=g).float()
    for ep in range(2):
        f.update_last_epoch_stats(ep)
        f.update_running_stats(torch.rand(500, 64, device="cuda", generator=g) + 0.01 * lab[:, None], lab, ep)
    assert float(f.num_samples_tracked.sum()) > 0 and float(f.epoch) == 1.0
    ptrs = {b: getattr(f, b).data_ptr() for b in FDS_BUFFERS}
    x = torch.rand(16, 64, device="cuda", generator=g)
    before = x.clone()
    y = f.smooth(x, lab[:16, None].contiguous(), 2)
    assert not torch.equal(y, before)                                     # calibration is live
    f.reset()
    assert float(f.epoch) == 1.0
    for b in FDS_BUFFERS[1:]:
        t = getattr(f, b)
        assert t.data_ptr() == ptrs[b], b                                 # in place (the reference uses zero_ / fill_)
        want = 1.0 if "var" in b else 0.0
        assert torch.all(t == want), b
    x2 = before.clone()
    assert torch.equal(f.smooth(x2, lab[:16, None].contiguous(), 2), before)      # mean 0 / var 1 tables: the identity again


def test_epoch_tail_leaves_padded_rows_out_of_the_statistics():
    """wrap-around duplicates that pad a data-parallel shard run through the network with their batch but must
    not be counted by FDS.update_running_stats."""
    from dirhip.parallel import DataParallelEngine
    from dirhip.resnet import resnet50
    from dirhip.train_loop import EpochFeatures, epoch_tail
    torch.manual_seed(3)
    model = resnet50(fds=True, **FDS_KW).cuda()
    eng = DataParallelEngine(model, amp_dtype=torch.bfloat16, channels_last=True)
    eng.train()
    x = torch.randn(8, 3, 224, 224, device="cuda")
    y = torch.tensor([[20.0], [20.0], [31.0], [31.0], [31.0], [47.0], [47.0], [20.0]], device="cuda")
    valid = torch.tensor([True] * 6 + [False] * 2)
    store = EpochFeatures(8, 2048, x.device)
    epoch_tail(eng, [(x, y, valid)], 0, store)
    tracked = model.FDS.num_samples_tracked.cpu().numpy()
    assert tracked[20] == 2 and tracked[31] == 3 and tracked[47] == 1 and tracked.sum() == 6


# ---- two processes on the one GPU, gloo on DEVICE tensors ----------------------------------------------------------
def _fds_worker(rank, world, port, tmp):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from dirhip.fds import FDS
    d = np.load(os.path.join(tmp, "fds_in.npz"))
    f = FDS(64, bucket_num=30, bucket_start=3, start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=2, momentum=0.9).cuda()
    for ep in range(2):
        feats, labels = torch.tensor(d[f"feats{ep}"]).cuda(), torch.tensor(d[f"labels{ep}"]).cuda()
        mine = torch.tensor(d[f"owner{ep}"]) == rank
        f.update_last_epoch_stats(ep)
        f.update_running_stats(feats[mine.cuda()].contiguous(), labels[mine.cuda()].contiguous(), ep)
    torch.cuda.synchronize()
    np.savez(os.path.join(tmp, f"fds_out{rank}.npz"), **{b: getattr(f, b).cpu().numpy() for b in FDS_BUFFERS})
    dist.destroy_process_group()


def test_two_rank_fds_update_equals_single_process_bit_for_bit(tmp_path):
    """FDS.update_running_stats on a batch split over two ranks (device tensors, gloo) == one process on the whole batch,
    for all 8 buffers on both ranks — including the boundary lumping of SURVEY A.3 when the boundary label lives on ONE rank
    only (the presence flags are MAX-reduced across ranks) and a bin that only one rank sees."""
    import torch.multiprocessing as mp
    from dirhip.fds import FDS
    rng = np.random.default_rng(12)
    data = {}
    for ep in range(2):
        n = 900
        labels = np.clip(np.round(np.abs(rng.normal(0, 8, n)) + 3), 0, 34).astype(np.float32)
        labels[:6] = [1, 2, 3, 29, 31, 33]                   # below / at bucket_start, at / above bucket_num - 1
        owner = rng.integers(0, 2, n)
        owner[:6] = [1, 1, 0, 0, 1, 1]                       # label 3 (lower boundary) and 29 (upper boundary) only on rank 0
        owner[labels == 17] = 1                              # a bin only rank 1 sees
        feats = (np.abs(rng.normal(0, 1, (n, 64))) * 0.5 + 0.01 * labels[:, None]).astype(np.float32)
        feats[:, 5] = 0.25                                   # constant column: variance exactly 0 after the merge
        data.update({f"feats{ep}": feats, f"labels{ep}": labels, f"owner{ep}": owner})
    np.savez(tmp_path / "fds_in.npz", **data)
    port = 33000 + int(rng.integers(0, 2000))
    mp.spawn(_fds_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    f = FDS(64, bucket_num=30, bucket_start=3, start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=2, momentum=0.9).cuda()
    f.sync_across_ranks = False
    for ep in range(2):
        f.update_last_epoch_stats(ep)
        f.update_running_stats(torch.tensor(data[f"feats{ep}"]).cuda(), torch.tensor(data[f"labels{ep}"]).cuda(), ep)
    o0, o1 = np.load(tmp_path / "fds_out0.npz"), np.load(tmp_path / "fds_out1.npz")
    for b in FDS_BUFFERS:
        assert np.array_equal(o0[b], o1[b]), f"ranks disagree on {b}"
        single = getattr(f, b).cpu().numpy()
        if b in ("epoch", "num_samples_tracked"):
            assert np.array_equal(o0[b], single), b
        else:
            # the two-rank merge adds (count, mean, M2) triples in float64 (Chan), the single process sums all rows at once
            assert_close(o0[b], single, rtol=1e-6, atol_scale=1e-7, msg=b)
    # A.3: labels below bucket_start are lumped into bin 0 on BOTH ranks although only rank 0 holds a label == bucket_start
    assert o0["num_samples_tracked"][0] == float((data["labels0"] <= 3).sum() + (data["labels1"] <= 3).sum())
    assert np.array_equal(o0["running_var"][:, 5] == 0.0, f.running_var[:, 5].cpu().numpy() == 0.0)
    assert (o0["running_var"][:, 5] == 0.0).sum() >= 10                   # constant column: exactly zero variance survives the merge


def _train_worker(rank, world, port, tmp):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from dirhip.parallel import DataParallelEngine
    from dirhip.resnet import resnet50
    from dirhip.train_loop import resolve_loss, train_step
    torch.manual_seed(5 + rank)                               # different initial weights: rank 0's must win (broadcast)
    model = resnet50(fds=True, **FDS_KW).cuda()
    eng = DataParallelEngine(model, amp_dtype=torch.bfloat16, channels_last=True, bucket_mb=8)
    eng.train()
    opt = torch.optim.SGD(eng.parameters(), lr=1e-2, momentum=0.9)
    d = np.load(os.path.join(tmp, "train_in.npz"))
    for step in range(2):
        sl = slice(rank * 8, rank * 8 + 8)
        x, y, w = (torch.tensor(d[k][step, sl]).cuda() for k in ("x", "y", "w"))
        train_step(eng, opt, x, y, w, 0, resolve_loss("l1"))
    assert len(eng._buckets) >= 3
    torch.cuda.synchronize()
    torch.save({k: v.cpu() for k, v in model.state_dict().items()}, os.path.join(tmp, f"model{rank}.pt"))
    # ---- what the engine itself launches per step: a third step under the profiler (the saved parameters are those of two steps)
    import json
    from torch.profiler import ProfilerActivity, profile
    eng.measure_comm = True
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        sl = slice(rank * 8, rank * 8 + 8)
        x, y, w = (torch.tensor(d[k][1, sl]).cuda() for k in ("x", "y", "w"))
        train_step(eng, opt, x, y, w, 0, resolve_loss("l1"))
        torch.cuda.synchronize()
    ops = {}
    for e in prof.events():
        if e.name in ("aten::copy_", "aten::mul", "aten::mul_", "aten::add_", "aten::zero_"):
            ops[e.name] = ops.get(e.name, 0) + 1
    rep = eng.comm_report()
    rep["aten_ops_in_one_train_step"] = ops
    rep["grad_is_bucket_view"] = all(p.grad is not None and p.grad.data_ptr() == eng._buckets[eng._bucket_of[id(p)][0]].views[eng._bucket_of[id(p)][1]].data_ptr()
                                     for p in eng.parameters())
    with open(os.path.join(tmp, f"comm{rank}.json"), "w") as f:
        json.dump(rep, f)
    dist.destroy_process_group()


def test_two_rank_train_steps_equal_single_process_emulation(tmp_path):
    """Two train_steps through DataParallelEngine with the REAL fused ResNet nodes on two ranks (bucketed gradient
    all-reduce from the post-accumulate hooks) == one process that runs the two half-batches separately through the same
    replica weights and averages their gradients (each rank normalises over its own half, like the reference's DataParallel
    replicas). Deterministic kernels + a two-term sum: bit-identical parameters."""
    import torch.multiprocessing as mp
    from dirhip.resnet import resnet50
    from dirhip.loss import weighted_l1_loss
    rng = np.random.default_rng(4)
    x = rng.normal(0, 1, (2, 16, 3, 224, 224)).astype(np.float32)
    y = np.clip(np.round(np.abs(rng.normal(0, 18, (2, 16, 1))) + 20), 0, 99).astype(np.float32)
    w = rng.uniform(0.5, 1.5, (2, 16, 1)).astype(np.float32)
    np.savez(tmp_path / "train_in.npz", x=x, y=y, w=w)
    port = 35000 + int(rng.integers(0, 2000))
    mp.spawn(_train_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    m0, m1 = torch.load(tmp_path / "model0.pt"), torch.load(tmp_path / "model1.pt")
    params = [k for k in m0 if "running_" not in k and "num_batches" not in k and "FDS" not in k]
    for k in params:
        assert torch.equal(m0[k], m1[k]), f"ranks diverged on {k}"
    # the engine adds NO per-parameter work to a step: every gradient kernel wrote straight into its bucket slot (no copy_ per
    # parameter: 161 of them before), the mean over ranks costs one [B, 1] multiply (gloo) or nothing (RCCL: AVG in the collective)
    import json
    for rank in (0, 1):
        rep = json.load(open(tmp_path / f"comm{rank}.json"))
        assert rep["ranks"] == 2 and rep["grad_copies"] == 0 and rep["bucket_scale_kernels"] == 0 and rep["grad_is_bucket_view"], rep
        assert len(rep["buckets_MB"]) >= 3 and abs(sum(rep["buckets_MB"]) - 23510081 * 4 / 2 ** 20) < 0.1, rep
        ops = rep["aten_ops_in_one_train_step"]
        # (copy_: the three host -> device input copies of this worker, the bf16 input cast, and gloo's device <-> host staging of each
        # bucket — nothing per parameter)
        assert ops.get("aten::copy_", 0) <= 8 + 2 * len(rep["buckets_MB"]) and ops.get("aten::mul", 0) <= 2 and ops.get("aten::mul_", 0) == 0, rep
        assert "exposed_comm_ms_last_step" in rep
    from conftest import records_dir
    out = records_dir()
    with open(os.path.join(out, "two_rank_comm_report.json"), "w") as f:
        json.dump(json.load(open(tmp_path / "comm0.json")), f, indent=1)
    # single-process emulation: replicas = two BN-buffer sets over shared parameters
    torch.manual_seed(5)
    replicas = [resnet50(fds=True, **FDS_KW).cuda().to(memory_format=torch.channels_last) for _ in range(2)]
    replicas[1].load_state_dict(replicas[0].state_dict())
    for r in replicas:
        r.train()
    opt = torch.optim.SGD(replicas[0].parameters(), lr=1e-2, momentum=0.9)
    for step in range(2):
        grads = []
        for rank, rep in enumerate(replicas):
            sl = slice(rank * 8, rank * 8 + 8)
            xb, yb, wb = (torch.tensor(a[step, sl]).cuda() for a in (x, y, w))
            rep.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred, _ = rep(xb.contiguous(memory_format=torch.channels_last), yb, 0)
            weighted_l1_loss(pred, yb, wb).backward()
            grads.append([p.grad.detach().clone() for p in rep.parameters()])
        for p, g0, g1 in zip(replicas[0].parameters(), *grads):
            p.grad = (g0 + g1) * 0.5
        opt.step()
        with torch.no_grad():
            for p1, p0 in zip(replicas[1].parameters(), replicas[0].parameters()):
                p1.copy_(p0)
        from dirhip.conv import invalidate_weight_cache
        invalidate_weight_cache()                             # replica 1's weights were edited out of band (conv.py)
    want = replicas[0].state_dict()
    worst = 0.0
    for k in params:
        a, b = m0[k].double(), want[k].cpu().double()
        worst = max(worst, float((a - b).norm() / b.norm().clamp_min(1e-30)))
    assert worst <= 1e-6, worst
    # each rank's BatchNorm statistics are those of ITS half (no SyncBN in the reference)
    assert_close(m0["bn1.running_mean"].numpy(), replicas[0].state_dict()["bn1.running_mean"].cpu().numpy(), rtol=1e-5, atol_scale=1e-6)
    assert_close(m1["bn1.running_mean"].numpy(), replicas[1].state_dict()["bn1.running_mean"].cpu().numpy(), rtol=1e-5, atol_scale=1e-6)


def test_bench_multi_rank_control_flow_two_processes_one_gpu(tmp_path):
    """`python bench.py --gpus 2 ...` started WITHOUT a launcher (bench.py spawns its own ranks under
    torch.distributed.run), both ranks sharing the one GPU over gloo (--backend gloo --share-gpu): barrier + MAX-over-ranks
    timing, per-rank batches, gradient buckets from the real fused nodes, the FDS statistic merge in the epoch tail, one JSON
    line from rank 0 with the whole-job aggregate — and the N > 1 line carries what the N = 1 line is judged on: the in-situ
    `roofline`, `cpu_baseline`, `comm`. --steps 5 with --epoch-len 2: one tail forward per trained batch for any step count.
    (RCCL itself runs in tests/test_hip_rccl.py with one rank: the lease has one GPU.)"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    detail = str(tmp_path / "bench_detail.json")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--batch", "16",
           "--epoch-len", "2", "--backend", "gloo", "--share-gpu", "--detail", detail]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 8192, p.stdout[-2000:]          # rank 0 only, one SHORT line (a 20 KB line was not parsed)
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 5 and r["warmup"] == 2 and r["scaling"] == "weak" and r["unit"] == "images/sec"
    assert r["config"]["global_batch"] == 32 and r["config"]["parallelism"] == "dp2"
    assert r["config"]["tail_forward_batches_in_timed_region"] == 5
    assert abs(r["value"] - 5 * 16 * 2 / (r["ms_per_step"] * 5 / 1e3)) <= 1e-3 * r["value"]     # whole-job aggregate over both ranks (5 significant digits)
    assert np.isfinite(r["config"]["final_loss"])
    # N > 1 observability in the line: ranks, bucket sizes (94 MB in all) and bus bandwidths, exposed communication per step
    c = r["comm"]
    assert c["rccl_ranks"] == 2 and c["backend"] == "gloo" and len(c["bucket_MB"]) == 3 and len(c["bucket_bus_GBs"]) == 3, c
    assert abs(sum(c["bucket_MB"]) - 23510081 * 4 / 2 ** 20) < 0.1 and all(v > 0 for v in c["bucket_bus_GBs"]) and c["exposed_comm_ms_per_step"] is not None, c
    rf = r["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["achieved"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 * rf["frac"], rf
    assert rf["launches_per_step"] >= 100
    assert "cpu_baseline" not in r                                           # (rank 0 at N = 1 only: the contract)
    # ... and the untrimmed result in the detail file: no per-parameter gradient copies / bucket-wide scale kernels from the engine
    d = json.load(open(detail))
    assert d["comm"]["grad_copies_per_step"] == 0 and d["comm"]["bucket_scale_kernels"] == 0 and len(d["comm"]["buckets"]) == 3, d["comm"]
    assert "step_breakdown_in_situ" in d and "peaks" in d and d["value"] > 0
    r = d
    from conftest import records_dir
    out = records_dir()
    with open(os.path.join(out, "bench_two_ranks_one_gpu.json"), "w") as f:
        json.dump(r, f, indent=1)
