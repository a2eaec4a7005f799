"""val-MAE proxy for BASELINE.json's "val-MAE within +-0.02 of the reference" (reference metric: imdb-wiki-dir/train.py:286-335
``validate`` + :338-391 ``shot_metrics``). The real datasets are not in the image, so the comparison that CAN be made is: does the
benchmarked arithmetic (bf16 conv stack) reach the same validated MAE as the parity-exact arithmetic (float32 mode, loss bit-equal to
the reference's float32 CPU run at step 0, tests/test_hip_step0_parity.py) when both train the same learnable task with the same
schedule, initialisation and data order?

Task: image -> age through a FIXED random teacher: x = sum_k c_k(age) * P_k + N(0, 1), six fixed random low-resolution colour patterns
P_k (7 x 7, upsampled to 224 x 224) with fixed smooth coefficient functions c_k of the age; train labels drawn from the long-tailed
AgeDB-DIR train histogram (tests/golden/lds_weights.npz: in_labels_agedb), validation labels balanced over the ages present (as the
reference's curated val sets are). ResNet-50 + LDS (sqrt_inv, gaussian 5/2) + FDS (agedb defaults), l1 loss, Adam 1e-3, the drop-in
train_step / epoch_tail / validate / shot_metrics. For every seed the two modes share the initial weights and the batch order.

Round 5: arms instead of two modes — ``bf16`` (the product path), ``float32`` (the product's parity-exact mode), ``lib_f32`` (the
reference's own arithmetic on this GPU: the same network as plain torch modules on the vendor library's float32 kernels — tools/library_resnet.py)
and ``lib_bf16`` (that network under ``torch.autocast(bfloat16)``: bf16 in general, not this repo's graph); every arm shares FDS / LDS / loss /
Adam arithmetic, the initial weights and the batch order per seed. Deltas are paired against ``--ref`` (default lib_f32).
``--branch E0``: variance-reduced form — ONE run of the reference arm up to epoch E0 is the common starting point (weights, BatchNorm and FDS
buffers, Adam moments); every arm then trains the remaining epochs from it with the same batch order per seed, so the arms differ by their
arithmetic over the converging phase only and the paired difference is not buried under the chaotic divergence of whole trajectories.

Round 6, split arithmetics: arms ``fp32x3`` / ``fp32x2`` (whole schedule on the split-bf16 kernels) and ``mixed<E>x3`` / ``mixed<E>x2`` (split arithmetic for
epochs < E, bf16 after) — `train.py --amp fp32x2`, `--amp_switch_epoch E --amp_early fp32x2`.

Round 6: arm ``mixed<E>`` = the product's precision schedule (`train.py --amp_switch_epoch E`): ONE engine, float32 tile kernels for
epochs < E, bf16 from epoch E on; its first E epochs are bit-identical to the ``float32`` arm's (deterministic kernels, same seed), so the paired
difference measures exactly what the switch costs.

    python tools/valmae_proxy.py --seeds 10 --epochs 24 --decay-at 16 --arms float32,mixed8,bf16 --ref float32 --out valmae_mixed
    python tools/valmae_proxy.py --seeds 10 --epochs 24 --decay-at 16 --batch 256 --arms bf16,lib_f32,lib_bf16 [--branch 16] [--out name]
    ->  gpurun_out/<name>.json (commit as profiles/rNN_<name>.json)"""
import argparse
import copy
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def make_task(device, n_train, n_val, seed=1234):
    g = torch.Generator().manual_seed(seed)
    rng = np.random.default_rng(seed)
    agedb = np.load(os.path.join(ROOT, "tests", "golden", "lds_weights.npz"))["in_labels_agedb"].astype(np.float32)
    y_train = rng.choice(agedb, n_train, replace=True).astype(np.float32)
    ages = np.unique(agedb)
    y_val = rng.choice(ages, n_val, replace=True).astype(np.float32)
    k = 6
    pats = torch.nn.functional.interpolate(torch.randn(k, 3, 7, 7, generator=g), size=(224, 224), mode="bilinear", align_corners=False).to(device)
    freq = torch.tensor(rng.uniform(0.5, 2.0, k), dtype=torch.float32, device=device)
    phase = torch.tensor(rng.uniform(0, 6.28, k), dtype=torch.float32, device=device)

    def images(y, seed_x):
        yt = torch.as_tensor(y, device=device)
        a = yt[:, None] / 100.0
        c = torch.sin(6.2831853 * freq[None, :] * a + phase[None, :])
        c[:, 0] = 2.0 * a[:, 0] - 1.0                                   # one monotone channel: the task is learnable in a few hundred steps
        out = torch.empty(len(y), 3, 224, 224, device=device)
        gx = torch.Generator(device=device).manual_seed(seed_x)
        for s in range(0, len(y), 512):
            e = slice(s, s + 512)
            out[e] = torch.einsum("bk,kchw->bchw", c[e], pats) + torch.randn(out[e].shape, device=device, generator=gx)
        return out.contiguous(memory_format=torch.channels_last)
    return y_train, y_val, images(y_train, seed + 1), images(y_val, seed + 2)


ARMS = ("bf16", "float32", "float32_sepstats", "fp32x3", "fp32x2", "lib_f32", "lib_bf16")


def switch_epoch_of(arm):
    """``mixed<E>`` / ``mixed<E>x3`` / ``mixed<E>x2``: float32 (exact / split-bf16 x3 / x2) for epochs < E, bf16 after."""
    return int(arm[5:].split("x")[0]) if arm.startswith("mixed") else None


def arith_of(arm):
    """Arithmetic of the arm's float32 epochs: "exact" (float32, mixed<E>), "x3" / "x2" (fp32x3 / fp32x2, mixed<E>x3 / mixed<E>x2)."""
    return "x3" if arm.endswith("x3") else ("x2" if arm.endswith("x2") else "exact")


def build_arm(arm, seed, device, init_state=None):
    """Model wrapper (call / .module / .parameters / .train / .eval) + optimizer for one arm; identical initial weights per seed."""
    from dirhip.optim import Adam
    from dirhip.parallel import DataParallelEngine
    from dirhip.resnet import resnet50
    torch.manual_seed(seed)
    model = resnet50(fds=True, bucket_num=100, bucket_start=3, start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=2, momentum=0.9).to(device)
    if init_state is not None:
        model.load_state_dict(init_state["model"])
    if arm in ("bf16", "float32", "float32_sepstats", "fp32x3", "fp32x2") or arm.startswith("mixed"):
        eng = DataParallelEngine(model, amp_dtype=torch.bfloat16 if arm == "bf16" else None, channels_last=True, f32_arith=arith_of(arm))
        opt = Adam(eng.parameters(), lr=1e-3)
    else:
        from library_resnet import AutocastModel, LibraryResNet50
        lib = LibraryResNet50(fds=model.FDS).to(device)
        lib.load_state_dict(model.state_dict())                          # same weights, BatchNorm and FDS buffers
        lib.to(memory_format=torch.channels_last)
        eng = AutocastModel(lib, torch.bfloat16 if arm == "lib_bf16" else None)
        opt = torch.optim.Adam(eng.parameters(), lr=1e-3)                  # (dirhip.optim.Adam is bit-equal to it: tests/test_hip_optim.py)
    if init_state is not None and init_state.get("optimizer") is not None:
        opt.load_state_dict(init_state["optimizer"])
    return eng, opt


def train_epochs(eng, opt, task, seed, epoch0, epochs, batch, decay_at, lr0=1e-3, switch_epoch=None):
    from dirhip import lds
    from dirhip.train_loop import EpochFeatures, epoch_tail, resolve_loss, train_step
    y_train, y_val, x_train, x_val = task
    device = x_train.device
    w_all = torch.as_tensor(np.asarray(lds.prepare_weights(y_train, "sqrt_inv", lds=True, lds_kernel="gaussian", lds_ks=5, lds_sigma=2), dtype=np.float32),
                            device=device).view(-1, 1)
    yt = torch.as_tensor(y_train, device=device).view(-1, 1)
    loss_fn = resolve_loss("l1")
    n = len(y_train)
    store = EpochFeatures(n, 2048, device)
    order_gen = torch.Generator().manual_seed(10_000 + seed)
    for epoch in range(epochs):
        perm = torch.randperm(n, generator=order_gen).to(device)          # (drawn for every epoch, also the skipped ones: the order of epoch e is the same in every form)
        if epoch < epoch0:
            continue
        for gp in opt.param_groups:                                       # the reference's step schedule (train.py: adjust_learning_rate, x0.1), once
            gp["lr"] = lr0 * (0.1 if (decay_at and epoch >= decay_at) else 1.0)
        if switch_epoch is not None:                                      # train.py --amp_switch_epoch
            eng.set_amp_dtype(None if epoch < switch_epoch else torch.bfloat16)
        eng.train()
        idx = [perm[s:s + batch] for s in range(0, n - batch + 1, batch)]
        for ix in idx:
            train_step(eng, opt, x_train[ix], yt[ix], w_all[ix], epoch, loss_fn)
        epoch_tail(eng, ((x_train[ix], yt[ix]) for ix in idx), epoch, store)


def evaluate(eng, task, batch):
    from dirhip.train_main import shot_metrics, validate
    y_train, y_val, x_train, x_val = task
    device = x_val.device
    yv = torch.as_tensor(y_val, device=device).view(-1, 1)

    def val_batches():
        for s in range(0, len(y_val), batch):
            yield x_val[s:s + batch], yv[s:s + batch], None
    args = types.SimpleNamespace(print_freq=10 ** 9)
    mse, l1, gm = validate(val_batches, (len(y_val) + batch - 1) // batch, eng, args, train_labels=y_train, prefix="Val")
    eng.eval()
    with torch.no_grad():
        preds = torch.cat([eng(x_val[s:s + batch]).float() for s in range(0, len(y_val), batch)]).view(-1).cpu().numpy()
    shots = shot_metrics(preds, y_val, y_train)
    return {"all": float(l1), "mse": float(mse), "gmean": float(gm), "many": float(shots["many"]["l1"]), "med": float(shots["median"]["l1"]),
            "few": float(shots["low"]["l1"])}


def snapshot(eng, opt):
    mod = eng.module
    return {"model": {k: v.detach().clone() for k, v in mod.state_dict().items()}, "optimizer": copy.deepcopy(opt.state_dict())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=5)
    ap.add_argument("--seed0", type=int, default=0)
    ap.add_argument("--epochs", type=int, default=12)
    ap.add_argument("--decay-at", type=int, default=8)
    ap.add_argument("--n-train", type=int, default=8192)
    ap.add_argument("--n-val", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--arms", default="bf16,lib_f32,lib_bf16")
    ap.add_argument("--ref", default="lib_f32")
    ap.add_argument("--branch", type=int, default=0, help="epoch at which the arms branch off ONE run of the reference arm (0 = whole schedules)")
    ap.add_argument("--out", default="valmae_proxy")
    a = ap.parse_args()
    arms = [x for x in a.arms.split(",") if x]
    assert all(x in ARMS or x.startswith("mixed") for x in arms)
    assert a.ref in arms or not a.branch, "--branch needs its reference arm in --arms"       # (whole schedules: an absent --ref only drops the paired deltas;
                                                                                              #  the arms are deterministic per seed, pair them with an earlier run's)
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    task = make_task(device, a.n_train, a.n_val)
    res = {x: [] for x in arms}
    t0 = time.time()
    start = None
    if a.branch:
        eng, opt = build_arm(a.ref, 1000, device)
        train_epochs(eng, opt, task, 1000, 0, a.branch, a.batch, a.decay_at)
        start = snapshot(eng, opt)
        start["metrics_at_branch"] = evaluate(eng, task, a.batch)
        print(f"common start ({a.ref}, {a.branch} epochs): {start['metrics_at_branch']}", flush=True)
        del eng, opt
    for seed in range(a.seed0, a.seed0 + a.seeds):
        for arm in arms:
            t1 = time.time()
            # float32_sepstats: the float32 graph with the BatchNorm statistics from a separate pass over y (the build before the float32 tile kernels formed them
            # in their store loop): the same arithmetic class, another float32 summation order — the arm that measures what the proxy can resolve
            import dirhip.conv_f32 as _cf
            if not hasattr(_cf, "_stats_fusable_product"):
                _cf._stats_fusable_product = _cf.stats_fusable
            _cf.stats_fusable = (lambda x, w: False) if arm == "float32_sepstats" else _cf._stats_fusable_product
            eng, opt = build_arm(arm, seed if not a.branch else 1000, device, init_state=start)
            train_epochs(eng, opt, task, seed, a.branch, a.epochs, a.batch, a.decay_at, switch_epoch=switch_epoch_of(arm))
            r = evaluate(eng, task, a.batch)
            r["seconds"] = time.time() - t1
            res[arm].append(r)
            print(f"seed {seed} {arm}: {r}", flush=True)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", a.out + ".partial.json"), "w") as f:     # (a run cut short still leaves its seeds)
                json.dump({"args": vars(a), "per_seed": res}, f)
            del eng, opt
            torch.cuda.empty_cache()
    keys = ("all", "many", "med", "few")
    out = {"task": "synthetic teacher task (tools/valmae_proxy.py): 224x224 images = fixed random low-resolution patterns with age-dependent coefficients + N(0,1) noise; "
                   f"{a.n_train} train labels from the AgeDB-DIR train histogram (long-tailed), {a.n_val} balanced validation labels; ResNet-50 + LDS + FDS, l1, Adam 1e-3, "
                   f"batch {a.batch}, {a.epochs} epochs" + (f" (lr x0.1 from epoch {a.decay_at})" if a.decay_at else "") + ", the drop-in train_step / epoch_tail / validate / shot_metrics",
           "arms": {"bf16": "the product path (this repo's bf16 graph)", "float32": "the product's parity-exact float32 mode",
                    "lib_f32": "the reference's arithmetic on this GPU: plain torch modules, vendor-library float32 kernels (tools/library_resnet.py)",
                    "lib_bf16": "the same library network under torch.autocast(bfloat16)",
                    "float32_sepstats": "float32 with the BatchNorm statistics from a separate pass (the build before commit 1521e9c): same arithmetic class, other summation order",
                    "fp32x3": "the product's float32 graph on the split-bf16 x3 kernels (train.py --amp fp32x3)",
                    "fp32x2": "the product's float32 graph on the split-bf16 x2 kernels (train.py --amp fp32x2)",
                    **{x: f"the product's precision schedule (train.py --amp_switch_epoch {switch_epoch_of(x)} --amp_early {'fp32' if arith_of(x) == 'exact' else 'fp32' + arith_of(x)}): "
                          f"float32 tile kernels ({arith_of(x)}) for epochs < {switch_epoch_of(x)}, bf16 after"
                       for x in arms if x.startswith("mixed")}},
           "form": (f"branch: every arm continues ONE {a.ref} run from epoch {a.branch} (weights, buffers, Adam moments), same batch order per seed" if a.branch
                    else "whole schedules, same initial weights and batch order per seed"),
           "seeds": a.seeds, "seed0": a.seed0, "reference_arm": a.ref, "per_seed": res,
           "metric": "validation L1 (= val MAE, years): all / many-shot / median-shot / few-shot (train.py:286-391)"}
    if start is not None:
        out["metrics_at_branch"] = start["metrics_at_branch"]
    out["summary"] = {arm: {k: {"mean": float(np.mean([r[k] for r in rows])), "std": float(np.std([r[k] for r in rows], ddof=1)) if len(rows) > 1 else 0.0}
                            for k in keys} for arm, rows in res.items()}
    out["paired_delta_vs_reference"] = {}
    for arm in arms:
        if arm == a.ref or a.ref not in res:
            continue
        d = {}
        for k in keys:
            v = np.array([x[k] - y[k] for x, y in zip(res[arm], res[a.ref])])
            d[k] = {"mean": float(v.mean()), "std": float(v.std(ddof=1)) if len(v) > 1 else 0.0, "stderr": float(v.std(ddof=1) / np.sqrt(len(v))) if len(v) > 1 else 0.0}
        out["paired_delta_vs_reference"][arm] = d
    if "bf16" in arms and "lib_bf16" in arms:
        v = np.array([x["all"] - y["all"] for x, y in zip(res["bf16"], res["lib_bf16"])])
        out["paired_delta_bf16_minus_lib_bf16_all"] = {"mean": float(v.mean()), "std": float(v.std(ddof=1)) if len(v) > 1 else 0.0,
                                                       "stderr": float(v.std(ddof=1) / np.sqrt(len(v))) if len(v) > 1 else 0.0}
    out["wall_seconds"] = time.time() - t0
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", a.out + ".json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in out if k not in ("per_seed", "task", "arms")}, indent=1))


if __name__ == "__main__":
    main()
