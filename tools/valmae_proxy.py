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

    python tools/valmae_proxy.py [seeds=5] [epochs=4] [n_train=8192] [batch=64] [lr decay at epoch, 0 = none]   ->  gpurun_out/valmae_proxy.json (commit as profiles/rNN_valmae_proxy.json)"""
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


def run_one(mode, seed, task, epochs, batch, device, decay_at=0):
    from dirhip import lds
    from dirhip.optim import Adam
    from dirhip.parallel import DataParallelEngine
    from dirhip.resnet import resnet50
    from dirhip.train_loop import EpochFeatures, epoch_tail, resolve_loss, train_step
    from dirhip.train_main import shot_metrics, validate
    y_train, y_val, x_train, x_val = task
    torch.manual_seed(seed)
    model = resnet50(fds=True, bucket_num=100, bucket_start=3, start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=2, momentum=0.9).to(device)
    eng = DataParallelEngine(model, amp_dtype=torch.bfloat16 if mode == "bf16" else None, channels_last=True)
    opt = Adam(eng.parameters(), lr=1e-3)
    w_all = torch.as_tensor(np.asarray(lds.prepare_weights(y_train, "sqrt_inv", lds=True, lds_kernel="gaussian", lds_ks=5, lds_sigma=2), dtype=np.float32),
                            device=device).view(-1, 1)
    yt = torch.as_tensor(y_train, device=device).view(-1, 1)
    yv = torch.as_tensor(y_val, device=device).view(-1, 1)
    loss_fn = resolve_loss("l1")
    n = len(y_train)
    store = EpochFeatures(n, 2048, device)
    order_gen = torch.Generator().manual_seed(10_000 + seed)
    for epoch in range(epochs):
        if decay_at and epoch == decay_at:                            # the reference's step schedule (train.py: adjust_learning_rate, x0.1), once
            for gp in opt.param_groups:
                gp["lr"] = gp["lr"] * 0.1
        eng.train()
        perm = torch.randperm(n, generator=order_gen).to(device)
        idx = [perm[s:s + batch] for s in range(0, n - batch + 1, batch)]
        for ix in idx:
            train_step(eng, opt, x_train[ix], yt[ix], w_all[ix], epoch, loss_fn)
        epoch_tail(eng, ((x_train[ix], yt[ix]) for ix in idx), epoch, store)

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


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    n_train = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 64
    decay_at = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    task = make_task(device, n_train, 2048)
    res = {"bf16": [], "float32": []}
    t0 = time.time()
    for seed in range(seeds):
        for mode in ("bf16", "float32"):
            t1 = time.time()
            r = run_one(mode, seed, task, epochs, batch, device, decay_at)
            r["seconds"] = time.time() - t1
            res[mode].append(r)
            print(f"seed {seed} {mode}: {r}", flush=True)
    out = {"task": "synthetic teacher task (tools/valmae_proxy.py): 224x224 images = fixed random low-resolution patterns with age-dependent coefficients + N(0,1) noise; "
                   f"{n_train} train labels from the AgeDB-DIR train histogram (long-tailed), 2048 balanced validation labels; ResNet-50 + LDS + FDS, l1, Adam 1e-3, "
                   f"batch {batch}, {epochs} epochs" + (f" (lr x0.1 from epoch {decay_at})" if decay_at else "") + ", the drop-in train_step / epoch_tail / validate / shot_metrics",
           "seeds": seeds, "per_seed": res, "metric": "validation L1 (= val MAE, years): all / many-shot / median-shot / few-shot (train.py:286-391)"}
    summ = {}
    for mode, rows in res.items():
        summ[mode] = {k: {"mean": float(np.mean([r[k] for r in rows])), "std": float(np.std([r[k] for r in rows], ddof=1)) if len(rows) > 1 else 0.0}
                      for k in ("all", "many", "med", "few")}
    out["summary"] = summ
    d = {k: summ["bf16"][k]["mean"] - summ["float32"][k]["mean"] for k in ("all", "many", "med", "few")}
    sig = {k: float(np.sqrt((summ["bf16"][k]["std"] ** 2 + summ["float32"][k]["std"] ** 2) / 2)) for k in d}
    paired = np.array([a["all"] - b["all"] for a, b in zip(res["bf16"], res["float32"])])
    out["delta_mean_bf16_minus_float32"] = d
    out["seed_sigma"] = sig
    out["paired_delta_all"] = {"mean": float(paired.mean()), "std": float(paired.std(ddof=1)) if len(paired) > 1 else 0.0,
                               "stderr": float(paired.std(ddof=1) / np.sqrt(len(paired))) if len(paired) > 1 else 0.0}
    out["statement"] = {k: ("|delta mean| <= 0.02" if abs(d[k]) <= 0.02 else
                            ("inside one seed-sigma" if abs(d[k]) <= sig[k] else "OUTSIDE one seed-sigma")) + f" (delta {d[k]:+.4f}, seed sigma {sig[k]:.4f})"
                        for k in d}
    out["wall_seconds"] = time.time() - t0
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "valmae_proxy.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ("summary", "delta_mean_bf16_minus_float32", "seed_sigma", "paired_delta_all", "statement", "wall_seconds")}, indent=1))


if __name__ == "__main__":
    main()
