#!/usr/bin/env python
"""Generate the golden vectors in this directory by RUNNING THE UPSTREAM REFERENCE.

Run in the build container only (needs /root/reference):  python tests/golden/gen_golden.py
The reference modules are imported through ``oracle/refshim.py`` (``.cuda()`` -> identity,
stub torchvision). Outputs are committed as small ``.npz`` fixtures; the GPU box never
sees /root/reference, only these files. Environment the vectors were produced under is
recorded in ``MANIFEST.json`` (torch / numpy / scipy versions = the de-facto pin of the
reference's unpinned dependencies).

Every array named ``ref_*`` is an output of reference code; every ``in_*`` is an input.
"""
import hashlib
import json
import os
import sys

import numpy as np
import scipy
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import refshim  # noqa: E402

torch.set_num_threads(4)
BUFFERS = ("epoch", "running_mean", "running_var", "running_mean_last_epoch",
           "running_var_last_epoch", "smoothed_mean_last_epoch",
           "smoothed_var_last_epoch", "num_samples_tracked")
WINDOW_GRID = [("gaussian", 5, 2), ("gaussian", 5, 1), ("gaussian", 9, 1), ("gaussian", 9, 2),
               ("gaussian", 3, 0.5), ("triang", 5, 1), ("triang", 9, 2), ("laplace", 5, 2),
               ("laplace", 9, 1.5), ("gaussian", 1, 2)]


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    return path


def gen_windows(ref):
    out = {}
    for i, (k, ks, s) in enumerate(WINDOW_GRID):
        with refshim.cuda_identity():
            out[f"ref_fds_{i}"] = ref.fds.FDS._get_kernel_window(k, ks, s).numpy()
        out[f"ref_lds_{i}"] = np.asarray(ref.utils.get_lds_kernel_window(k, ks, s), dtype=np.float64)
    out["grid"] = np.array([f"{k},{ks},{s}" for k, ks, s in WINDOW_GRID])
    save("windows.npz", **out)


LDS_CONFIGS = [("sqrt_inv", True, "gaussian", 5, 2), ("sqrt_inv", False, "gaussian", 5, 2),
               ("inverse", False, "gaussian", 5, 2), ("inverse", True, "gaussian", 9, 1),
               ("inverse", True, "triang", 9, 1), ("sqrt_inv", True, "laplace", 7, 1.5),
               ("sqrt_inv", True, "gaussian", 9, 1), ("none", False, "gaussian", 5, 2)]


def gen_lds():
    import pandas as pd
    df = pd.read_csv(os.path.join(refshim.REFERENCE_ROOT, "agedb-dir/data/agedb.csv"))
    ages = df[df["split"] == "train"]["age"].values.astype(np.int64)
    rng = np.random.default_rng(7)
    # synthetic long-tailed IMDB-WIKI-like label set incl. ages > 120 (clipped to bin 120)
    synth = np.clip(np.round(np.abs(rng.normal(0, 18, 20000)) + 20), 0, 140).astype(np.int64)
    synth[:5] = [0, 120, 121, 139, 1]
    frac = np.round(rng.uniform(0, 125, 3000), 1)          # non-integer labels -> int() truncation
    sets = {"agedb": ages, "synth": synth, "frac": frac, "tiny": np.array([5, 5, 7], dtype=np.int64)}
    out = {}
    for sname, labels in sets.items():
        out[f"in_labels_{sname}"] = labels
        for ci, (rw, lds, k, ks, s) in enumerate(LDS_CONFIGS):
            w = refshim.prepare_weights(labels, reweight=rw, lds=lds, lds_kernel=k, lds_ks=ks, lds_sigma=s)
            if w is None:
                out[f"ref_w_{sname}_{ci}"] = np.zeros((0,), np.float32)
                continue
            assert all(type(x) is np.float32 for x in w[:4]), type(w[0])
            out[f"ref_w_{sname}_{ci}"] = np.asarray(w, dtype=np.float32)
    out["configs"] = np.array([f"{rw},{int(lds)},{k},{ks},{s}" for rw, lds, k, ks, s in LDS_CONFIGS])
    save("lds_weights.npz", **out)
    # known-answer prefixes quoted in SURVEY.md §8c
    w0 = out["ref_w_agedb_0"]
    return hashlib.sha256(w0.tobytes()).hexdigest()[:16]


def make_epoch_inputs(rng, n, c, start, num, epoch, variant):
    """Synthetic labels/features exercising the quirks of SURVEY Appendix A."""
    labels = np.clip(np.round(np.abs(rng.normal(0, 0.25 * num, n)) + start), 0, num + 6)
    if variant == "boundary_absent":          # A.3: out-of-range rows but no boundary label
        labels[labels == start] = start + 1
        labels[labels == num - 1] = num - 2
        labels[:6] = [start - 1, start - 2, num, num + 3, start + 2, start + 2]
    elif variant == "boundary_present":
        labels[:8] = [start, start - 1, start - 2, num - 1, num, num + 3, start, num - 1]
    labels = np.maximum(labels, 0).astype(np.float32)
    feats = (np.abs(rng.normal(0, 1, (n, c))) * 0.5 + 0.01 * labels[:, None]).astype(np.float32)
    feats[:, 3] = 0.0                          # dead channel: zero variance, zero mean (A.9)
    feats[:, 5] = 0.37109375                   # constant non-zero column (exact in f32)
    one = start + 7                            # single-sample bin (A.9): biased var = 0
    labels[labels == one] = one + 1
    labels[n // 2] = one
    if epoch == 2:
        feats[:, 8] = F32(1.0) + F32(1e-4) * rng.normal(0, 1, n).astype(np.float32)   # tight column
    return feats, labels


F32 = np.float32
FDS_CASES = [
    dict(name="imdb", subdir="imdb-wiki-dir", n=500, c=40, variant="plain",
         kw=dict(bucket_num=100, bucket_start=0, start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=2, momentum=0.9)),
    dict(name="agedb", subdir="agedb-dir", n=400, c=24, variant="boundary_present",
         kw=dict(bucket_num=30, bucket_start=3, start_update=0, start_smooth=1, kernel="gaussian", ks=9, sigma=1, momentum=0.9)),
    dict(name="absent", subdir="agedb-dir", n=300, c=16, variant="boundary_absent",
         kw=dict(bucket_num=24, bucket_start=3, start_update=0, start_smooth=1, kernel="triang", ks=5, sigma=1, momentum=0.9)),
    dict(name="nomomentum", subdir="imdb-wiki-dir", n=300, c=16, variant="boundary_present",
         kw=dict(bucket_num=20, bucket_start=0, start_update=1, start_smooth=2, kernel="laplace", ks=3, sigma=1.5, momentum=None)),
]
N_EPOCHS = 5
SMOOTH_B = 48


def gen_fds_traces():
    for case in FDS_CASES:
        rng = np.random.default_rng(sum(map(ord, case["name"])))
        kw = dict(feature_dim=case["c"], **case["kw"])
        R = refshim.make_fds(case["subdir"], **kw)
        out = {"kw": np.array(json.dumps(kw))}
        for epoch in range(N_EPOCHS):
            feats, labels = make_epoch_inputs(rng, case["n"], case["c"], kw["bucket_start"], kw["bucket_num"], epoch, case["variant"])
            # --- smooth() on the first SMOOTH_B rows (fwd + autograd) with the tables as they are now
            xb = feats[:SMOOTH_B].copy()
            lb = labels[:SMOOTH_B, None].copy()
            gy = rng.normal(0, 1, xb.shape).astype(np.float32)
            xt = torch.tensor(xb, requires_grad=True)
            with refshim.cuda_identity():
                yt = R.smooth(xt.clone(), torch.tensor(lb), epoch)
            if yt.requires_grad:
                yt.backward(torch.tensor(gy))
                gx = xt.grad.numpy().copy()
            else:
                gx = gy.copy()
            out[f"e{epoch}_in_x"] = xb
            out[f"e{epoch}_in_labels_b"] = lb
            out[f"e{epoch}_in_gy"] = gy
            out[f"e{epoch}_ref_smooth"] = yt.detach().numpy().copy()
            out[f"e{epoch}_ref_gx"] = gx
            for k in BUFFERS:       # tables smooth() saw (lets a checker feed identical tables)
                out[f"e{epoch}_pre_{k}"] = getattr(R, k).detach().numpy().copy()
            # --- epoch tail: train.py:280-281
            with refshim.cuda_identity():
                R.update_last_epoch_stats(epoch)
                for k in BUFFERS:
                    out[f"e{epoch}_mid_{k}"] = getattr(R, k).detach().numpy().copy()
                R.update_running_stats(torch.tensor(feats), torch.tensor(labels), epoch)
            out[f"e{epoch}_in_feats"] = feats
            out[f"e{epoch}_in_labels"] = labels
            out[f"e{epoch}_alias"] = np.array(R.running_mean_last_epoch is R.running_mean)
            for k in BUFFERS:
                out[f"e{epoch}_post_{k}"] = getattr(R, k).detach().numpy().copy()
        save(f"fds_trace_{case['name']}.npz", **out)


def gen_bins():
    """A.3 examples + randoms: row -> bin index as implied by which rows the reference touches.
    Derived by running reference ``smooth`` with tables that shift every touched row by (bin+1)."""
    ref_cases = [([1, 2, 5, 5, 11, 12], 3, 10), ([1, 2, 3, 9, 11, 12], 3, 10), ([0, 99, 100, 150, 50], 0, 100),
                 ([0, 98, 100, 150, 50], 0, 100), ([4.5, 4.0, 8.99, 9.0, 9.5, 3.0, 2.5], 3, 10)]
    rng = np.random.default_rng(3)
    for _ in range(6):
        start, num = int(rng.integers(0, 4)), int(rng.integers(8, 40))
        ref_cases.append((list(rng.integers(0, num + 4, 64).astype(float)), start, num))
    out = {}
    for i, (lab, start, num) in enumerate(ref_cases):
        nb = num - start
        R = refshim.make_fds("imdb-wiki-dir", feature_dim=2, bucket_num=num, bucket_start=start)
        R.running_mean_last_epoch.zero_(); R.running_var_last_epoch.fill_(1)
        R.smoothed_var_last_epoch.fill_(1)
        R.smoothed_mean_last_epoch.copy_(torch.arange(1, nb + 1, dtype=torch.float32)[:, None].expand(nb, 2))
        lab_t = torch.tensor(lab, dtype=torch.float32)[:, None]
        with refshim.cuda_identity():
            y = R.smooth(torch.zeros(len(lab), 2), lab_t, 5)
        out[f"in_labels_{i}"] = lab_t.numpy()
        out[f"params_{i}"] = np.array([start, num])
        out[f"ref_bins_{i}"] = (y[:, 0].numpy().round().astype(np.int32) - 1)
    out["n"] = np.array(len(ref_cases))
    save("bin_index.npz", **out)


def gen_calibrate(ref):
    rng = np.random.default_rng(11)
    out = {}
    c = 32
    cases = []
    m1, m2 = rng.normal(0, 1, c).astype(np.float32), rng.normal(0, 1, c).astype(np.float32)
    v1, v2 = rng.uniform(0.01, 2, c).astype(np.float32), rng.uniform(0.01, 2, c).astype(np.float32)
    cases.append((m1, v1, m2, v2, 0.1, 10))                      # all columns
    v1z = v1.copy(); v1z[[2, 9]] = 0
    cases.append((m1, v1z, m2, v2, 0.1, 10))                     # some v1 == 0
    cases.append((m1, np.full(c, 1e-13, np.float32), m2, v2, 0.1, 10))   # sum(v1) < 1e-10
    v2w = v2.copy(); v2w[:4] = [1e-6, 1e6, -1.0, 0.0]
    cases.append((m1, v1, m2, v2w, 0.1, 10))                     # both clip edges, negative v2
    cases.append((m1, v1, m2, v2, 0.5, 2))                       # STS-B style clip
    for i, (a, b, cc, d, lo, hi) in enumerate(cases):
        x = rng.normal(0, 1, (9, c)).astype(np.float32)
        y = ref.utils.calibrate_mean_var(torch.tensor(x.copy()), torch.tensor(a), torch.tensor(b),
                                         torch.tensor(cc), torch.tensor(d), lo, hi)
        out.update({f"in_x_{i}": x, f"in_m1_{i}": a, f"in_v1_{i}": b, f"in_m2_{i}": cc, f"in_v2_{i}": d,
                    f"clip_{i}": np.array([lo, hi]), f"ref_y_{i}": y.numpy()})
    out["n"] = np.array(len(cases))
    save("calibrate.npz", **out)


LOSS_VARIANTS = [("mse", {}), ("l1", {}), ("focal_mse", {}), ("focal_mse", {"activate": "tanh", "beta": 0.3, "gamma": 2}),
                 ("focal_l1", {}), ("focal_l1", {"activate": "tanh", "beta": 0.3, "gamma": 2}),
                 ("huber", {}), ("huber", {"beta": 0.5})]


def gen_losses(ref):
    rng = np.random.default_rng(5)
    out = {}
    for b in (1, 8, 256, 1000):
        x = rng.normal(35, 12, (b, 1)).astype(np.float32)
        y = np.round(rng.uniform(0, 100, (b, 1))).astype(np.float32)
        w = rng.uniform(0.2, 4, (b, 1)).astype(np.float32)
        x[0] = y[0]                                  # exact zero error: sign(0)/abs'(0) conventions
        out[f"in_x_{b}"], out[f"in_y_{b}"], out[f"in_w_{b}"] = x, y, w
        for vi, (kind, extra) in enumerate(LOSS_VARIANTS):
            for use_w in (0, 1):
                xt = torch.tensor(x, requires_grad=True)
                fn = getattr(ref.loss, f"weighted_{kind}_loss")
                loss = fn(xt, torch.tensor(y), torch.tensor(w) if use_w else None, **extra)
                loss.backward()
                out[f"ref_loss_{b}_{vi}_{use_w}"] = loss.detach().numpy()
                out[f"ref_grad_{b}_{vi}_{use_w}"] = xt.grad.numpy().copy()
    out["variants"] = np.array([json.dumps([k, e]) for k, e in LOSS_VARIANTS])
    save("losses.npz", **out)


def gen_resnet():
    """Seeded reference resnet50 forward on CPU fp32 (resnet.py:127-153): pins architecture,
    init order (same torch RNG stream) and the (pred, encoding) contract."""
    torch.manual_seed(1234)
    kw = dict(fds=True, bucket_num=100, bucket_start=0, start_update=0, start_smooth=1,
              kernel="gaussian", ks=5, sigma=2, momentum=0.9)
    model = refshim.make_resnet50("imdb-wiki-dir", **kw)
    g = torch.Generator().manual_seed(99)
    x = torch.randn(2, 3, 224, 224, generator=g)
    t = torch.tensor([[31.0], [64.0]])
    sd = model.state_dict()
    keys = list(sd.keys())
    sums = np.array([float(v.double().sum()) for v in sd.values()])       # right after construction
    model.eval()
    with torch.no_grad():
        pred_eval = model(x)                                               # fresh BN running stats
    model.train()
    with torch.no_grad(), refshim.cuda_identity():
        pred, enc = model(x, t, 0)                                         # batch statistics; epoch 0 -> no smoothing
    save("resnet50_forward.npz", ref_pred_train=pred.numpy(), ref_enc_train=enc.numpy(),
         ref_pred_eval=pred_eval.numpy(), keys=np.array(keys), ref_param_sums=sums,
         shapes=np.array([json.dumps(list(v.shape)) for v in sd.values()]),
         n_params=np.array(sum(p.numel() for p in model.parameters())))


def gen_train_trajectory():
    """End-to-end: the reference's own resnet50 + FDS + weighted_l1_loss + Adam driven by train.py:246-262 /
    :269-281 verbatim on CPU fp32 (a) BASELINE configs[0]: AgeDB defaults, LDS only, B=8; (b) IMDB-WIKI defaults,
    LDS + FDS, 3 epochs x 2 steps of B=6 + epoch tails. Losses per step + FDS buffers of the bins that occur."""
    ref = refshim.load("agedb-dir")
    import pandas as pd
    df = pd.read_csv(os.path.join(refshim.REFERENCE_ROOT, "agedb-dir/data/agedb.csv"))
    ages = df[df["split"] == "train"]["age"].values
    w_all = np.asarray(refshim.prepare_weights(ages, "agedb-dir", reweight="sqrt_inv", lds=True, lds_kernel="gaussian", lds_ks=5, lds_sigma=2), dtype=np.float32)
    out = {}
    # ---- (a) config 1
    rows = np.random.default_rng(0).choice(len(ages), 8, replace=False)
    y = torch.tensor(ages[rows], dtype=torch.float32).view(-1, 1)
    w = torch.tensor(w_all[rows]).view(-1, 1)

    def run_a(noise):
        torch.manual_seed(11)
        model = refshim.make_resnet50("agedb-dir", fds=False, bucket_num=100, bucket_start=3, start_update=0, start_smooth=1,
                                      kernel="gaussian", ks=9, sigma=1, momentum=0.9)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        x = torch.randn(8, 3, 224, 224, generator=torch.Generator().manual_seed(0))
        if noise:
            x = x * (1 + noise * torch.randn(x.shape, generator=torch.Generator().manual_seed(5)))
        losses = []
        model.train()
        for _ in range(4):
            o = model(x, y, 0)
            loss = ref.loss.weighted_l1_loss(o, y, w)
            opt.zero_grad(); loss.backward(); opt.step()
            losses.append(loss.item())
        return np.array(losses)
    out.update(a_rows=rows, a_labels=y.numpy(), a_weights=w.numpy(), a_ref_losses=run_a(0.0), a_pert_losses=run_a(1e-6))
    # ---- (b) LDS + FDS, run twice: exact inputs and inputs perturbed by 1e-6 relative noise. ResNet-50 + BN at
    # B=6 is chaotic (the second run diverges from the first by ~1e-4 after one step and ~1e-2 after five), so the
    # perturbed run is the yardstick for what "the same trajectory" can mean across two conv libraries.
    kw = dict(bucket_num=100, bucket_start=0, start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=2, momentum=0.9)
    bins = np.array([25, 26, 27, 31, 40])
    labels = [torch.tensor([[25.0], [40.0], [25.0], [31.0], [40.0], [26.0 + i]]) for i in range(2)]

    def run_b(noise):
        torch.manual_seed(12)
        model = refshim.make_resnet50("imdb-wiki-dir", fds=True, **kw)
        # --optimizer sgd (train.py:163-164): updates proportional to the gradient (Adam's g/sqrt(g^2) first steps
        # turn 1e-6 gradient noise into +-lr sign flips, which is even more chaotic)
        opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
        g = torch.Generator().manual_seed(1)
        batches = [(torch.randn(6, 3, 224, 224, generator=g), labels[i], torch.rand(6, 1, generator=g) + 0.5) for i in range(2)]
        weights = np.stack([b[2].numpy() for b in batches])
        if noise:
            gn = torch.Generator().manual_seed(5)
            batches = [(x * (1 + noise * torch.randn(x.shape, generator=gn)), y, w) for x, y, w in batches]
        losses = []
        for epoch in range(3):
            model.train()
            for xb, yb, wb in batches:
                with refshim.cuda_identity():
                    o, _ = model(xb, yb, epoch)
                loss = ref.loss.weighted_l1_loss(o, yb, wb)
                opt.zero_grad(); loss.backward(); opt.step()
                losses.append(loss.item())
            enc, lab = [], []
            with torch.no_grad(), refshim.cuda_identity():
                for xb, yb, _ in batches:
                    _, f = model(xb, yb, epoch)
                    enc.extend(f.data.squeeze().cpu().numpy()); lab.extend(yb.data.squeeze().cpu().numpy())
                model.FDS.update_last_epoch_stats(epoch)
                model.FDS.update_running_stats(torch.from_numpy(np.vstack(enc)), torch.from_numpy(np.hstack(lab)), epoch)
        F = model.FDS
        return dict(losses=np.array(losses), weights=weights, running_mean=F.running_mean[bins].numpy().copy(),
                    running_var=F.running_var[bins].numpy().copy(), smoothed_mean=F.smoothed_mean_last_epoch[bins].numpy().copy(),
                    tracked=F.num_samples_tracked.numpy().copy(), epoch=F.epoch.numpy().copy())
    r0, r1 = run_b(0.0), run_b(1e-6)
    out.update(b_weights=r0["weights"], b_labels=np.stack([l.numpy() for l in labels]), b_bins=bins,
               b_ref_losses=r0["losses"], b_ref_running_mean=r0["running_mean"], b_ref_running_var=r0["running_var"],
               b_ref_smoothed_mean=r0["smoothed_mean"], b_ref_tracked=r0["tracked"], b_ref_epoch=r0["epoch"],
               b_pert_losses=r1["losses"], b_pert_running_mean=r1["running_mean"], b_pert_running_var=r1["running_var"],
               b_pert_smoothed_mean=r1["smoothed_mean"])
    save("train_trajectory.npz", **out)
    print("trajectory losses", out["a_ref_losses"], out["b_ref_losses"])


def gen_stsb():
    """STS-B FDS variant (sts-b-dir/fds.py + util.py:63-73): state-machine trace with empty buckets (incl. both ends and a
    run of consecutive empties), labels 0 / 5 / <0.1, and calibrate_mean_var cases with the v1 <= 0 / v2 < 0 guards."""
    import warnings
    warnings.filterwarnings("ignore")
    ref = refshim.load_stsb()
    rng = np.random.default_rng(77)
    kw = dict(feature_dim=24, bucket_num=50, bucket_start=0, start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=2, momentum=0.9)
    with refshim.cuda_identity():
        R = ref.fds.FDS(**kw)
    out = {"kw": np.array(json.dumps(kw))}
    n = 260
    for epoch in range(4):
        labels = np.round(np.clip(rng.normal(2.6, 1.1, n), 0, 5), 3).astype(np.float32)
        labels[:6] = [0.0, 5.0, 4.95, 0.0999, 0.1, 2.5]
        labels[(labels >= 1.0) & (labels < 1.3)] = 1.35          # buckets 10..12 empty (a run)
        if epoch % 2 == 0:
            labels[labels < 0.1] = 0.15                           # bucket 0 empty on even epochs
            labels[labels >= 4.9] = 4.85                          # bucket 49 empty on even epochs
        feats = (np.abs(rng.normal(0, 1, (n, 24))) * 0.5 + 0.1 * labels[:, None]).astype(np.float32)
        feats[:, 3] = 0.0
        xb, lb = feats[:40].copy(), labels[:40, None].copy()
        gy = rng.normal(0, 1, xb.shape).astype(np.float32)
        xt = torch.tensor(xb, requires_grad=True)
        with refshim.cuda_identity():
            yt = R.smooth(xt.clone(), torch.tensor(lb), epoch)
        yt.backward(torch.tensor(gy))
        out.update({f"e{epoch}_in_x": xb, f"e{epoch}_in_labels_b": lb, f"e{epoch}_in_gy": gy,
                    f"e{epoch}_ref_smooth": yt.detach().numpy().copy(), f"e{epoch}_ref_gx": xt.grad.numpy().copy()})
        for k in BUFFERS:
            out[f"e{epoch}_pre_{k}"] = getattr(R, k).detach().numpy().copy()
        with refshim.cuda_identity():
            R.update_last_epoch_stats(epoch)
            R.update_running_stats(torch.tensor(feats), torch.tensor(labels), epoch)
        out[f"e{epoch}_in_feats"], out[f"e{epoch}_in_labels"] = feats, labels
        out[f"e{epoch}_ref_buckets"] = np.array([R._get_bucket_idx(l) for l in labels])
        for k in BUFFERS:
            out[f"e{epoch}_post_{k}"] = getattr(R, k).detach().numpy().copy()
    save("fds_trace_stsb.npz", **out)
    c = 32
    m1, m2 = rng.normal(0, 1, c).astype(np.float32), rng.normal(0, 1, c).astype(np.float32)
    v1, v2 = rng.uniform(0.01, 2, c).astype(np.float32), rng.uniform(0.01, 2, c).astype(np.float32)
    v1g, v2g = v1.copy(), v2.copy()
    v1g[[1, 7]] = [0.0, -0.5]; v2g[[4, 9]] = [-1e-3, 0.0]
    cases = [(m1, v1, m2, v2, 0.5, 2.0), (m1, v1g, m2, v2g, 0.5, 2.0), (m1, np.full(c, 1e-13, np.float32), m2, v2, 0.5, 2.0),
             (m1, v1, m2, v2 * np.float32(50), 0.5, 2.0), (m1, v1g, m2, v2g, 0.2, 5.0)]
    o2 = {"n": np.array(len(cases))}
    for i, (a, b, cc, d, lo, hi) in enumerate(cases):
        x = rng.normal(0, 1, (7, c)).astype(np.float32)
        y = ref.util.calibrate_mean_var(torch.tensor(x.copy()), torch.tensor(a), torch.tensor(b), torch.tensor(cc), torch.tensor(d), lo, hi)
        o2.update({f"in_x_{i}": x, f"in_m1_{i}": a, f"in_v1_{i}": b, f"in_m2_{i}": cc, f"in_v2_{i}": d, f"clip_{i}": np.array([lo, hi]), f"ref_y_{i}": y.numpy()})
    save("calibrate_stsb.npz", **o2)


def gen_nyud2():
    """NYUD2 dense FDS variant (nyud2-dir/models/fds.py + util.py:151-162): features [B,16,9,11], depth maps [B,1,9,11],
    buckets clamp(int(10 d), 7, 29); run with device-transfer shims that create new tensors (alias breaking, like the
    real .cpu()/.cuda() round trip) and np.bool restored."""
    import warnings
    warnings.filterwarnings("ignore")
    ref = refshim.load_nyud2()
    rng = np.random.default_rng(91)
    kw = dict(feature_dim=16, bucket_num=30, bucket_start=7, start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=2, momentum=0.9)
    with refshim.device_transfer_clones():
        R = ref.fds.FDS(**kw)
    out = {"kw": np.array(json.dumps(kw))}
    for epoch in range(4):
        depth = rng.uniform(0.5, 3.4, (6, 1, 9, 11)).astype(np.float32)
        depth[0, 0, 0, :4] = [0.7, 0.6999, 2.9, 3.3]                          # below start, at the edges, above num-1
        feats = (np.abs(rng.normal(0, 1, (6, 16, 9, 11))) * 0.5 + 0.2 * depth).astype(np.float32)
        if epoch >= 2:
            feats[:, 3] = 0.0                                                   # zero-variance column -> guard branch
        xb, lb = feats[:2].copy(), depth[:2].copy()
        gy = rng.normal(0, 1, xb.shape).astype(np.float32)
        xt = torch.tensor(xb, requires_grad=True)
        with refshim.device_transfer_clones():
            yt = R.smooth(xt * 1.0, torch.tensor(lb), epoch)
        yt.backward(torch.tensor(gy))
        out.update({f"e{epoch}_in_x": xb, f"e{epoch}_in_labels_b": lb, f"e{epoch}_in_gy": gy,
                    f"e{epoch}_ref_smooth": yt.detach().numpy().copy(), f"e{epoch}_ref_gx": xt.grad.numpy().copy()})
        for k in BUFFERS:
            out[f"e{epoch}_pre_{k}"] = getattr(R, k).detach().numpy().copy()
        with refshim.device_transfer_clones():
            R.update_last_epoch_stats(epoch)
            R.update_running_stats(torch.tensor(feats), torch.tensor(depth), epoch)
        out[f"e{epoch}_in_feats"], out[f"e{epoch}_in_labels"] = feats, depth
        out[f"e{epoch}_alias"] = np.array(R.running_mean_last_epoch is R.running_mean)
        for k in BUFFERS:
            out[f"e{epoch}_post_{k}"] = getattr(R, k).detach().numpy().copy()
    save("fds_trace_nyud2.npz", **out)


def main():
    ref = refshim.load("imdb-wiki-dir")
    gen_windows(ref)
    sha = gen_lds()
    gen_fds_traces()
    gen_bins()
    gen_calibrate(ref)
    gen_losses(ref)
    gen_resnet()
    gen_train_trajectory()
    gen_stsb()
    gen_nyud2()
    manifest = {"torch": torch.__version__, "numpy": np.__version__, "scipy": scipy.__version__,
                "agedb_sqrtinv_lds_g52_sha256_prefix": sha,
                "files": sorted(f for f in os.listdir(HERE) if f.endswith(".npz"))}
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print(json.dumps(manifest, indent=1))


if __name__ == "__main__":
    main()
