#!/usr/bin/env python
"""Round-2 golden vectors, again produced by RUNNING THE UPSTREAM REFERENCE in the build container
(needs /root/reference):  python tests/golden/gen_golden_r2.py

  step0_b64.npz   one non-chaotic training step of the reference's own resnet50 + FDS + weighted_l1_loss on CPU fp32 at
                  B=64, epoch 2 with FDS tables populated by two reference update rounds (so FDS.smooth is live): the loss,
                  the predictions, the calibrated encoding, and for each of the 161 parameter tensors the gradient's L2 norm
                  plus 512 sampled elements (fixed seeded indices). Also the FDS buffers after the reference's epoch tail
                  on GIVEN features (update_last_epoch_stats(2) + update_running_stats(features, labels, 2)).
  validate.npz    the reference's own `validate` / `shot_metrics` (imdb-wiki-dir/train.py:286-391, extracted from the
                  file with `ast` — train.py itself cannot be imported) on seeded predictions.

Inputs that are too large to store are regenerated from seeds (torch CPU generators; versions in MANIFEST_r2.json).
"""
import ast
import json
import os
import sys
import types

import numpy as np
import scipy
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import refshim  # noqa: E402

torch.set_num_threads(8)

FDS_KW = dict(bucket_num=100, bucket_start=0, start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=2, momentum=0.9)
STEP0 = dict(seed_model=31, seed_x=32, seed_lab=33, seed_fds=34, batch=64, n_fds=4000, n_samples=512, epoch=2)


def long_tail(rng, n):
    return np.clip(np.round(np.abs(rng.normal(0, 18, n)) + 20), 0, 120).astype(np.float32)


def step0_inputs():
    """Everything the GPU test regenerates from seeds (must stay in sync with tests/test_hip_step0_parity.py)."""
    c = STEP0
    g = torch.Generator().manual_seed(c["seed_x"])
    x = torch.randn(c["batch"], 3, 224, 224, generator=g)
    rng = np.random.default_rng(c["seed_lab"])
    y = torch.tensor(long_tail(rng, c["batch"])).view(-1, 1)
    w = torch.tensor(rng.uniform(0.5, 1.5, c["batch"]).astype(np.float32)).view(-1, 1)
    rounds = []
    for ep in range(2):
        rr = np.random.default_rng(c["seed_fds"] + ep)
        lab = long_tail(rr, c["n_fds"])
        feats = (np.abs(rr.normal(0, 1, (c["n_fds"], 2048))) * 0.5 + 0.01 * lab[:, None]).astype(np.float32)
        rounds.append((torch.tensor(feats), torch.tensor(lab)))
    rr = np.random.default_rng(c["seed_fds"] + 7)
    lab_t = long_tail(rr, c["n_fds"])
    feats_t = (np.abs(rr.normal(0, 1, (c["n_fds"], 2048))) * 0.4 + 0.012 * lab_t[:, None]).astype(np.float32)
    return x, y, w, rounds, (torch.tensor(feats_t), torch.tensor(lab_t))


def sample_indices(numel, n, seed):
    rng = np.random.default_rng(seed)
    if numel <= n:
        return np.arange(numel, dtype=np.int64)
    return np.sort(rng.choice(numel, n, replace=False)).astype(np.int64)


def _step0_run(dtype):
    """The reference's resnet50 + FDS + weighted_l1_loss, one forward/backward in `dtype` (float32 = what the reference
    runs; float64 = the same modules cast to double: the yardstick for what float32 can resolve)."""
    ref = refshim.load("imdb-wiki-dir")
    c = STEP0
    torch.manual_seed(c["seed_model"])
    model = refshim.make_resnet50("imdb-wiki-dir", fds=True, **FDS_KW)
    x, y, w, rounds, tail = step0_inputs()
    with refshim.cuda_identity():
        for ep, (f, l) in enumerate(rounds):
            model.FDS.update_last_epoch_stats(ep)
            model.FDS.update_running_stats(f, l, ep)
    if dtype != torch.float32:
        model = model.to(dtype)
        model.FDS.kernel_window = model.FDS.kernel_window.to(dtype)
    model.train()
    with refshim.cuda_identity():
        pred, enc = model(x.to(dtype), y.to(dtype), c["epoch"])
    loss = ref.loss.weighted_l1_loss(pred, y.to(dtype), w.to(dtype))
    model.zero_grad()
    loss.backward()
    return model, loss, pred, enc, (x, y, w, tail)


def gen_step0():
    c = STEP0
    model, loss, pred, enc, (x, y, w, (feats_t, lab_t)) = _step0_run(torch.float32)
    out = dict(ref_loss=np.array(loss.item(), dtype=np.float64), ref_pred=pred.detach().numpy(), ref_encoding=enc.detach().numpy(),
               in_labels=y.numpy(), in_weights=w.numpy())
    names, norms, samples, idxs = [], [], [], []
    for i, (name, p) in enumerate(model.named_parameters()):
        gflat = p.grad.detach().reshape(-1)
        idx = sample_indices(gflat.numel(), c["n_samples"], 1000 + i)
        names.append(name)
        norms.append(float(gflat.double().norm()))
        pad = np.zeros(c["n_samples"], np.float32)
        pad[:len(idx)] = gflat[torch.from_numpy(idx)].numpy()
        ipad = np.full(c["n_samples"], -1, np.int64)
        ipad[:len(idx)] = idx
        samples.append(pad)
        idxs.append(ipad)
    out.update(param_names=np.array(names), ref_grad_norms=np.array(norms), ref_grad_samples=np.stack(samples),
               grad_sample_idx=np.stack(idxs))
    # BatchNorm running statistics after the one train-mode forward (momentum blend of batch statistics)
    out["ref_bn1_running_mean"] = model.bn1.running_mean.numpy().copy()
    out["ref_bn1_running_var"] = model.bn1.running_var.numpy().copy()
    out["ref_l4_bn3_running_var"] = model.layer4[2].bn3.running_var.numpy().copy()
    # epoch tail on GIVEN features (train.py:280-281)
    with refshim.cuda_identity():
        model.FDS.update_last_epoch_stats(c["epoch"])
        model.FDS.update_running_stats(feats_t, lab_t, c["epoch"])
    bins = np.array([20, 21, 25, 33, 47, 60, 85, 99])
    F = model.FDS
    out.update(tail_bins=bins, ref_tail_running_mean=F.running_mean[bins].numpy().copy(), ref_tail_running_var=F.running_var[bins].numpy().copy(),
               ref_tail_smoothed_mean=F.smoothed_mean_last_epoch[bins].numpy().copy(),
               ref_tail_smoothed_var=F.smoothed_var_last_epoch[bins].numpy().copy(),
               ref_tail_tracked=F.num_samples_tracked.numpy().copy(), ref_tail_epoch=F.epoch.numpy().copy(),
               ref_tail_sum_running_mean=np.array(float(F.running_mean.double().sum())),
               ref_tail_sum_running_var=np.array(float(F.running_var.double().sum())))
    # ---- the same reference modules in float64: what the float32 run can resolve. ReLU masks flip under 1e-5 forward
    # noise, so float32 GRADIENTS of this 53-layer network carry percent-level noise (measured below) while the loss and the
    # activations agree to 1e-5; the GPU test holds our float32 gradients to the reference's own float32-vs-float64 error.
    m64, loss64, pred64, enc64, _ = _step0_run(torch.float64)
    s64, n64, r32 = [], [], []
    p32 = dict(model.named_parameters())
    for i, (name, p) in enumerate(m64.named_parameters()):
        gflat = p.grad.detach().reshape(-1)
        idx = idxs[i][idxs[i] >= 0]
        pad = np.zeros(c["n_samples"], np.float64)
        pad[:len(idx)] = gflat[torch.from_numpy(idx)].numpy()
        s64.append(pad)
        n64.append(float(gflat.norm()))
        g32 = p32[name].grad.detach().double().reshape(-1)
        r32.append(float((g32 - gflat).norm() / gflat.norm()))
    out.update(ref64_loss=np.array(loss64.item()), ref64_grad_samples=np.stack(s64), ref64_grad_norms=np.array(n64),
               ref32_vs_ref64_grad_rel_l2=np.array(r32),
               ref32_vs_ref64_encoding_rel_l2=np.array(float((enc.detach().double() - enc64.detach()).norm() / enc64.detach().norm())),
               ref64_encoding=enc64.detach().numpy().astype(np.float32))
    out["config"] = np.array(json.dumps(dict(STEP0, **FDS_KW)))
    np.savez_compressed(os.path.join(HERE, "step0_b64.npz"), **out)
    print("step0: loss", loss.item(), loss64.item(), "grad norms", norms[0], norms[-2], norms[-1],
          "fp32-vs-fp64 grad rel l2 median/max", float(np.median(r32)), float(np.max(r32)))


def reference_functions(names):
    """Compile the named top-level functions of the reference's imdb-wiki-dir/train.py into a private namespace
    (train.py cannot be imported: argparse / folder creation / logging at import time)."""
    import logging
    import time
    from collections import defaultdict
    import torch.nn as nn
    from scipy.stats import gmean
    ref = refshim.load("imdb-wiki-dir")
    path = os.path.join(refshim.REFERENCE_ROOT, "imdb-wiki-dir", "train.py")
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(body) == len(names), [n.name for n in body]
    ns = dict(torch=torch, nn=nn, np=np, gmean=gmean, defaultdict=defaultdict, time=time, print=logging.info,
              AverageMeter=ref.utils.AverageMeter, ProgressMeter=ref.utils.ProgressMeter,
              args=types.SimpleNamespace(print_freq=3))
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


class _Echo(torch.nn.Module):
    """Stand-in network for `validate`: the 'image' batch carries the prediction in its first element per sample."""

    def forward(self, inputs):
        return inputs.reshape(inputs.shape[0], -1)[:, :1].clone()


def gen_validate():
    ns = reference_functions(("validate", "shot_metrics"))
    rng = np.random.default_rng(77)
    train_labels = long_tail(rng, 6000)
    cases = {}
    for ci, (n, bs, noise) in enumerate(((1000, 64, 6.0), (333, 100, 15.0), (150, 7, 2.0))):
        labels = long_tail(rng, n)
        labels[:3] = [100.0, 110.0, 5.0]                 # every shot class must be present: the reference's shot_metrics raises on an empty one
        preds = (labels + rng.normal(0, noise, n)).astype(np.float32)
        loader = []
        for i in range(0, n, bs):
            inp = torch.zeros(len(labels[i:i + bs]), 3, 2, 2)
            inp[:, 0, 0, 0] = torch.tensor(preds[i:i + bs])
            loader.append((inp, torch.tensor(labels[i:i + bs]).view(-1, 1), torch.ones(len(labels[i:i + bs]), 1)))
        with refshim.cuda_identity():
            mse, l1, gm = ns["validate"](loader, _Echo(), train_labels=train_labels)
        sd = ns["shot_metrics"](preds, labels, train_labels)
        cases[f"in_preds_{ci}"] = preds
        cases[f"in_labels_{ci}"] = labels
        cases[f"batch_{ci}"] = np.array(bs)
        cases[f"ref_validate_{ci}"] = np.array([mse, l1, gm], dtype=np.float64)
        cases[f"ref_shot_{ci}"] = np.array([[sd[s][k] for k in ("mse", "l1", "gmean")] for s in ("many", "median", "low")], dtype=np.float64)
    cases["in_train_labels"] = train_labels
    np.savez_compressed(os.path.join(HERE, "validate.npz"), **cases)
    print("validate:", cases["ref_validate_0"], cases["ref_shot_0"])


def gen_nyud2_lds():
    """nyud2_lds_weights.npz: the reference's depthDataset._get_bucket_weights / get_bin_idx / _get_weights
    (nyud2-dir/loaddata.py:29-69), compiled out of the file with `ast` (the module imports torchvision at the top)."""
    import logging
    from scipy.ndimage import convolve1d
    path = os.path.join(refshim.REFERENCE_ROOT, "nyud2-dir", "loaddata.py")
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if (isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "TRAIN_BUCKET_NUM")
            or (isinstance(n, ast.ClassDef) and n.name == "depthDataset")]
    assert len(body) == 2
    nyu = refshim.load_nyud2()
    ns = dict(torch=torch, np=np, logging=logging, convolve1d=convolve1d, get_lds_kernel_window=nyu.util.get_lds_kernel_window,
              Dataset=object, pd=None, os=os)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    cls = ns["depthDataset"]
    out = {"ref_train_bucket_num": np.asarray(ns["TRAIN_BUCKET_NUM"], dtype=np.int64)}
    configs = [("sqrt_inv", True, "gaussian", 5, 2), ("inverse", True, "gaussian", 5, 2), ("sqrt_inv", False, "gaussian", 5, 2),
               ("inverse", False, "gaussian", 5, 2), ("inverse", True, "triang", 9, 1), ("sqrt_inv", True, "laplace", 7, 1.5)]
    rng = np.random.default_rng(21)
    depth = rng.uniform(0.0, 10.5, (2, 1, 19, 23)).astype(np.float32)
    depth[0, 0, 0, :6] = [0.7, 0.69999, 9.9, 9.95, 10.0, 0.0]
    out["in_depth"] = depth
    for ci, (rw, lds, k, ks, s) in enumerate(configs):
        args = types.SimpleNamespace(reweight=rw, lds=lds, lds_kernel=k, lds_ks=ks, lds_sigma=s, bucket_num=100, bucket_start=7)
        ds = cls.__new__(cls)
        bw = ds._get_bucket_weights(args)
        assert all(type(x) is np.float32 for x in bw)
        ds.bucket_weights = bw
        out[f"ref_bucket_weights_{ci}"] = np.asarray(bw, dtype=np.float32)
        out[f"ref_pixel_weights_{ci}"] = ds._get_weights(torch.tensor(depth)).numpy()
    out["configs"] = np.array([f"{rw},{int(lds)},{k},{ks},{s}" for rw, lds, k, ks, s in configs])
    ds = cls.__new__(cls)
    ds.bucket_weights = None
    out["ref_pixel_weights_none"] = ds._get_weights(torch.tensor(depth)).numpy()
    np.savez_compressed(os.path.join(HERE, "nyud2_lds_weights.npz"), **out)
    print("nyud2 lds:", out["ref_bucket_weights_0"][:10])


def main():
    which = sys.argv[1:] or ["step0", "validate", "nyud2"]
    if "nyud2" in which:
        gen_nyud2_lds()
    if "validate" in which:
        gen_validate()
    if "step0" in which:
        gen_step0()
    manifest = {"torch": torch.__version__, "numpy": np.__version__, "scipy": scipy.__version__,
                "files": ["step0_b64.npz", "validate.npz", "nyud2_lds_weights.npz"], "generator": "tests/golden/gen_golden_r2.py"}
    with open(os.path.join(HERE, "MANIFEST_r2.json"), "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()
