#!/usr/bin/env python
"""Round-3 golden vectors, produced by RUNNING THE UPSTREAM REFERENCE in the build container (needs /root/reference):

    python tests/golden/gen_golden_r3.py [b64emul] [b256]

  step0_b64_bf16emul.npz   the reference's own resnet50 + FDS + weighted_l1_loss modules (the run of step0_b64.npz: same
                           seeds, same inputs) executed in float64 with a round-to-bfloat16 inserted at exactly the points
                           where the bf16 product path stores bfloat16: the input image, every convolution weight, every
                           convolution output, relu(bn1(.)) / relu(bn2(.)) of every block and of the stem, and every block
                           output relu(bn3(.) + shortcut). (bn3 / the downsample BatchNorm are NOT rounded on their own: the
                           product's join kernel adds them in float32.) Gradients flowing through those points are rounded
                           the same way (the product stores activation gradients in bfloat16). The pool / FDS / linear /
                           loss tail stays unrounded, as in the product. This is the bf16 product path's arithmetic with
                           exact accumulation: what the hand-written MFMA kernels must reproduce to bf16-rounding-flip noise.
  step0_b256.npz           BASELINE configs[1]'s own batch size: the float32 and the float64 run of the reference at B=256
                           (as step0_b64.npz, sampled), plus the bf16-rounding-emulated float64 run.

Inputs are regenerated from seeds by the tests (torch CPU generators; versions in MANIFEST_r3.json).
"""
import json
import os
import sys
import time

import numpy as np
import scipy
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from oracle import refshim  # noqa: E402
import gen_golden_r2 as r2  # noqa: E402

torch.set_num_threads(os.cpu_count() or 8)

FDS_KW = r2.FDS_KW
STEP0_B256 = dict(seed_model=41, seed_x=42, seed_lab=43, seed_fds=44, batch=256, n_fds=4000, n_samples=512, epoch=2)
ENC_COLS = 512          # sampled encoding columns stored for the B=256 golden


from oracle.bf16_emul import bf16_points  # noqa: E402


def inputs(c):
    saved = dict(r2.STEP0)
    r2.STEP0.update(c)
    try:
        return r2.step0_inputs()
    finally:
        r2.STEP0.clear()
        r2.STEP0.update(saved)


def checkpoint_blocks(model):
    """Recompute each Bottleneck in the backward pass instead of keeping its activations (B=256 in float64 would need
    > 60 GB otherwise). Same arithmetic; BatchNorm running statistics are blended twice, so they are read right after the forward."""
    from torch.utils.checkpoint import checkpoint
    for m in model.modules():
        if type(m).__name__ == "Bottleneck":
            m.forward = (lambda x, f=m.forward: checkpoint(f, x, use_reentrant=False))


def run(c, dtype, emulate, ckpt=False):
    ref = refshim.load("imdb-wiki-dir")
    torch.manual_seed(c["seed_model"])
    model = refshim.make_resnet50("imdb-wiki-dir", fds=True, **FDS_KW)
    x, y, w, rounds, tail = inputs(c)
    with refshim.cuda_identity():
        for ep, (f, l) in enumerate(rounds):
            model.FDS.update_last_epoch_stats(ep)
            model.FDS.update_running_stats(f, l, ep)
    if dtype != torch.float32:
        model = model.to(dtype)
        model.FDS.kernel_window = model.FDS.kernel_window.to(dtype)
    if emulate:
        bf16_points(model)
        x = x.bfloat16().float()
    if ckpt:
        checkpoint_blocks(model)
    model.train()
    xin = x.to(dtype)
    if ckpt:
        xin.requires_grad_(True)          # (non-reentrant checkpoints need a graph input; the gradient is discarded)
    with refshim.cuda_identity():
        pred, enc = model(xin, y.to(dtype), c["epoch"])
    bn_stats = dict(bn1_running_mean=model.bn1.running_mean.detach().clone(), bn1_running_var=model.bn1.running_var.detach().clone(),
                    l4_bn3_running_var=model.layer4[2].bn3.running_var.detach().clone())
    loss = ref.loss.weighted_l1_loss(pred, y.to(dtype), w.to(dtype))
    model.zero_grad()
    loss.backward()
    model.bn_stats_after_forward = bn_stats
    return model, loss, pred, enc, (x, y, w)


def sampled_grads(model, c, idxs=None):
    names, norms, samples, out_idx = [], [], [], []
    for i, (name, p) in enumerate(model.named_parameters()):
        gflat = p.grad.detach().reshape(-1)
        idx = r2.sample_indices(gflat.numel(), c["n_samples"], 1000 + i) if idxs is None else idxs[i][idxs[i] >= 0]
        pad = np.zeros(c["n_samples"], np.float64)
        pad[:len(idx)] = gflat[torch.from_numpy(idx)].double().numpy()
        ipad = np.full(c["n_samples"], -1, np.int64)
        ipad[:len(idx)] = idx
        names.append(name); norms.append(float(gflat.double().norm())); samples.append(pad); out_idx.append(ipad)
    return np.array(names), np.array(norms), np.stack(samples), np.stack(out_idx)


def gen_b64_emul():
    c = dict(r2.STEP0)
    t0 = time.time()
    m, loss, pred, enc, (x, y, w) = run(c, torch.float64, True)
    names, norms, samples, idx = sampled_grads(m, c)
    out = dict(emul_loss=np.array(loss.item()), emul_pred=pred.detach().numpy(), emul_encoding=enc.detach().numpy().astype(np.float32),
               param_names=names, emul_grad_norms=norms, emul_grad_samples=samples, grad_sample_idx=idx,
               emul_linear_weight_grad=m.linear.weight.grad.numpy().copy(), emul_linear_bias_grad=m.linear.bias.grad.numpy().copy(),
               emul_bn1_running_mean=m.bn_stats_after_forward["bn1_running_mean"].numpy(), emul_bn1_running_var=m.bn_stats_after_forward["bn1_running_var"].numpy(),
               emul_l4_bn3_running_var=m.bn_stats_after_forward["l4_bn3_running_var"].numpy(),
               in_labels=y.numpy(), in_weights=w.numpy(), config=np.array(json.dumps(dict(c, **FDS_KW))))
    # the SAME emulation with float32 arithmetic between the rounding points: two evaluations of one bf16 graph that differ only in
    # accumulation precision. A bf16 rounding turns a relative difference d << 2^-8 into sqrt(d * 2^-8) (a flipped rounding is a
    # whole ulp), so such differences grow towards the bf16 ulp within a few layers whatever their origin, and the network's depth
    # amplifies them further: this pair measures the resolution of ANY whole-network comparison of bf16 evaluations.
    m32, loss32, pred32, enc32, _ = run(c, torch.float32, True)
    _, n32, s32, _ = sampled_grads(m32, c, idx)
    out.update(emul32_loss=np.array(loss32.item(), dtype=np.float64), emul32_pred=pred32.detach().numpy(), emul32_encoding=enc32.detach().numpy(),
               emul32_grad_norms=n32, emul32_grad_samples=s32, emul32_linear_weight_grad=m32.linear.weight.grad.numpy().copy(),
               emul32_l4_bn3_running_var=m32.bn_stats_after_forward["l4_bn3_running_var"].numpy())
    np.savez_compressed(os.path.join(HERE, "step0_b64_bf16emul.npz"), **out)
    print(f"b64 emul: loss {loss.item():.9f} (float32 arithmetic: {loss32.item():.9f})  ({time.time() - t0:.0f} s)")


def gen_b256():
    c = STEP0_B256
    t0 = time.time()
    m32, loss32, pred32, enc32, (x, y, w) = run(c, torch.float32, False, ckpt=True)
    names, n32, s32, idx = sampled_grads(m32, c)
    cols = np.sort(np.random.default_rng(4242).choice(2048, ENC_COLS, replace=False)).astype(np.int64)
    out = dict(ref_loss=np.array(loss32.item(), dtype=np.float64), ref_pred=pred32.detach().numpy(),
               ref_encoding_cols=enc32.detach().numpy()[:, cols], encoding_cols=cols,
               ref_encoding_rowsum=enc32.detach().double().sum(1).numpy(),
               param_names=names, ref_grad_norms=n32, ref_grad_samples=s32.astype(np.float32), grad_sample_idx=idx,
               ref_bn1_running_mean=m32.bn_stats_after_forward["bn1_running_mean"].numpy(), ref_bn1_running_var=m32.bn_stats_after_forward["bn1_running_var"].numpy(),
               ref_l4_bn3_running_var=m32.bn_stats_after_forward["l4_bn3_running_var"].numpy(),
               in_labels=y.numpy(), in_weights=w.numpy())
    print(f"b256 float32: loss {loss32.item():.9f}  ({time.time() - t0:.0f} s)", flush=True)
    g32 = {n: p.grad.detach().double().reshape(-1).clone() for n, p in m32.named_parameters()}
    del m32
    m64, loss64, pred64, enc64, _ = run(c, torch.float64, False, ckpt=True)
    _, n64, s64, _ = sampled_grads(m64, c, idx)
    r32 = [float((g32[n] - p.grad.detach().reshape(-1)).norm() / p.grad.detach().norm()) for n, p in m64.named_parameters()]
    out.update(ref64_loss=np.array(loss64.item()), ref64_grad_samples=s64, ref64_grad_norms=n64, ref32_vs_ref64_grad_rel_l2=np.array(r32),
               ref32_vs_ref64_encoding_rel_l2=np.array(float((enc32.detach().double() - enc64.detach()).norm() / enc64.detach().norm())),
               ref64_encoding_cols=enc64.detach().numpy()[:, cols].astype(np.float32), ref64_pred=pred64.detach().numpy())
    print(f"b256 float64: loss {loss64.item():.12f}  ({time.time() - t0:.0f} s)", flush=True)
    del m64, g32
    me, losse, prede, ence, _ = run(c, torch.float64, True, ckpt=True)
    _, ne, se, _ = sampled_grads(me, c, idx)
    out.update(emul_loss=np.array(losse.item()), emul_pred=prede.detach().numpy(), emul_encoding_cols=ence.detach().numpy()[:, cols].astype(np.float32),
               emul_encoding_rowsum=ence.detach().sum(1).numpy(), emul_grad_norms=ne, emul_grad_samples=se,
               emul_linear_weight_grad=me.linear.weight.grad.numpy().copy(), emul_linear_bias_grad=me.linear.bias.grad.numpy().copy())
    out["config"] = np.array(json.dumps(dict(c, **FDS_KW)))
    np.savez_compressed(os.path.join(HERE, "step0_b256.npz"), **out)
    print(f"b256 emul: loss {losse.item():.9f}  ({time.time() - t0:.0f} s)")


def gen_b256_emul32():
    """Adds the float32-arithmetic emulation (see gen_b64_emul) to step0_b256.npz."""
    c = STEP0_B256
    path = os.path.join(HERE, "step0_b256.npz")
    old = dict(np.load(path, allow_pickle=False))
    t0 = time.time()
    m32, loss32, pred32, enc32, _ = run(c, torch.float32, True, ckpt=True)
    cols = old["encoding_cols"]
    _, n32, s32, _ = sampled_grads(m32, c, old["grad_sample_idx"])       # the noise floor of the per-tensor gradient norms / samples
    old.update(emul32_loss=np.array(loss32.item(), dtype=np.float64), emul32_pred=pred32.detach().numpy(),
               emul32_encoding_cols=enc32.detach().numpy()[:, cols], emul32_encoding_rowsum=enc32.detach().double().sum(1).numpy(),
               emul32_linear_weight_grad=m32.linear.weight.grad.numpy().copy(), emul32_grad_norms=n32, emul32_grad_samples=s32.astype(np.float32))
    np.savez_compressed(path, **old)
    print(f"b256 emul32: loss {loss32.item():.9f}  ({time.time() - t0:.0f} s)")


MULTI = dict(seed_model=51, seed_x=52, seed_lab=53, seed_fds=54, batch=32, n_fds=4000, steps=4, lr=1e-3, momentum=0.9, weight_decay=1e-4)


def multistep_inputs(c):
    g = torch.Generator().manual_seed(c["seed_x"])
    xs = [torch.randn(c["batch"], 3, 224, 224, generator=g) for _ in range(c["steps"])]
    x_eval = torch.randn(16, 3, 224, 224, generator=g)
    rng = np.random.default_rng(c["seed_lab"])
    ys = [torch.tensor(r2.long_tail(rng, c["batch"])).view(-1, 1) for _ in range(c["steps"])]
    ws = [torch.tensor(rng.uniform(0.5, 1.5, c["batch"]).astype(np.float32)).view(-1, 1) for _ in range(c["steps"])]
    rounds = []
    for ep in range(2):
        rr = np.random.default_rng(c["seed_fds"] + ep)
        lab = r2.long_tail(rr, c["n_fds"])
        feats = (np.abs(rr.normal(0, 1, (c["n_fds"], 2048))) * 0.5 + 0.01 * lab[:, None]).astype(np.float32)
        rounds.append((torch.tensor(feats), torch.tensor(lab)))
    return xs, ys, ws, x_eval, rounds


def gen_multistep():
    """multistep_b32.npz: FOUR optimizer steps of the reference (train.py:246-262: resnet50 + live FDS calibration + weighted_l1_loss +
    SGD with momentum and weight decay, lr 1e-3 — steps small enough that float32 round-off does not decide the trajectory), then
    an eval-mode forward of a held-out batch (BatchNorm running statistics of all four steps in use) and the epoch tail on the
    last batch: pins the interplay of the optimizer, the bf16/float32 weight caches, BatchNorm running statistics and FDS over
    several steps, not just step 0. Run in float32 (the reference) and float64 (what float32 can resolve)."""
    c = MULTI
    ref = refshim.load("imdb-wiki-dir")
    xs, ys, ws, x_eval, rounds = multistep_inputs(c)
    out = {}
    for dtype, tag in ((torch.float32, "ref"), (torch.float64, "ref64")):
        torch.manual_seed(c["seed_model"])
        model = refshim.make_resnet50("imdb-wiki-dir", fds=True, **FDS_KW)
        with refshim.cuda_identity():
            for ep, (f, l) in enumerate(rounds):
                model.FDS.update_last_epoch_stats(ep)
                model.FDS.update_running_stats(f, l, ep)
        if dtype != torch.float32:
            model = model.to(dtype)
            model.FDS.kernel_window = model.FDS.kernel_window.to(dtype)
        opt = torch.optim.SGD(model.parameters(), lr=c["lr"], momentum=c["momentum"], weight_decay=c["weight_decay"])
        losses = []
        model.train()
        for x, y, w in zip(xs, ys, ws):
            with refshim.cuda_identity():
                pred, enc = model(x.to(dtype), y.to(dtype), 2)
            loss = ref.loss.weighted_l1_loss(pred, y.to(dtype), w.to(dtype))
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(loss.item())
        with torch.no_grad(), refshim.cuda_identity():                      # train.py:269-281 on the last batch
            _, feat = model(xs[-1].to(dtype), ys[-1].to(dtype), 2)
            model.FDS.update_last_epoch_stats(2)
            model.FDS.update_running_stats(feat.squeeze().float() if dtype == torch.float32 else feat.squeeze(), ys[-1].squeeze().to(dtype), 2)
        model.eval()
        with torch.no_grad():
            pe = model(x_eval.to(dtype))
        out[f"{tag}_losses"] = np.array(losses, dtype=np.float64)
        out[f"{tag}_pred_eval"] = pe.double().numpy()
        out[f"{tag}_bn1_running_mean"] = model.bn1.running_mean.double().numpy()
        out[f"{tag}_bn1_running_var"] = model.bn1.running_var.double().numpy()
        out[f"{tag}_l3_bn2_running_var"] = model.layer3[2].bn2.running_var.double().numpy()
        out[f"{tag}_l4_bn3_running_mean"] = model.layer4[2].bn3.running_mean.double().numpy()
        out[f"{tag}_linear_weight"] = model.linear.weight.detach().double().numpy()
        out[f"{tag}_l1_conv1_weight_sample"] = model.layer1[0].conv1.weight.detach().double().reshape(-1)[:512].numpy()
        out[f"{tag}_fds_tracked"] = model.FDS.num_samples_tracked.double().numpy()
        out[f"{tag}_fds_running_mean_sum"] = np.array(float(model.FDS.running_mean.double().sum()))
        print(tag, losses, flush=True)
    out["in_labels"] = np.stack([y.numpy() for y in ys])
    out["in_weights"] = np.stack([w.numpy() for w in ws])
    out["config"] = np.array(json.dumps(dict(c, **FDS_KW)))
    np.savez_compressed(os.path.join(HERE, "multistep_b32.npz"), **out)


def main():
    which = sys.argv[1:] or ["b64emul", "b256", "b256emul32", "multistep"]
    if "b64emul" in which:
        gen_b64_emul()
    if "b256" in which:
        gen_b256()
    if "b256emul32" in which:
        gen_b256_emul32()
    if "multistep" in which:
        gen_multistep()
    manifest = {"torch": torch.__version__, "numpy": np.__version__, "scipy": scipy.__version__,
                "files": ["step0_b64_bf16emul.npz", "step0_b256.npz", "multistep_b32.npz"], "generator": "tests/golden/gen_golden_r3.py"}
    with open(os.path.join(HERE, "MANIFEST_r3.json"), "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()
