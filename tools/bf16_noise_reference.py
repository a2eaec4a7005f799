"""Documentation probe (NOT product code; uses torch's library convolutions on purpose): how far apart are bf16-autocast and
float32 gradients of the reference architecture itself, at random init on the step0 golden inputs? Answers "is the O(1)
bf16-vs-float32 gradient difference of the hand-written path a property of the network or of our kernels?".
Run on the GPU box: python tools/bf16_noise_reference.py  ->  JSON on stdout."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import torch_oracle  # noqa: E402


def main():
    dev = "cuda"
    g = torch.Generator().manual_seed(32)
    x = torch.randn(64, 3, 224, 224, generator=g).to(dev)
    rng = np.random.default_rng(33)
    y = torch.tensor(np.clip(np.round(np.abs(rng.normal(0, 18, 64)) + 20), 0, 120).astype(np.float32)).view(-1, 1).to(dev)
    w = torch.tensor(rng.uniform(0.5, 1.5, 64).astype(np.float32)).view(-1, 1).to(dev)
    grads, encs, losses = {}, {}, {}
    for tag in ("float32", "bf16", "float32_again"):
        torch.manual_seed(31)
        model = torch_oracle.RefResNet50(fds=False, bucket_num=100, bucket_start=0, start_update=0, start_smooth=1, kernel="gaussian",
                                         ks=5, sigma=2, momentum=0.9).to(dev)
        model.train()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(tag == "bf16")):
            pred = model(x, y, 0)
        loss = (torch.abs(pred.float() - y) * w).mean()
        loss.backward()
        grads[tag] = {n: p.grad.detach().double().reshape(-1).clone() for n, p in model.named_parameters()}
        losses[tag] = float(loss.item())
    out = {"loss": losses}
    for a, b in (("bf16", "float32"), ("float32_again", "float32")):
        rel = [float((grads[a][n] - grads[b][n]).norm() / grads[b][n].norm().clamp_min(1e-300)) for n in grads[b]]
        out[f"{a}_vs_{b}_grad_rel_l2"] = {"median": float(np.median(rel)), "max": float(np.max(rel)), "min": float(np.min(rel)),
                                           "last_three": rel[-3:]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
