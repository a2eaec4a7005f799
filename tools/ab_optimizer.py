"""Same-process A/B of whole training steps (B = 256, bf16 product path): dirhip.optim.Adam (one launch incl. the bf16 weight operands)
vs torch.optim.Adam(fused=True) + dir_conv_prep_weights_batched, two independent model replicas, alternating rounds.
    python tools/ab_optimizer.py [rounds] [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
import bench  # noqa: E402
from dirhip.train_loop import resolve_loss, train_step  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12

    class A:
        batch, epoch_len, gpus = 256, 4, 1
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    model, engine, opt_ours, batches = bench.build(A, device, 0)
    opt_torch = torch.optim.Adam(engine.parameters(), lr=1e-3, fused=True)
    loss_fn = resolve_loss("l1")
    res = {"ours": [], "torch_fused": []}
    for r in range(rounds + 1):
        for name, opt in (("ours", opt_ours), ("torch_fused", opt_torch)):
            for i in range(2):
                train_step(engine, opt, *batches[i % len(batches)], 2, loss_fn)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                train_step(engine, opt, *batches[i % len(batches)], 2, loss_fn)
            torch.cuda.synchronize()
            if r:
                res[name].append((time.perf_counter() - t0) / steps * 1e3)
    a, b = min(res["ours"]), min(res["torch_fused"])
    print(f"optimizer [train]: dirhip.optim.Adam {a:.3f} ms/step (rounds {[round(v, 3) for v in res['ours']]})  torch fused Adam + batched prep {b:.3f} ms/step "
          f"(rounds {[round(v, 3) for v in res['torch_fused']]})  ratio {a / b:.4f}", flush=True)


if __name__ == "__main__":
    main()
