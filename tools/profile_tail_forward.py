"""Kernel breakdown of the FDS epoch-tail forward (no-grad, train-mode forward of one B=256 batch + store append), in situ."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
import bench  # noqa: E402
from dirhip.train_loop import EpochFeatures, epoch_tail, resolve_loss, train_step  # noqa: E402


def main():
    class A:
        batch, epoch_len, gpus = 256, 4, 1
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    model, engine, optimizer, batches = bench.build(A, device, 0)
    loss_fn = resolve_loss("l1")
    store = EpochFeatures(len(batches) * A.batch, 2048, device)
    for i in range(2):
        train_step(engine, optimizer, *batches[i], 2, loss_fn)
    tb = [(b[0], b[1]) for b in batches]
    epoch_tail(engine, tb, 2, store)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        epoch_tail(engine, tb, 2, store)
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages():
        tot = getattr(e, "device_time_total", None) or getattr(e, "cuda_time_total", 0.0)
        if tot:
            rows.append((tot / len(tb), e.count / len(tb), e.key))
    rows.sort(reverse=True)
    print(f"per forward: {sum(r[0] for r in rows) / 1e3:.3f} ms busy")
    for t, c, k in rows[:32]:
        print(f"{k[:80]:80s} {c:6.1f}/fwd {t:8.1f} us")


if __name__ == "__main__":
    main()
