"""Every conv launch of ONE training step in situ (B=256, bf16 product path), in launch order: kernel, grid, microseconds —
from a rocprofv3 kernel trace of this script (run it under `rocprofv3 --kernel-trace --output-format csv`), or standalone it
prints the torch-profiler view grouped by (kernel, grid)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
import bench  # noqa: E402
from dirhip.train_loop import resolve_loss, train_step  # noqa: E402


def main():
    class A:
        batch, epoch_len, gpus = 256, 4, 1
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    model, engine, optimizer, batches = bench.build(A, device, 0)
    loss_fn = resolve_loss("l1")
    for i in range(6):
        train_step(engine, optimizer, *batches[i % len(batches)], 2, loss_fn)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
