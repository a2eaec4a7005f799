"""Per-kernel device time of the float32-mode training step (amp_dtype=None), from the profiler's kernel trace.
    python tools/profile_f32_step.py [batch] [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
import bench  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2

    class A:
        pass
    A.batch, A.epoch_len, A.gpus = batch, 2, 1
    device = torch.device("cuda", 0)
    from dirhip.train_loop import resolve_loss, train_step
    model, engine, optimizer, batches = bench.build(A, device, 0, amp_dtype=None)
    loss_fn = resolve_loss("l1")
    for s in range(2):
        train_step(engine, optimizer, *batches[s % 2], 2, loss_fn)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for s in range(steps):
        train_step(engine, optimizer, *batches[s % 2], 2, loss_fn)
    torch.cuda.synchronize()
    print(f"float32 mode, B={batch}: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms per train step (wall)")
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for s in range(steps):
            train_step(engine, optimizer, *batches[s % 2], 2, loss_fn)
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages():
        tot = getattr(e, "device_time_total", None) or getattr(e, "cuda_time_total", 0.0)
        if tot:
            rows.append((tot / steps, e.count / steps, e.key))
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)
    print(f"busy {total / 1e3:.2f} ms per step")
    for us, cnt, name in rows[:40]:
        print(f"{us / 1e3:9.3f} ms {cnt:7.1f} launches  {us / total * 100:5.1f}%  {name[:110]}")


if __name__ == "__main__":
    main()
