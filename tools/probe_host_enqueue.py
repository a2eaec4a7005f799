"""How far ahead of the GPU is the host? Per train step (B=256, bf16 product path): host time to ENQUEUE the step (train_step returns
without a sync) vs the device-bound wall time per step, and the same with the weight gradients on a side stream.
    python tools/probe_host_enqueue.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
import bench  # noqa: E402
import variant_switches as VS  # noqa: E402
from dirhip import conv as C  # noqa: E402
from dirhip.train_loop import resolve_loss, train_step  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24

    class A:
        batch, epoch_len, gpus = 256, 8, 1
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    model, engine, optimizer, batches = bench.build(A, device, 0)
    loss_fn = resolve_loss("l1")
    for side in (False, True, False):
        VS.set_wgrad_side_stream(side)
        for i in range(4):
            train_step(engine, optimizer, *batches[i % len(batches)], 2, loss_fn)
        torch.cuda.synchronize()
        host = []
        t0 = time.perf_counter()
        for i in range(steps):
            a = time.perf_counter()
            train_step(engine, optimizer, *batches[i % len(batches)], 2, loss_fn)
            host.append(time.perf_counter() - a)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        host.sort()
        print(f"side_stream={side}: wall {wall / steps * 1e3:.3f} ms/step; host enqueue: first 4 steps {[round(h * 1e3, 2) for h in host[:4]]} ms (fastest), "
              f"median {host[len(host) // 2] * 1e3:.2f} ms, all steps enqueued after {t_enq / steps * 1e3:.3f} ms/step", flush=True)
    VS.set_wgrad_side_stream(False)


if __name__ == "__main__":
    main()
