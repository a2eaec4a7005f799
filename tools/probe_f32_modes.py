"""GPU probe: the float32 graph of the bench loop in its three conv arithmetics — exact float32 MFMA, split-bf16 x3, split-bf16 x2 — at B = 256:
train step / loop time and the step-0 loss against the reference's own float32 CPU run (tests/golden/step0_b256.npz).
    python tools/probe_f32_modes.py [exact,x3,x2] > gpurun_out/f32_modes.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
sys.modules.setdefault("bench", bench)
import bench_extras  # noqa: E402
from dirhip.train_loop import resolve_loss  # noqa: E402


class Args:
    batch, epoch_len, gpus, steps = 256, 8, 1, 8


def main():
    modes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["exact", "x3", "x2"]
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    out = {}
    for m in modes:
        out[m] = bench_extras.float32_mode_probe(device, Args, resolve_loss("l1"), m)
        print(m, json.dumps({k: out[m].get(k) for k in ("train_only_ms_per_step", "ms_per_step", "step0_loss", "loss_rel_err_vs_golden")}), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
