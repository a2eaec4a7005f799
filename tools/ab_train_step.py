"""In-process A/B of whole training steps (B=256, bf16 product path) with one switch flipped: alternating rounds of `steps`
train steps each, wall clock per step. Switches: wgrad3 (conv.set_wgrad3_all_taps), bnfuse (resnet.set_bn_bwd_fusion),
relubits (bn.set_relu_bits), joinbwd, wgradside, stemtail, wgradbatch (conv.set_wgrad_batched_reduce).   python tools/ab_train_step.py wgrad3 [rounds] [steps] [train|tail]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
import bench  # noqa: E402
import variant_switches as VS  # noqa: E402
from dirhip import _lib as L  # noqa: E402
from dirhip import bn as B  # noqa: E402
from dirhip import conv as C  # noqa: E402
from dirhip import resnet as R  # noqa: E402
from dirhip.train_loop import EpochFeatures, epoch_tail, resolve_loss, train_step  # noqa: E402


def main():
    which = sys.argv[1]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    what = sys.argv[4] if len(sys.argv) > 4 else "train"          # "train" steps or epoch-"tail" forwards (one per batch)
    from dirhip import pool as P
    # Python-level wiring switches only: the C-ABI has no process-wide kernel switches any more (kernel variants are per-launch arguments;
    # to A/B two builds of a kernel use tools/ab_two_libs.py)
    setter = {"joinbwd": VS.set_join_bwd, "wgrad3": VS.set_wgrad3_all_taps, "bnfuse": VS.set_bn_bwd_fusion, "relubits": VS.set_relu_bits,
              "wgradside": VS.set_wgrad_side_stream, "stemtail": VS.set_stem_tail_xmax, "wgradbatch": VS.set_wgrad_batched_reduce}[which]

    class A:
        batch, epoch_len, gpus = 256, 8, 1
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    model, engine, optimizer, batches = bench.build(A, device, 0)
    loss_fn = resolve_loss("l1")
    store = EpochFeatures(len(batches) * A.batch, 2048, device)

    def one(i):
        if what == "train":
            train_step(engine, optimizer, *batches[i % len(batches)], 2, loss_fn)
        else:
            b = batches[i % len(batches)]
            epoch_tail(engine, [(b[0], b[1])], 2, store)
    for i in range(3):
        one(i)
    res = {True: [], False: []}
    for r in range(rounds):
        for flag in (True, False):
            setter(flag)
            for i in range(2):
                one(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                one(i)
            torch.cuda.synchronize()
            res[flag].append((time.perf_counter() - t0) / steps * 1e3)
    setter(True)
    on, off = min(res[True]), min(res[False])
    print(f"{which} [{what}]: on {on:.3f} ms/step (rounds {[round(v, 3) for v in res[True]]})  off {off:.3f} ms/step (rounds {[round(v, 3) for v in res[False]]})  "
          f"on/off {on / off:.4f}", flush=True)


if __name__ == "__main__":
    main()
