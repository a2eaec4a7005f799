"""Same-box A/B of two builds of libdir_hip.so on whole training steps: alternating child processes, each loads ONE library
(the product's, or the file given) and times `steps` train steps (B=256, bf16 product path).
    python tools/ab_two_libs.py imbalanced-regression_amd/dirhip/libdir_hip_alt.so [rounds] [steps]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, torch
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip import _lib as L
if sys.argv[2] != "-":
    L.LIB_PATH = sys.argv[2]
import bench
from dirhip.train_loop import EpochFeatures, epoch_tail, resolve_loss, train_step
class A: batch, epoch_len, gpus = 256, 8, 1
device = torch.device("cuda", 0); torch.cuda.set_device(device)
model, engine, optimizer, batches = bench.build(A, device, 0)
loss_fn = resolve_loss("l1")
steps = int(sys.argv[3])
store = EpochFeatures(len(batches) * A.batch, 2048, device)
def tail(i):
    b = batches[i % len(batches)]; epoch_tail(engine, [(b[0], b[1])], 2, store)
for i in range(4): train_step(engine, optimizer, *batches[i % len(batches)], 2, loss_fn)
for i in range(2): tail(i)
best = best_t = 1e9
for r in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): train_step(engine, optimizer, *batches[i % len(batches)], 2, loss_fn)
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    t0 = time.perf_counter()
    for i in range(steps): tail(i)
    torch.cuda.synchronize(); best_t = min(best_t, (time.perf_counter() - t0) / steps * 1e3)
print(f"{best:.3f} {best_t:.3f}")
'''


def main():
    alt = os.path.abspath(sys.argv[1])
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    res = {"product": [], "alt": []}
    tails = {"product": [], "alt": []}
    for r in range(rounds):
        for name, path in (("product", "-"), ("alt", alt)):
            out = subprocess.run([sys.executable, "-c", CHILD, ROOT, path, str(steps)], capture_output=True, text=True, timeout=600)
            vals = out.stdout.strip().splitlines()[-1].split() if out.returncode == 0 else ["nan", "nan"]
            res[name].append(float(vals[0]))
            tails[name].append(float(vals[1]))
            if out.returncode != 0:
                print(out.stderr[-2000:])
    print(f"train step: product {min(res['product']):.3f} ms {res['product']}   alt ({os.path.basename(alt)}) {min(res['alt']):.3f} ms {res['alt']}", flush=True)
    print(f"epoch-tail forward: product {min(tails['product']):.3f} ms {tails['product']}   alt {min(tails['alt']):.3f} ms {tails['alt']}", flush=True)


if __name__ == "__main__":
    main()
