"""Where does the conv family's in-situ time (whole training steps) exceed the same launches measured alone (bench --extras: conv_layers)?
VERDICT r5 item 5: toggle the fused epilogues one at a time (tools/variant_switches.py) and read the per-family device times of whole
steps from the profiler's kernel trace (bench.in_situ_breakdown): what leaves the conv family when a fusion is off is what that fusion costs
inside it; the BatchNorm family shows where the work went.      python tools/attribute_in_situ_gap.py [steps]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
import variant_switches as VS  # noqa: E402


def run(tag, setters, steps):
    class A:
        pass
    A.batch, A.epoch_len, A.gpus = 256, 4, 1
    prev = [(s, s(v)) for s, v in setters]
    try:
        device = torch.device("cuda", 0)
        from dirhip.train_loop import resolve_loss
        model, engine, optimizer, batches = bench.build(A, device, 0)
        fam = bench.in_situ_breakdown(engine, optimizer, batches, resolve_loss("l1"), 2, steps=steps)
        del model, engine, optimizer, batches
        torch.cuda.empty_cache()
    finally:
        for s, p in prev:
            s(p)
    busy = sum(f["us_per_step"] for f in fam.values())
    row = {"variant": tag, "busy_ms": busy / 1e3}
    for k in ("conv_igemm", "conv_wgrad", "batchnorm", "stem", "stem_tail"):
        row[k + "_ms"] = fam.get(k, {}).get("us_per_step", 0.0) / 1e3
        row[k + "_launches"] = fam.get(k, {}).get("launches_per_step", 0.0)
    print(json.dumps(row), flush=True)
    return row


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    rows = [run("product (all fusions)", [], steps),
            run("BatchNorm-backward sums NOT in the data-gradient epilogues", [(VS.set_bn_bwd_fusion, False)], steps),
            run("ReLU mask from the bf16 tensor instead of the bit mask", [(VS.set_relu_bits, False)], steps),
            run("no graph fusion (plain conv -> BatchNorm nodes: no fused addend / deferred ReLU / projection pair / join)",
                [(VS.set_graph_fusion, False), (VS.set_bn_bwd_fusion, False)], steps),
            run("product again (box drift)", [], steps)]
    out = os.path.join(ROOT, "gpurun_out", "in_situ_gap_attribution.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rows, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
