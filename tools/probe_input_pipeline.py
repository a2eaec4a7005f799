"""GPU probe of the real-file input pipeline (tools/bench_extras.py: input_pipeline).   python tools/probe_input_pipeline.py [e2e [gil switch interval] [depth] [pinned staging 0/1]]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_extras as bench  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "e2e":
    kw = {}
    if len(sys.argv) > 2 and float(sys.argv[2]) > 0:
        kw["switch_interval"] = float(sys.argv[2])
    if len(sys.argv) > 3:
        kw["depth"] = int(sys.argv[3])
    if len(sys.argv) > 4:
        kw["pinned"] = bool(int(sys.argv[4]))
    if len(sys.argv) > 5:
        kw["diagnose"] = sys.argv[5]
    print(json.dumps(bench.input_pipeline_probe(torch.device("cuda", 0), 10300.0, only_end_to_end=True, e2e_kw=kw)["end_to_end"], indent=1))
else:
    print(json.dumps(bench.input_pipeline_probe(torch.device("cuda", 0), 10300.0), indent=1))
