#!/bin/bash
# LDS bank-conflict counters of the all-taps 3x3 weight-gradient kernel (one --pmc pass over tools/probe_wgrad3.py).
R=$(pwd)
O=$R/gpurun_out/${1:-pmc_w3}
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d "$O/pmc" -o p -- python "$R/tools/probe_wgrad3.py" 64 > "$O/pmc.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$O/pmc2" -o p -- python "$R/tools/probe_wgrad3.py" 64 > "$O/pmc2.log" 2>&1
python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for sub in ("pmc", "pmc2"):
    f = glob.glob(f"{O}/{sub}/**/*counter_collection.csv", recursive=True)
    if not f:
        print(sub, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "wgrad" not in k: continue
        k = k[:60] + "|" + r.get("Grid_Size", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] in ("SQ_INSTS_LDS", "SQ_BUSY_CYCLES"): cnt[k] += 1
    for k, v in agg.items():
        print(sub, k, "launches", cnt[k], {a: b / max(1, cnt[k]) for a, b in v.items()})
PY
find "$O" -name "*.db" -delete
