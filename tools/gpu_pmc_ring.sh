#!/bin/bash
# SQ counter passes over tools/pmc_ring_pass.py -> gpurun_out/<dir>/cc/*.csv + a per-dispatch table
R=$(pwd); O=$R/gpurun_out/${1:-pmc_ring}; mkdir -p "$O/cc"; cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|TA_[A-Z_0-9]*" | sort -u > "$O/counters.txt"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d "$O/p1" -o p1 -- python "$R/tools/pmc_ring_pass.py" 256 > "$O/p1.log" 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/p2" -o p2 -- python "$R/tools/pmc_ring_pass.py" 256 > "$O/p2.log" 2>&1
find "$O" -name "*counter_collection.csv" -exec cp {} "$O/cc/" \;
find "$O" -name "*.db" -delete
cd "$R"
python - "$O" <<'PY'
import csv, sys, glob, collections
O = sys.argv[1]
for f in sorted(glob.glob(O + "/cc/*.csv")):
    rows = list(csv.DictReader(open(f)))
    by = collections.OrderedDict()
    for r in rows:
        k = (r["Dispatch_Id"], r["Kernel_Name"][:60], r.get("Grid_Size", ""))
        by.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    print(f)
    for (d, name, grid), c in by.items():
        if "conv" not in name: continue
        print(f"  {d:>4} {name:60s} grid {grid:>8} " + " ".join(f"{k}={v:.4g}" for k, v in c.items()))
PY
