#!/bin/bash
# L2 (TCC) / L1 (TCP) counter passes over tools/pmc_l2_pass.py -> gpurun_out/<dir>/cc/*.csv + a per-dispatch table
R=$(pwd); O=$R/gpurun_out/${1:-pmc_l2}; mkdir -p "$O/cc"; cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Za-z_0-9]*\|TCP_[A-Za-z_0-9]*\|TA_[A-Za-z_0-9]*\|TD_[A-Za-z_0-9]*" | sort -u > "$O/counters.txt"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --kernel-trace --output-format csv -d "$O/p1" -o p1 -- python "$R/tools/pmc_l2_pass.py" 256 > "$O/p1.log" 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/p2" -o p2 -- python "$R/tools/pmc_l2_pass.py" 256 > "$O/p2.log" 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_TAG_STALL_sum --kernel-trace --output-format csv -d "$O/p3" -o p3 -- python "$R/tools/pmc_l2_pass.py" 256 > "$O/p3.log" 2>&1
for p in p1 p2 p3; do f=$(find "$O/$p" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/cc/$p.csv"; tail -3 "$O/$p.log"; done
find "$O" -name "*.db" -delete
cd "$R"
python - "$O" <<'PY'
import csv, sys, glob, collections
O = sys.argv[1]
for f in sorted(glob.glob(O + "/cc/*.csv")):
    rows = list(csv.DictReader(open(f)))
    by = collections.OrderedDict()
    for r in rows:
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"][:48], r.get("Grid_Size", ""))
        by.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    print(f)
    for (d, name, grid), c in by.items():
        if "conv" not in name: continue
        print(f"  {d:>4} {name:48s} grid {grid:>8} " + " ".join(f"{k}={v:.4g}" for k, v in c.items()))
PY
