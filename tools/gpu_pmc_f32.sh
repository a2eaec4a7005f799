#!/bin/bash
# SQ wave-state + MFMA-busy counters of the float32 tile kernels:   bash tools/gpu_pmc_f32.sh [variant 2 | 3 | 4] -> gpurun_out/pmc_f32_v<variant>
V=${1:-2}; R=$(pwd); O=$R/gpurun_out/pmc_f32_v$V; mkdir -p "$O"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$O/a" -o a -- python "$R/tools/pmc_f32_pass.py" $V > "$O/a.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace --output-format csv -d "$O/b" -o b -- python "$R/tools/pmc_f32_pass.py" $V > "$O/b.log" 2>&1
find "$O" -name "*.db" -delete
O=$O python - <<'PY'
import csv, glob, collections, os
for tag in "ab":
    f = glob.glob(os.environ["O"] + f"/{tag}/**/*counter_collection.csv", recursive=True)
    if not f: print("no csv", tag); continue
    by = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        if "conv_f32_tile" not in r["Kernel_Name"]: continue
        d = by.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"][40:75], "t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    for k, d in by.items():
        print(k, d["name"], d["t"], {a: round(b) for a, b in d.items() if a not in ("name", "t")})
PY
