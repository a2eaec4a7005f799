#!/bin/bash
# per-launch durations of the conv kernels of the LAST training step of tools/in_situ_conv_launches.py, grouped by (kernel, grid)
R=$(pwd); O=$R/gpurun_out/${1:-insitu}; mkdir -p "$O"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$O/t" -o k -- python "$R/tools/in_situ_conv_launches.py" > "$O/log.txt" 2>&1
python - "$O" <<'PY'
import csv, glob, sys, collections, re
O = sys.argv[1]
f = glob.glob(f"{O}/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "FusedAdam" in r["Kernel_Name"]]
last = rows[adam[-6] + 1:adam[-1] + 1]            # the last whole step (5 Adam launches per step)
agg = collections.OrderedDict()
for r in last:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    if not any(s in k for s in ("conv_igemm", "conv3x3_patch", "conv_wgrad")): continue
    k = re.sub(r"\(.*", "", k).replace("void ", "")
    key = (k, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
out = open(f"{O}/conv_launches_in_situ.txt", "w")
for (k, g), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    line = f"{k:42s} WGs {g:6d}  x{c:2d}  {t / c:7.1f} us each  {t:8.1f} us/step"
    print(line); out.write(line + "\n")
PY
find "$O" -name "*.db" -delete; find "$O/t" -name "*kernel_trace.csv" -delete
