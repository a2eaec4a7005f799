"""rocprofv3 kernel_trace.csv of `bench.py` -> per-train-step kernel breakdown of the train-only timed region
(step boundaries = the optimizer launch of each step: adam_step_kernel; in older traces the 5 launches of torch's fused Adam)."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps = [i for i, r in enumerate(rows) if "adam_step_kernel" in r["Kernel_Name"]]
if not steps:
    adam = [i for i, r in enumerate(rows) if "FusedAdam" in r["Kernel_Name"]]
    steps = [adam[i] for i in range(4, len(adam), 5)]
n = 23
seg = rows[steps[-(n + 1)] + 1:steps[-1] + 1]
wall = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    k = re.sub(r"\(.*", "", k)[-70:]
    agg[k][0] += 1
    agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
busy = sum(v[1] for v in agg.values()) / 1e6
print(f"train-only steps: {n}  wall/step {wall / n:.2f} ms  busy/step {busy / n:.2f} ms")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:48]:
    print(f"{k:<70s} {c / n:6.1f}/step {t / n / 1e3:8.1f} us/step {100 * t / busy / 1e6:5.1f}%")
