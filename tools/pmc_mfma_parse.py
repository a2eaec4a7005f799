"""MFMA utilisation of the conv_igemm kernels from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE`
pass over tools/pmc_conv_pass.py (one launch pair per conv configuration of ResNet-50 at batch 256).
    python tools/pmc_mfma_parse.py <counter_collection.csv> [name.json]  ->  profiles/r04_conv_mfma_util.json
Utilisation per launch = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): the MFMA counter counts cycles, 32
per v_mfma_f32_32x32x16_bf16, summed over all SIMDs (MI355X_MICROARCH.md; checked here: it equals FLOP / 32768 x 32 of the
launch to 3 digits, `busy_over_ideal`), GRBM_GUI_ACTIVE is summed over the 8 XCDs (active / 8 / duration = 2.2 GHz, the
shader clock)."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import RESNET50_CONVS  # noqa: E402

B = 256
cfgs = []
for cin, cout, k, st, h, cnt in RESNET50_CONVS:
    pad = k // 2
    ho = (h + 2 * pad - k) // st + 1
    cfgs.append((f"{cin}->{cout} k{k} s{st} H{h} fwd", cnt, 2.0 * B * ho * ho * cout * cin * k * k))
    if st == 1:
        cfgs.append((f"{cout}->{cin} k{k} s1 H{ho} dgrad", cnt, 2.0 * B * ho * ho * cout * cin * k * k))
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "conv_igemm" in r["Kernel_Name"] or "conv3x3_patch" in r["Kernel_Name"]]     # (conv_igemm_big_kernel included)
by_disp = collections.OrderedDict()
for r in rows:
    d = by_disp.setdefault(r["Dispatch_Id"], {"_t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
disp = list(by_disp.values())
assert len(disp) == 2 * len(cfgs), (len(disp), len(cfgs))
out = {"layers": []}
tot_busy = tot_active = tot_ideal = 0.0
for (name, cnt, flop), d in zip(cfgs, disp[1::2]):                 # second launch of each configuration
    busy, active = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), d.get("GRBM_GUI_ACTIVE", 0.0)
    ideal = flop / 32768.0 * 32.0
    out["layers"].append({"layer": name, "count": cnt, "mfma_busy_cycles": busy, "gui_active_cycles": active, "duration_ns": d["_t"],
                          "sq_busy_cycles": d.get("SQ_BUSY_CYCLES"),
                          "mfma_util": busy / (active / 8.0 * 1024.0) if active else None, "clock_GHz": active / 8.0 / d["_t"] if d["_t"] else None, "ideal_busy_cycles": ideal})
    tot_busy += cnt * busy
    tot_active += cnt * active
    tot_ideal += cnt * ideal
out["weighted_mfma_util"] = tot_busy / (tot_active / 8.0 * 1024.0) if tot_active else None
out["busy_over_ideal"] = tot_busy / tot_ideal
out["note"] = "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace over tools/pmc_conv_pass.py 256"
json.dump(out, open(os.path.join(ROOT, "profiles", sys.argv[2] if len(sys.argv) > 2 else "r04_conv_mfma_util.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "layers"}, indent=1))
for l in out["layers"]:
    u = l["mfma_util"]
    print(f"{l['layer']:32s} x{l['count']} util {u if u is None else round(u, 3)}  busy/ideal {l['mfma_busy_cycles'] / l['ideal_busy_cycles']:.2f}  "
          f"active {l['gui_active_cycles']:.0f} cyc  {l['duration_ns'] / 1e3:.1f} us")
