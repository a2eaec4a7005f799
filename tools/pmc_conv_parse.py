"""Parse rocprofv3 counter_collection CSVs of tools/pmc_conv_pass.py -> per-step / per-launch HBM traffic of the conv
forward + data-gradient kernels (conv_igemm*, conv3x3_patch*).
    python tools/pmc_conv_parse.py <dir with *FETCH_SIZE*counter_collection.csv and *WRITE_SIZE*counter_collection.csv>
writes profiles/r02_conv_pmc_traffic.json; `parse(dir)` is also what bench.py calls after its own two counter passes."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = 256


def parse(directory, batch=B):
    sys.path.insert(0, ROOT)
    from bench import RESNET50_CONVS
    cfgs = []
    for cin, cout, k, st, h, cnt in RESNET50_CONVS:
        pad = k // 2
        ho = (h + 2 * pad - k) // st + 1
        cfgs.append((cnt, 2.0 * batch * (h * h * cin + ho * ho * cout) + 2.0 * cout * cin * k * k))
        if st == 1:
            cfgs.append((cnt, 2.0 * batch * (ho * ho * cout + h * h * cin) + 2.0 * cout * cin * k * k))
    out = {"batch": batch, "launches_per_step": sum(c for c, _ in cfgs), "algorithmic_bytes_per_step": sum(c * b for c, b in cfgs)}
    for name in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(os.path.join(directory, "**", f"*{name}*counter_collection.csv"), recursive=True)
        rows = [r for r in csv.DictReader(open(files[0]))
                if ("conv_igemm" in r["Kernel_Name"] or "conv3x3_patch" in r["Kernel_Name"]) and r["Counter_Name"] == name]
        vals = [float(r["Counter_Value"]) for r in rows]
        assert len(vals) == 2 * len(cfgs), (len(vals), len(cfgs))
        last = vals[1::2]                                        # second launch of each configuration
        out[name + "_KB_per_step_raw"] = sum(c * v for (c, _), v in zip(cfgs, last))
    # MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads
    out["hbm_read_bytes_per_step"] = out["FETCH_SIZE_KB_per_step_raw"] * 1024 * 2
    out["hbm_write_bytes_per_step"] = out["WRITE_SIZE_KB_per_step_raw"] * 1024
    out["traffic_bytes_per_step"] = out["hbm_read_bytes_per_step"] + out["hbm_write_bytes_per_step"]
    out["traffic_bytes_per_launch"] = out["traffic_bytes_per_step"] / out["launches_per_step"]
    out["note"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/pmc_conv_pass.py; FETCH_SIZE doubled per MI355X_MICROARCH.md"
    return out


if __name__ == "__main__":
    res = parse(sys.argv[1])
    json.dump(res, open(os.path.join(ROOT, "profiles", sys.argv[2] if len(sys.argv) > 2 else "r04_conv_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))
