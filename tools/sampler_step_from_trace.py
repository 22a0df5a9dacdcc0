#!/usr/bin/env python3
"""Per-sampler-step kernel table out of a rocprofv3 kernel trace of tools/bench_sampler_configs3.py: the launches between two
consecutive predictor kernels (sampler_predict), corrections = 0.
    python tools/sampler_step_from_trace.py <kernel_trace.csv> [step index from the end, default 2]"""
import csv
import sys

path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "predict" in r["Kernel_Name"]]
i0, i1 = marks[-back - 1], marks[-back]
span = (int(rows[i1]["End_Timestamp"]) - int(rows[i0]["End_Timestamp"])) / 1e6
agg = {}
for r in rows[i0 + 1: i1 + 1]:
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    n = n.split("(")[0] if not n.startswith("_Z") else n
    grid = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))
    wgs = int(r["Workgroup_Size"]) if "Workgroup_Size" in r else int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1)) * int(r.get("Workgroup_Size_Z", 1))
    wg = grid // max(wgs, 1)
    a = agg.setdefault((n[:90], wg), [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
busy = sum(v[1] for v in agg.values()) / 1e3
print(f"{path}: sampler step {back} from the end: {i1 - i0} launches, {span:.3f} ms between predictor kernels, {busy:.3f} ms of kernel time")
print(f"{'ms/step':>9} {'launches':>8} {'avg us':>9} {'workgroups':>10}  kernel")
for (k, wg), v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[1] / 1e3:9.3f} {v[0]:8d} {v[1] / v[0]:9.1f} {wg:10d}  {k}")
